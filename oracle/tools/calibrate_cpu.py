#!/usr/bin/env python3
"""Build container only: the CPU-baseline calibration SURVEY.md section 8d(iii) asks for.

The vendored pomegranate (the reference's own Viterbi path) can only be timed where /root/reference exists, i.e. here;
on the GPU box bench.py times the C restatement oracle/viterbi_oracle.c instead.  This script runs BOTH on the same
2 000-read REF150 batch (config C1: seed 20240601, 150-base reads), one thread each, on this container's CPU, checks
that they return the same log-probabilities, and writes the ratio to profiles/cpu_calibration.json so that a GPU-box
`cpu_baseline.value` can be read as "pomegranate-equivalent" (value / ratio).  Only numbers are written.

    python oracle/tools/build_reference.py && python oracle/tools/calibrate_cpu.py [n_reads]
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "nx111"), os.path.join(HERE, "stubs"), os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build"), REPO]

import numpy as np                                               # noqa: E402
from advntr import settings as ref_settings, hmm_utils as ref_hmm_utils      # noqa: E402  (the reference)
from advntr_amd import _lib, workloads                           # noqa: E402  (read generator + encoder only)
from oracle.oracle import OracleModel                            # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def run(shape, locus):
    ref_settings.MAX_ERROR_RATE = 0.05
    m = ref_hmm_utils.get_read_matcher_model(locus.left, locus.right, locus.units, locus.copies)
    reads = workloads.make_reads(np.random.default_rng(20240601), locus, n_reads, 150)
    t0 = time.process_time()
    ref_logp = [m.viterbi(r)[0] for r in reads]
    t_ref = time.process_time() - t0
    idx = {s: i for i, s in enumerate(m.states)}
    edges = [(idx[a], idx[b], d["probability"]) for a, b, d in m.graph.edges_iter(data=True)]
    emis = np.array([[s.distribution.log_probability(c) for c in "ACGT"] for s in m.states[:m.silent_start]])
    O = OracleModel(len(m.states), m.silent_start, m.start_index, m.end_index, edges, emis)
    bases, off = _lib.encode_reads(reads)
    t0 = time.process_time()
    logp, _ = O.viterbi_many(bases, off)
    t_or = time.process_time() - t0
    assert np.array_equal(np.array(ref_logp), logp), "oracle and reference disagree"
    return {"shape": shape, "states": len(m.states), "emitting": int(m.silent_start), "edges": len(edges), "reads": n_reads,
            "pomegranate_reads_per_s": n_reads / t_ref, "oracle_reads_per_s": n_reads / t_or,
            "oracle_over_pomegranate": t_ref / t_or, "logp_bit_equal": True}


ref150 = run("REF150", workloads.ref150())
s300 = run("S300", workloads.s300())
out = {"what": "oracle/viterbi_oracle.c vs the reference's vendored pomegranate 0.6.1 (Model.viterbi incl. its Python path "
               "list), same reads, 1 thread each, time.process_time()",
       "cpu_model": cpu_model(), "host_threads": os.cpu_count(),
       "oracle_over_pomegranate": ref150["oracle_over_pomegranate"], "ref150": ref150, "s300": s300,
       "use": "pomegranate-equivalent reads/s on another host = that host's oracle reads/s / oracle_over_pomegranate"}
json.dump(out, open(os.path.join(REPO, "profiles", "cpu_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
