#!/usr/bin/env python3
"""Generate tests/golden/model_update.json.gz by RUNNING THE REFERENCE: the model re-estimation step of
VNTRFinder.iteratively_update_model (advntr/vntr_finder.py:667-697) -- reads and the reference repeat units are scored,
their Viterbi paths go back into hmm_utils.get_read_matcher_model(left, right, None, copies, vpaths)
(hmm_utils.py:424-431: profile parameters from the multiple alignment of the paths' repeat units).  TEST
INFRASTRUCTURE; only data is written: the (sequence, path state names) pairs and the baked model they produce.

    python oracle/tools/build_reference.py && python tests/golden/make_update_golden.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build")
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"), os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD]

import numpy as np                                    # noqa: E402
from advntr import settings, hmm_utils                # noqa: E402


def dump_model(m):
    idx = {s: i for i, s in enumerate(m.states)}
    edges = [[idx[a], idx[b], d["probability"]] for a, b, d in m.graph.edges_iter(data=True)]
    emis = [{"prob": [s.distribution.parameters[0][c] for c in "ACGT"],
             "logp": [s.distribution.log_probability(c) for c in "ACGT"]} for s in m.states[:m.silent_start]]
    return {"state_names": [s.name for s in m.states], "silent_start": m.silent_start, "start_index": m.start_index,
            "end_index": m.end_index, "edges": edges, "emissions": emis}


def main():
    rng = np.random.default_rng(606)
    dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    settings.MAX_ERROR_RATE = 0.05
    hmm_utils.build_profile_hmm_for_repeats = \
        lambda repeats, error_rate: hmm_utils.build_profile_hmm_pseudocounts_for_alignment(error_rate, repeats)
    pattern, left, right = dna(13), dna(200), dna(200)
    segs = [pattern] * 4
    copies = int(round(100.0 / 13 + 0.5))
    hmm = hmm_utils.get_read_matcher_model(left[-100:], right[:100], segs, copies)
    # the sample carries a variant unit: one base inserted in some units, one substituted in others
    var_ins = pattern[:6] + "G" + pattern[6:]
    var_sub = pattern[:3] + ("A" if pattern[3] != "A" else "C") + pattern[4:]
    allele = left + pattern + var_ins + pattern + var_sub + pattern + right
    sequences = []
    for _ in range(40):
        st = int(rng.integers(110, len(allele) - 210))
        sequences.append(allele[st:st + 100])
    sequences += [s.upper() for s in segs]                                  # reference_repeats (vntr_finder.py:673-677)
    vpaths = []
    for s in sequences:
        logp, vpath = hmm.viterbi(s)
        if vpath is not None:
            vpaths.append((s, vpath))
    updated = hmm_utils.get_read_matcher_model(left[-100:], right[:100], None, copies, vpaths)
    alignment = hmm_utils.get_multiple_alignment_of_repeats_from_reads(vpaths)
    out = {"left": left[-100:], "right": right[:100], "copies": copies, "error_rate": 0.05,
           "vpaths": [[s, [st.name for _, st in vp]] for s, vp in vpaths], "alignment": alignment,
           "model": dump_model(updated)}
    with gzip.open(os.path.join(HERE, "model_update.json.gz"), "wt") as fh:
        json.dump(out, fh)
    print("vpaths %d, alignment %d rows x %d, states %d" % (len(vpaths), len(alignment), len(alignment[0]), len(updated.states)))


if __name__ == "__main__":
    main()
