#!/usr/bin/env python3
"""Generate tests/golden/bake_merge.json.gz by RUNNING THE REFERENCE's vendored pomegranate (hmm.pyx:673-1123):
random small models with orphan states, out-edges that do not sum to one and silent states with probability-1
transitions, baked with merge='All' and merge='Partial'; plus hmm_utils.build_reference_repeat_finder_hmm
(hmm_utils.py:598-680, baked with the default merge) with ReferenceVNTR.find_repeat_segments' Viterbi segmentation
(reference_vntr.py:80-87) of a few regions.  TEST INFRASTRUCTURE; only data is written.

    python oracle/tools/build_reference.py && python tests/golden/make_merge_golden.py
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
from advntr import hmm_utils                          # noqa: E402
from pomegranate import HiddenMarkovModel, State, DiscreteDistribution      # noqa: E402


def dump(m):
    idx = {s: i for i, s in enumerate(m.states)}
    return {"state_names": [s.name for s in m.states], "silent_start": m.silent_start, "start_index": m.start_index,
            "end_index": m.end_index,
            "edges": [[idx[a], idx[b], d["probability"]] for a, b, d in m.graph.edges_iter(data=True)],
            "emissions": [{"logp": [s.distribution.log_probability(c) for c in "ACGT"]} for s in m.states[:m.silent_start]]}


def random_spec(rng):
    n_emit, n_silent = int(rng.integers(1, 6)), int(rng.integers(1, 9))
    names = ["E%d" % i for i in range(n_emit)] + ["S%d" % i for i in range(n_silent)]
    dists = []
    for _ in range(n_emit):
        p = rng.random(4) + 0.05
        p = p / p.sum()
        dists.append([float(x) for x in p])
    n = n_emit + n_silent
    edges = []          # (a, b, prob) with a/b in -1 (start), 0..n-1, n (end)
    order = rng.permutation(n_silent)
    rank = {n_emit + int(s): r for r, s in enumerate(order)}          # silent->silent edges follow this order: no cycles
    for a in [-1] + list(range(n)):
        targets = [b for b in list(range(n)) + [n] if b != -1]
        k = int(rng.integers(0, 4))
        chosen = list(rng.choice(len(targets), size=min(k, len(targets)), replace=False)) if k else []
        probs = rng.random(len(chosen)) + 0.05
        style = rng.random()
        if style < 0.45 and len(chosen):
            probs = probs / probs.sum()                    # a proper row
        elif style < 0.6 and len(chosen) >= 1:
            probs[:] = 0.0
            probs[0] = 1.0                                 # a probability-1 edge (merge candidate if `a` is silent)
            chosen = chosen[:1]
            probs = probs[:1]
        for ci, pr in zip(chosen, probs):
            b = targets[int(ci)]
            if a >= n_emit and b != n and b >= n_emit and rank[a] >= rank[b]:
                continue                                   # keep the silent sub-graph acyclic
            if a == b and a >= n_emit:
                continue
            edges.append([int(a), int(b), float(pr)])
    return {"names": names, "n_emit": n_emit, "dists": dists, "edges": edges}


def build(spec):
    m = HiddenMarkovModel("rnd")
    states = []
    for i, nm in enumerate(spec["names"]):
        d = DiscreteDistribution(dict(zip("ACGT", spec["dists"][i]))) if i < spec["n_emit"] else None
        states.append(State(d, name=nm))
    m.add_states(states)
    n = len(states)
    for a, b, p in spec["edges"]:
        m.add_transition(m.start if a == -1 else states[a], m.end if b == n else states[b], p)
    return m


def main():
    rng = np.random.default_rng(20241)
    cases = []
    tries = 0
    while len(cases) < 60 and tries < 2000:
        tries += 1
        spec = random_spec(rng)
        for merge in ("All", "Partial"):
            try:
                m = build(spec)
                m.bake(merge=merge)
                if m.silent_start == 0:
                    continue
                cases.append({"merge": merge, "spec": spec, "model": dump(m)})
            except Exception:
                pass
    # the repeat finder and its segmentation
    finder = []
    for pattern, copies in (("ACGTTGCA", 5), ("GATTACAGATTACCA", 3), ("CAG", 8)):
        m = hmm_utils.build_reference_repeat_finder_hmm([pattern], copies=copies)
        regions = []
        for k in range(4):
            units = []
            for c in range(copies - (k % 2)):
                u = list(pattern)
                if rng.random() < 0.4:
                    u[int(rng.integers(0, len(u)))] = "ACGT"[int(rng.integers(0, 4))]
                if rng.random() < 0.2 and len(u) > 2:
                    u.pop(int(rng.integers(0, len(u))))
                units.append("".join(u))
            region = "".join(units)
            if k == 3:
                region = "TTGACA" + region + "GGTACC"
            logp, path = m.viterbi(region)
            names = [st.name for _, st in path[1:-1]]
            regions.append({"region": region, "logp": logp, "path": [i for i, _ in path],
                            "segments": hmm_utils.get_repeat_segments_from_visited_states_and_region(names, region)})
        finder.append({"pattern": pattern, "copies": copies, "model": dump(m), "regions": regions})
    with gzip.open(os.path.join(HERE, "bake_merge.json.gz"), "wt") as fh:
        json.dump({"cases": cases, "repeat_finder": finder}, fh)
    print("random merge cases %d (of %d tries), repeat finder models %d" % (len(cases), tries, len(finder)))


if __name__ == "__main__":
    main()
