#!/usr/bin/env python3
"""A/B of the fused short-read kernel against the split launch (ADVNTR_FLAG_SPLIT_FINISH: sweep kernel, then finish kernel) inside ONE
process on one GPU box: the bench shapes REF150 and S300, 100 000 reads each; every record of the split launch must equal the fused
launch's (logp bit for bit, all eight summary integers); kernel-region ms from HIP events on the launch stream, interleaved rounds.
  python3 scripts/ab_split.py [--reads 100000] [--rounds 3] [--iters 10]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--extra-flags", type=int, default=0)
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from advntr_amd import _lib, workloads
    _lib.require_gpu()
    out = {}
    shapes = [("ref150", workloads.ref150()), ("s300", workloads.s300())]
    for name, locus in shapes:
        reads = workloads.make_reads(np.random.default_rng(20240601), locus, args.reads, 150)
        bases, off = _lib.encode_reads(reads)
        rm = np.zeros(args.reads, np.int32)
        dm = [locus.model.device_model()]
        fused = _lib.DeviceBatch(dm, bases, off, rm, flags=args.extra_flags)
        split = _lib.DeviceBatch(dm, bases, off, rm, flags=args.extra_flags | _lib.FLAG_SPLIT_FINISH)
        for b in (fused, split):
            b.run()
            b.sync()
        lf, sf = fused.fetch()
        ls, ss = split.fetch()
        same = bool(np.array_equal(lf, ls) and np.array_equal(sf, ss))
        rec = {"identical": same, "fused_ms": [], "split_ms": [], "fused_kernels": fused.kernels(), "split_kernels": split.kernels(),
               "fused_bytes": fused.device_bytes(), "split_bytes": split.device_bytes()}
        if not same:
            bad = np.nonzero((lf != ls) | (sf != ss).any(axis=1))[0]
            rec["first_bad"] = [int(x) for x in bad[:8]]
            rec["n_bad"] = int(len(bad))
        for _ in range(args.rounds):
            rec["fused_ms"].append(round(fused.run_timed(args.iters), 4))
            rec["split_ms"].append(round(split.run_timed(args.iters), 4))
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
        fused.close()
        split.close()
    return 0 if all(r["identical"] for r in out.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
