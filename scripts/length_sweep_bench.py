"""Kernel time per 100 k reads by read length: the row-blocked kernels (default routing) against the anti-diagonal kernel
(ADVNTR_FLAG_ANTIDIAGONAL) on the REF150 model; results must be identical.  Usage: python scripts/length_sweep_bench.py"""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads

loc = workloads.ref150()
dm = loc.model.device_model()
out = []
for n in (40, 64, 76, 100, 124, 125, 150, 155):
    reads = workloads.make_reads(np.random.default_rng(n), loc, 100000, n)
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    res = {}
    for name, flags in (("rows", 0), ("antidiagonal", _lib.FLAG_ANTIDIAGONAL)):
        B = _lib.DeviceBatch([dm], bases, off, which, flags=flags)
        B.run()
        ms = B.run_timed(3)
        res[name] = (ms, B.fetch())
        B.close()
    same = np.array_equal(res["rows"][1][0], res["antidiagonal"][1][0]) and np.array_equal(res["rows"][1][1], res["antidiagonal"][1][1])
    row = {"read_len": n, "rows_ms": res["rows"][0], "antidiagonal_ms": res["antidiagonal"][0],
           "speedup": res["antidiagonal"][0] / res["rows"][0], "identical": bool(same)}
    print(json.dumps(row))
    out.append(row)
    assert same
json.dump(out, open("gpurun_out/length_sweep.json", "w"), indent=1)
