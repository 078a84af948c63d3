#!/usr/bin/env python3
"""How evenly the dynamic tile dequeue ends a launch: per-workgroup clocks of one rank's share of a multi-GPU job.

    scripts/build_variant.sh wgclocks -DADVNTR_WG_CLOCKS          # measurement build of the engine (exp/wgclocks.so)
    python scripts/wg_clocks.py c4 1120                            # an 8-rank share of BASELINE config 5 (1 120 loci)
    python scripts/wg_clocks.py c2 840                             # an 8-rank share of the 6 719-locus set
    python scripts/wg_clocks.py ref150 | s300                      # the bench line's batches

The measurement build stamps, per workgroup of the row-blocked kernels, when it started, when it took its last tile, when it
ended and how many tiles it ran (s_memrealtime, 100 MHz; viterbi_rows.h WG_CLOCKS_*).  Prints one JSON line.  Workgroups that
only became resident when others left (grid > what fits at once) ran no tile and are left out."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ADVNTR_HIP_LIB", os.path.abspath("exp/wgclocks.so"))
import numpy as np                                                                                      # noqa: E402
from advntr_amd import _lib, workloads                                                                  # noqa: E402
from advntr_amd.pomegranate import device_models                                                        # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else (1120 if kind == "c4" else 840)
    if kind in ("ref150", "s300"):                    # the bench line's own batches: 100 000 reads of one model
        locus = workloads.s300() if kind == "s300" else workloads.ref150()
        n_loci = 1
        loci = [locus]
        reads = workloads.make_reads(np.random.default_rng(20240601), locus, 100000, 150)
        which = np.zeros(len(reads), np.int32)
    elif kind == "c2":
        loci, reads, which = workloads.make_c2_parallel(n_loci, seed=20240602)
    else:
        loci, reads, which = workloads.make_c4(n_loci, seed=20240603)
        workloads.build_models(loci)
    dms = device_models([l.model for l in loci])
    bases, off = _lib.encode_reads(reads)
    batch = _lib.DeviceBatch(dms, bases, off, np.asarray(which, dtype=np.int32))
    for _ in range(3):
        batch.run()
    batch.sync()
    ms = batch.run_timed(1)
    lib = ctypes.CDLL(os.environ["ADVNTR_HIP_LIB"])
    if not hasattr(lib, "advntr_debug_wg_clocks"):
        raise SystemExit("not a measurement build: scripts/build_variant.sh wgclocks -DADVNTR_WG_CLOCKS")
    lib.advntr_debug_wg_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((4096, 4), np.uint64)
    n = lib.advntr_debug_wg_clocks(batch._h, buf.ctypes.data, len(buf))
    t = buf[:n].astype(np.float64)
    ran = t[:, 3] > 0
    t0 = t[ran, 0].min()
    end, last, tiles = (t[ran, 1] - t0) / 1e5, (t[ran, 2] - t0) / 1e5, t[ran, 3]                        # 100 MHz -> ms
    idle = end.max() - end
    print(json.dumps({
        "workload": "%s share of %d loci, %d calls" % (kind, n_loci, len(reads)),
        "kernel_ms": round(ms, 3), "workgroups": int(n), "workgroups_that_ran_tiles": int(ran.sum()),
        "tiles_per_workgroup": {"min": int(tiles.min()), "median": float(np.median(tiles)), "max": int(tiles.max())},
        "queue_empty_ms": round(float(last.max()), 3),
        "workgroup_end_ms": {"min": round(float(end.min()), 3), "median": round(float(np.median(end)), 3),
                             "p90": round(float(np.percentile(end, 90)), 3), "max": round(float(end.max()), 3)},
        "last_tile_ms": {"median": round(float(np.median(end - last)), 3), "max": round(float((end - last).max()), 3)},
        "idle_at_end_mean_ms": round(float(idle.mean()), 3),
        "idle_at_end_frac_of_launch": round(float(idle.mean() / end.max()), 4)}))


if __name__ == "__main__":
    main()
