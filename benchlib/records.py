"""Sub-records of the N = 1 line on the bench shapes themselves: `s300` and `log_probability`."""
import time

import numpy as np

from .common import CLOCK_GHZ, HBM_PEAK_GBPS, SIMDS, algorithmic_bytes, measured_clock_ghz, oracle_model

F64_PEAK_TFLOPS = SIMDS * CLOCK_GHZ * 1e9 * 16 * 2 / 1e12      # 16 fp64 lanes per cycle and SIMD (a wave64 fp64 instruction
                                                               # issues over 4 cycles, profiles/r02_f64_issue_ubench.txt), fused
                                                               # multiply-add = 2 flop: 78.6 TFLOP/s
FORWARD_FMA_PER_CELL = 11      # csrc/forward_rows.h: the linear-domain cell, three states


def forward_record(_lib, locus, batch, bases, off, n_reads, n, args):
    """Model.log_probability (the sum-product twin of the scored path) on the same batch.  `kernel_ms`:
    the sum-product kernels on the RESIDENT reads (advntr_batch_forward_timed, HIP events on the launch stream), priced
    against fp64 multiply-add issue: cells x fused multiply-adds per cell x 2 flop / time vs the vector fp64 peak.
    `ms_per_call`: the one-shot C-ABI call from host buffers (upload, kernel, download), best of three.  On the bench
    sample the values are within 1e-9 relative of the oracle's log-domain forward."""
    dm = locus.model.device_model()
    which = np.zeros(n_reads, np.int32)
    nc = dm.n_columns() if hasattr(dm, "n_columns") else None
    batch.forward()
    batch.sync()
    kernel_ms = batch.forward_timed(max(1, args.steps))
    lp_resident, _ = batch.fetch()
    _lib.forward_batch([dm], bases[:off[64]], off[:65], which[:64])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        lp = _lib.forward_batch([dm], bases, off, which)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.array_equal(lp, lp_resident), "resident and one-shot log_probability differ"
    rec = {"reads": n_reads, "value": n_reads / (kernel_ms * 1e-3), "unit": "reads/s", "kernel_ms": kernel_ms,
           "kernel": "forward_rows_kernel<5, 2>", "timing": "advntr_batch_forward_timed on the resident batch (HIP events)",
           "one_shot": {"value": n_reads / best, "ms_per_call": best * 1e3,
                        "timing": "advntr_forward_batch from host buffers (PCIe inclusive), best of 3"}}
    if nc:
        cells = float(n_reads) * n * nc
        tflops = cells * FORWARD_FMA_PER_CELL * 2 / (kernel_ms * 1e-3) / 1e12
        ghz = measured_clock_ghz()
        rec["roofline"] = {"bound": "valu_f64", "achieved": tflops, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": tflops / F64_PEAK_TFLOPS, "cells": cells, "fma_per_cell": FORWARD_FMA_PER_CELL,
                           "columns": nc, "clock_ghz_measured": ghz,
                           "frac_at_measured_clock": tflops / (F64_PEAK_TFLOPS * ghz / CLOCK_GHZ) if ghz else None,
                           "note": "trellis cells (read length x model columns x reads) x 11 fused multiply-adds x 2 flop over the "
                                   "HIP-event kernel time, against the fp64 vector peak (16 lanes/cycle/SIMD x 1024 SIMDs x 2.4 GHz)"}
    if not args.no_cpu:
        O = oracle_model(locus)
        k = min(200, n_reads)
        worst = 0.0
        for i in range(k):
            want = O.forward(bases[off[i]:off[i + 1]])
            worst = max(worst, abs(lp[i] - want) / max(1.0, abs(want)))
        rec["max_rel_diff_vs_oracle"] = worst
        rec["oracle_sample"] = k
        assert worst <= 1e-9, "GPU/oracle log_probability mismatch on the bench sample"
    return rec


def s300_record(_lib, workloads, flags, args):
    """The label-matching shape of BASELINE's metric: flank 30, 12-bp pattern, 3 copies -> 315 states / 197 emitting /
    1004 edges, the same 100 000 synthetic 150-bp reads recipe (report both shapes with m/P/E stated)."""
    locus = workloads.s300()
    a = locus.model.baked_arrays()
    m, P, E = a["m"], a["silent_start"], len(a["in_src"])
    n, n_reads = 150, args.reads
    reads = workloads.make_reads(np.random.default_rng(20240601), locus, n_reads, n)
    bases, off = _lib.encode_reads(reads)
    batch = _lib.DeviceBatch([locus.model.device_model()], bases, off, np.zeros(n_reads, np.int32), flags=flags)
    batch.run()
    batch.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
    batch.sync()
    dt = (time.perf_counter() - t0) / args.steps
    kernel_ms = batch.run_timed(max(1, args.steps))
    kinfo = batch.kernel_info()
    kernels = [k[:3] for k in kinfo]
    B = algorithmic_bytes(n, m)
    rec = {"states": int(m), "emitting": int(P), "edges": int(E), "reads": n_reads, "read_len": n,
           "value": n_reads / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "kernel_ms": kernel_ms,
           "kernel": max(kernels, key=lambda k: k[1])[0], "bytes_per_read": B,
           "useful_lane_steps": max(kinfo, key=lambda k: k[1])[3],
           "achieved_gbps": B * n_reads / (kernel_ms * 1e-3) / 1e9, "frac": B * n_reads / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "relaxations_per_s": n_reads / dt * (n + 1) * E}
    if not args.no_cpu:
        O = oracle_model(locus)
        k = min(args.cpu_sample, 2000, n_reads)          # (a sub-record: the contract's bounded CPU sample is C1's)
        t0 = time.perf_counter()
        cpu_logp, _ = O.viterbi_many(bases[:off[k]], off[:k + 1])
        rec["cpu_1thread_reads_per_s"] = k / (time.perf_counter() - t0)
        logp, _ = batch.fetch()
        assert np.array_equal(cpu_logp, logp[:k]), "GPU/oracle log-prob mismatch on the S300 sample"
    # the sum-product twin on the same resident batch (the back-to-back sweeps matter most on this narrow model)
    dm = locus.model.device_model()
    nc = dm.n_columns()
    batch.forward()
    batch.sync()
    fwd_ms = batch.forward_timed(max(1, args.steps))
    lp, _ = batch.fetch()
    tflops = float(n_reads) * n * nc * FORWARD_FMA_PER_CELL * 2 / (fwd_ms * 1e-3) / 1e12
    rec["log_probability"] = {"kernel_ms": fwd_ms, "value": n_reads / (fwd_ms * 1e-3), "unit": "reads/s",
                              "kernel": "forward_rows_kernel<5, 2>", "columns": nc,
                              "roofline": {"bound": "valu_f64", "achieved": tflops, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": tflops / F64_PEAK_TFLOPS}}
    if not args.no_cpu:
        worst = 0.0
        for i in range(min(100, n_reads)):
            want = O.forward(bases[off[i]:off[i + 1]])
            worst = max(worst, abs(lp[i] - want) / max(1.0, abs(want)))
        rec["log_probability"]["max_rel_diff_vs_oracle"] = worst
        assert worst <= 1e-9, "GPU/oracle log_probability mismatch on the S300 sample"
    batch.close()
    return rec
