#!/usr/bin/env python3
"""bench.py -- reads/s Viterbi-scored on the BASELINE.json workload, one process per GPU.

A "step" is one pass of the hot path (Viterbi DP + traceback + path summaries) over one resident batch
of synthetic reads.  Workload at every N: config C1 of BASELINE.json / SURVEY 8d -- one REF150 locus
(flank 150, 14-bp pattern, 11 copies: 1413 states / 921 emitting / 4626 edges), 100 000 synthetic 150-bp
reads PER GPU (weak scaling: the read x locus batch shards with no data-path collective; the only RCCL
call is the gather of the 40-B/read result records to rank 0, inside the timed region; the gather of pass i
runs from staging buffers while the kernel of pass i+1 computes, and all of them complete before the clock stops).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, m):
    """B = n [read] + (n+1)*m [1-byte back-pointer per cell] + (n+m) [traceback reads] + 32."""
    return n + (n + 1) * m + (n + m) + 32


class _CudaArray(object):
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr, "version": 2}


def cpu_baseline(locus, bases, off, n_sample):
    """The oracle (C restatement of the reference loop, full tables calloc'd per call) on a bounded sample
    of the same reads, 1 thread -- the reference path is single-threaded (GIL held, hmm.pyx:1958)."""
    from oracle.oracle import OracleModel
    a = locus.model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    sub_off = off[:n_sample + 1]
    t0 = time.perf_counter()
    logp, _ = O.viterbi_many(bases[:sub_off[-1]], sub_off)
    dt = time.perf_counter() - t0
    return n_sample / dt, logp, O


def ru_concordance(O, locus, reads, summ, n_check):
    """RU-count concordance (the second half of BASELINE.json's metric): repeat-unit counts the kernel derived on
    the GPU vs. advntr/hmm_utils.py:155-188 applied to the oracle's Viterbi path, read by read."""
    from oracle import oracle as Or
    names = [s.name for s in locus.model.states]
    same = 0
    for i in range(n_check):
        _, path = O.viterbi(reads[i])
        ru = Or.number_of_repeats([names[j] for j in path][1:-1]) if path else 0
        same += int(ru == int(summ[i][0]))
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=2000)
    ap.add_argument("--generic", action="store_true", help="force the generic-CSR kernel")
    ap.add_argument("--stream", action="store_true", help="experimental stream-packed column kernel")
    ap.add_argument("--workload", default="c1", choices=["c1", "c2", "c3", "c4"],
                    help="c1 (default, the bench line): 1 REF150 locus x --reads per GPU; c2: --loci synthetic loci x ~160 "
                         "calls per GPU (weak); c3: ONE set of --loci loci partitioned over the GPUs by estimated work "
                         "(strong scaling, BASELINE config 3), per-call records gathered to rank 0 over RCCL; "
                         "c4: --loci PacBio loci (flank 100, error 0.3) x 20 trimmed spanning reads per GPU")
    ap.add_argument("--loci", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is exercised even with one rank, so the
    # gather code is tested on a single-GPU box too
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("ADVNTR_BENCH_DIST", "1") == "1")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as entry
    entry.build()
    from advntr_amd import _lib, workloads
    _lib.check(_lib.load().advntr_set_device(local_rank))

    n = 150
    flags = _lib.FLAG_FORCE_GENERIC if args.generic else (_lib.FLAG_STREAM if args.stream else 0)
    total_calls = None
    if args.workload in ("c2", "c3", "c4"):
        if args.workload == "c4":
            loci, reads, which = workloads.make_c4(args.loci, seed=20240603 + rank)
        elif args.workload == "c3":
            # every rank derives the same plan and the same LPT partition without communicating
            from advntr_amd import sharding
            plan = workloads.c2_plan(args.loci, seed=20240602)
            work = [calls * 151 * states for calls, states in plan]
            mine = sharding.partition_loci(work, world)[rank]
            total_calls = int(sum(c for c, _ in plan))
            loci, reads, which = workloads.make_c2_parallel(args.loci, seed=20240602, build=False, only=mine)
        else:
            loci, reads, which = workloads.make_c2_parallel(args.loci, seed=20240602 + rank, build=False)
        t_build = time.perf_counter()
        workloads.build_models(loci)           # native builder, all host cores
        t_build = time.perf_counter() - t_build
        locus = loci[0]
        bases, off = _lib.encode_reads(reads)
        from advntr_amd.pomegranate import device_models
        dms = device_models([l.model for l in loci])          # one allocation + one copy for the whole model set
        args.reads = len(reads)
        batch = _lib.DeviceBatch(dms, bases, off, which, flags=flags)
        dm = dms[0]
        ms = np.array([d.m for d in dms])
        m = int(round(float(np.mean(ms[which]))))
        P, E = locus.model.silent_start, int(np.mean([l.model.n_edges for l in loci]))
        lens = np.diff(off)
        n = int(round(float(lens.mean())))
        # exact sums over the calls (models and read lengths differ per call)
        alg_bytes_total = float(np.sum(lens + (lens + 1) * ms[which] + (lens + ms[which]) + 32))
        relax_total = float(np.sum((lens + 1) * np.array([l.model.n_edges for l in loci], np.int64)[which]))
        args.no_cpu = True
    else:
        locus = workloads.ref150()
        a = locus.model.baked_arrays()
        m, P, E = a["m"], a["silent_start"], len(a["in_src"])
        reads = workloads.make_reads(np.random.default_rng(20240601 + rank), locus, args.reads, n)
        bases, off = _lib.encode_reads(reads)
        dm = locus.model.device_model()
        batch = _lib.DeviceBatch([dm], bases, off, np.zeros(args.reads, np.int32), flags=flags)
    kernel = "viterbi_columns" if (dm.has_column_program() and not args.generic) else "viterbi_generic"
    # reads of up to 155 bases go to the row-blocked kernels (engine.hip: use_rows); the name is what
    # rocprofv3 --kernel-trace shows for the dominant kernel of this command
    if (kernel == "viterbi_columns" and not args.stream and n <= 155
            and "ADVNTR_ROWS_MIN" not in os.environ and "ADVNTR_ROWS_MIN_READ" not in os.environ):
        kernel = "viterbi_rows"

    gathered = None
    if use_dist:
        p_logp, p_sum = batch.result_ptrs()
        e_logp = torch.as_tensor(_CudaArray(p_logp, (args.reads,), "<f8"), device="cuda")
        e_sum = torch.as_tensor(_CudaArray(p_sum, (args.reads, 8), "<i4"), device="cuda")
        # ranks may hold different numbers of calls (c3): gather fixed-size buffers padded to the largest share
        cap = torch.tensor([args.reads], dtype=torch.int64, device="cuda")
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        cap = int(cap.item())
        # staging buffers: the gather of step i runs from them while the kernel of step i+1 refills the engine's
        # result arrays (the only collective of the path overlaps the next pass instead of serialising with it)
        t_logp = torch.zeros(cap, dtype=torch.float64, device="cuda")
        t_sum = torch.zeros((cap, 8), dtype=torch.int32, device="cuda")
        if rank == 0:
            gathered = ([torch.empty_like(t_logp) for _ in range(world)],
                        [torch.empty_like(t_sum) for _ in range(world)])
    pending = []

    def step():
        batch.run()
        if use_dist:
            batch.sync()                                   # this pass's records are final
            for w in pending:                              # the previous gather has had a whole pass to finish
                w.wait()
            del pending[:]
            t_logp[:args.reads].copy_(e_logp)
            t_sum[:args.reads].copy_(e_sum)
            torch.cuda.current_stream().synchronize()      # engine buffers may be overwritten by the next pass
            pending.append(dist.gather(t_logp, gathered[0] if rank == 0 else None, dst=0, async_op=True))
            pending.append(dist.gather(t_sum, gathered[1] if rank == 0 else None, dst=0, async_op=True))

    def drain():
        for w in pending:
            w.wait()
        del pending[:]

    for _ in range(args.warmup):
        step()
    drain()
    batch.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                # every gather of the timed steps completes inside the region
    batch.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside
    # the bench); None when the profile does not describe this workload/kernel
    traffic, valu_insts = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        if args.workload == "c1" and args.reads == 100000 and kernel == "viterbi_rows":
            traffic = pmc["viterbi_rows"]["hbm_bytes_per_launch_fetch_x2"] / 1e9
            valu_insts = pmc["viterbi_rows"].get("valu_insts_per_launch")
        elif args.workload == "c1" and args.reads == 100000 and kernel == "viterbi_columns":
            traffic = pmc["hbm_bytes_per_launch_fetch_x2"] / 1e9
            valu_insts = pmc.get("valu_insts_per_launch")
    except Exception:
        traffic = None

    # kernel-only duration, HIP events on the engine's launch stream
    kernel_ms = batch.run_timed(max(1, min(args.steps, 3)))
    logp, summ = batch.fetch()
    if use_dist and rank == 0:
        # the gathered copy of rank 0's own records must equal what the engine holds
        assert np.array_equal(gathered[0][0][:args.reads].cpu().numpy(), logp), "RCCL gather returned different log-probs"
        assert np.array_equal(gathered[1][0][:args.reads].cpu().numpy(), summ), "RCCL gather returned different summaries"

    if rank == 0:
        total_reads = total_calls if total_calls is not None else args.reads * world
        value = total_reads * args.steps / elapsed
        B = algorithmic_bytes(n, m)
        if args.workload == "c1":
            alg_bytes_total, relax_total = float(B) * args.reads, float(args.reads) * (n + 1) * E
        achieved = alg_bytes_total / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": ("reads/sec Viterbi-scored (150 bp reads, REF150 profile HMM: 1413 states / 4626 edges)"
                       if args.workload == "c1" else
                       "calls/sec Viterbi-scored (150 bp reads, %d per-locus profile HMMs partitioned over %d GPUs)" % (args.loci, world)
                       if args.workload == "c3" else
                       "calls/sec Viterbi-scored (PacBio: trimmed spanning reads, mean %d bases, %d per-locus profile HMMs, "
                       "mean %d states)" % (n, args.loci, m) if args.workload == "c4" else
                       "calls/sec Viterbi-scored (150 bp reads, %d per-locus profile HMMs, mean %d states)" % (args.loci, m)),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.workload == "c3" else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C1: 1 VNTR locus REF150 (flank 150, 14-bp pattern, 11 copies) x 100k synthetic "
                                    "150-bp reads per GPU, seed 20240601") if args.workload == "c1" else
                                   ("C4: %d synthetic PacBio loci (pattern 10-60 bp, VNTR 100-1000 bp, flank 100, error rate 0.3) x "
                                    "20 trimmed spanning reads at +-20 %% of the reference copy number, 12 %% indel/substitution "
                                    "noise, seed 20240603; host model build %.2f s" % (args.loci, t_build))
                                   if args.workload == "c4" else
                                   ("%s: %d synthetic loci (pattern 6-100 bp, 2-20 repeat units, flank 150) x "
                                    "~Poisson(80) mapped + 2*Poisson(40) unmapped-strand calls, seed 20240602%s; "
                                    "host model build %.2f s (native builder, %d threads)"
                                    % (args.workload.upper(), args.loci,
                                       " (whole loci assigned to ranks by LPT on calls x states)" if args.workload == "c3" else "",
                                       t_build, os.cpu_count() or 1)),
                       "states": int(m), "emitting": int(P), "edges": int(E), "reads_per_gpu": args.reads,
                       "read_len": n, "kernel": kernel, "outputs": "logp + RU count + 6 path summaries per read",
                       "relaxations_per_s": value * relax_total / max(args.reads, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "GB per launch (profiles/r01_pmc_summary.json: WRITE_SIZE + 2 x FETCH_SIZE)",
                         "algorithmic_gb_per_launch": alg_bytes_total / 1e9,
                         "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_read": B,
                         "note": "algorithmic bytes (SURVEY 8d) / HIP-event kernel time; the max-plus recurrence is "
                                 "fp64-VALU/LDS-issue bound long before HBM (see DESIGN.md)"},
        }
        if valu_insts:
            # what actually bounds the kernel: every wave64 VALU instruction holds its SIMD for 4 cycles (MI355X: 256 CUs x
            # 4 SIMDs at 2.4 GHz); SQ_INSTS_VALU per launch from the committed PMC pass of this same command
            bound_ms = valu_insts * 4 / (1024 * 2.4e9) * 1e3
            out["roofline"]["valu_issue"] = {"valu_insts_per_launch": valu_insts, "cycles_per_inst": 4, "simds": 1024,
                                             "clock_ghz": 2.4, "issue_bound_ms": bound_ms, "kernel_ms": kernel_ms,
                                             "frac": bound_ms / kernel_ms}
        if not args.no_cpu:
            cps, cpu_logp, O = cpu_baseline(locus, bases, off, min(args.cpu_sample, args.reads))
            assert np.array_equal(cpu_logp, logp[:len(cpu_logp)]), "GPU/oracle log-prob mismatch on the bench sample"
            n_ru = min(500, len(cpu_logp))
            same = ru_concordance(O, locus, reads, summ, n_ru)
            out["ru_concordance"] = {"reads": n_ru, "identical_ru_counts": same, "fraction": same / n_ru,
                                     "note": "GPU path summaries vs hmm_utils.get_number_of_repeats_in_vpath on the oracle path"}
            out["cpu_baseline"] = {"value": cps, "unit": "reads/s", "cores": 1, "kind": "port",
                                   "sample": "first %d reads of rank 0's batch, oracle/viterbi_oracle.c, 1 thread; "
                                             "GPU logp bit-equal on the sample" % len(cpu_logp)}
            out["config"]["speedup_vs_cpu_1thread"] = value / cps
            # the same restatement on every host core (the reference has no such mode; stated for scale only)
            cores = os.cpu_count() or 1
            n_mt = min(args.reads, max(2000, 150 * cores))
            t0 = time.perf_counter()
            mt_logp = O.viterbi_many_threads(bases[:off[n_mt]], off[:n_mt + 1], cores)
            dt = time.perf_counter() - t0
            assert np.array_equal(mt_logp, logp[:n_mt]), "GPU/oracle log-prob mismatch on the all-cores sample"
            out["cpu_baseline_all_cores"] = {"value": n_mt / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                                             "sample": "first %d reads, oracle/viterbi_oracle.c on %d pthreads; GPU logp "
                                                       "bit-equal on the sample" % (n_mt, cores)}
        print(json.dumps(out), flush=True)
    batch.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
