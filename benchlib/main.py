"""The bench line itself (contract in the task statement; bench.py is the entry point)."""
import json
import os
import sys
import time

import numpy as np

from .cli import emulation_argv, parse_args, spawn_ranks
from .hostinfo import HostRegion, cpu_quota_cores
from .common import (CLOCK_GHZ, HBM_PEAK_GBPS, SIMDS, algorithmic_bytes, cpu_baseline, cpu_model_name, load_json,
                     measured_clock_ghz, pmc_section, ru_concordance)
from .illumina import illumina_pipeline_input, illumina_pipeline_record, target_configuration_records
from .passes import Passes, passes_of, two_in_flight_ms
from .records import forward_record, s300_record
from .rehearsal import scale_rehearsal
from .upstream import (c4_record, flank_align_record, pacbio_end_to_end_record, prefilter_record, upstream_inputs)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "RANK" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, argv)
    if "RANK" not in os.environ and args.emulate_ranks > 1 and args.processes:
        # N rank processes sharing this box's GPU(s) through the host communicator: the host side of an N-rank job on one box
        return spawn_ranks(args, emulation_argv(argv, args.emulate_ranks), n=args.emulate_ranks,
                           extra_env={"ADVNTR_DIST_BACKEND": "host", "ADVNTR_EMULATED_RANKS": str(args.emulate_ranks)})

    # stdout carries exactly ONE line, the JSON record: anything a library prints while the bench runs (RCCL's version
    # banner, HIP warnings) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        os.write(json_fd, (json.dumps(record) + "\n").encode())

    from advntr_amd import comm as comm_mod
    rank, local_rank, world = comm_mod.env_world()
    quota_total = cpu_quota_cores()          # CPUs of the whole job's control group (before a rank takes its 1 / world of them)
    if args.fault:
        kind, _, who = args.fault.partition(":")
        if int(who or -1) == rank:
            if kind == "exit":
                return 3
            if kind == "hang":
                time.sleep(10 ** 6)
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks; using %d\n" % (args.gpus, world, world))
    workload = args.workload or ("c1" if world == 1 else "c3")
    n_loci = args.loci if args.loci is not None else (8960 if workload == "c4" else 6719)

    import __graft_entry__ as entry
    entry.build()
    from advntr_amd import _lib, sharding, workloads

    # The GPU and RCCL are touched only AFTER the synthetic reads exist: their generators fork a process pool, and a child
    # forked from a process that holds an initialised HIP runtime and an RCCL communicator (proxy threads, shared-memory
    # segments) must not be left to tear those down at its exit.
    def join_job():
        if world > 1 or "RANK" in os.environ:
            # (under a launcher the RCCL path is exercised even with one rank, so the gather code runs on a 1-GPU box too)
            return comm_mod.init_from_env(backend="host" if args.dry_run else None, set_device=not args.dry_run)
        if not args.dry_run:
            _lib.check(_lib.load().advntr_set_device(local_rank))
        return None

    comm = join_job() if args.dry_run else None

    # ---------------------------------------------------------------- workload
    n = 150
    flags = (_lib.FLAG_FORCE_GENERIC if args.generic else _lib.FLAG_STREAM if args.stream else
             _lib.FLAG_ANTIDIAGONAL if args.antidiagonal else 0)
    total_calls, t_build, plan_info = None, 0.0, {}
    # passes queued at a time (class Passes): the strong-scaling lines alternate between two copies of a rank's device batch
    in_flight = args.in_flight if args.in_flight else (2 if workload in ("c3", "c4") else 1)
    # (process-pool workers of the synthetic-read generators: out of the CPUs this process GROUP may use -- the control group's
    # quota, 16 cores on the GPU boxes of this pool, not the 256 hardware threads -- divided among the ranks of the host)
    host_workers = max(1, min(32, quota_total // world - 1))
    if workload in ("c3", "c4"):
        # every rank derives the same plan and the same LPT partition without communicating
        if workload == "c3":
            plan = workloads.c2_plan(n_loci, seed=20240602)
            work = [calls * 151 * states for calls, states in plan]
        else:
            # (PacBio: what is known of a locus before its reads are extracted -- pattern and reference VNTR length -- prices it)
            plan = workloads.c4_plan(n_loci, seed=20240603)
            work = [calls * (length + 1) * states for calls, length, states in plan]
        capacity = [args.root_capacity] + [1.0] * (world - 1) if world > 1 else None
        parts = sharding.partition_loci(work, world, capacity)
        mine = parts[rank]
        total_calls = int(sum(p[0] for p in plan))
        loads = [float(sum(work[int(k)] for k in p)) for p in parts]
        plan_info = {"loci_per_rank": [int(len(p)) for p in parts],
                     "calls_per_rank": [int(sum(plan[int(k)][0] for k in p)) for p in parts],
                     "load_imbalance_max_over_mean": max(loads) / (sum(loads) / world),
                     "root_capacity": capacity[0] if capacity else None, "root_load_over_mean": loads[0] / (sum(loads) / world),
                     "per_locus_work_max_over_min": float(max(work)) / max(float(min(work)), 1.0)}
    if args.dry_run:
        counts = comm.allgather_i64(plan_info["calls_per_rank"][rank] if plan_info else args.reads) if comm else [args.reads]
        per_rank = None
        with HostRegion() as h:
            if comm:
                comm.barrier()
                got = comm.allreduce_max(float(rank))
                assert got == float(world - 1), got
        if comm:
            # the per-rank records of a real line, with the fields a dry run can fill (no GPU work: no timings)
            mine_rec = json.dumps({"rank": rank, "calls": int(counts[rank]), "loop_ms": None, "kernel_ms": None, "gather_ms": None,
                                   "cells": None, "host": h.record()}).encode()
            parts_json = comm.gather_bytes(mine_rec, 0)
            if rank == 0:
                per_rank = [json.loads(p) for p in parts_json]
        if rank == 0:
            strong = workload in ("c3", "c4")
            line = {"metric": "dry run: plan and rendezvous only", "value": None, "unit": "reads/s", "n_gpus": world,
                    "steps": 0, "warmup": 0, "dry_run": True, "scaling": "strong" if strong else "weak", "host": h.record(),
                    "config": dict({"workload": workload, "loci": n_loci, "calls_seen_by_ranks": counts,
                                    "comm": comm.backend if comm else None, "per_rank": per_rank}, **plan_info)}
            if strong:
                line["same_workload_n1"] = {"value": None, "where": "dry run",
                                            "command": "python bench.py --workload %s --gpus 1 --loci %d --steps %d --warmup %d --in-flight %d"
                                                       % (workload, n_loci, args.steps, args.warmup, in_flight)}
                line["efficiency_measured"] = None
            emit(order_line(line))
        if comm:
            comm.close()
        return 0
    whole_input = None
    if workload in ("c2", "c3", "c4"):
        if workload in ("c3", "c4") and world > 1 and rank == 0 and not args.no_n1:
            # the line's own baseline: rank 0 will score the WHOLE set alone after the timed region (same_workload_n1), so its reads
            # are made here, before the GPU is touched like every other synthetic input
            t_whole = time.perf_counter()
            # (its peers are done with their shares after an Nth of this: rank 0's pool takes half of the job's CPUs)
            whole_workers = max(host_workers, min(32, quota_total // 2))
            whole_input = (workloads.make_c4(n_loci, seed=20240603, workers=whole_workers) if workload == "c4" else
                           workloads.make_c2_parallel(n_loci, seed=20240602, build=False, workers=whole_workers))
            whole_input = whole_input + (time.perf_counter() - t_whole,)
        if workload == "c4":
            loci, reads, which = workloads.make_c4(n_loci, seed=20240603, workers=host_workers, only=mine)
        elif workload == "c3":
            loci, reads, which = workloads.make_c2_parallel(n_loci, seed=20240602, build=False, only=mine, workers=host_workers)
        else:
            loci, reads, which = workloads.make_c2_parallel(n_loci, seed=20240602 + rank, build=False, workers=host_workers)
        t_build = time.perf_counter()
        workloads.build_models(loci)           # native builder, host threads
        t_build = time.perf_counter() - t_build
        locus = loci[0]
        bases, off = _lib.encode_reads(reads)
        comm = join_job()
        _lib.require_gpu()
        from advntr_amd.pomegranate import device_models
        dms = device_models([l.model for l in loci])          # one allocation + one copy for the whole model set
        n_reads = len(reads)
        make_batch = lambda extra=0: _lib.DeviceBatch(dms, bases, off, which, flags=flags | extra)          # noqa: E731
        passes = Passes(make_batch, in_flight)
        batch = passes.batches[0]
        ms = np.array([d.m for d in dms])
        m = int(round(float(np.mean(ms[which]))))
        edges_per_locus = np.array([l.model.n_edges for l in loci], np.int64)
        P, E = locus.model.silent_start, int(np.mean(edges_per_locus))
        lens = np.diff(off)
        n = int(round(float(lens.mean())))
        # exact sums over the calls (models and read lengths differ per call)
        alg_bytes_total = float(np.sum(lens + (lens + 1) * ms[which] + (lens + ms[which]) + 32))
        relax_total = float(np.sum((lens + 1) * edges_per_locus[which]))
    else:
        c2_input = upstream_input = pipeline_input = None
        if world == 1 and workload == "c1" and not args.no_c2 and not args.no_s300:
            # the target configuration of the north star rides on the C1 line as sub-records `c2` / `end_to_end`; its
            # synthetic reads come out of a process pool, which must have gone before the GPU is touched (see above)
            t_gen = time.perf_counter()
            c2_input = workloads.make_c2_parallel(args.c2_loci, seed=20240602, build=False, workers=host_workers,
                                                  return_counts=True) + (time.perf_counter() - t_gen,)
            # (a million read strings: out of the garbage collector's sight, or its passes land in the timed loops)
            upstream_input = None if args.no_upstream else upstream_inputs(workloads, host_workers, args)
            pipeline_input = illumina_pipeline_input(workloads, c2_input, args) if args.pipeline_reads > 0 else None
            import gc
            gc.collect()
            gc.freeze()
        locus = workloads.s300() if workload == "s300" else workloads.ref150()
        a = locus.model.baked_arrays()
        m, P, E = a["m"], a["silent_start"], len(a["in_src"])
        n_reads = args.reads
        reads = workloads.make_reads(np.random.default_rng(20240601 + rank), locus, n_reads, n)
        bases, off = _lib.encode_reads(reads)
        comm = join_job()
        _lib.require_gpu()
        c1_model = locus.model.device_model()
        make_batch = lambda extra=0: _lib.DeviceBatch([c1_model], bases, off, np.zeros(n_reads, np.int32), flags=flags | extra)    # noqa: E731
        passes = Passes(make_batch, in_flight)
        batch = passes.batches[0]
        alg_bytes_total = float(algorithmic_bytes(n, m)) * n_reads
        relax_total = float(n_reads) * (n + 1) * E
    kinfo = batch.kernel_info()                 # what the engine launches for this batch (advntr_batch_info)
    kernels = [k[:3] for k in kinfo]
    kernel = max(kernels, key=lambda k: k[1])[0] if kernels else "none"

    # ---------------------------------------------------------------- timed region
    counts = comm.allgather_i64(n_reads) if comm else [n_reads]
    use_gather = comm is not None and comm.backend == "rccl"
    state = {"pending": False}

    # (only when RCCL could not be set up on a multi-GPU node and comm.py fell back: the gather then goes through host
    # memory inside the timed region, unoverlapped -- slower, but the line stays a measurement of the whole path)
    host_gather = comm is not None and comm.backend == "host" and world > 1 and comm.fallback_reason is not None

    def step():
        b = passes.run()                                    # (two in flight: the copy whose previous pass is the older one)
        if use_gather:
            if state["pending"]:                            # the previous gather has had a whole pass to finish
                comm.gather_results_finish(fetch=False)
                state["gather_ms"] = comm.last_gather_ms()
            # queued behind this pass on its copy's stream; the next pass overlaps it (the copy's own next pass leaves the
            # slots the gather asks for: with two copies that is the pass that overlaps the NEXT gather -- every pass but the
            # first two leaves them)
            comm.gather_results_start(b, counts, root=0)
            state["pending"] = True
        elif host_gather:
            state["host"] = comm.gather_results(b, counts, root=0)

    def drain(fetch=False):
        out = (None, None)
        if state["pending"]:
            out = comm.gather_results_finish(fetch=fetch)
            state["pending"] = False
            state["gather_ms"] = comm.last_gather_ms()
        return out

    for _ in range(args.warmup):
        step()
    drain()
    passes.sync()
    if comm:
        comm.barrier()
    host_region = HostRegion().__enter__()                  # the control group's throttle counters around the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                # every gather of the timed steps completes inside the region
    passes.sync()                                          # ... and so does every pass, on either copy
    loop_mine = time.perf_counter() - t0                    # this rank's own passes and gathers, before it waits for its peers
    if comm:
        comm.barrier()
    elapsed_mine = time.perf_counter() - t0
    host_region.__exit__(None, None, None)
    host_rec = host_region.record()
    elapsed = comm.allreduce_max(elapsed_mine) if comm else elapsed_mine

    # one more pass outside the timed region whose gathered records rank 0 checks against what its engine holds
    gathered = (None, None)
    if use_gather:
        step()
        gathered = drain(fetch=True)
    elif host_gather:
        step()
        gathered = state["host"]
    # kernel-only duration, HIP events on the engine's launch stream
    # (as many passes as the timed region had: a burst of two or three passes after a pause runs 1-3 % faster than the
    # sustained loop on this part, and the roofline is about the sustained rate)
    kernel_ms = batch.run_timed(max(1, args.steps))
    logp, summ = batch.fetch()
    per_rank = None
    if comm:
        rec = json.dumps({"rank": rank, "calls": n_reads, "loop_ms": loop_mine / max(args.steps, 1) * 1e3,
                          "loop_ms_per_step": elapsed_mine / max(args.steps, 1) * 1e3,
                          "kernel_ms": kernel_ms, "model_build_s": t_build, "host": host_rec,
                          # the share's ACTUAL work (the plan prices a locus before its reads exist)
                          "relaxations": relax_total, "cells": float(np.sum((np.diff(off) + 1) * ms[which])) if workload in ("c2", "c3", "c4") else None,
                          # the last gather of the timed region on the communicator's stream (HIP events): the transfer alone
                          # when it ran beside the next pass, about a pass when it had to wait for that pass's kernels
                          "gather_ms": state.get("gather_ms")}).encode()
        parts_json = comm.gather_bytes(rec, 0)
        if rank == 0:
            per_rank = [json.loads(p) for p in parts_json]
    if (use_gather or host_gather) and rank == 0:
        at = 0                                              # rank 0's own records sit first
        assert np.array_equal(gathered[0][at:at + n_reads], logp), "RCCL gather returned different log-probabilities"
        assert np.array_equal(gathered[1][at:at + n_reads], summ), "RCCL gather returned different summaries"
        assert len(gathered[0]) == sum(counts)
        if workload in ("c3", "c4"):
            assert sum(counts) == total_calls, (sum(counts), total_calls)

    if args.dump_records:
        if workload in ("c3", "c4"):                        # global call id = position in the whole set's locus order
            first = np.concatenate([[0], np.cumsum([p[0] for p in plan])])
            ids = np.concatenate([np.arange(first[int(k)], first[int(k) + 1]) for k in mine]) if len(mine) else np.zeros(0, np.int64)
        else:
            ids = np.arange(n_reads, dtype=np.int64) + rank * n_reads
        res = sharding.gather_records(comm, ids, logp, summ) if comm else (ids, logp, summ)
        if rank == 0:
            np.savez(args.dump_records, ids=res[0], logp=res[1], summary=res[2])

    # ---------------------------------------------------------------- the line's own N = 1 baseline
    # An N-rank strong-scaling line is only worth its same-workload one-rank rate: rank 0 scores the WHOLE set alone, here, after
    # the timed region, in the same process on the same GPU, with the line's own steps / warm-up / passes in flight; its peers wait
    # at the barrier.  (`--gpus 1` without --workload runs C1, a different workload: the driver's curve across N mixes the two.)
    n1 = None
    if whole_input is not None:
        passes.close()                                      # (rank 0 has its GPU to itself for this)
        n1 = same_workload_n1(_lib, workloads, whole_input, flags, args, in_flight, workload)
    if comm and world > 1 and workload in ("c3", "c4") and not args.no_n1:
        comm.barrier()

    rc = 0
    if rank == 0:
        total_reads = total_calls if total_calls is not None else n_reads * world
        value = total_reads * args.steps / elapsed
        B = algorithmic_bytes(n, m)
        achieved = alg_bytes_total / (kernel_ms * 1e-3) / 1e9
        pmc = pmc_section(workload, n_reads, kernel) or {}
        traffic = pmc.get("hbm_bytes_per_launch_fetch_x2")
        traffic = traffic / 1e9 if traffic else None
        valu_insts = pmc.get("valu_insts_per_launch")
        if workload == "s300":
            metric = "reads/sec Viterbi-scored (150 bp reads, S300 profile HMM: %d states / %d edges)" % (m, E)
            wl = "S300: 1 VNTR locus (flank 30, 12-bp pattern, 3 copies) x 100k synthetic 150-bp reads per GPU, seed 20240601"
        elif workload == "c1":
            metric = "reads/sec Viterbi-scored (150 bp reads, REF150 profile HMM: 1413 states / 4626 edges)"
            wl = ("C1: 1 VNTR locus REF150 (flank 150, 14-bp pattern, 11 copies) x 100k synthetic 150-bp reads per GPU, "
                  "seed 20240601")
        elif workload == "c4":
            metric = ("calls/sec Viterbi-scored (PacBio: trimmed spanning reads, mean %d bases, %d per-locus profile HMMs%s, "
                      "mean %d states)" % (n, n_loci, " partitioned over %d GPUs" % world, m))
            wl = ("C4: %d synthetic PacBio loci (pattern 10-60 bp, VNTR 100-1000 bp, flank 100, error rate 0.3) x 20 trimmed "
                  "spanning reads at +-20 %% of the reference copy number, 12 %% indel/substitution noise, seed 20240603 "
                  "(ONE set; whole loci assigned to ranks by LPT on calls x (reference VNTR length + 201) x expected states; "
                  "%d calls in total); host model build %.2f s" % (n_loci, total_calls, t_build))
        else:
            metric = ("calls/sec Viterbi-scored (150 bp reads, %d per-locus profile HMMs partitioned over %d GPUs)" % (n_loci, world)
                      if workload == "c3" else
                      "calls/sec Viterbi-scored (150 bp reads, %d per-locus profile HMMs, mean %d states)" % (n_loci, m))
            wl = ("%s: %d synthetic loci (pattern 6-100 bp, 2-20 repeat units, flank 150) x ~Poisson(80) mapped + "
                  "2*Poisson(40) unmapped-strand calls, seed 20240602%s; host model build %.2f s (native builder)"
                  % (workload.upper(), n_loci,
                     " (whole loci assigned to ranks by LPT on calls x states; %d calls in total)" % total_calls
                     if workload == "c3" else "", t_build))
        out = {
            "metric": metric, "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if workload in ("c3", "c4") else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": dict({"workload": wl, "states": int(m), "emitting": int(P), "edges": int(E),
                            "calls_this_rank": int(n_reads), "read_len": n, "kernel": kernel,
                            "kernels": [{"name": k, "reads": r, "tiles": t, "useful_lane_steps": u} for k, r, t, u in kinfo],
                            "outputs": "logp + RU count + 6 path summaries per read",
                            "passes_in_flight": in_flight,
                            "relaxations_per_s": value * relax_total / max(n_reads, 1)}, **plan_info),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "GB per launch (WRITE_SIZE + 2 x FETCH_SIZE of the committed PMC passes)",
                         "traffic_source": pmc.get("file"),
                         "algorithmic_gb_per_launch": alg_bytes_total / 1e9,
                         "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_read": B,
                         "note": "tier rule: algorithmic bytes (SURVEY 8d) / HIP-event kernel time against HBM; the "
                                 "roof that actually binds this max-plus recurrence is fp64 VALU issue -> bound_actual"},
        }
        if workload in ("c3", "c4"):
            # which one-rank run this line compares with -- and, for N > 1, that run's rate measured inside this very job
            n1_cmd = "python bench.py --workload %s --gpus 1 --loci %d --steps %d --warmup %d --in-flight %d" % (
                workload, n_loci, args.steps, args.warmup, in_flight)
            if world == 1:
                out["same_workload_n1"] = {"value": value, "unit": "reads/s", "ms_per_step": elapsed / args.steps * 1e3,
                                           "kernel_ms": kernel_ms, "calls": int(n_reads), "passes_in_flight": in_flight,
                                           "command": n1_cmd, "where": "this line"}
                out["efficiency_measured"] = 1.0
            elif n1 is not None:
                out["same_workload_n1"] = dict(n1, command=n1_cmd)
                out["efficiency_measured"] = value / (world * n1["value"])
            else:
                out["same_workload_n1"] = {"value": None, "command": n1_cmd, "where": "not measured (--no-n1)"}
                out["efficiency_measured"] = None
            out["efficiency_note"] = ("value / (n_gpus x same_workload_n1.value): the same locus set, the same metric string family, "
                                      "steps, warm-up and passes in flight; the N = 1 figure has no gather, an N-rank one has it "
                                      "inside its timed region")
        out["host"] = host_rec
        if os.environ.get("ADVNTR_EMULATED_RANKS"):
            out["emulated_ranks"] = {"ranks": world, "how": "%d rank processes share this box's GPU(s) through the host communicator "
                                                            "(--emulate-ranks N --processes)" % world,
                                     "what_it_shows": "the HOST side of an N-rank job against the box's one CPU quota: per-rank "
                                                      "host.nr_throttled_delta / throttled_usec_delta around the timed region; the "
                                                      "rates are those of ranks sharing a GPU, not a scaling measurement"}
        if comm:
            # what carried the gather, at the top level of the line: "rccl", or "host" when ADVNTR_COMM_FALLBACK=1 let the
            # ranks drop to the file rendezvous (without that variable a job whose RCCL cannot be set up ends with an error)
            out["comm"] = comm.backend
            out["rccl"] = comm.backend == "rccl"
            out["config"]["comm"] = comm.backend if comm.fallback_reason is None else "host (RCCL unavailable: %s)" % comm.fallback_reason
            out["config"]["world_size_seen_by_comm"] = comm.world
            out["config"]["per_rank"] = per_rank
        if valu_insts:
            # what actually bounds the kernel: every wave64 VALU instruction holds its SIMD for >= 4 cycles (fp64: 16
            # lanes per cycle); SQ_INSTS_VALU per launch from the committed PMC pass of this same command
            bound_ms = valu_insts * 4 / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
            out["roofline"]["bound_actual"] = {"bound": "valu_f64", "valu_insts_per_launch": valu_insts, "cycles_per_inst": 4,
                                               "simds": SIMDS, "clock_ghz": CLOCK_GHZ, "issue_bound_ms": bound_ms,
                                               "kernel_ms": kernel_ms, "frac": bound_ms / kernel_ms,
                                               "source": pmc.get("file"), "stale": pmc.get("stale")}
            ghz = measured_clock_ghz()
            if ghz:
                # at the clock the chip really holds under this kernel, and at the rate it really issues 64-bit-encoded vector
                # instructions (fp64 arithmetic, DPP, three-operand forms: ~4.5 cycles each at 3-4 wavefronts per SIMD,
                # profiles/r01_valu_ubench.txt, r02_f64_issue_ubench.txt) -- nominal: 4 cycles at 2.4 GHz
                b = out["roofline"]["bound_actual"]
                b["clock_ghz_measured"] = ghz
                b["clock_source"] = "profiles/r04_clock_summary.json (GRBM_GUI_ACTIVE / dispatch duration)"
                b["issue_bound_ms_at_measured_clock"] = valu_insts * 4 / (SIMDS * ghz * 1e9) * 1e3
                b["frac_at_measured_clock"] = b["issue_bound_ms_at_measured_clock"] / kernel_ms
                b["frac_at_measured_clock_and_4p5_cycles_per_inst"] = b["issue_bound_ms_at_measured_clock"] * 4.5 / 4 / kernel_ms
        if workload in ("c1", "s300") and world == 1 and in_flight == 1 and not args.no_s300:
            # (not with --no-s300: the profiler passes of scripts/profile_round5.sh trace one launch at a time only)
            # the same batch with two passes queued at a time (class Passes; what `--in-flight 2` makes the line itself): the
            # next pass starts while the last workgroups of this one drain.  Reported beside the line, not as its value: the
            # line's kernel time, roofline and profiles are those of one launch at a time
            ms2 = two_in_flight_ms(batch, make_batch, max(1, args.steps))
            out["two_passes_in_flight"] = {"ms_per_step": ms2, "value": total_reads / (ms2 * 1e-3), "unit": "reads/s",
                                           "note": "consecutive passes alternate between two copies of the device batch "
                                                   "(own scratch, results and stream); every pass scores every read"}
        if workload == "c1" and not args.no_s300:
            out["s300"] = s300_record(_lib, workloads, flags, args)
            out["log_probability"] = forward_record(_lib, locus, batch, bases, off, n_reads, n, args)
            if c2_input is not None:
                out["end_to_end"], out["c2"] = target_configuration_records(_lib, workloads, c2_input, flags, args)
                out["scale_rehearsal"] = out["c2"].pop("scale_rehearsal")
                if pipeline_input is not None:
                    out["illumina_pipeline"] = illumina_pipeline_record(_lib, pipeline_input, args)
            if upstream_input is not None:
                out["c4"] = c4_record(_lib, workloads, upstream_input, flags, args)
                out["c4_scale_rehearsal"] = out["c4"].pop("scale_rehearsal")
                out["pacbio_end_to_end"] = pacbio_end_to_end_record(_lib, upstream_input, args)
                out["flank_align"] = flank_align_record(_lib, upstream_input, args)
                out["prefilter"] = prefilter_record(_lib, upstream_input, args)
        if args.emulate_ranks > 1 and world == 1 and workload in ("c2", "c3", "c4"):
            whole = {"calls": int(n_reads), "kernel_ms": kernel_ms,
                     "loop_ms": Passes.ms_per_pass(passes_of(batch), max(1, args.steps)),
                     "loop_ms_two_passes_in_flight": (elapsed / args.steps * 1e3 if in_flight == 2 else
                                                      two_in_flight_ms(batch, make_batch, max(1, args.steps)))}
            passes.close()                                  # (a rank has its GPU to itself: see c4_record)
            out["scale_rehearsal"] = scale_rehearsal(_lib, sharding, loci, dms, bases, off, which, args.emulate_ranks,
                                                     whole, flags, max(1, args.steps),
                                                     planned_work=work if workload == "c4" else None,
                                                     root_capacity=args.root_capacity)
        if workload == "c1" and world == 1 and not args.no_cpu:
            cps, cpu_logp, O = cpu_baseline(locus, bases, off, min(args.cpu_sample, n_reads))
            assert np.array_equal(cpu_logp, logp[:len(cpu_logp)]), "GPU/oracle log-prob mismatch on the bench sample"
            n_ru = min(500, len(cpu_logp))
            same = ru_concordance(O, locus, reads, summ, n_ru)
            out["ru_concordance"] = {"reads": n_ru, "identical_ru_counts": same, "fraction": same / n_ru,
                                     "note": "GPU path summaries vs hmm_utils.get_number_of_repeats_in_vpath on the oracle path"}
            cal = load_json("profiles", "cpu_calibration.json") or {}
            ratio = cal.get("oracle_over_pomegranate")
            out["cpu_baseline"] = {"value": cps, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": cpu_model_name(),
                                   "host_threads_available": os.cpu_count(),
                                   "sample": "first %d reads of rank 0's batch, oracle/viterbi_oracle.c, 1 thread; "
                                             "GPU logp bit-equal on the sample" % len(cpu_logp),
                                   "pomegranate_equivalent": (cps / ratio) if ratio else None,
                                   "calibration": ("oracle / vendored pomegranate = %.2f on %s, same 2000-read REF150 batch, 1 "
                                                   "thread (profiles/cpu_calibration.json, oracle/tools/calibrate_cpu.py)"
                                                   % (ratio, cal.get("cpu_model", "?"))) if ratio else None}
            out["config"]["speedup_vs_cpu_1thread"] = value / cps
            # the same restatement on every core this process may use (the reference has no such mode; stated for scale only).
            # `cores` = the threads really used = the CPU quota of the process's control group (advntr_host_threads: 16 on the GPU
            # boxes of this pool) -- NOT the 256 hardware threads of the host, which the quota would only throttle
            cores = max(1, quota_total)
            n_mt = min(n_reads, max(2000, 600 * cores))
            with HostRegion() as h_mt:
                t0 = time.perf_counter()
                mt_logp = O.viterbi_many_threads(bases[:off[n_mt]], off[:n_mt + 1], cores)
                dt = time.perf_counter() - t0
            assert np.array_equal(mt_logp, logp[:n_mt]), "GPU/oracle log-prob mismatch on the all-cores sample"
            mt_host = h_mt.record()
            out["cpu_baseline_all_cores"] = {"value": n_mt / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                                             "cpu_model": cpu_model_name(), "host_threads_available": os.cpu_count(),
                                             "nr_throttled": mt_host.get("nr_throttled_delta"),
                                             "throttled_usec": mt_host.get("throttled_usec_delta"),
                                             "sample": "first %d reads, oracle/viterbi_oracle.c on %d pthreads (= the CPU quota of the "
                                                       "process's control group); GPU logp bit-equal on the sample" % (n_mt, cores)}
        emit(order_line(out))
    passes.close()
    if comm:
        comm.close()
    return rc


def same_workload_n1(_lib, workloads, whole_input, flags, args, in_flight, workload):
    """Rank 0 of an N-rank strong-scaling job scores the WHOLE locus set alone: what `--workload c3|c4 --gpus 1` measures, taken
    inside the N-rank job (same box, same GPU, same build) with the line's own steps, warm-up and passes in flight."""
    from advntr_amd.pomegranate import device_models
    loci, reads, which, t_gen = whole_input
    t0 = time.perf_counter()
    workloads.build_models(loci)
    t_build = time.perf_counter() - t0
    bases, off = _lib.encode_reads(reads)
    dms = device_models([l.model for l in loci])
    make = lambda extra=0: _lib.DeviceBatch(dms, bases, off, which, flags=flags | extra)          # noqa: E731
    passes = Passes(make, in_flight)
    try:
        for _ in range(args.warmup):
            passes.run()
        passes.sync()
        with HostRegion() as h:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                passes.run()
            passes.sync()
            dt = time.perf_counter() - t0
        kernel_ms = passes.batches[0].run_timed(max(1, args.steps))
    finally:
        passes.close()
    return {"value": len(reads) * args.steps / dt, "unit": "reads/s", "ms_per_step": dt / args.steps * 1e3, "kernel_ms": kernel_ms,
            "calls": int(len(reads)), "loci": int(len(loci)), "steps": args.steps, "warmup": args.warmup,
            "passes_in_flight": in_flight, "model_build_s": t_build, "reads_generated_s": t_gen, "host": h.record(),
            "where": "rank 0 alone on its GPU, after the timed region of the N-rank job (its peers wait at a barrier)"}


# keys of the line in the order they are written: the compact summary FIRST (a driver that keeps only the head or a few parsed
# keys of a long line still has every headline number), the contract's keys, then the sub-records, the bulky ones (per-share
# lists of the rehearsals) last
LINE_ORDER = ("summary", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "same_workload_n1", "efficiency_measured", "efficiency_note",
              "comm", "rccl", "host", "config", "ru_concordance", "cpu_baseline_all_cores", "two_passes_in_flight", "s300",
              "log_probability", "c2", "end_to_end", "illumina_pipeline", "c4", "pacbio_end_to_end", "prefilter", "flank_align")
LINE_LAST = ("scale_rehearsal", "c4_scale_rehearsal")


def line_summary(out):
    """The headline numbers of a line, a few hundred bytes."""
    def pick(rec, *keys):
        return {k: rec.get(k) for k in keys if rec and rec.get(k) is not None} if rec else None

    def frac_of(rec):
        return (rec.get("roofline") or {}).get("frac", rec.get("frac")) if rec else None
    s = {"value": out.get("value"), "unit": out.get("unit"), "n_gpus": out.get("n_gpus"), "ms_per_step": out.get("ms_per_step"),
         "roofline_frac": (out.get("roofline") or {}).get("frac"), "kernel_ms": (out.get("roofline") or {}).get("kernel_ms")}
    if out.get("same_workload_n1") is not None:
        s["same_workload_n1_value"] = out["same_workload_n1"].get("value")
        s["efficiency_measured"] = out.get("efficiency_measured")
    per_rank = (out.get("config") or {}).get("per_rank")
    if per_rank:
        s["slowest_rank_loop_ms"] = max(r.get("loop_ms") or 0.0 for r in per_rank)
        s["slowest_rank_kernel_ms"] = max(r.get("kernel_ms") or 0.0 for r in per_rank)
        gathers = [r["gather_ms"] for r in per_rank if r.get("gather_ms") is not None]
        s["max_gather_ms"] = max(gathers) if gathers else None
        throttled = [(r.get("host") or {}).get("nr_throttled_delta") for r in per_rank]
        s["throttled_periods_in_timed_region"] = max([t for t in throttled if t is not None], default=None)
    elif out.get("host"):
        s["throttled_periods_in_timed_region"] = out["host"].get("nr_throttled_delta")
    for key in ("s300", "c2", "c4"):
        if out.get(key):
            s[key] = {"value": out[key].get("value"), "frac": frac_of(out[key]), "kernel_ms": out[key].get("kernel_ms")}
    if out.get("log_probability"):
        s["log_probability"] = pick(out["log_probability"], "value", "kernel_ms")
    if out.get("end_to_end"):
        s["end_to_end_total_s"] = out["end_to_end"].get("total_s")
    if out.get("illumina_pipeline"):
        s["illumina_pipeline_total_s"] = out["illumina_pipeline"].get("total_s")
    if out.get("pacbio_end_to_end"):
        s["pacbio_end_to_end_total_s"] = out["pacbio_end_to_end"].get("total_s")
    for key, name in (("scale_rehearsal", "projected_efficiency_c3_8_ranks"), ("c4_scale_rehearsal", "projected_efficiency_c4_8_ranks")):
        if out.get(key):
            s[name] = out[key].get("projected_efficiency")
            s[name + "_one_pass_in_flight"] = out[key].get("projected_efficiency_one_pass_in_flight")
    if out.get("cpu_baseline"):
        s["cpu_baseline_value"] = out["cpu_baseline"].get("value")
    return s


def order_line(out):
    out = dict(out)
    out["summary"] = line_summary(out)
    ordered = {}
    for k in LINE_ORDER:
        if k in out:
            ordered[k] = out[k]
    for k in out:
        if k not in ordered and k not in LINE_LAST:
            ordered[k] = out[k]
    for k in LINE_LAST:
        if k in out:
            ordered[k] = out[k]
    return ordered
