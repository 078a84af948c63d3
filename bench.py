#!/usr/bin/env python3
"""bench.py -- reads/s Viterbi-scored on the BASELINE.json workloads, one process per GPU, no PyTorch.

A "step" is one pass of the hot path (Viterbi DP + traceback + path summaries) over one resident batch of synthetic
reads; inputs are in HBM before the timed region.  Prints ONE JSON line on rank 0.

  --gpus 1 (default): config C1 of BASELINE.json / SURVEY 8d -- one REF150 locus (flank 150, 14-bp pattern, 11 copies:
      1413 states / 921 emitting / 4626 edges), 100 000 synthetic 150-bp reads.  The line also carries the roofline object,
      the VALU bound that actually binds (nominal, at the measured clock, at the measured issue rate), the CPU baseline (the C
      oracle on the host cores, with its calibration against the vendored pomegranate), and as sub-records everything else that
      has a number: `s300` (the "~300-state" label of the metric, Viterbi and log_probability), `log_probability`, `c2` and
      `end_to_end` (BASELINE config 2: 6 719 loci, kernel alone and candidate reads -> genotypes), `scale_rehearsal` (the 8-rank
      strong-scaling line projected from this one GPU), `c4`, `c4_scale_rehearsal` and `pacbio_end_to_end` (BASELINE config 5:
      8 960 PacBio loci, its 8-rank split rehearsed; whole 5-15 kb reads -> genotypes), `prefilter` and `flank_align` (the two
      kernels upstream of the scoring path), `two_passes_in_flight` (the C1 batch with the next pass queued on a second copy's
      stream).  About 40 s on the GPU box.
  --gpus N > 1: config C3 -- ONE set of 6 719 synthetic Illumina loci (~1.07 M calls) partitioned over the N GPUs by
      estimated work (strong scaling; whole loci per rank, LPT), every rank scores its share with no exchange, and the
      per-call result records are gathered to rank 0 over RCCL inside the timed region (the gather of pass i overlaps
      the kernels of pass i+1; all gathers complete before the clock stops).  `value` = calls of the WHOLE set per second.
      The strong-scaling lines keep TWO passes in flight (--in-flight, class Passes): consecutive passes alternate between two
      copies of the rank's device batch (scratch, results and stream of their own), so pass i+1 starts while the last
      workgroups of pass i drain -- the end of a launch is most of what separates an 8-rank share from an eighth of the set.
  --workload c4 --gpus N: BASELINE config 5 the same way -- ONE set of 8 960 PacBio loci (179 200 calls), whole loci to ranks
      by LPT on what is known of a locus before its reads exist (workloads.c4_plan), records gathered over RCCL, "strong".

Launching: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the driver's way; only the
launcher is torch, this file imports none of it) the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; started
plainly with --gpus N > 1 this process starts the N ranks itself as CHILD processes, before anything touches a GPU,
forwards rank 0's line and exits with their status (it never replaces itself by another program).
--dry-run does the planning and the rendezvous without any GPU work (CPU-side check of the multi-rank plumbing).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS, CLOCK_GHZ = 1024, 2.4      # 256 CUs x 4 SIMDs, 2.4 GHz


def algorithmic_bytes(n, m):
    """B = n [read] + (n+1)*m [1-byte back-pointer per cell] + (n+m) [traceback reads] + 32."""
    return n + (n + 1) * m + (n + m) + 32


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=100000, help="c1: reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=2000)
    ap.add_argument("--generic", action="store_true", help="force the generic-CSR kernel")
    ap.add_argument("--stream", action="store_true", help="experimental stream-packed column kernel")
    ap.add_argument("--antidiagonal", action="store_true", help="one-read-per-wavefront anti-diagonal kernel")
    ap.add_argument("--workload", default=None, choices=["c1", "s300", "c2", "c3", "c4"],
                    help="default: c1 at --gpus 1, c3 at --gpus > 1.  c1: 1 REF150 locus x --reads per GPU (weak); s300: the same "
                         "recipe on the metric's ~300-state shape (the launch the `s300` sub-record times, alone: for profilers); c2: --loci "
                         "synthetic loci x ~160 calls per GPU (weak); c3: ONE set of --loci loci partitioned over the GPUs by "
                         "estimated work (strong scaling, BASELINE config 3), records gathered to rank 0 over RCCL; "
                         "c4: ONE set of --loci PacBio loci (flank 100, error 0.3) x 20 trimmed spanning reads, partitioned over "
                         "the GPUs like c3 (strong scaling, BASELINE config 5)")
    ap.add_argument("--loci", type=int, default=None, help="c2/c3: default 6719; c4: default 8960")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-s300", action="store_true")
    ap.add_argument("--no-c2", action="store_true",
                    help="c1 at --gpus 1: leave out the `c2` and `end_to_end` sub-records (the 6719-locus target configuration)")
    ap.add_argument("--c2-loci", type=int, default=6719, help="loci of the `c2` / `end_to_end` sub-records")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="passes queued at a time (0 = the workload's default: 2 for the strong-scaling lines c3 / c4, else 1): "
                         "with 2, consecutive passes alternate between two copies of the device batch (own scratch, own stream) "
                         "and pass k + 1 starts while the last workgroups of pass k drain")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one process, one GPU (--workload c3 or c4): partition the locus set for this many ranks (LPT, as --gpus N does), run every "
                         "rank's share as its own resident batch with the multi-GPU launch parameters and print a "
                         "`scale_rehearsal` record -- a PROJECTION of the strong-scaling line, not a measurement of it")
    ap.add_argument("--root-capacity", type=float, default=0.99,
                    help="c3/c4 with more than one rank: rank 0 (the root of the result gather, which also hosts the receive side "
                         "of every peer's records) gets this fraction of an equal share of the planned work (1.0 = equal shares)")
    ap.add_argument("--no-upstream", action="store_true",
                    help="c1 at --gpus 1: leave out the `c4`, `pacbio_end_to_end`, `prefilter` and `flank_align` sub-records")
    ap.add_argument("--c4-loci", type=int, default=8960, help="loci of the `c4` sub-record (BASELINE config 5)")
    ap.add_argument("--pacbio-loci", type=int, default=896,
                    help="loci of the `pacbio_end_to_end` sub-record (whole 5-15 kb reads: 896 loci are 180 MB of read text)")
    ap.add_argument("--filter-reads", type=int, default=2000000, help="reads of the `prefilter` sub-record")
    ap.add_argument("--flank-reads", type=int, default=4000, help="reads of the `flank_align` sub-record")
    ap.add_argument("--dry-run", action="store_true", help="plan + rendezvous only, no GPU work (host communicator)")
    ap.add_argument("--launch-timeout", type=float, default=1800.0,
                    help="--gpus N > 1 started without a launcher: seconds after which the ranks are ended and the status is non-zero")
    ap.add_argument("--fault", default=None,
                    help="fault injection for the launcher's tests: 'exit:R' makes rank R leave with status 3 before the "
                         "rendezvous, 'hang:R' makes it sleep instead of joining")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 writes every call's (global id, logp, summary), gathered from all ranks after the timed "
                         "region, to this .npz (parity of an N-rank run with a 1-rank run: tests/test_gpu_parity.py)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes (only when no launcher set RANK for us)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """Start the N ranks as children, forward rank 0's line, exit with their status.  The children are watched together:
    the first one to fail, or the overall deadline (--launch-timeout), ends the job -- the children this process started
    are killed (exactly those) and the status is non-zero; a rank left waiting in a collective for a peer that is gone
    must not keep the launcher alive."""
    import shutil
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rdzv = tempfile.mkdtemp(prefix="advntr_rdzv_")             # mode 0700, this launch's alone
    fd, out_path = tempfile.mkstemp(prefix="advntr_rank0_")    # (not inside rdzv: rank 0 removes that directory when it leaves)
    os.close(fd)
    procs = []
    rc, why = 0, None

    def end_rank(p):
        """A rank and whatever it started (its process group: the workload generators' pool workers hold each other's pipe
        ends and would never see them close)."""
        import signal
        try:
            os.killpg(p.pid, signal.SIGKILL)                   # the group this launcher created for exactly that rank
        except (ProcessLookupError, PermissionError):
            pass
        if p.poll() is None:
            p.kill()
    try:
        with open(out_path, "wb") as out0:
            for r in range(args.gpus):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                           MASTER_PORT=str(port), ADVNTR_RDZV_DIR=rdzv)
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                              stdout=out0 if r == 0 else subprocess.DEVNULL, start_new_session=True))
        deadline = time.time() + args.launch_timeout
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = bad[0][1], "rank %d exited with status %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc, why = 124, "the ranks did not finish within %.0f s (--launch-timeout)" % args.launch_timeout
                break
            time.sleep(0.05)
        if why is not None:
            sys.stderr.write("bench.py: %s; ending the other ranks\n" % why)
        with open(out_path, "rb") as fh:
            sys.stdout.write(fh.read().decode())
        sys.stdout.flush()
    finally:
        # also when the launcher itself is interrupted: no rank, pool worker, output file or rendezvous directory stays behind
        # (only ranks that have not been reaped: their process-group id is still theirs.  The id of a rank that has exited and
        # been waited for may have been recycled for somebody else's group)
        for p in procs:
            if p.returncode is None and p.poll() is None:
                end_rank(p)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        try:
            os.unlink(out_path)
        except OSError:
            pass
        shutil.rmtree(rdzv, ignore_errors=True)
    return rc or 0


# ------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle is the checker, timed here as the reported baseline
# ------------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_model(locus):
    from oracle.oracle import OracleModel
    a = locus.model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    return OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])


def cpu_baseline(locus, bases, off, n_sample):
    """The oracle (C restatement of the reference loop, full tables calloc'd per call) on a bounded sample
    of the same reads, 1 thread -- the reference path is single-threaded (GIL held, hmm.pyx:1958)."""
    O = oracle_model(locus)
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


def load_json(*parts):
    try:
        return json.load(open(os.path.join(ROOT, *parts)))
    except (OSError, ValueError):
        return None


def measured_clock_ghz():
    """The shader clock under the bench kernel, measured once per round with GRBM_GUI_ACTIVE over the dispatch duration
    (scripts/clock_measure.sh -> profiles/r04_clock_summary.json; MI355X_MICROARCH.md, DVFS); None without the profile."""
    d = load_json("profiles", "r04_clock_summary.json") or {}
    for k, v in d.items():
        if "viterbi_rows_kernel" in k and v.get("effective_clock_mhz"):
            # (the counter is summed over the chip's 8 XCDs)
            return v["effective_clock_mhz"] / 8.0 / 1e3
    return None


def pmc_section(workload, n_calls, kernel):
    """Counters per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the bench);
    None when no committed profile describes this workload / kernel / size."""
    for name in ("r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        pmc = load_json("profiles", name)
        if not pmc:
            continue
        for sec in pmc.get("sections", []):
            if sec.get("workload") == workload and sec.get("calls") == n_calls and sec.get("kernel") == kernel:
                return dict(sec, file="profiles/" + name)
        if name == "r01_pmc_summary.json" and workload == "c1" and n_calls == 100000 and kernel == "viterbi_rows_kernel<5, 2>":
            s = pmc.get("viterbi_rows", {})
            if s:
                return {"hbm_bytes_per_launch_fetch_x2": s.get("hbm_bytes_per_launch_fetch_x2"),
                        "valu_insts_per_launch": s.get("valu_insts_per_launch"), "file": "profiles/" + name,
                        "stale": "counters of the round-1 build of this kernel"}
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "RANK" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, argv)

    # stdout carries exactly ONE line, the JSON record: anything a library prints while the bench runs (RCCL's version
    # banner, HIP warnings) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        os.write(json_fd, (json.dumps(record) + "\n").encode())

    from advntr_amd import comm as comm_mod
    rank, local_rank, world = comm_mod.env_world()
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
    host_workers = max(1, min(32, (os.cpu_count() or 2) // world - 1))
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
        if comm:
            comm.barrier()
            got = comm.allreduce_max(float(rank))
            assert got == float(world - 1), got
        if rank == 0:
            emit({"metric": "dry run: plan and rendezvous only", "value": None, "unit": "reads/s", "n_gpus": world,
                  "steps": 0, "warmup": 0, "dry_run": True, "scaling": "strong" if workload in ("c3", "c4") else "weak",
                  "config": dict({"workload": workload, "loci": n_loci, "calls_seen_by_ranks": counts,
                                  "comm": comm.backend if comm else None}, **plan_info)})
        if comm:
            comm.close()
        return 0
    if workload in ("c2", "c3", "c4"):
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
        c2_input = upstream_input = None
        if world == 1 and workload == "c1" and not args.no_c2 and not args.no_s300:
            # the target configuration of the north star rides on the C1 line as sub-records `c2` / `end_to_end`; its
            # synthetic reads come out of a process pool, which must have gone before the GPU is touched (see above)
            t_gen = time.perf_counter()
            c2_input = workloads.make_c2_parallel(args.c2_loci, seed=20240602, build=False, workers=host_workers,
                                                  return_counts=True) + (time.perf_counter() - t_gen,)
            # (a million read strings: out of the garbage collector's sight, or its passes land in the timed loops)
            upstream_input = None if args.no_upstream else upstream_inputs(workloads, host_workers, args)
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                # every gather of the timed steps completes inside the region
    passes.sync()                                          # ... and so does every pass, on either copy
    if comm:
        comm.barrier()
    elapsed_mine = time.perf_counter() - t0
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
        rec = json.dumps({"rank": rank, "calls": n_reads, "loop_ms_per_step": elapsed_mine / max(args.steps, 1) * 1e3,
                          "kernel_ms": kernel_ms, "model_build_s": t_build,
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
                      "mean %d states)" % (n, n_loci, " partitioned over %d GPUs" % world if world > 1 else "", m))
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
            # the same restatement on every host core (the reference has no such mode; stated for scale only)
            cores = os.cpu_count() or 1
            n_mt = min(n_reads, max(2000, 150 * cores))
            t0 = time.perf_counter()
            mt_logp = O.viterbi_many_threads(bases[:off[n_mt]], off[:n_mt + 1], cores)
            dt = time.perf_counter() - t0
            assert np.array_equal(mt_logp, logp[:n_mt]), "GPU/oracle log-prob mismatch on the all-cores sample"
            out["cpu_baseline_all_cores"] = {"value": n_mt / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                                             "cpu_model": cpu_model_name(),
                                             "sample": "first %d reads, oracle/viterbi_oracle.c on %d pthreads; GPU logp "
                                                       "bit-equal on the sample" % (n_mt, cores)}
        emit(out)
    passes.close()
    if comm:
        comm.close()
    return rc


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


SECOND_QUEUE = 128        # ADVNTR_FLAG_SECOND_QUEUE (include/advntr_hip.h)


class Passes(object):
    """Consecutive passes over ONE resident batch, one or two of them queued at a time.  With two, the passes alternate between
    two copies of the device batch -- same models, same reads, scratch, result arrays and stream of their own: pass k + 1 is
    queued behind nothing but its own copy's previous pass and starts while the last workgroups of pass k drain (the dynamic
    dequeue of a launch ends on single sweeps: 2-4 % of a launch, most of what separates an 8-rank share from an eighth of the
    whole set).  Every pass scores every read; the copies hold identical results."""

    def __init__(self, make, in_flight):
        # make(extra_flags) -> device batch; the second copy's stream is of a class of its own (ADVNTR_FLAG_SECOND_QUEUE): two
        # streams of one class can land on the same hardware queue, where their kernels would run strictly one after the other
        self.batches = [make(SECOND_QUEUE if i else 0) for i in range(max(1, int(in_flight)))]
        self.k = 0

    def run(self, reserve=0):
        b = self.batches[self.k % len(self.batches)]
        self.k += 1
        if reserve:
            b.reserve_next(reserve)
        b.run()
        return b

    def sync(self):
        for b in self.batches:
            b.sync()

    def ms_per_pass(self, steps, warm=2, reserve=0):
        for _ in range(warm):
            self.run(reserve)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run(reserve)
        self.sync()
        return (time.perf_counter() - t0) / steps * 1e3

    def close(self):
        for b in self.batches:
            b.close()


def passes_of(batch):
    """One pass at a time over an existing device batch."""
    one = Passes(lambda extra: None, 0)
    one.batches = [batch]
    return one


def two_in_flight_ms(batch, make, steps, reserve=0):
    """ms per pass with two passes in flight: `batch` and a second copy of it made here (and given back)."""
    twin = make(SECOND_QUEUE)
    try:
        both = passes_of(batch)
        both.batches.append(twin)
        return both.ms_per_pass(steps, reserve=reserve)
    finally:
        twin.close()


def scale_rehearsal(_lib, sharding, loci, dms, bases, off, which, n_ranks, whole, flags, steps, planned_work=None,
                    root_capacity=0.99):
    """What a 1-GPU lease can say about the north star's "strong scaling to 8 GPUs": the C3 locus set partitioned for
    n_ranks ranks exactly as `--gpus N` partitions it (whole loci, LPT on calls x (n+1) x states, sharding.partition_loci), and
    every rank's share run on THIS GPU as its own resident batch with the launch parameters of the multi-GPU job (the slots
    the gather asks each pass to leave free: advntr_batch_reserve_next(8)).  projected_efficiency = T(whole set, 1 rank) /
    (n_ranks x slowest share): what load balance, the per-launch costs that do not shrink with the batch and the partial last
    round of resident wavefronts leave of perfect strong scaling, BEFORE the gather (43 MB over xGMI per pass, overlapped with
    the next pass by design) and before any difference between GPUs.  A projection, labelled as such; the measured curve is
    the driver's SCALE run."""
    lens = np.diff(off)
    ms = np.array([d.m for d in dms])
    calls = np.bincount(which, minlength=len(dms))
    # planned_work: the per-locus estimates the multi-GPU job partitions by when it cannot know a locus's calls exactly (C4:
    # workloads.c4_plan); otherwise the plan is exact (C3: calls x 151 x states)
    work = list(planned_work) if planned_work is not None else [int(calls[k]) * 151 * int(ms[k]) for k in range(len(dms))]
    parts = sharding.partition_loci(work, n_ranks, [root_capacity] + [1.0] * (n_ranks - 1))
    loads = [float(sum(work[int(k)] for k in p)) for p in parts]
    cells = np.bincount(which, weights=(lens + 1) * ms[which], minlength=len(dms))       # actual work: trellis cells per locus
    actual = [float(cells[p].sum()) for p in parts]
    uniform = bool(len(lens) and lens.min() == lens.max())
    shares = []
    for r, mine in enumerate(parts):
        remap = np.full(len(dms), -1, np.int32)
        remap[mine] = np.arange(len(mine), dtype=np.int32)
        sel = remap[which] >= 0
        if uniform:
            sub_bases = bases.reshape(len(lens), -1)[sel].reshape(-1)
        else:
            sub_bases = bases[np.repeat(sel, lens)]
        sub_off = np.zeros(int(sel.sum()) + 1, np.int64)
        np.cumsum(lens[sel], out=sub_off[1:])
        make = lambda extra=0: _lib.DeviceBatch([dms[int(k)] for k in mine], sub_bases, sub_off, remap[which[sel]],      # noqa: E731
                                                flags=flags | extra)
        one = Passes(make, 1)
        batch = one.batches[0]
        loop_ms = one.ms_per_pass(steps, reserve=8)
        loop2_ms = two_in_flight_ms(batch, make, steps, reserve=8)
        kernel_ms = batch.run_timed(steps)                  # (no reservation: the kernel alone)
        shares.append({"rank": r, "loci": int(len(mine)), "calls": int(sel.sum()), "loop_ms": loop_ms,
                       "loop_ms_two_passes_in_flight": loop2_ms, "kernel_ms": kernel_ms})
        one.close()
    worst_loop = max(x["loop_ms"] for x in shares)
    worst_loop2 = max(x["loop_ms_two_passes_in_flight"] for x in shares)
    worst_kernel = max(x["kernel_ms"] for x in shares)
    whole2 = whole.get("loop_ms_two_passes_in_flight")
    return {"projection": True, "ranks": n_ranks, "whole_set": whole, "shares": shares,
            "sum_of_shares_loop_ms": sum(x["loop_ms"] for x in shares), "slowest_share_loop_ms": worst_loop,
            "load_imbalance_max_over_mean": max(loads) / (sum(loads) / n_ranks),
            "root_capacity": root_capacity, "root_load_over_mean": loads[0] / (sum(loads) / n_ranks),
            "root_share_loop_ms_over_slowest": shares[0]["loop_ms"] / worst_loop,
            "actual_cells_imbalance_max_over_mean": max(actual) / (sum(actual) / n_ranks),
            "per_locus_work_max_over_min": float(max(work)) / max(float(min(work)), 1.0),
            # as the strong-scaling lines run (bench.py --workload c3|c4: two passes in flight, class Passes) ...
            "projected_efficiency": (whole2 / (n_ranks * worst_loop2)) if whole2 else whole["loop_ms"] / (n_ranks * worst_loop),
            "passes_in_flight": 2 if whole2 else 1,
            # ... and with one pass at a time (rounds 3-5: a share's launch ends on single sweeps that nothing overlaps)
            "projected_efficiency_one_pass_in_flight": whole["loop_ms"] / (n_ranks * worst_loop),
            "projected_efficiency_kernels_only": whole["kernel_ms"] / (n_ranks * worst_kernel),
            "projected_value_calls_per_s": float(len(lens)) / ((worst_loop2 if whole2 else worst_loop) * 1e-3),
            "partitioned_by": ("estimated work per locus (calls x (reference VNTR length + 201) x expected states, workloads.c4_plan)"
                               if planned_work is not None else "exact work per locus (calls x 151 x states)"),
            "excludes": "the RCCL gather of the result records (40 B per call to rank 0, queued behind pass i and overlapped with "
                        "pass i + 1) and differences between the GPUs of a node",
            "note": "ONE GPU ran the %d shares one after the other; each share is a rank's whole batch (its models, its calls), "
                    "launched as the multi-GPU job launches it" % n_ranks}


def target_configuration_records(_lib, workloads, c2_input, flags, args):
    """The north star's target configuration on one GPU (BASELINE config 2: 6719 Illumina loci x a 30x-equivalent read
    batch) as two sub-records of the C1 line.

    `end_to_end`: candidate reads -> genotypes, what the reference's per-locus loop does (genome_analyzer.py:280-297 ->
    vntr_finder.py:727-767, 807-887): models built by the native builder, both strands of every candidate scored, recruit
    rule, aggregation, maximum-likelihood genotype -- the host stages of one piece of the locus set overlapped with the
    scoring of the previous one (vntr_finder.genotype_loci_pipelined), next to the same stages run one after the other.
    `c2`: the scoring kernel alone over the whole set's calls, resident in HBM, with its roofline object."""
    from advntr_amd import hmm_utils, vntr_finder
    from advntr_amd.pomegranate import device_models
    loci, reads, which, counts, t_gen = c2_input
    n_loci = len(loci)
    desc = [(l.left, l.right, l.units, l.copies) for l in loci]
    first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
    candidates = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]      # forward strands only
    n_cand = int(sum(len(c) for c in candidates))
    hmm_utils.build_read_matcher_models(desc[:4])                                                   # warm-up
    vntr_finder.score_reads_arrays(hmm_utils.build_read_matcher_models(desc[:1]), [candidates[0][:8]])
    # the stages one after the other
    T = {}
    t0 = time.perf_counter()
    models = hmm_utils.build_read_matcher_models(desc)
    T["build_models"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    dms = device_models(models)
    T["upload_models"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    res = vntr_finder.score_reads_arrays(models, candidates, None, compute_reverse=True)
    T["encode_score_recruit"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    plain = vntr_finder._genotypes_from_scores(res, n_loci, False, False, 0)
    T["aggregate_genotype"] = time.perf_counter() - t1
    T["total"] = time.perf_counter() - t0
    recruited = int((res["recruited"] & (res["summary"][:, _lib.SUM_REPEAT_BP] > 2)).sum())
    # ... and overlapped
    # (three passes: host threads, page cache and the PCIe path make a single pass vary by +-15 %; the fastest one is reported,
    # all three totals are listed)
    P, totals = None, []
    for _ in range(3):
        Pk = {}
        piped = vntr_finder.genotype_loci_pipelined(desc, candidates, timings=Pk)
        totals.append(Pk["total"])
        if P is None or Pk["total"] < P["total"]:
            P = Pk
    same = sum(a.copy_numbers == b.copy_numbers and a.recruited_reads_count == b.recruited_reads_count
               for a, b in zip(plain, piped))
    assert same == n_loci, "pipelined and stage-by-stage genotypes differ on %d loci" % (n_loci - same)
    e2e = {"loci": n_loci, "candidate_reads": n_cand, "viterbi_calls": 2 * n_cand, "recruited_reads": recruited,
           "loci_with_genotype": sum(g.copy_numbers is not None for g in piped),
           "value": 2 * n_cand / P["total"], "unit": "calls/s", "total_s": P["total"], "total_s_of_each_pass": totals,
           "stage_s_overlapped": {k: v for k, v in P.items() if k != "total"},
           "stages_one_after_the_other": dict(T),
           "genotypes_identical_to_stage_by_stage": same == n_loci,
           "note": "from candidate reads in Python lists to RU-count genotypes; overlapped = model build / upload / read "
                   "encoding of locus piece k+1 on host threads while piece k is scored (12 pieces, the first one in 3 growing parts; a piece's kernels are queued before the previous piece's are waited for); synthetic input "
                   "generated in %.1f s (not timed)" % t_gen}
    # the kernel over the whole set's calls (mapped forward + unmapped on both strands, as BASELINE config 2 counts them)
    bases, off = _lib.encode_reads(reads)
    batch = _lib.DeviceBatch(dms, bases, off, which, flags=flags)
    batch.run()
    batch.sync()
    steps = max(1, min(args.steps, 5))
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    batch.sync()
    dt = (time.perf_counter() - t0) / steps
    kernel_ms = batch.run_timed(steps)
    dt2_ms = two_in_flight_ms(batch, lambda extra: _lib.DeviceBatch(dms, bases, off, which, flags=flags | extra), steps)
    logp, summ = batch.fetch()
    kinfo = batch.kernel_info()
    kernels = [k[:3] for k in kinfo]
    kernel = max(kernels, key=lambda k: k[1])[0]
    ms = np.array([d.m for d in dms])
    lens = np.diff(off)
    alg = float(np.sum(lens + (lens + 1) * ms[which] + (lens + ms[which]) + 32))
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    pmc = pmc_section("c2", len(reads), kernel) or {}
    traffic = pmc.get("hbm_bytes_per_launch_fetch_x2")
    c2 = {"loci": n_loci, "calls": len(reads), "mean_states": float(np.mean(ms[which])), "read_len": int(round(float(lens.mean()))),
          "value": len(reads) / dt, "unit": "calls/s", "ms_per_step": dt * 1e3, "steps": steps, "kernel_ms": kernel_ms,
          "ms_per_step_two_passes_in_flight": dt2_ms,
          "kernel": kernel, "kernels": [{"name": k, "reads": r, "tiles": t, "useful_lane_steps": u} for k, r, t, u in kinfo],
          "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBPS, "algorithmic_gb_per_launch": alg / 1e9,
                       "traffic": traffic / 1e9 if traffic else None, "traffic_source": pmc.get("file"),
                       "note": "exact sum over the calls of n + (n+1) m + (n+m) + 32 bytes (SURVEY 8d) / HIP-event kernel time"}}
    from advntr_amd import sharding
    c2["scale_rehearsal"] = scale_rehearsal(_lib, sharding, loci, dms, bases, off, which, 8,
                                            {"calls": len(reads), "loop_ms": dt * 1e3, "kernel_ms": kernel_ms,
                                             "loop_ms_two_passes_in_flight": dt2_ms}, flags, steps,
                                            root_capacity=args.root_capacity)
    if not args.no_cpu:
        # per-locus sample against the oracle: log-probabilities bit for bit, repeat-unit counts as hmm_utils derives them
        # from the oracle's path; its single-thread rate on these models prices the whole set for the reference
        from oracle import oracle as Or
        sample = np.linspace(0, n_loci - 1, 48).astype(int)
        n_chk = same_ru = 0
        t_cpu = 0.0
        for k in sample:
            arr = models[k].baked_arrays()
            edges = [(int(arr["in_src"][e]), l, float(arr["in_logp"][e]))
                     for l in range(arr["m"]) for e in range(arr["in_ptr"][l], arr["in_ptr"][l + 1])]
            O = Or.OracleModel(arr["m"], arr["silent_start"], arr["start_index"], arr["end_index"], edges, arr["emis_logp"])
            names = [st.name for st in models[k].states]
            for i in range(int(first[k]), int(first[k]) + 6):
                t1 = time.perf_counter()
                olp, opath = O.viterbi(reads[i])
                t_cpu += time.perf_counter() - t1
                assert logp[i] == olp, "GPU/oracle log-prob mismatch on the C2 sample (locus %d)" % k
                ru = Or.number_of_repeats([names[j] for j in opath][1:-1]) if opath else 0
                same_ru += int(ru == int(summ[i][0]))
                n_chk += 1
        cal = load_json("profiles", "cpu_calibration.json") or {}
        ratio = cal.get("oracle_over_pomegranate")
        cps = n_chk / t_cpu
        c2["ru_concordance"] = {"loci": len(sample), "calls": n_chk, "identical_ru_counts": same_ru, "logp_bit_equal": True}
        c2["cpu_baseline"] = {"value": cps, "unit": "calls/s", "cores": 1, "kind": "port", "cpu_model": cpu_model_name(),
                              "sample": "6 calls of each of 48 loci spread over the set, oracle/viterbi_oracle.c through its "
                                        "per-call entry, 1 thread",
                              "pomegranate_equivalent": cps / ratio if ratio else None}
        if ratio:
            c2["speedup_vs_pomegranate_equivalent_1thread"] = c2["value"] / (cps / ratio)
            e2e["reference_scoring_alone_s_pomegranate_equivalent"] = 2 * n_cand / (cps / ratio)
    batch.close()
    return e2e, c2


# ------------------------------------------------------------------------------------------------
# BASELINE config 5 (PacBio) and the two stages upstream of the scoring path, as sub-records of the N = 1 line
# ------------------------------------------------------------------------------------------------
REF_FILTER = os.path.join(ROOT, "oracle", "_ref", "adVNTR-Filtering")


def upstream_inputs(workloads, host_workers, args):
    """Synthetic inputs of the `c4`, `pacbio_end_to_end`, `prefilter` and `flank_align` sub-records -- made BEFORE the GPU
    is touched (their generators fork a process pool).  The reference's own filter binary (oracle/_ref, the prefilter's
    CPU baseline, kind "reference") is started here on a bounded sample and collected at the end: its 20 s run beside
    the GPU work instead of in front of it."""
    import tempfile
    inp = {}
    t = time.perf_counter()
    inp["c4"] = workloads.make_c4(args.c4_loci, seed=20240603, workers=host_workers)
    inp["pacbio"] = workloads.make_pacbio_whole_reads(args.pacbio_loci, seed=20240603, workers=host_workers)
    inp["flank"] = workloads.make_flank_align_workload(args.flank_reads)
    lines, fasta, rec_len = workloads.make_prefilter_workload(6719, args.filter_reads)
    inp["prefilter"] = (lines, fasta, rec_len)
    inp["gen_s"] = time.perf_counter() - t
    inp["ref_filter"] = None
    if os.path.exists(REF_FILTER) and not args.no_cpu:
        sample = min(args.filter_reads, 50000)
        d = tempfile.mkdtemp(prefix="advntr_reffilter_")
        kw, fa, empty = os.path.join(d, "kw.txt"), os.path.join(d, "s.fa"), os.path.join(d, "e.fa")
        with open(kw, "w") as fh:
            fh.write("".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines))
        with open(fa, "wb") as fh:
            fh.write(fasta[:sample * rec_len])
        with open(empty, "w") as fh:
            fh.write(">x\nACGT\n")
        import threading
        took = {}

        def timed(path):
            # the child is started HERE, before anything touches the GPU; the thread only waits for it
            t0 = time.perf_counter()
            with open(kw) as fin, open(path + ".out", "wb") as fout:
                child = subprocess.Popen([REF_FILTER, path], stdin=fin, stdout=fout)

            def wait():
                child.wait()
                took[path] = (time.perf_counter() - t0, child.returncode)
            th = threading.Thread(target=wait)
            th.start()
            return th
        threads = [timed(f) for f in (empty, fa)]
        inp["ref_filter"] = {"dir": d, "threads": threads, "took": took, "sample": sample, "fa": fa, "empty": empty}
    return inp


def c4_record(_lib, workloads, inp, flags, args):
    """BASELINE config 5 on one GPU: 8 960 PacBio loci (flank 100, error rate 0.3), 20 trimmed spanning reads each -- the
    batch get_dominant_copy_numbers_from_spanning_reads scores (vntr_finder.py:550-555), resident in HBM."""
    from advntr_amd.pomegranate import device_models
    loci, reads, which = inp["c4"]
    t0 = time.perf_counter()
    workloads.build_models(loci)
    t_build = time.perf_counter() - t0
    dms = device_models([l.model for l in loci])
    bases, off = _lib.encode_reads(reads)
    batch = _lib.DeviceBatch(dms, bases, off, which, flags=flags)
    batch.run()
    batch.sync()
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    batch.sync()
    dt = (time.perf_counter() - t0) / steps
    kernel_ms = batch.run_timed(steps)
    dt2_ms = two_in_flight_ms(batch, lambda extra: _lib.DeviceBatch(dms, bases, off, which, flags=flags | extra), steps)
    logp, summ = batch.fetch()
    kinfo = batch.kernel_info()
    kernel = max(kinfo, key=lambda k: k[1])[0]
    ms = np.array([d.m for d in dms])
    edges = np.array([l.model.n_edges for l in loci], np.int64)
    lens = np.diff(off)
    alg = float(np.sum(lens + (lens + 1) * ms[which] + (lens + ms[which]) + 32))
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    pmc = pmc_section("c4", len(reads), kernel) or {}
    traffic, valu = pmc.get("hbm_bytes_per_launch_fetch_x2"), pmc.get("valu_insts_per_launch")
    rec = {"loci": len(loci), "calls": len(reads), "mean_states": float(np.mean(ms[which])), "read_len_mean": float(lens.mean()),
           "read_len_min_max": [int(lens.min()), int(lens.max())], "model_build_s": t_build,
           "value": len(reads) / dt, "unit": "calls/s", "ms_per_step": dt * 1e3, "steps": steps, "kernel_ms": kernel_ms,
           "ms_per_step_two_passes_in_flight": dt2_ms,
           "kernel": kernel, "kernels": [{"name": k, "reads": r, "tiles": t, "useful_lane_steps": u} for k, r, t, u in kinfo],
           "relaxations_per_s": float(np.sum((lens + 1) * edges[which])) / dt,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "algorithmic_gb_per_launch": alg / 1e9, "traffic": traffic / 1e9 if traffic else None,
                        "traffic_source": pmc.get("file"),
                        "note": "exact sum over the calls of n + (n+1) m + (n+m) + 32 bytes (SURVEY 8d) / HIP-event kernel time"}}
    if valu:
        bound_ms = valu * 4 / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
        rec["roofline"]["bound_actual"] = {"bound": "valu_f64", "valu_insts_per_launch": valu, "cycles_per_inst": 4,
                                           "issue_bound_ms": bound_ms, "kernel_ms": kernel_ms, "frac": bound_ms / kernel_ms,
                                           "source": pmc.get("file")}
    from advntr_amd import sharding
    plan = workloads.c4_plan(len(loci), seed=20240603)
    # (a rank of the multi-GPU job has its GPU to itself: the whole set's batch -- 100 GB of long-read scratch -- gives its memory
    # back before the shares are laid out, or their launches would be sized for what is left)
    batch.close()
    rec["scale_rehearsal"] = scale_rehearsal(_lib, sharding, loci, dms, bases, off, which, 8,
                                             {"calls": len(reads), "loop_ms": dt * 1e3, "kernel_ms": kernel_ms,
                                              "loop_ms_two_passes_in_flight": dt2_ms}, flags, steps,
                                             planned_work=[c * (ln + 1) * st for c, ln, st in plan],
                                             root_capacity=args.root_capacity)
    if not args.no_cpu:
        from oracle import oracle as Or
        sample = np.linspace(0, len(loci) - 1, 12).astype(int)
        n_chk = same_ru = 0
        t_cpu = 0.0
        for k in sample:
            model = loci[k].model
            arr = model.baked_arrays()
            edge_list = [(int(arr["in_src"][e]), l, float(arr["in_logp"][e]))
                         for l in range(arr["m"]) for e in range(arr["in_ptr"][l], arr["in_ptr"][l + 1])]
            O = Or.OracleModel(arr["m"], arr["silent_start"], arr["start_index"], arr["end_index"], edge_list, arr["emis_logp"])
            names = [st.name for st in model.states]
            first = int(np.searchsorted(which, k))
            for i in (first, first + 7):
                t1 = time.perf_counter()
                olp, opath = O.viterbi(reads[i])
                t_cpu += time.perf_counter() - t1
                assert logp[i] == olp, "GPU/oracle log-prob mismatch on the C4 sample (locus %d)" % k
                ru = Or.number_of_repeats([names[j] for j in opath][1:-1]) if opath else 0
                same_ru += int(ru == int(summ[i][0]))
                n_chk += 1
        cal = load_json("profiles", "cpu_calibration.json") or {}
        ratio = cal.get("oracle_over_pomegranate")
        cps = n_chk / t_cpu
        rec["ru_concordance"] = {"loci": len(sample), "calls": n_chk, "identical_ru_counts": same_ru, "logp_bit_equal": True}
        rec["cpu_baseline"] = {"value": cps, "unit": "calls/s", "cores": 1, "kind": "port", "cpu_model": cpu_model_name(),
                               "sample": "2 calls of each of 12 loci spread over the set, oracle/viterbi_oracle.c, 1 thread",
                               "pomegranate_equivalent": cps / ratio if ratio else None}
    batch.close()
    return rec


def pacbio_end_to_end_record(_lib, inp, args):
    """find_repeat_count_from_pacbio_reads (vntr_finder.py:652-665) for a tenth of config 5's loci, from WHOLE 5-15 kb reads
    to RU-count genotypes: flank alignment of both strands (advntr_flank_align), trimming, one model per locus sized for its
    longest spanning read, Viterbi, maximum-likelihood copy numbers -- the stages of locus piece k + 1 overlapped with the
    scoring of piece k (vntr_finder.genotype_pacbio_loci).  Three passes, the fastest reported."""
    from advntr_amd import settings, vntr_finder
    loci, read_lists = inp["pacbio"]
    n_reads = sum(len(r) for r in read_lists)
    n_bases = sum(len(s) for r in read_lists for s in r)
    old = settings.MAX_ERROR_RATE
    settings.MAX_ERROR_RATE = 0.3
    try:
        vntr_finder.genotype_pacbio_loci(loci[:8], read_lists[:8], chunks=2)                       # warm-up
        P, totals, res = None, [], None
        for _ in range(3):
            Pk = {}
            got = vntr_finder.genotype_pacbio_loci(loci, read_lists, timings=Pk)
            totals.append(Pk["total"])
            if P is None or Pk["total"] < P["total"]:
                P, res = Pk, got
        # a sample of loci the way the reference walks them, one at a time: same spanning reads, same genotype
        sample = np.linspace(0, len(loci) - 1, 8).astype(int)
        for k in sample:
            left, right, segments, pattern = loci[k]
            spanning, _ = vntr_finder.extract_spanning_reads(left, right, read_lists[k])
            want, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(left, right, segments, pattern,
                                                                                   [s[0] for s in spanning])
            assert res[k].copy_numbers == want and res[k].maximum_likelihood == prob and res[k].spanning_reads_count == len(spanning), \
                "pipelined PacBio route differs from the per-locus route on locus %d" % k
    finally:
        settings.MAX_ERROR_RATE = old
    n_span = int(sum(g.spanning_reads_count for g in res))
    return {"loci": len(loci), "whole_reads": n_reads, "read_bases": n_bases, "flank_alignments": 4 * n_reads,
            "spanning_reads_scored": n_span, "loci_with_genotype": sum(g.copy_numbers is not None for g in res),
            "value": n_reads / P["total"], "unit": "whole reads/s", "loci_per_s": len(loci) / P["total"],
            "total_s": P["total"], "total_s_of_each_pass": totals,
            "stage_s_overlapped": {k: v for k, v in P.items() if k != "total"},
            "per_locus_route_identical_on_sample": len(sample),
            "note": "a tenth of BASELINE config 5's loci (the c4 recipe, seed 20240603) with WHOLE reads of 5-15 kb, either "
                    "strand, one in ten unrelated; extraction parity with biopython's pairwise2 is unpinned (absent from the "
                    "image), kernel == restatement in tests/test_flank_align.py"}


def flank_align_record(_lib, inp, args):
    """advntr_flank_align on PacBio-sized input (what scripts/flank_align_bench.py prints): reads of 5-15 kb, two 100-base
    flanks, both strands = 4 alignments per read; int32 VALU issue is the roof that binds."""
    from oracle import oracle as Or
    from advntr_amd import vntr_finder
    left, right, reads = inp["flank"]
    n = len(reads)
    strand_read = np.arange(2 * n, dtype=np.int32) // 2 + (np.arange(2 * n, dtype=np.int32) & 1) * n
    pr = np.repeat(strand_read, 2)
    pf = np.tile(np.array([0, 1], np.int32), 2 * n)
    _lib.flank_align(reads[:8], [left, right], np.arange(16, dtype=np.int32) // 2, pf[:16])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        score, begin, end, ms = _lib.flank_align(reads, [left, right], pr, pf)
        wall = time.perf_counter() - t0
        if best is None or ms < best[0]:
            best = (ms, wall)
    ms, wall = best
    lens = np.fromiter(map(len, reads), dtype=np.int64, count=n)
    cells = float(lens.sum()) * 4 * 100
    bytes_alg = float(lens.sum()) * 4
    # the sweep of a pair takes n + lf - 1 steps of 128 cells (both 64-column chunks of a lane in the halves of one register);
    # instruction census of a step (ISA of flank_align_kernel, pass 1): 18 wave64 vector instructions -- 5 DPP operations, 7 packed
    # 16-bit operations, 2 byte permutes and a three-way maximum (all 64-bit encodings: ~4.5 cycles each on this part,
    # profiles/r01_valu_ubench.txt) and 3 plain 32-bit ones (~2.6)
    steps = float((lens + 99).sum()) * 4
    valu_per_step = 18
    peak = 128.0 / (valu_per_step * 2) * SIMDS * CLOCK_GHZ * 1e9
    rec = {"alignments": int(len(pr)), "reads": n, "value": len(pr) / (ms * 1e-3), "unit": "alignments/s", "dtype": "i16 (packed pairs)",
           "kernel_ms": ms, "call_ms_incl_pcie_and_host": wall * 1e3, "cells_per_s": cells / (ms * 1e-3),
           "spanning_found": int(((score[0::2] >= 70) & (score[1::2] >= 70) & (begin[1::2] >= begin[0::2])).sum()),
           "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                        "note": "tier rule (HBM: each alignment streams its read once) only; the binding roof is VALU issue",
                        "bound_actual": {"bound": "valu_int", "unit": "DP cells/s", "achieved": cells / (ms * 1e-3),
                                         "valu_per_step_of_128_cells": valu_per_step, "peak": peak,
                                         "peak_note": "every instruction at the nominal 2 cycles per wave64 instruction",
                                         "frac": cells / (ms * 1e-3) / peak,
                                         "cycles_per_step_measured": ms * 1e-3 * CLOCK_GHZ * 1e9 * SIMDS / steps,
                                         "cycles_per_step_at_measured_issue_rates": 15 * 4.5 + 3 * 2.6}}}
    if not args.no_cpu:
        n_cpu = 24
        t0 = time.perf_counter()
        for p in range(n_cpu):
            r = int(pr[p])
            s = reads[r] if r < n else vntr_finder.reverse_complement(reads[r - n])
            got = Or.flank_align(s, [left, right][pf[p]])
            assert got == (int(score[p]), int(begin[p]), int(end[p])), "flank alignment differs from its restatement (pair %d)" % p
        rec["cpu_baseline"] = {"value": n_cpu / (time.perf_counter() - t0), "unit": "alignments/s", "cores": 1, "kind": "port",
                               "sample": "first %d alignments, oracle/flank_align_oracle.c (biopython is absent: parity unpinned); "
                                         "results equal to the GPU's" % n_cpu}
    return rec


def prefilter_record(_lib, inp, args):
    """The keyword prefilter (adVNTR-Filtering's scan, filtering/main.cc:247-283) at model-database scale, from the bytes of the
    FASTA file to the (read, VNTR, count) records; HBM read of one byte per base is the roof.  CPU baseline: the REFERENCE binary
    itself on a bounded sample of the same file, its stdout compared byte for byte with the GPU path's."""
    from advntr_amd import filtering
    lines, fasta, rec_len = inp["prefilter"]
    n_reads = len(fasta) // rec_len
    read_len = rec_len - 11
    n_kw = sum(len(k) for _, k in lines)
    t0 = time.perf_counter()
    f = filtering.KeywordFilter(lines)
    t_build = time.perf_counter() - t0
    starts = np.arange(n_reads, dtype=np.int64) * rec_len + 10
    ends = starts + read_len

    def scan_fasta(text):
        idx = _lib.line_index(text)                       # the line index is part of the call: the file is all the caller has
        k = (len(idx) - 1) // 2
        return f.scan_text(text, idx[1:2 * k:2], idx[2:2 * k + 1:2] - 1)
    scan_fasta(fasta[:1000 * rec_len])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        recs = scan_fasta(fasta)
        wall = time.perf_counter() - t0
        if best is None or f.kernel_ms < best[0]:
            best = (f.kernel_ms, wall)
    kernel_ms, wall = best
    bases = float(n_reads) * read_len
    gbps = bases / (kernel_ms * 1e-3) / 1e9
    rec = {"keywords": n_kw, "loci": len(lines), "reads": n_reads, "read_len": read_len, "value": bases / (kernel_ms * 1e-3),
           "unit": "bases/s", "dtype": "u8", "kernel_ms": kernel_ms, "kernel": "keyword_filter_short_kernel",
           "filter_build_s": t_build, "call_ms_from_fasta_bytes_incl_pcie_and_host": wall * 1e3, "fasta_bytes": len(fasta),
           "reads_with_hits": int(len(np.unique(recs[0]))),
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                        "traffic": None, "bytes_per_base": 1}}
    ref = inp.get("ref_filter")
    if ref:
        import shutil
        for th in ref["threads"]:
            th.join()
        (t_start, rc0), (t_run, rc1) = ref["took"][ref["empty"]], ref["took"][ref["fa"]]
        assert rc0 == 0 and rc1 == 0, "oracle/_ref/adVNTR-Filtering failed"
        out = open(ref["fa"] + ".out", "rb").read()
        mine = f.select_fasta(fasta[:ref["sample"] * rec_len])
        same = mine.encode("latin-1") == out
        assert same, "prefilter stdout differs from the reference binary's on the bench sample"
        rec["cpu_baseline"] = {"value": ref["sample"] * read_len / max(t_run - t_start, 1e-9), "unit": "bases/s", "cores": 1,
                               "kind": "reference", "cpu_model": cpu_model_name(),
                               "sample": "first %d reads through oracle/_ref/adVNTR-Filtering (filtering/main.cc); start-up "
                                         "(automaton build + 1.9 GB memset) %.1f s subtracted from %.1f s; stdout identical to the "
                                         "GPU path: %s" % (ref["sample"], t_start, t_run, same)}
        shutil.rmtree(ref["dir"], ignore_errors=True)
    f.close()
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
        k = min(args.cpu_sample, n_reads)
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


if __name__ == "__main__":
    sys.exit(main())
