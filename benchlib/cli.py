"""Command line of bench.py and its own launcher: N ranks as child processes when no launcher set RANK for us."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=100000, help="c1: reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=8000,
                    help="reads of the bounded sample the CPU baseline (the C oracle, 1 thread) is timed on: 8 000 REF150 reads are "
                         "about 11 s of CPU work on the GPU box's host")
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
    ap.add_argument("--processes", action="store_true",
                    help="with --emulate-ranks N: start N rank PROCESSES that share this box's GPU(s) through the host communicator and "
                         "run the N-rank strong-scaling line as they would on N GPUs -- what one GPU can show of the HOST side of an "
                         "N-rank job: every rank's waits, helper threads and launch loop against the one CPU quota of the box "
                         "(per-rank `host.nr_throttled_delta`); the rates of such a line are those of ranks sharing a GPU")
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
    ap.add_argument("--pipeline-reads", type=int, default=10000000,
                    help="reads of the FASTA file of the `illumina_pipeline` sub-record (file bytes -> prefilter -> selection -> scoring "
                         "-> genotypes -> VCF rows on one timeline; 10 M reads are 1.6 GB of text); 0 leaves the record out")
    ap.add_argument("--no-n1", action="store_true",
                    help="c3 / c4 with more than one rank: leave out `same_workload_n1` (rank 0 scoring the WHOLE set alone after the "
                         "timed region, the line's own baseline) and `efficiency_measured`")
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


def emulation_argv(argv, n):
    """argv of the rank processes of `--emulate-ranks N --processes`: the same line as `--gpus N`."""
    out, skip = [], 0
    for a in argv:
        if skip:
            skip -= 1
            continue
        if a in ("--emulate-ranks", "--gpus"):
            skip = 1
            continue
        if a.startswith("--emulate-ranks=") or a.startswith("--gpus=") or a == "--processes":
            continue
        out.append(a)
    return out + ["--gpus", str(n), "--no-n1"]


def spawn_ranks(args, argv, n=None, extra_env=None):
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
            n_ranks = int(n or args.gpus)
            for r in range(n_ranks):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                           MASTER_PORT=str(port), ADVNTR_RDZV_DIR=rdzv)
                env.update(extra_env or {})
                procs.append(subprocess.Popen([sys.executable, BENCH] + argv, env=env,
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
