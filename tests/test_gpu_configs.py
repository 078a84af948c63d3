"""BASELINE.json configs C2 / C3 / C4 under -m gpu: the bench's own generators (advntr_amd/workloads.py) scored through
the C ABI, (a) at a size the oracle replays in seconds -- log-probabilities `==`, paths and RU counts exact on a
subsample of the calls -- and (b) at BASELINE's full sizes through size-independent properties (run-to-run determinism,
split / permutation invariance, the two kernel families agree on every call, value ranges, base-count conservation).
C3 (the set partitioned over ranks): a 2-rank run of bench.py on one GPU gathers exactly the records of the 1-rank run,
and the RCCL communicator is exercised with the world size a 1-GPU box allows.
Reference: the per-locus scoring loops /root/reference/advntr/vntr_finder.py:727-767 (Illumina), :534-585 (PacBio).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _oracle(model):
    from oracle.oracle import OracleModel
    a = model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    return OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])


def _check_calls_against_oracle(loci, reads, which, logp, summ, paths, every):
    from advntr_amd import _lib
    from oracle import oracle as Or
    cache = {}
    checked = 0
    for i in range(0, len(reads), every):
        k = int(which[i])
        if k not in cache:
            cache[k] = (_oracle(loci[k].model), [s.name for s in loci[k].model.states])
        O, names = cache[k]
        olp, opath = O.viterbi(reads[i])
        assert logp[i] == olp, (i, k, logp[i], olp)
        assert abs(logp[i] - olp) <= 1e-4                                      # the north star's tolerance, implied by ==
        assert paths[i] == opath, (i, k)
        inner = [names[j] for j in opath][1:-1]
        assert summ[i][_lib.SUM_RU] == Or.number_of_repeats(inner), (i, k)
        assert summ[i][_lib.SUM_MATCHES] == Or.number_of_matches(inner)
        assert summ[i][_lib.SUM_REPEAT_BP] == Or.repeat_bp_matches(inner)
        checked += 1
    return checked


def _properties(dms, reads, which, flags_other, n_split):
    """Size-independent properties of one multi-locus batch; returns (logp, summ)."""
    from advntr_amd import _lib
    bases, off = _lib.encode_reads(reads)
    lens = np.diff(off)
    B = _lib.DeviceBatch(dms, bases, off, which)
    B.run()
    logp, summ = B.fetch()
    B.run()
    logp2, summ2 = B.fetch()
    kernels = B.kernels()
    B.close()
    assert np.array_equal(logp, logp2) and np.array_equal(summ, summ2)                    # deterministic / idempotent
    assert np.all(np.isfinite(logp)) and np.all(logp <= 0)
    assert np.all(summ[:, _lib.SUM_PATH_LEN] >= lens + 2)
    assert np.all(summ[:, _lib.SUM_LEFT_BP] + summ[:, _lib.SUM_RIGHT_BP] + summ[:, _lib.SUM_REPEAT_BP] == lens)
    assert np.all(summ[:, _lib.SUM_MATCHES] <= lens) and np.all(summ[:, _lib.SUM_RU] >= 0)
    # the other kernel family (one read per wavefront, anti-diagonal) on every call
    lp_a, sm_a, _ = _lib.viterbi_batch(dms, bases, off, which, flags=flags_other)
    assert np.array_equal(lp_a, logp) and np.array_equal(sm_a, summ)
    # permutation + split invariance: shuffle the calls, score in two parts, un-shuffle
    rng = np.random.default_rng(7)
    perm = rng.permutation(len(reads))
    got = np.zeros_like(logp)
    got_sum = np.zeros_like(summ)
    for part in (perm[:n_split], perm[n_split:]):
        pb, po = _lib.encode_reads([reads[i] for i in part])
        lp, sm, _ = _lib.viterbi_batch(dms, pb, po, which[part])
        got[part], got_sum[part] = lp, sm
    assert np.array_equal(got, logp) and np.array_equal(got_sum, summ)
    assert int(got_sum[:, _lib.SUM_RU].sum()) == int(summ[:, _lib.SUM_RU].sum())         # checksum of checksums
    return logp, summ, kernels


def test_c2_200_loci_vs_oracle():
    """C2's generator at 200 loci (~32 k calls, pattern 6-100, 2-20 units, flank 150) in ONE batch; oracle on every 50th call."""
    from advntr_amd import _lib, workloads
    from advntr_amd.pomegranate import device_models
    loci, reads, which = workloads.make_c2_parallel(200, seed=20240602, build=True)
    dms = device_models([l.model for l in loci])
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch(dms, bases, off, which, want_paths=True)
    n = _check_calls_against_oracle(loci, reads, which, logp, summ, paths, every=50)
    assert n >= 600
    # every call's log-probability against the oracle as well (threads over calls, per locus)
    for k in range(0, len(loci), 9):
        idx = np.flatnonzero(which == k)
        kb, ko = _lib.encode_reads([reads[i] for i in idx])
        assert np.array_equal(_oracle(loci[k].model).viterbi_many_threads(kb, ko, 16), logp[idx]), k


def test_c2_full_size_properties():
    """C2 at BASELINE's size: 6 719 loci, ~1.07 M calls in one batch."""
    from advntr_amd import _lib, workloads
    from advntr_amd.pomegranate import device_models
    loci, reads, which = workloads.make_c2_parallel(6719, seed=20240602, build=True)
    assert len(loci) == 6719 and 1.0e6 < len(reads) < 1.15e6
    dms = device_models([l.model for l in loci])
    logp, summ, kernels = _properties(dms, reads, which, _lib.FLAG_ANTIDIAGONAL, 400001)
    assert kernels[0][0] == "viterbi_rows_kernel<5, 2>" and kernels[0][1] == len(reads)     # what the bench line runs on
    # loci are independent: a locus scored alone gives the records it got inside the big batch
    for k in (0, 3333, 6718):
        idx = np.flatnonzero(which == k)
        kb, ko = _lib.encode_reads([reads[i] for i in idx])
        lp, sm, _ = _lib.viterbi_batch([dms[k]], kb, ko, np.zeros(len(idx), np.int32))
        assert np.array_equal(lp, logp[idx]) and np.array_equal(sm, summ[idx])


def test_c4_50_loci_vs_oracle():
    """C4's generator at 50 PacBio loci (flank 100, error 0.3, 20 trimmed spanning reads each, 12 % noise): every call's
    log-probability against the oracle, path and RU count on every 25th call."""
    from advntr_amd import _lib, workloads
    from advntr_amd.pomegranate import device_models
    loci, reads, which = workloads.make_c4(50, seed=20240603)
    workloads.build_models(loci)
    dms = device_models([l.model for l in loci])
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch(dms, bases, off, which, want_paths=True)
    assert _check_calls_against_oracle(loci, reads, which, logp, summ, paths, every=25) == 40
    for k in range(0, 50, 5):
        idx = np.flatnonzero(which == k)
        kb, ko = _lib.encode_reads([reads[i] for i in idx])
        assert np.array_equal(_oracle(loci[k].model).viterbi_many_threads(kb, ko, 20), logp[idx]), k
    # genotype-level: the dominant copy number of a locus's spanning reads is within the planted +-20 % band
    ru = summ[:, _lib.SUM_RU]
    assert np.all(ru > 0)


def test_c4_full_size_properties():
    """C4 at BASELINE's size: 8 960 PacBio loci x 20 reads (179 200 calls, mean ~750 bases) in one batch."""
    from advntr_amd import _lib, workloads
    from advntr_amd.pomegranate import device_models
    loci, reads, which = workloads.make_c4(8960, seed=20240603)
    workloads.build_models(loci)
    assert len(reads) == 8960 * 20
    dms = device_models([l.model for l in loci])
    logp, summ, kernels = _properties(dms, reads, which, _lib.FLAG_ANTIDIAGONAL, 70001)
    assert any(k[0] == "viterbi_rows_long_kernel<5>" for k in kernels)


def _bench(args, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         check=True, timeout=timeout).stdout
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_c3_two_ranks_gather_equals_one_rank(tmp_path):
    """bench.py --gpus 2 on the C3 set (300 loci here): two ranks (sharing this box's one GPU through the host
    communicator) score their LPT shares and rank 0 gathers every call's record; the result equals the 1-rank run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    common = ["--workload", "c3", "--loci", "300", "--steps", "2", "--warmup", "1", "--no-cpu"]
    d1 = _bench(common + ["--gpus", "1", "--dump-records", one])
    d2 = _bench(common + ["--gpus", "2", "--dump-records", two], env=dict(os.environ, ADVNTR_DIST_BACKEND="host"))
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    assert d2["config"]["comm"] == "host" and len(d2["config"]["per_rank"]) == 2
    assert d2["comm"] == "host" and d2["rccl"] is False
    assert sum(d2["config"]["calls_per_rank"]) == d1["config"]["calls_this_rank"]
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["ids"], np.arange(len(a["ids"]))) and np.array_equal(b["ids"], a["ids"])
    assert np.array_equal(a["logp"], b["logp"]) and np.array_equal(a["summary"], b["summary"])
    # one metric string family per workload for every N, and the 2-rank line carries its own one-rank baseline: rank 0 scored
    # the WHOLE set alone after the timed region (here both ranks share one GPU, so the "efficiency" is about one half)
    assert d1["metric"].replace("over 1 GPUs", "over 2 GPUs") == d2["metric"]
    n1 = d2["same_workload_n1"]
    assert n1["calls"] == d1["config"]["calls_this_rank"] and n1["value"] > 0 and n1["kernel_ms"] > 0 and n1["loci"] == 300
    assert d2["efficiency_measured"] == d2["value"] / (2 * n1["value"]) and 0.1 < d2["efficiency_measured"] < 1.5
    assert d2["summary"]["efficiency_measured"] == d2["efficiency_measured"] and list(d2.keys())[0] == "summary"
    assert all(r["loop_ms"] > 0 and r["host"]["cpu_quota_cores"] >= 1 for r in d2["config"]["per_rank"])
    assert d1["efficiency_measured"] == 1.0 and d1["same_workload_n1"]["where"] == "this line"


def test_emulated_rank_processes_run_the_n_rank_line():
    """--emulate-ranks 2 --processes: two rank PROCESSES share this box's GPU through the host communicator and run the 2-rank line
    itself -- what one GPU can show of the host side of an N-rank job: every rank sleeps in its device waits, takes its share of the
    host threads, and reports the control group's throttle counters around its timed region."""
    d = _bench(["--workload", "c3", "--loci", "120", "--steps", "3", "--warmup", "1", "--no-cpu", "--emulate-ranks", "2", "--processes"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["comm"] == "host" and d["emulated_ranks"]["ranks"] == 2
    assert d["same_workload_n1"]["value"] is None and d["efficiency_measured"] is None          # (--no-n1: ranks share one GPU)
    ranks = d["config"]["per_rank"]
    assert len(ranks) == 2 and sum(r["calls"] for r in ranks) == sum(d["config"]["calls_per_rank"])
    for r in ranks:
        assert r["loop_ms"] > 0 and r["kernel_ms"] > 0 and r["host"]["nr_throttled_delta"] is not None
        assert r["host"]["cpu_quota_cores"] >= 1
    assert d["summary"]["throttled_periods_in_timed_region"] is not None


def test_c4_two_ranks_gather_equals_one_rank(tmp_path):
    """bench.py --workload c4 --gpus 2 (BASELINE config 5 in its sharded form, 60 loci here): ONE PacBio locus set, whole
    loci to ranks by the planned work, each rank's share scored by the long-read kernel, every call's record gathered to rank
    0; the result equals the 1-rank run record for record."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    common = ["--workload", "c4", "--loci", "60", "--steps", "2", "--warmup", "1", "--no-cpu"]
    d1 = _bench(common + ["--gpus", "1", "--dump-records", one])
    d2 = _bench(common + ["--gpus", "2", "--dump-records", two], env=dict(os.environ, ADVNTR_DIST_BACKEND="host"))
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d1["scaling"] == d2["scaling"] == "strong"
    assert d2["config"]["comm"] == "host" and len(d2["config"]["per_rank"]) == 2
    assert sum(d2["config"]["calls_per_rank"]) == d1["config"]["calls_this_rank"] == 1200
    assert sum(r["cells"] for r in d2["config"]["per_rank"]) > 0
    assert d2["config"]["kernel"].startswith("viterbi_rows_long_kernel")
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["ids"], np.arange(1200)) and np.array_equal(b["ids"], a["ids"])
    assert np.array_equal(a["logp"], b["logp"]) and np.array_equal(a["summary"], b["summary"])
    assert np.all(a["summary"][:, 0] > 0)


def test_c4_scale_rehearsal_reports_the_planned_and_the_actual_imbalance():
    """bench.py --workload c4 --emulate-ranks 4: the shares are the partition `--gpus 4` makes from workloads.c4_plan (whole
    loci, every call once); the record carries the planned imbalance, the imbalance of the ACTUAL work and the spread of the
    per-locus work (more than an order of magnitude on this set)."""
    d = _bench(["--workload", "c4", "--loci", "120", "--steps", "2", "--warmup", "1", "--no-cpu", "--emulate-ranks", "4"])
    r = d["scale_rehearsal"]
    assert r["projection"] is True and r["ranks"] == 4 and len(r["shares"]) == 4
    assert sum(x["calls"] for x in r["shares"]) == 2400 and sum(x["loci"] for x in r["shares"]) == 120
    assert 1.0 <= r["load_imbalance_max_over_mean"] < 1.05 and 1.0 <= r["actual_cells_imbalance_max_over_mean"] < 1.15
    assert r["per_locus_work_max_over_min"] > 5 and "c4_plan" in r["partitioned_by"]
    from advntr_amd import sharding, workloads
    plan = workloads.c4_plan(120, seed=20240603)
    parts = sharding.partition_loci([c * (n + 1) * m for c, n, m in plan], 4, [0.99, 1, 1, 1])      # bench.py --root-capacity
    assert sorted(x["loci"] for x in r["shares"]) == sorted(len(p) for p in parts)


def test_scale_rehearsal_partitions_the_set_like_the_multi_gpu_job():
    """bench.py --workload c3 --emulate-ranks 4 (one process, one GPU): the shares are the LPT partition `--gpus 4` makes --
    whole loci, every call exactly once --, each runs as its own resident batch with the multi-GPU launch parameters, and the
    record is labelled a projection."""
    d = _bench(["--workload", "c3", "--loci", "240", "--steps", "2", "--warmup", "1", "--no-cpu", "--emulate-ranks", "4"])
    r = d["scale_rehearsal"]
    assert r["projection"] is True and r["ranks"] == 4 and len(r["shares"]) == 4
    assert sum(x["calls"] for x in r["shares"]) == d["config"]["calls_this_rank"] == r["whole_set"]["calls"]
    assert sum(x["loci"] for x in r["shares"]) == 240
    assert all(x["loop_ms"] > 0 and x["kernel_ms"] > 0 and x["loop_ms_two_passes_in_flight"] > 0 for x in r["shares"])
    assert 0.0 < r["projected_efficiency"] <= 1.5 and r["load_imbalance_max_over_mean"] >= 1.0
    # the strong-scaling lines keep two passes in flight (two copies of the device batch): the line says so, both projections
    # are in the record, and a pass of either copy gives the records the line's own check compares
    assert d["config"]["passes_in_flight"] == 2 and r["passes_in_flight"] == 2
    assert 0.0 < r["projected_efficiency_one_pass_in_flight"] <= 1.5 and r["whole_set"]["loop_ms_two_passes_in_flight"] > 0
    one = _bench(["--workload", "c3", "--loci", "240", "--steps", "2", "--warmup", "1", "--no-cpu", "--in-flight", "1"])
    assert one["config"]["passes_in_flight"] == 1 and one["config"]["calls_this_rank"] == d["config"]["calls_this_rank"]
    from advntr_amd import sharding, workloads
    plan = workloads.c2_plan(240, seed=20240602)
    parts = sharding.partition_loci([c * 151 * m for c, m in plan], 4, [0.99, 1, 1, 1])
    assert sorted(x["calls"] for x in r["shares"]) == sorted(int(sum(plan[int(k)][0] for k in p)) for p in parts)


def test_reserved_workgroups_cover_one_pass_only():
    """advntr_batch_reserve_next (what the gather of a multi-GPU job asks for): same results, and a small batch -- whose grid
    does not fill the device -- is not cut down to a single workgroup."""
    from advntr_amd import _lib, workloads
    locus = workloads.s300()
    reads = workloads.make_reads(np.random.default_rng(3), locus, 3000, 150)
    bases, off = _lib.encode_reads(reads)
    batch = _lib.DeviceBatch([locus.model.device_model()], bases, off, np.zeros(len(reads), np.int32))
    batch.run()
    want = batch.fetch()
    batch.reserve_next(8)
    batch.run()
    got = batch.fetch()
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    t_plain = batch.run_timed(5)
    batch.reserve_next(8)
    t_reserved = batch.run_timed(1)
    assert t_reserved < 3 * t_plain + 1.0
    with pytest.raises(_lib.EngineError):
        batch.reserve_next(-1)
    batch.close()


def test_c3_rccl_communicator_world_size_1(tmp_path):
    """The RCCL path with the one rank a 1-GPU box allows (what torch.distributed.run --nproc-per-node 1 sets up): unique
    id through the rendezvous, ncclCommInitRank, the small collectives, the overlapped gather of the result records inside
    the timed loop (bench.py asserts the gathered copy equals the engine's) and the ragged byte gather."""
    rec, ref = str(tmp_path / "rccl.npz"), str(tmp_path / "plain.npz")
    common = ["--workload", "c3", "--loci", "120", "--steps", "3", "--warmup", "1", "--no-cpu"]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641",
               ADVNTR_RDZV_DIR=str(tmp_path / "rdzv"))
    d = _bench(common + ["--dump-records", rec], env=env)
    assert d["config"]["comm"] == "rccl" and d["config"]["world_size_seen_by_comm"] == 1
    assert d["comm"] == "rccl" and d["rccl"] is True                      # at the top level of the line
    assert d["config"]["per_rank"][0]["gather_ms"] is not None and d["config"]["per_rank"][0]["gather_ms"] >= 0.0
    # the line is self-contained (see tests/test_bench_launch.py): summary first, the same-workload one-rank figure (at world
    # size 1: this line), per-rank loop / kernel / gather times and cells, the host's CPU quota and throttle counters
    assert list(d.keys())[0] == "summary" and d["summary"]["value"] == d["value"] and d["summary"]["max_gather_ms"] is not None
    assert d["efficiency_measured"] == 1.0 and d["same_workload_n1"]["value"] == d["value"]
    assert "--workload c3 --gpus 1" in d["same_workload_n1"]["command"] and "partitioned over 1 GPUs" in d["metric"]
    r0 = d["config"]["per_rank"][0]
    assert r0["loop_ms"] > 0 and r0["kernel_ms"] > 0 and r0["cells"] > 0 and r0["loop_ms"] <= r0["loop_ms_per_step"] + 1e-6
    assert r0["host"]["cpu_quota_cores"] >= 1 and r0["host"]["nr_throttled_delta"] is not None
    assert d["host"]["cpu_quota_cores"] >= 1
    _bench(common + ["--dump-records", ref])
    a, b = np.load(rec), np.load(ref)
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["logp"], b["logp"]) and np.array_equal(a["summary"], b["summary"])
    from advntr_amd import comm
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT="29642", ADVNTR_RDZV_DIR=str(tmp_path / "rdzv2"))
    try:
        c = comm.init_from_env()
        assert c.backend == "rccl" and (c.rank, c.world) == (0, 1)
        c.barrier()
        assert c.allreduce_max(2.5) == 2.5 and c.allgather_i64(41) == [41]
        assert c.gather_bytes(b"per-locus rows") == [b"per-locus rows"] and c.gather_bytes(b"") == [b""]
        c.close()
    finally:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ADVNTR_RDZV_DIR"):
            os.environ.pop(k, None)


def test_both_strands_flag_equals_explicit_reverse_complements():
    """ADVNTR_FLAG_BOTH_STRANDS (reverse complements made on the device, calls n .. 2n-1) == scoring the forward reads and
    their host-made reverse complements explicitly -- what process_unmapped_read does (vntr_finder.py:239-242); ragged
    lengths, several loci, paths included, both the one-shot and the device-resident API."""
    from advntr_amd import _lib, workloads, vntr_finder
    from advntr_amd.pomegranate import device_models
    rng = np.random.default_rng(808)
    loci = [workloads.make_locus(rng, 150, int(L), vntr_finder.get_copies_for_hmm(150, int(L))) for L in (7, 23, 61)]
    workloads.build_models(loci)
    dms = device_models([l.model for l in loci])
    reads, which = [], []
    for k, loc in enumerate(loci):
        for n in (150, 150, 150, 149, 90, 64, 33, 1, 200, 301):
            reads += workloads.make_reads(rng, loc, 3, n, locus_fraction=0.7)
            which += [k] * 3
    which = np.asarray(which, np.int32)
    bases, off = _lib.encode_reads(reads)
    lp2, sm2, paths2 = _lib.viterbi_batch(dms, bases, off, which, flags=_lib.FLAG_BOTH_STRANDS, want_paths=True)
    n = len(reads)
    assert len(lp2) == 2 * n and sm2.shape == (2 * n, 8) and len(paths2) == 2 * n
    both = reads + [vntr_finder.reverse_complement(s) for s in reads]
    b2, o2 = _lib.encode_reads(both)
    lp, sm, paths = _lib.viterbi_batch(dms, b2, o2, np.concatenate([which, which]), want_paths=True)
    assert np.array_equal(lp, lp2) and np.array_equal(sm, sm2) and paths == paths2
    B = _lib.DeviceBatch(dms, bases, off, which, flags=_lib.FLAG_BOTH_STRANDS)
    B.run()
    lp3, sm3 = B.fetch()
    B.close()
    assert np.array_equal(lp3, lp) and np.array_equal(sm3, sm)


def test_pipelined_genotyping_equals_stage_by_stage():
    """vntr_finder.genotype_loci_pipelined (model build / upload / read encoding of locus piece k+1 on a host thread while
    piece k is scored) gives the genotypes, probabilities and read counts of the stage-by-stage route on the same loci --
    incl. a locus without reads, reads holding N, more pieces than loci -- and its per-read scores are the oracle's."""
    from advntr_amd import hmm_utils, vntr_finder, workloads
    loci, reads, which, counts = workloads.make_c2_parallel(37, seed=77, build=False, workers=2, return_counts=True)
    desc = [(l.left, l.right, l.units, l.copies) for l in loci]
    first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
    cand = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
    cand[5] = []                                           # a locus nobody mapped to
    cand[7] = cand[7][:3] + ["ACGTN" * 30] + cand[7][3:]   # dropped before scoring (vntr_finder.py:237)
    models = hmm_utils.build_read_matcher_models(desc)
    plain = vntr_finder.genotype_loci(models, cand)
    for chunks, extra in ((1, {}), (4, {}), (64, {}), (4, dict(ramp=0, stage_threads=(2, 1, 1))), (3, dict(ramp=9))):
        T = {}
        piped = vntr_finder.genotype_loci_pipelined(desc, cand, chunks=chunks, timings=T, **extra)
        assert len(piped) == len(plain) == 37
        for a, b in zip(plain, piped):
            assert a.copy_numbers == b.copy_numbers
            assert a.maximum_likelihood == b.maximum_likelihood
            assert (a.recruited_reads_count, a.spanning_reads_count, a.flanking_reads_count) == \
                (b.recruited_reads_count, b.spanning_reads_count, b.flanking_reads_count)
        assert T["total"] > 0 and set(T) >= {"build_models", "upload_models", "encode_reads", "score_recruit"}
    assert plain[5].copy_numbers is None
    assert sum(g.copy_numbers is not None for g in plain) >= 30
    # a failure in the preparation thread reaches the caller
    bad = list(desc)
    bad[20] = (desc[20][0], desc[20][1], ["ACGTXX"], 3)
    with pytest.raises(Exception):
        vntr_finder.genotype_loci_pipelined(bad, cand, chunks=4)


def test_resident_log_probability_equals_one_shot_and_oracle():
    """advntr_batch_forward on a resident batch (what bench.py times) == the one-shot advntr_forward_batch, within 1e-9
    relative of the oracle's log-domain forward; a Viterbi run on the same batch afterwards is unaffected."""
    from advntr_amd import _lib, workloads
    locus = workloads.s300()
    reads = workloads.make_reads(np.random.default_rng(5), locus, 300, 150) + ["ACGT", "A" * 200]
    bases, off = _lib.encode_reads(reads)
    dm = locus.model.device_model()
    which = np.zeros(len(reads), np.int32)
    B = _lib.DeviceBatch([dm], bases, off, which)
    B.run()
    v_logp, v_summ = B.fetch()
    B.forward()
    f_logp, f_summ = B.fetch()
    assert np.array_equal(f_summ, v_summ)                   # summaries are the Viterbi run's
    one_shot = _lib.forward_batch([dm], bases, off, which)
    assert np.array_equal(f_logp, one_shot)
    assert B.forward_timed(2) > 0
    O = _oracle(locus.model)
    for i in range(0, len(reads), 11):
        want = O.forward(bases[off[i]:off[i + 1]])
        assert abs(f_logp[i] - want) <= 1e-9 * max(1.0, abs(want))
        assert f_logp[i] >= v_logp[i] - 1e-9                # the sum over paths is at least the best path
    B.run()
    again, _ = B.fetch()
    assert np.array_equal(again, v_logp)
    B.close()


def test_device_recruit_equals_the_rule_applied_on_the_host():
    """advntr_batch_recruit (strand choice, recruit_read, more than two repeat bases -- vntr_finder.py:179-190, 242-254 -- applied
    to the records in HBM, survivors compacted in read order) against the same rule in numpy on the downloaded records
    (recruit_mask, which the golden verdicts pin): loci with and without a trained score, both strands and forward only, reads
    of several lengths, a batch whose last wavefront is partial, a batch nobody survives."""
    from advntr_amd import _lib, vntr_finder, workloads
    from advntr_amd.pomegranate import device_models
    rng = np.random.default_rng(4242)
    loci = [workloads.make_locus(rng, 150, int(L), vntr_finder.get_copies_for_hmm(150, int(L))) for L in (9, 14, 33, 57)]
    workloads.build_models(loci)
    dms = device_models([l.model for l in loci])
    reads, which = [], []
    for k, loc in enumerate(loci):
        for n, cnt in ((150, 230), (149, 17), (100, 40), (60, 9)):
            rs = workloads.make_reads(rng, loc, cnt, n, locus_fraction=0.7)
            reads += [r if rng.random() < 0.5 else vntr_finder.reverse_complement(r) for r in rs]
            which += [k] * cnt
    which = np.asarray(which, np.int32)
    bases, off = _lib.encode_reads(reads)
    lens = np.diff(off)
    nf = len(reads)
    assert nf % 64 != 0
    for scaled in (None, [-1.0, None, -0.9, 0], [-0.5, -0.5, -0.5, -0.5]):
        for both in (True, False):
            B = _lib.DeviceBatch(dms, bases, off, which, flags=_lib.FLAG_BOTH_STRANDS if both else 0)
            B.run()
            logp, summ = B.fetch()
            idx, lp, sm, rev = B.recruit(scaled, 2)
            B.close()
            if both:
                use_rev = logp[:nf] < logp[nf:]
                c_lp, c_sm = np.where(use_rev, logp[nf:], logp[:nf]), np.where(use_rev[:, None], summ[nf:], summ[:nf])
            else:
                use_rev, c_lp, c_sm = np.zeros(nf, bool), logp, summ
            sc = np.array([np.nan if (s is None or s == 0) else s for s in (scaled or [None] * 4)], np.float64)
            want = vntr_finder.recruit_mask(c_lp, c_sm, lens, sc[which] * lens) & (c_sm[:, _lib.SUM_REPEAT_BP] > 2)
            keep = np.flatnonzero(want)
            assert np.array_equal(idx, keep), (scaled, both, len(idx), len(keep))
            assert np.array_equal(lp, c_lp[keep]) and np.array_equal(sm, c_sm[keep]) and np.array_equal(rev, use_rev[keep])
            if scaled is None and both:
                assert 50 < len(keep) < nf and rev.any() and not rev.all()
    # nobody survives / an empty batch
    junk = [workloads.rand_seq(rng, 150) for _ in range(70)]
    jb, jo = _lib.encode_reads(junk)
    B = _lib.DeviceBatch(dms[:1], jb, jo, np.zeros(70, np.int32), flags=_lib.FLAG_BOTH_STRANDS)
    B.run()
    idx, lp, sm, rev = B.recruit(None, 2)
    assert len(idx) == len(lp) == len(sm) == len(rev) == 0
    with pytest.raises(ValueError):
        B.recruit([1.0, 2.0], 2)
    B.close()
    # the selection the pipelined genotyper makes == score_reads_arrays + recruit_mask
    models = [l.model for l in loci]
    lists = [[r for r, w in zip(reads, which) if w == k] for k in range(4)]
    lists[2] = lists[2][:5] + ["ACGTN" * 30] + lists[2][5:]
    prep = vntr_finder._prepare_reads(lists)
    res = vntr_finder._score_prepared(models, prep, [-1.0, None, -0.9, None])
    pos, locus, sm, rev = vntr_finder._select_prepared(models, prep, [-1.0, None, -0.9, None])
    keep = np.flatnonzero(res["recruited"] & (res["summary"][:, _lib.SUM_REPEAT_BP] > 2))
    assert np.array_equal(pos, keep) and np.array_equal(locus, res["locus"][keep])
    assert np.array_equal(sm, res["summary"][keep]) and np.array_equal(rev, res["reversed"][keep])
