"""bench.py's multi-rank entry point on CPU: `--gpus N` really starts N ranks (as child processes, or under
torch.distributed.run as the driver does), they rendezvous, derive the same C3 plan and rank 0 prints one line with
n_gpus = N.  --dry-run stops before any GPU work, so this runs without a GPU; the full path runs in the GPU suite."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--loci", "48"],
                         cwd=ROOT, stdout=subprocess.PIPE, check=True, timeout=300).stdout
    d = _line(out)
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["scaling"] == "strong"
    cfg = d["config"]
    assert cfg["workload"] == "c3" and cfg["comm"] == "host"
    assert cfg["calls_seen_by_ranks"] == cfg["calls_per_rank"] and len(cfg["calls_per_rank"]) == 2      # both ranks reported in
    assert sum(cfg["loci_per_rank"]) == 48 and cfg["load_imbalance_max_over_mean"] < 1.05
    # a multi-GPU line is self-contained: it names the one-rank run of the SAME workload it compares with, carries that run's rate
    # (measured inside the N-rank job; not in a dry run) and the efficiency derived from it, per-rank records with the host's CPU
    # quota and throttle counters, and a compact summary FIRST so that a truncated record keeps the headline numbers
    assert list(d.keys())[0] == "summary"
    n1 = d["same_workload_n1"]
    assert "--workload c3 --gpus 1" in n1["command"] and "--loci 48" in n1["command"] and "efficiency_measured" in d
    assert set(d["summary"]) >= {"value", "n_gpus", "same_workload_n1_value", "efficiency_measured", "slowest_rank_loop_ms",
                                 "max_gather_ms", "throttled_periods_in_timed_region"}
    ranks = cfg["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["calls"] for r in ranks] == cfg["calls_per_rank"]
    for r in ranks:
        assert set(r) >= {"loop_ms", "kernel_ms", "gather_ms", "cells", "host"}
        assert set(r["host"]) >= {"cpu_quota_cores", "nr_throttled_delta", "throttled_usec_delta"}
        assert 1 <= r["host"]["cpu_quota_cores"] <= max(1, (os.cpu_count() or 1) // 2)       # a rank's share of the host's CPUs
    assert len(json.dumps(d["summary"])) < 1000


def test_c4_gpus_2_is_one_set_partitioned():
    """--workload c4 --gpus 2 (BASELINE config 5): ONE PacBio locus set split by the planned work, 20 calls per locus."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "c4",
                          "--loci", "40"], cwd=ROOT, stdout=subprocess.PIPE, check=True, timeout=300).stdout
    d = _line(out)
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and cfg["workload"] == "c4"
    assert sum(cfg["loci_per_rank"]) == 40 and sum(cfg["calls_per_rank"]) == 800 == sum(cfg["calls_seen_by_ranks"])
    assert cfg["load_imbalance_max_over_mean"] < 1.05 and cfg["per_locus_work_max_over_min"] > 3
    assert "--workload c4 --gpus 1" in d["same_workload_n1"]["command"] and list(d.keys())[0] == "summary"


def test_gpus_3_under_torch_distributed_run():
    """The driver's launch line: torch is only the launcher; the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "bench.py"),
                          "--gpus", "3", "--dry-run", "--loci", "30"], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True, timeout=300).stdout
    d = _line(out)
    assert d["n_gpus"] == 3 and len(d["config"]["calls_per_rank"]) == 3 and sum(d["config"]["loci_per_rank"]) == 30


def test_single_process_default_is_c1():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], cwd=ROOT, stdout=subprocess.PIPE,
                         check=True, timeout=300).stdout
    d = _line(out)
    assert d["n_gpus"] == 1 and d["config"]["workload"] == "c1" and d["scaling"] == "weak"


def test_a_rank_that_exits_early_ends_the_job_with_an_error():
    """One rank leaves before the rendezvous, the other waits for it: the launcher sees the failure, ends the waiting
    rank (a child it started) and exits non-zero, long before any rendezvous timeout."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--loci", "24",
                        "--fault", "exit:1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-400:])
    assert b"rank 1 exited with status 3" in p.stderr
    assert time.time() - t0 < 120
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]       # no line that looks like a result


def test_a_hanging_rank_is_ended_at_the_launch_deadline():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--loci", "24",
                        "--fault", "hang:0", "--launch-timeout", "8"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 124, (p.returncode, p.stderr[-400:])
    assert b"--launch-timeout" in p.stderr


def test_emulate_ranks_as_processes_is_the_n_rank_line():
    """--emulate-ranks N --processes: N rank processes (sharing the box's GPU through the host communicator when there is one) run
    the N-rank line itself, so that one box shows the host side of an N-rank job -- every rank with its share of the host's CPUs and
    the control group's throttle counters around its timed region."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--emulate-ranks", "3", "--processes",
                          "--dry-run", "--loci", "30"], cwd=ROOT, stdout=subprocess.PIPE, check=True, timeout=300).stdout
    d = _line(out)
    assert d["n_gpus"] == 3 and d["config"]["comm"] == "host" and sum(d["config"]["loci_per_rank"]) == 30
    assert len(d["config"]["per_rank"]) == 3 and all("nr_throttled_delta" in r["host"] for r in d["config"]["per_rank"])
    from benchlib.cli import emulation_argv
    assert emulation_argv(["--workload", "c3", "--emulate-ranks", "8", "--processes", "--steps", "5", "--gpus=1"], 8) == \
        ["--workload", "c3", "--steps", "5", "--gpus", "8", "--no-n1"]


def test_line_order_summary_first_bulky_records_last():
    """The line a driver may keep only the head of: `summary` first (every headline number, < 1 KB), the contract's keys next, the
    per-share lists of the rehearsals last; keys nobody listed keep their place in between."""
    from benchlib.main import line_summary, order_line
    out = {"metric": "m", "value": 11.9e6, "unit": "reads/s", "n_gpus": 1, "ms_per_step": 8.4, "config": {"workload": "C1"},
           "scale_rehearsal": {"projected_efficiency": 0.98, "projected_efficiency_one_pass_in_flight": 0.95, "shares": [{}] * 8},
           "roofline": {"frac": 0.32, "kernel_ms": 8.4}, "s300": {"value": 37.5e6, "frac": 0.237, "kernel_ms": 2.54},
           "c2": {"value": 11.9e6, "kernel_ms": 90.4, "roofline": {"frac": 0.33}}, "something_new": 1,
           "end_to_end": {"total_s": 0.21}, "illumina_pipeline": {"total_s": 0.54}, "cpu_baseline": {"value": 716.0},
           "host": {"nr_throttled_delta": 0}}
    line = order_line(out)
    keys = list(line)
    assert keys[0] == "summary" and keys[-1] == "scale_rehearsal" and keys.index("metric") == 1
    assert keys.index("roofline") < keys.index("config") < keys.index("s300") < keys.index("something_new") < keys.index("scale_rehearsal")
    s = line["summary"]
    assert s["value"] == 11.9e6 and s["roofline_frac"] == 0.32 and s["s300"] == {"value": 37.5e6, "frac": 0.237, "kernel_ms": 2.54}
    assert s["c2"]["frac"] == 0.33 and s["end_to_end_total_s"] == 0.21 and s["illumina_pipeline_total_s"] == 0.54
    assert s["projected_efficiency_c3_8_ranks"] == 0.98 and s["cpu_baseline_value"] == 716.0
    assert s["throttled_periods_in_timed_region"] == 0 and len(json.dumps(s)) < 1000
    multi = line_summary({"value": 80e6, "n_gpus": 8, "same_workload_n1": {"value": 11.7e6}, "efficiency_measured": 0.85,
                          "config": {"per_rank": [{"loop_ms": 11.0, "kernel_ms": 10.5, "gather_ms": 0.4, "host": {"nr_throttled_delta": 0}},
                                                  {"loop_ms": 11.6, "kernel_ms": 10.9, "gather_ms": 0.7, "host": {"nr_throttled_delta": 2}}]}})
    assert multi["efficiency_measured"] == 0.85 and multi["same_workload_n1_value"] == 11.7e6
    assert multi["slowest_rank_loop_ms"] == 11.6 and multi["max_gather_ms"] == 0.7 and multi["throttled_periods_in_timed_region"] == 2


def test_committed_bench_line_keeps_the_contract():
    """The N = 1 line committed under profiles/ (taken on an MI355X with the driver's flags) carries what the contract asks of it:
    the standard keys, `roofline` (bound / achieved / peak / unit / frac / traffic, achieved < peak, frac = achieved / peak, the kernel
    not slower than the step), `cpu_baseline` (value / unit / cores / kind / sample), a `summary` that agrees with the records it
    condenses, and no model vocabulary in `config`."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_c1_bench_driver_flags.json")))
    assert list(d)[0] == "summary"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"] and "REF150" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["achieved"] < r["peak"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] is not None and r["traffic"] < r["algorithmic_gb_per_launch"]
    assert abs(r["achieved"] - 215108 * 100000 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.01 and abs(d["value"] - 100000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "reads/s" and c["value"] > 0 and "8000 reads" in c["sample"]
    a = d["cpu_baseline_all_cores"]
    assert a["cores"] == 16 and a["host_threads_available"] == 256 and a["nr_throttled"] is not None     # the cgroup quota, not the host
    s = d["summary"]
    assert s["value"] == d["value"] and s["roofline_frac"] == r["frac"] and s["s300"]["kernel_ms"] == d["s300"]["kernel_ms"]
    assert s["c2"]["frac"] == d["c2"]["roofline"]["frac"] and s["illumina_pipeline_total_s"] == d["illumina_pipeline"]["total_s"]
    assert d["illumina_pipeline"]["genotypes_identical_to_stage_by_stage"] and d["illumina_pipeline"]["reference_filter"]["stdout_identical"]
    assert d["illumina_pipeline"]["fasta_reads"] >= 10000000 and len(json.dumps(s)) < 1500
