"""Shared by the bench modules: the roofline constants, SURVEY 8(d)'s algorithmic bytes, the CPU baseline (the oracle is the
checker, timed here as the reported baseline), and what the committed profiles say about a launch."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS, CLOCK_GHZ = 1024, 2.4      # 256 CUs x 4 SIMDs, 2.4 GHz


def algorithmic_bytes(n, m):
    """B = n [read] + (n+1)*m [1-byte back-pointer per cell] + (n+m) [traceback reads] + 32."""
    return n + (n + 1) * m + (n + m) + 32

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
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
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
