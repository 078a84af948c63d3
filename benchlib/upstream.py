"""BASELINE config 5 (PacBio) and the two stages upstream of the scoring path, as sub-records of the N = 1 line."""
import os
import subprocess
import time

import numpy as np

from .common import CLOCK_GHZ, HBM_PEAK_GBPS, ROOT, SIMDS, cpu_model_name, load_json, pmc_section
from .passes import two_in_flight_ms
from .rehearsal import scale_rehearsal

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
