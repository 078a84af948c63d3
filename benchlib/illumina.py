"""BASELINE config 2 (6 719 Illumina loci) on one GPU: the `end_to_end`, `c2` and `illumina_pipeline` sub-records of the N = 1 line."""
import os
import subprocess
import tempfile
import time

import numpy as np

from .common import HBM_PEAK_GBPS, ROOT, cpu_model_name, load_json, pmc_section
from .hostinfo import HostRegion
from .passes import two_in_flight_ms
from .rehearsal import scale_rehearsal


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
    P, totals, throttled = None, [], []
    for _ in range(3):
        Pk = {}
        with HostRegion() as h:
            piped = vntr_finder.genotype_loci_pipelined(desc, candidates, timings=Pk)
        totals.append(Pk["total"])
        throttled.append(h.record())
        if P is None or Pk["total"] < P["total"]:
            P = Pk
    same = sum(a.copy_numbers == b.copy_numbers and a.recruited_reads_count == b.recruited_reads_count
               for a, b in zip(plain, piped))
    assert same == n_loci, "pipelined and stage-by-stage genotypes differ on %d loci" % (n_loci - same)
    e2e = {"loci": n_loci, "candidate_reads": n_cand, "viterbi_calls": 2 * n_cand, "recruited_reads": recruited,
           "loci_with_genotype": sum(g.copy_numbers is not None for g in piped),
           "value": 2 * n_cand / P["total"], "unit": "calls/s", "total_s": P["total"], "total_s_of_each_pass": totals,
           # why the passes differ: a pass whose host stages exhaust the control group's CPU quota in a 100-ms period has ALL its
           # threads stopped for the rest of that period (the one that launches kernels included)
           "host_of_each_pass": [{k: t.get(k) for k in ("nr_throttled_delta", "throttled_usec_delta", "process_cpu_s", "wall_s")}
                                 for t in throttled],
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
# `advntr genotype` on ONE timeline: file bytes -> prefilter -> selection -> scoring -> recruit -> genotypes -> rows
# ------------------------------------------------------------------------------------------------
REF_FILTER = os.path.join(ROOT, "oracle", "_ref", "adVNTR-Filtering")


def illumina_pipeline_input(workloads, c2_input, args):
    """Made BEFORE the GPU is touched: the FASTA bytes (the C2 loci's candidate reads planted among background reads), the keyword
    lines, and -- on a bounded head of the same file -- the REFERENCE's own filter binary started as a child (oracle/_ref,
    kind "reference"), collected when the record is made."""
    loci, reads, which, counts = c2_input[:4]
    first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
    candidates = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
    t0 = time.perf_counter()
    lines, fasta, rec_len, planted = workloads.make_illumina_pipeline_workload(loci, candidates, args.pipeline_reads)
    inp = {"loci": loci, "lines": lines, "fasta": fasta, "rec_len": rec_len, "planted": planted,
           "n_candidates_planted": int(len(planted)), "gen_s": time.perf_counter() - t0, "ref": None}
    if os.path.exists(REF_FILTER) and not args.no_cpu:
        sample = min(len(fasta) // rec_len, 50000)
        d = tempfile.mkdtemp(prefix="advntr_pipeline_ref_")
        kw, fa = os.path.join(d, "kw.txt"), os.path.join(d, "head.fa")
        with open(kw, "w") as fh:
            fh.write("".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines))
        with open(fa, "wb") as fh:
            fh.write(fasta[:sample * rec_len])
        t0 = time.perf_counter()
        with open(kw) as fin, open(fa + ".out", "wb") as fout:
            child = subprocess.Popen([REF_FILTER, fa], stdin=fin, stdout=fout)
        inp["ref"] = {"dir": d, "child": child, "t0": t0, "sample": sample, "out": fa + ".out", "kw": kw}
    return inp


def illumina_pipeline_record(_lib, inp, args):
    """The reference's `advntr genotype` flow for unmapped short reads on one timeline (genome_analyzer.py:172-208: keywords ->
    adVNTR-Filtering -> per-VNTR read ids; :262-297: per locus, the filtered reads of that VNTR -> both strands scored ->
    recruit_read -> repeat counts -> genotype -> a row): FASTA BYTES of >= 10 M reads -> line index + advntr_kwfilter_scan_text ->
    the selection of filtering/main.cc:286-331 on arrays (filtering.select_candidates) -> the candidates as spans of the same
    bytes (vntr_finder.TextReads: encoded on host threads piece by piece, no Python object per read) -> genotype_loci_pipelined
    (native model builder, both strands in one device batch, recruit rule on the device, advntr_genotype_illumina) -> VCF rows.
    Checked: genotypes == the stage-by-stage route (the filter's stdout TEXT parsed as genome_analyzer.py:183-197 does, reads as
    str, vntr_finder.genotype_loci on models built up front); the filter's stdout == the reference binary's on the file's head."""
    from advntr_amd import filtering, genome_analyzer, hmm_utils, models as models_mod, vntr_finder
    loci, lines, fasta = inp["loci"], inp["lines"], inp["fasta"]
    desc = [(l.left, l.right, l.units, l.copies) for l in loci]
    vntrs = []
    for k, l in enumerate(loci):
        v = models_mod.ReferenceVNTR(k + 1, l.units[0], 10000 * k, "chr%d" % (1 + k % 22), None, None, len(l.units))
        v.init_from_xml(list(l.units), l.left, l.right)
        vntrs.append(v)
    import threading
    best, totals = None, []
    for _ in range(3):                              # (the first pass also sizes the process's buffer caches; all totals are listed)
        T, F, P = {}, {}, {}
        h0 = HostRegion().__enter__()
        t0 = time.perf_counter()
        # Two strands of work from the first moment: the PREFILTER -- keyword tables to the device beside the line index of the
        # file (both in the library, the interpreter lock released), then scan and selection -- on a thread of its own, and the
        # SCORING pipeline on this one, whose model-building and upload stages need no reads and whose read-encoding stage waits
        # for the selection (TextReads.pending).  What the reference runs one after the other (genome_analyzer.py:172-208, then
        # the per-locus loop :262-297) overlaps wherever the data allow it.
        text_reads = vntr_finder.TextReads.pending(fasta, len(loci))
        picked = {}

        def prefilter():
            try:
                made = {}

                def keywords():
                    t = time.perf_counter()
                    made["kf"] = filtering.KeywordFilter(lines)
                    T["keywords_to_device"] = time.perf_counter() - t
                kw = threading.Thread(target=keywords, name="advntr-keywords")
                kw.start()
                t = time.perf_counter()
                spans = filtering.KeywordFilter.fasta_spans(fasta)
                T["line_index"] = time.perf_counter() - t
                kw.join()
                kf = made["kf"]
                try:
                    locus_off, ridx, ss, se = kf.candidate_spans(fasta, min_matches=5, timings=F, spans=spans)
                finally:
                    kf.close()
                picked["ridx"] = ridx
                T["prefilter_scan"], T["selection"] = F["scan"], F["select"]
                T["prefilter_done_at"] = time.perf_counter() - t0
                text_reads.fill(ss, se, locus_off)
            except BaseException as e:              # noqa: BLE001 -- handed to the thread that waits for the reads
                text_reads.fail(e)
        pf = threading.Thread(target=prefilter, name="advntr-prefilter")
        pf.start()
        genotypes = vntr_finder.genotype_loci_pipelined(desc, text_reads, timings=P)
        pf.join()
        T["genotypes_done_at"] = time.perf_counter() - t0
        t2 = time.perf_counter()
        rows = [genome_analyzer.vcf_header(vntrs, "reads.fa")] + \
               [genome_analyzer.genotype_row("vcf", v, v.id, g) for v, g in zip(vntrs, genotypes)]
        T["vcf_rows"] = time.perf_counter() - t2
        T["total"] = time.perf_counter() - t0
        T["host"] = {k: v for k, v in HostRegion.since(h0).items() if k in ("nr_throttled_delta", "throttled_usec_delta", "process_cpu_s")}
        totals.append(T["total"])
        if best is None or T["total"] < best[0]["total"]:
            best = (T, F, P, genotypes, rows, text_reads, picked["ridx"])
    T, F, P, genotypes, rows, text_reads, ridx = best
    n_reads = len(fasta) // inp["rec_len"]
    rec = {"loci": len(loci), "fasta_reads": n_reads, "fasta_bytes": len(fasta), "keywords": int(sum(len(k) for _, k in lines)),
           "planted_candidates": inp["n_candidates_planted"], "hit_records": F["hit_records"], "selected_candidates": F["candidates"],
           "viterbi_calls": 2 * F["candidates"], "loci_with_genotype": sum(g.copy_numbers is not None for g in genotypes),
           "vcf_rows": len(rows) - 1, "total_s": T["total"], "total_s_of_each_pass": totals,
           "value": n_reads / T["total"], "unit": "reads of the file/s",
           "stage_s": {k: v for k, v in T.items() if k != "total"}, "prefilter_kernel_ms": F["scan_kernel_ms"],
           "overlap": "the prefilter (keywords -> device beside the line index; scan; selection) on one thread, the scoring pipeline's "
                      "model building / upload on others from the first moment; its read encoding waits for the selection",
           "scoring_stage_s_overlapped": {k: v for k, v in P.items() if k not in ("total", "trace")},
           "note": "one process, one timeline: from the bytes of the read file to VCF rows; input generated in %.1f s (not timed)"
                   % inp["gen_s"]}
    # --- checks, outside the timed passes
    t0 = time.perf_counter()
    kw_text = "".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines)
    stdout_text = filtering.run(fasta, kw_text, 5)
    names, lists = {}, {}
    for line in stdout_text.split("\n"):           # genome_analyzer.py:183-197
        parts = line.split()
        if len(parts) < 2:
            continue
        if parts[0].isdigit() and parts[1].isdigit():
            lists[int(parts[0])] = set(parts[2:])
        else:
            names[parts[0]] = parts[1]
    # (a VNTR's reads in the order of the binary's read lines -- ascending name -- as the per-locus loop meets them, :283)
    read_lists = [[names[nm] for nm in sorted(lists.get(v.id, ()))] for v in vntrs]
    same_reads = sum(sorted(a) == sorted(b) for a, b in zip(read_lists, text_reads.read_lists()))
    models = hmm_utils.build_read_matcher_models(desc)
    plain = vntr_finder.genotype_loci(models, read_lists)
    same = sum(a.copy_numbers == b.copy_numbers and a.recruited_reads_count == b.recruited_reads_count
               for a, b in zip(plain, genotypes))
    rec["stage_by_stage_check_s"] = time.perf_counter() - t0
    rec["candidate_lists_identical_to_stdout_route"] = same_reads == len(loci)
    rec["genotypes_identical_to_stage_by_stage"] = same == len(loci)
    assert same_reads == len(loci), "candidate lists differ from the filter's stdout on %d loci" % (len(loci) - same_reads)
    assert same == len(loci), "one-timeline and stage-by-stage genotypes differ on %d loci" % (len(loci) - same)
    ref = inp.get("ref")
    if ref is not None:
        import shutil
        try:
            ref["child"].wait(timeout=600)
            took = time.perf_counter() - ref["t0"]
            want = open(ref["out"], "rb").read().decode("latin-1")
            got = filtering.run(fasta[:ref["sample"] * inp["rec_len"]], kw_text, 5)
            rec["reference_filter"] = {"kind": "reference", "binary": "oracle/_ref/adVNTR-Filtering (g++ -O2 filtering/main.cc)",
                                       "sample": "the first %d reads of the same file, all %d keyword lines" % (ref["sample"], len(lines)),
                                       "stdout_identical": got == want, "wall_s_beside_the_gpu_work": took, "returncode": ref["child"].returncode}
            assert got == want, "filter stdout differs from the reference binary's on the file's head"
        finally:
            shutil.rmtree(ref["dir"], ignore_errors=True)
    return rec
