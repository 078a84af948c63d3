"""The reference's `advntr genotype` flow for unmapped short reads on ONE timeline (-m gpu, 200 loci): the bytes of a FASTA file
-> advntr_kwfilter_scan_text -> the selection of filtering/main.cc:286-331 on arrays -> the candidates as spans of the same bytes
-> both strands scored, recruit rule on the device, advntr_genotype_illumina -> VCF rows
(/root/reference/advntr/genome_analyzer.py:172-208 and 262-297).  Held against the stage-by-stage route -- the filter's stdout
text parsed as the reference parses it, reads as str, models built up front -- and, where the prebuilt reference binary travelled
(oracle/_ref), against that binary's stdout."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _loci_and_candidates(n_loci):
    from advntr_amd import workloads
    loci, cands = [], []
    for k in range(n_loci):
        params, calls, (nm, nu) = workloads._c2_locus((k, 20240602, 150, 80, 40))
        loci.append(workloads.Locus(*params))
        cands.append(calls[:nm + nu])                           # forward strands: mapped-like + unmapped-like
    return loci, cands


def test_illumina_pipeline_one_timeline_200_loci(tmp_path):
    from advntr_amd import filtering, genome_analyzer, hmm_utils, models, vntr_finder, workloads
    loci, cands = _loci_and_candidates(200)
    lines, fasta, rec_len, planted = workloads.make_illumina_pipeline_workload(loci, cands, 120000)
    assert len(fasta) == 120000 * rec_len and len(planted) == sum(len(c) for c in cands)
    kw_text = "".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines)

    # --- the prefilter's selection as arrays == its stdout text, parsed as genome_analyzer.py:183-197 parses it
    kf = filtering.KeywordFilter(lines)
    T = {}
    locus_off, ridx, ss, se = kf.candidate_spans(fasta, min_matches=5, timings=T)
    kf.close()
    assert set(T) >= {"line_index", "scan", "select", "candidates"} and T["candidates"] == len(ridx) == locus_off[-1]
    stdout_text = filtering.run(fasta, kw_text, 5)
    names, lists = {}, {}
    for line in stdout_text.split("\n"):
        parts = line.split()
        if len(parts) < 2:
            continue
        if parts[0].isdigit() and parts[1].isdigit():
            lists[int(parts[0])] = parts[2:]
        else:
            names[parts[0]] = parts[1]
    text_reads = vntr_finder.TextReads(fasta, ss, se, locus_off)
    as_str = text_reads.read_lists()
    read_lists = [[names[nm] for nm in sorted(lists.get(k + 1, ()))] for k in range(200)]
    assert as_str == read_lists                                 # same reads, same (ascending-name) order, every locus
    got_names = [["r%07d" % r for r in ridx[locus_off[k]:locus_off[k + 1]]] for k in range(200)]
    assert got_names == [sorted(lists.get(k + 1, ())) for k in range(200)]
    # most of what the filter selects for a locus are reads planted for it; most planted locus-derived reads are found
    planted_of = np.repeat(np.arange(200), [len(c) for c in cands])
    own = sum(int(np.isin(ridx[locus_off[k]:locus_off[k + 1]], planted[planted_of == k]).sum()) for k in range(200))
    assert own >= 0.95 * len(ridx) and len(ridx) > 0.2 * len(planted)

    # --- genotypes: one timeline == stage by stage
    desc = [(l.left, l.right, l.units, l.copies) for l in loci]
    P = {}
    piped = vntr_finder.genotype_loci_pipelined(desc, text_reads, timings=P, chunks=5)
    plain = vntr_finder.genotype_loci(hmm_utils.build_read_matcher_models(desc), read_lists)
    from_str = vntr_finder.genotype_loci_pipelined(desc, read_lists, chunks=3)
    for a, b, c in zip(plain, piped, from_str):
        assert a.copy_numbers == b.copy_numbers == c.copy_numbers
        assert a.recruited_reads_count == b.recruited_reads_count == c.recruited_reads_count
        assert a.maximum_likelihood == b.maximum_likelihood
    assert sum(g.copy_numbers is not None for g in piped) > 100 and P["total"] > 0

    # --- the scoring pipeline started BEFORE the reads are known (its model building and upload need none; its encoding stage
    # waits): what the bench's `illumina_pipeline` record does while the prefilter still runs on another thread
    import threading
    import time
    late = vntr_finder.TextReads.pending(fasta, 200)
    threading.Timer(0.3, lambda: late.fill(ss, se, locus_off)).start()
    t0 = time.perf_counter()
    overlapped = vntr_finder.genotype_loci_pipelined(desc, late, chunks=5)
    assert time.perf_counter() - t0 >= 0.25
    assert [g.copy_numbers for g in overlapped] == [g.copy_numbers for g in plain]
    broken = vntr_finder.TextReads.pending(fasta, 200)
    threading.Timer(0.1, lambda: broken.fail(RuntimeError("the prefilter failed"))).start()
    with pytest.raises(RuntimeError, match="the prefilter failed"):
        vntr_finder.genotype_loci_pipelined(desc, broken, chunks=5)

    # --- rows
    vntrs = []
    for k, l in enumerate(loci):
        v = models.ReferenceVNTR(k + 1, l.units[0], 10000 * k, "chr%d" % (1 + k % 22), None, None, len(l.units))
        v.init_from_xml(list(l.units), l.left, l.right)
        vntrs.append(v)
    rows = [genome_analyzer.genotype_row("vcf", v, v.id, g) for v, g in zip(vntrs, piped)]
    assert len(rows) == 200 and all(r.count("\t") == 9 and ";VID=%d;" % (k + 1) in r for k, r in enumerate(rows))

    # --- the reference's own filter binary on the same file, where it travelled
    ref = os.path.join(ROOT, "oracle", "_ref", "adVNTR-Filtering")
    if os.path.exists(ref):
        fa, kw = tmp_path / "reads.fa", tmp_path / "kw.txt"
        fa.write_bytes(fasta)
        kw.write_text(kw_text)
        with open(str(kw)) as fin:
            want = subprocess.run([ref, str(fa)], stdin=fin, stdout=subprocess.PIPE, check=True, timeout=600).stdout.decode("latin-1")
        assert stdout_text == want
