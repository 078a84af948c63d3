// abi_genotype.h -- C-ABI entry point of the host-side genotype caller (genotype_caller.h).  Included by engine.hip.
#pragma once
#include "genotype_caller.h"

extern "C" int advntr_genotype_illumina(const int32_t *summaries, const int64_t *locus_off, int32_t n_loci, uint32_t flags,
                                        int32_t min_left_flank, int32_t min_right_flank, int32_t n_threads,
                                        int32_t *out_genotype, double *out_prob, int32_t *out_counts)
{
    if (n_loci < 0 || (n_loci && (!locus_off || !out_genotype || !out_prob)))
        return fail(ADVNTR_ERR_ARG, "advntr_genotype_illumina: bad argument");
    if (n_loci == 0) return ADVNTR_OK;
    if (locus_off[0] < 0) return fail(ADVNTR_ERR_ARG, "advntr_genotype_illumina: negative offset");
    for (int i = 0; i < n_loci; ++i)
        if (locus_off[i + 1] < locus_off[i]) return fail(ADVNTR_ERR_ARG, "advntr_genotype_illumina: locus_off not monotone at %d", i);
    if (locus_off[n_loci] > locus_off[0] && !summaries) return fail(ADVNTR_ERR_ARG, "advntr_genotype_illumina: null summaries");
    const bool accuracy = (flags & ADVNTR_GENOTYPE_ACCURACY_FILTER) != 0, haploid = (flags & ADVNTR_GENOTYPE_HAPLOID) != 0;
    if (n_threads <= 0) n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    n_threads = std::min(n_threads, std::max(1, n_loci / 64));
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i0 = next.fetch_add(64);
            if (i0 >= n_loci) return;
            for (int i = i0; i < std::min(n_loci, i0 + 64); ++i) {
                gt::Result r;
                gt::repeat_count_from_selected(summaries + 8 * locus_off[i], locus_off[i + 1] - locus_off[i], accuracy, haploid,
                                               min_left_flank, min_right_flank, r);
                out_genotype[2 * i] = r.a;
                out_genotype[2 * i + 1] = r.b;
                out_prob[i] = r.max_prob;
                if (out_counts) { out_counts[3 * i] = r.recruited; out_counts[3 * i + 1] = r.spanning; out_counts[3 * i + 2] = r.flanking; }
            }
        }
    };
    if (n_threads <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    return ADVNTR_OK;
}

// Host-side read encoding for genome-scale batches (the Python host spent 0.4 s per 0.8 M reads on this).
extern "C" int advntr_encode_ascii(const char *ascii, const int64_t *read_off, int32_t n_reads, int32_t n_threads,
                                   uint8_t *out_codes, uint8_t *out_bad)
{
    if (n_reads < 0 || !read_off || (n_reads && read_off[n_reads] > 0 && (!ascii || !out_codes)))
        return fail(ADVNTR_ERR_ARG, "advntr_encode_ascii: bad argument");
    if (n_reads == 0) return ADVNTR_OK;
    for (int r = 0; r < n_reads; ++r)
        if (read_off[r + 1] < read_off[r]) return fail(ADVNTR_ERR_ARG, "advntr_encode_ascii: read_off not monotone at %d", r);
    static const struct Table {
        uint8_t code[256];
        Table() { memset(code, 255, sizeof code); code['A'] = code['a'] = 0; code['C'] = code['c'] = 1; code['G'] = code['g'] = 2; code['T'] = code['t'] = 3; code['N'] = code['n'] = 254; }
    } table;
    if (n_threads <= 0) n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const int chunk = 4096;
    n_threads = std::max(1, std::min(n_threads, (n_reads + chunk - 1) / chunk));
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int r0 = next.fetch_add(chunk);
            if (r0 >= n_reads) return;
            const int r1 = std::min(n_reads, r0 + chunk);
            for (int r = r0; r < r1; ++r) {
                uint8_t any = 0, other = 0;
                for (int64_t i = read_off[r]; i < read_off[r + 1]; ++i) {
                    const uint8_t c = table.code[(uint8_t)ascii[i]];
                    out_codes[i] = c;
                    any |= c;
                    other |= (uint8_t)(c == 255);
                }
                if (out_bad) out_bad[r] = other ? 2 : ((any & 0xFC) ? 1 : 0);
            }
        }
    };
    if (n_threads == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    return ADVNTR_OK;
}
