// abi_genotype.h -- C-ABI entry point of the host-side genotype caller (genotype_caller.h).  Included by engine.hip.
#pragma once
#include <dlfcn.h>
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
    if (n_threads <= 0) n_threads = std::min(16, host_cpu_limit());
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

extern "C" int advntr_genotype_observed(const int32_t *ru_counts, const int64_t *locus_off, int32_t n_loci, uint32_t flags,
                                        int32_t n_threads, int32_t *out_genotype, double *out_prob)
{
    if (n_loci < 0 || (n_loci && (!locus_off || !out_genotype || !out_prob)))
        return fail(ADVNTR_ERR_ARG, "advntr_genotype_observed: bad argument");
    if (n_loci == 0) return ADVNTR_OK;
    if (locus_off[0] < 0) return fail(ADVNTR_ERR_ARG, "advntr_genotype_observed: negative offset");
    for (int i = 0; i < n_loci; ++i)
        if (locus_off[i + 1] < locus_off[i]) return fail(ADVNTR_ERR_ARG, "advntr_genotype_observed: locus_off not monotone at %d", i);
    if (locus_off[n_loci] > locus_off[0] && !ru_counts) return fail(ADVNTR_ERR_ARG, "advntr_genotype_observed: null ru_counts");
    const bool accuracy = (flags & ADVNTR_GENOTYPE_ACCURACY_FILTER) != 0, haploid = (flags & ADVNTR_GENOTYPE_HAPLOID) != 0;
    if (n_threads <= 0) n_threads = std::min(16, host_cpu_limit());
    n_threads = std::min(n_threads, std::max(1, n_loci / 64));
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i0 = next.fetch_add(64);
            if (i0 >= n_loci) return;
            for (int i = i0; i < std::min(n_loci, i0 + 64); ++i) {
                gt::Result r;
                gt::dominant_copy_numbers(ru_counts + locus_off[i], locus_off[i + 1] - locus_off[i], accuracy, haploid, r);
                out_genotype[2 * i] = r.a;
                out_genotype[2 * i + 1] = r.b;
                out_prob[i] = r.max_prob;
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

// Host-side text handling for genome-scale batches (the Python host spent 0.4 s per 0.8 M reads on encoding, and 0.7 s per
// 2 M reads around the prefilter kernel): line index of a FASTA text and ASCII -> base codes, on host threads.
static int host_text_threads(int n_threads, int64_t units, int64_t per_thread)
{
    if (n_threads <= 0) n_threads = std::min(16, host_cpu_limit());
    return (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (units + per_thread - 1) / per_thread));
}

template <class F> static void host_parallel(int n_threads, F &&work)
{
    if (n_threads <= 1) { work(0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
    for (auto &t : pool) t.join();
}

extern "C" int advntr_line_index(const char *text, int64_t n_bytes, int32_t n_threads, int64_t *line_start, int64_t capacity,
                                 int64_t *n_lines)
{
    if (n_bytes < 0 || (n_bytes && !text) || !n_lines || capacity < 0 || (capacity && !line_start))
        return fail(ADVNTR_ERR_ARG, "advntr_line_index: bad argument");
    const int T = host_text_threads(n_threads, n_bytes, (int64_t)4 << 20);
    std::vector<int64_t> count(T + 1, 0);
    const int64_t chunk = (n_bytes + T - 1) / std::max(T, 1);
    host_parallel(T, [&](int t) {
        const int64_t lo = t * chunk, hi = std::min(n_bytes, lo + chunk);
        int64_t c = 0;
        for (const char *p = text + lo; p < text + hi;) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(text + hi - p));
            if (!q) break;
            ++c;
            p = q + 1;
        }
        count[t + 1] = c;
    });
    for (int t = 0; t < T; ++t) count[t + 1] += count[t];
    // a line starts at 0 and after every newline that is not the last byte
    const int64_t newlines = count[T];
    const int64_t lines = n_bytes == 0 ? 0 : newlines + (text[n_bytes - 1] == '\n' ? 0 : 1);
    *n_lines = lines;
    if (lines > capacity) return fail(ADVNTR_ERR_TOO_LARGE, "advntr_line_index: %lld lines, capacity %lld", (long long)lines, (long long)capacity);
    if (lines == 0) return ADVNTR_OK;
    line_start[0] = 0;
    host_parallel(T, [&](int t) {
        const int64_t lo = t * chunk, hi = std::min(n_bytes, lo + chunk);
        int64_t at = count[t] + 1;                      // index of the line that starts after this chunk's first newline
        for (const char *p = text + lo; p < text + hi;) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(text + hi - p));
            if (!q) break;
            if (at < lines) line_start[at] = (q - text) + 1;
            ++at;
            p = q + 1;
        }
    });
    return ADVNTR_OK;
}

// ASCII -> base codes; source of read r given by `src_of(r)`, n bytes into out_codes[out_off[r] ..); work items sized by bytes
// (a batch may be a million 150-base reads or a few thousand 15 kb ones)
template <class Src>
static int encode_reads_host(const char *who, Src &&src_of, int32_t n_reads, uint32_t flags, int32_t n_threads,
                             const int64_t *out_off, uint8_t *out_codes, uint8_t *out_bad)
{
    // (clearing bit 5 folds a-z onto A-Z and maps nothing else onto a letter)
    const uint8_t fold_mask = (flags & ADVNTR_ENCODE_CASE_SENSITIVE) ? 0xFF : 0xDF;
    const int64_t total = out_off[n_reads] - out_off[0];
    const int64_t mean = std::max<int64_t>(1, total / std::max(n_reads, 1));
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(4096, ((int64_t)512 << 10) / mean));
    const int T = host_text_threads(n_threads, n_reads, chunk);
    std::atomic<int> next(0), bad_span(-1);
    host_parallel(T, [&](int) {
        for (;;) {
            const int r0 = next.fetch_add(chunk);
            if (r0 >= n_reads) return;
            const int r1 = std::min(n_reads, r0 + chunk);
            for (int r = r0; r < r1; ++r) {
                int64_t n = 0;
                const uint8_t *src = src_of(r, n);
                if (n < 0 || out_off[r + 1] - out_off[r] != n || (n && !src)) { bad_span = r; continue; }
                uint8_t *dst = out_codes + out_off[r];
                uint8_t any = 0, other = 0;
                // compare-and-select per byte instead of a table lookup: the loop vectorises (16-32 bases per instruction;
                // the lookup was a dependent load per base -- 1.8 core-seconds for the 1.8 GB of config 5's reads)
                const uint8_t keep = fold_mask;
                for (int64_t i = 0; i < n; ++i) {
                    const uint8_t f = (uint8_t)(src[i] & keep);
                    uint8_t c = 255;
                    c = f == 'A' ? (uint8_t)0 : c;
                    c = f == 'C' ? (uint8_t)1 : c;
                    c = f == 'G' ? (uint8_t)2 : c;
                    c = f == 'T' ? (uint8_t)3 : c;
                    c = f == 'N' ? (uint8_t)254 : c;
                    dst[i] = c;
                    any |= c;
                    other |= (uint8_t)(c == 255);
                }
                if (out_bad) out_bad[r] = other ? 2 : ((any & 0xFC) ? 1 : 0);
            }
        }
    });
    if (bad_span >= 0) return fail(ADVNTR_ERR_ARG, "%s: read %d does not match its output slot", who, (int)bad_span);
    return ADVNTR_OK;
}

extern "C" int advntr_encode_spans(const char *ascii, const int64_t *span_start, const int64_t *span_end, int32_t n_reads,
                                   uint32_t flags, int32_t n_threads, const int64_t *out_off, uint8_t *out_codes, uint8_t *out_bad)
{
    if (n_reads < 0 || (n_reads && (!span_start || !span_end || !out_off)))
        return fail(ADVNTR_ERR_ARG, "advntr_encode_spans: bad argument");
    if (n_reads == 0) return ADVNTR_OK;
    if (!ascii || !out_codes) {
        if (out_off[n_reads] > out_off[0]) return fail(ADVNTR_ERR_ARG, "advntr_encode_spans: null buffer");
    }
    return encode_reads_host("advntr_encode_spans", [&](int r, int64_t &n) {
        n = span_end[r] - span_start[r];
        return (const uint8_t *)ascii + span_start[r];
    }, n_reads, flags, n_threads, out_off, out_codes, out_bad);
}

extern "C" int advntr_encode_texts(const char *const *texts, int32_t n_reads, uint32_t flags, int32_t n_threads,
                                   const int64_t *out_off, uint8_t *out_codes, uint8_t *out_bad)
{
    if (n_reads < 0 || (n_reads && (!texts || !out_off))) return fail(ADVNTR_ERR_ARG, "advntr_encode_texts: bad argument");
    if (n_reads == 0) return ADVNTR_OK;
    for (int r = 0; r < n_reads; ++r)
        if (out_off[r + 1] < out_off[r]) return fail(ADVNTR_ERR_ARG, "advntr_encode_texts: out_off not monotone at %d", r);
    if (!out_codes && out_off[n_reads] > out_off[0]) return fail(ADVNTR_ERR_ARG, "advntr_encode_texts: null buffer");
    return encode_reads_host("advntr_encode_texts", [&](int r, int64_t &n) {
        n = out_off[r + 1] - out_off[r];
        return (const uint8_t *)texts[r];
    }, n_reads, flags, n_threads, out_off, out_codes, out_bad);
}

// A CPython host's list of str -> buffer pointers and lengths (include/advntr_pyhost.h: an optional helper beside the C ABI).  Called with the interpreter lock held.
extern "C" int64_t advntr_pylist_texts(void *list, const char **texts, int64_t *lengths, int64_t capacity)
{
    typedef long (*SizeFn)(void *);
    typedef void *(*ItemFn)(void *, long);
    typedef const char *(*Utf8Fn)(void *, long *);
    typedef void (*ClearFn)(void);
    struct Api {
        SizeFn list_size = nullptr, str_length = nullptr;
        ItemFn list_item = nullptr;
        Utf8Fn utf8 = nullptr;
        ClearFn clear = nullptr;
        bool ok = false;
        Api()
        {
            list_size = (SizeFn)dlsym(RTLD_DEFAULT, "PyList_Size");
            str_length = (SizeFn)dlsym(RTLD_DEFAULT, "PyUnicode_GetLength");
            list_item = (ItemFn)dlsym(RTLD_DEFAULT, "PyList_GetItem");
            utf8 = (Utf8Fn)dlsym(RTLD_DEFAULT, "PyUnicode_AsUTF8AndSize");
            clear = (ClearFn)dlsym(RTLD_DEFAULT, "PyErr_Clear");
            ok = list_size && str_length && list_item && utf8 && clear;
        }
    };
    static const Api api;
    if (!api.ok || !list || !texts || !lengths) return -1;
    const long n = api.list_size(list);
    if (n < 0) { api.clear(); return -1; }
    if (n > capacity) return -1;
    for (long i = 0; i < n; ++i) {
        void *item = api.list_item(list, i);                     // borrowed
        long size = 0;
        const char *p = item ? api.utf8(item, &size) : nullptr;
        if (!p) { api.clear(); return -(int64_t)i - 2; }         // not a str (or one without a UTF-8 form)
        if (api.str_length(item) != size) return -(int64_t)i - 2; // not ASCII: one byte per character does not hold
        texts[i] = p;
        lengths[i] = size;
    }
    return n;
}

extern "C" int advntr_cut_pieces(const uint8_t *codes, const int64_t *read_off, int32_t n_reads, const int32_t *piece_read,
                                 const int64_t *begin, const int64_t *end, const uint8_t *reverse, int64_t n_pieces,
                                 int32_t n_threads, const int64_t *out_off, uint8_t *out_codes)
{
    if (n_reads < 0 || n_pieces < 0 || (n_pieces && (!read_off || !piece_read || !begin || !end || !out_off)))
        return fail(ADVNTR_ERR_ARG, "advntr_cut_pieces: bad argument");
    if (n_pieces == 0) return ADVNTR_OK;
    for (int64_t p = 0; p < n_pieces; ++p) {
        const int32_t r = piece_read[p];
        if (r < 0 || r >= n_reads) return fail(ADVNTR_ERR_ARG, "advntr_cut_pieces: piece %lld names read %d of %d", (long long)p, r, n_reads);
        const int64_t n = read_off[r + 1] - read_off[r];
        if (begin[p] < 0 || end[p] < begin[p] || end[p] > n || out_off[p + 1] - out_off[p] != end[p] - begin[p])
            return fail(ADVNTR_ERR_ARG, "advntr_cut_pieces: piece %lld = [%lld, %lld) of a read of %lld bases into a slot of %lld",
                        (long long)p, (long long)begin[p], (long long)end[p], (long long)n, (long long)(out_off[p + 1] - out_off[p]));
    }
    if ((!codes || !out_codes) && out_off[n_pieces] > out_off[0]) return fail(ADVNTR_ERR_ARG, "advntr_cut_pieces: null buffer");
    const int64_t chunk = 256;
    const int T = host_text_threads(n_threads, n_pieces, chunk);
    std::atomic<int64_t> next(0);
    host_parallel(T, [&](int) {
        for (;;) {
            const int64_t p0 = next.fetch_add(chunk);
            if (p0 >= n_pieces) return;
            for (int64_t p = p0; p < std::min(n_pieces, p0 + chunk); ++p) {
                const uint8_t *src = codes + read_off[piece_read[p]];
                uint8_t *dst = out_codes + out_off[p];
                const int64_t b = begin[p], n = end[p] - b;
                if (reverse && reverse[p]) {
                    for (int64_t i = 0; i < n; ++i) {
                        const uint8_t c = src[b + n - 1 - i];
                        dst[i] = c < 4 ? (uint8_t)(3 - c) : (uint8_t)255;
                    }
                } else {
                    for (int64_t i = 0; i < n; ++i) {
                        const uint8_t c = src[b + i];
                        dst[i] = c < 4 ? c : (uint8_t)255;
                    }
                }
            }
        }
    });
    return ADVNTR_OK;
}

extern "C" int advntr_encode_ascii(const char *ascii, const int64_t *read_off, int32_t n_reads, int32_t n_threads,
                                   uint8_t *out_codes, uint8_t *out_bad)
{
    if (n_reads < 0 || !read_off) return fail(ADVNTR_ERR_ARG, "advntr_encode_ascii: bad argument");
    for (int r = 0; r < n_reads; ++r)
        if (read_off[r + 1] < read_off[r]) return fail(ADVNTR_ERR_ARG, "advntr_encode_ascii: read_off not monotone at %d", r);
    return advntr_encode_spans(ascii, read_off, read_off + 1, n_reads, 0, n_threads, read_off, out_codes, out_bad);
}
