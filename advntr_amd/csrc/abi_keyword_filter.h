// abi_keyword_filter.h -- C-ABI entry points of the keyword prefilter (advntr_kwfilter_*): table construction on the host, scan launch
// Included by engine.hip (same translation unit: uses its error helpers, device caches and HIP_TRY).
#pragma once

// ------------------------------------------------------------------------------------------------
// Keyword prefilter
// ------------------------------------------------------------------------------------------------
struct advntr_kwfilter {
    KwfDevice dev{};
    std::vector<void *> allocs;
    int64_t n_keys = 0;
};

extern "C" void advntr_kwfilter_destroy(advntr_kwfilter *F)
{
    if (!F) return;
    for (void *p : F->allocs) (void)hipFree(p);
    delete F;
}

extern "C" advntr_kwfilter *advntr_kwfilter_create(const uint8_t *kw_bases, const int64_t *kw_off,
                                                   const int32_t *kw_vntr, int32_t n_keywords)
{
    if (n_keywords < 0 || (n_keywords && (!kw_bases || !kw_off || !kw_vntr))) {
        fail(ADVNTR_ERR_ARG, "advntr_kwfilter_create: bad argument");
        return nullptr;
    }
    // group keyword strings: packed key -> owners (in keyword order).  A keyword of more than 29 bases is filed under the
    // key of its 29-base prefix (tag KWF_LONG_TAG) with a record of what follows the prefix.
    // (collected as (key, value) pairs and grouped by ONE stable sort -- ascending key, keyword order inside a key: the order a
    // std::map of vectors iterates in, which the first version used and which cost 0.3 of the 0.44 s this call took for the
    // 430 000 keywords of the 6 719-locus set)
    struct KwEntry { uint64_t key; int32_t val; };
    struct KwSpan {
        const KwEntry *b, *e;
        struct It {
            const KwEntry *p;
            int32_t operator*() const { return p->val; }
            It &operator++() { ++p; return *this; }
            bool operator!=(const It &o) const { return p != o.p; }
        };
        size_t size() const { return (size_t)(e - b); }
        It begin() const { return It{b}; }
        It end() const { return It{e}; }
        int32_t operator[](size_t i) const { return b[i].val; }
    };
    struct KwGroup { uint64_t first; KwSpan second; };       // short keys: owners; long-prefix keys: indices into long_recs_h
    std::vector<KwEntry> entries;
    entries.reserve((size_t)n_keywords);
    std::vector<KwfLongRec> long_recs_h;
    std::vector<uint8_t> long_bases;
    std::vector<std::pair<int, int>> lengths;                 // (window length, tag)
    for (int w = 0; w < n_keywords; ++w) {
        const int64_t L = kw_off[w + 1] - kw_off[w];
        if (L < 1 || L > 0x7fffffff) {
            fail(ADVNTR_ERR_ARG, "advntr_kwfilter_create: keyword %d has length %lld", w, (long long)L);
            return nullptr;
        }
        for (int64_t i = kw_off[w]; i < kw_off[w + 1]; ++i)
            if (kw_bases[i] > 3) {
                fail(ADVNTR_ERR_SYMBOL, "advntr_kwfilter_create: keyword %d holds a non-ACGT symbol", w);
                return nullptr;
            }
        const int win = (int)std::min<int64_t>(L, KWF_MAX_PACKED);
        const int tag = L > KWF_MAX_PACKED ? KWF_LONG_TAG : (int)L;
        uint64_t key = 0;
        for (int64_t i = kw_off[w]; i < kw_off[w] + win; ++i) key = (key << 2) | kw_bases[i];
        key |= (uint64_t)tag << 58;
        if (tag == KWF_LONG_TAG) {
            KwfLongRec rec{};
            rec.rest_off = (uint32_t)long_bases.size();
            rec.rest_len = (uint32_t)(L - win);
            rec.owner = kw_vntr[w];
            if (long_bases.size() + (size_t)(L - win) > 0xffffffffu) {
                fail(ADVNTR_ERR_TOO_LARGE, "advntr_kwfilter_create: more than 4 GiB of long keywords");
                return nullptr;
            }
            long_bases.insert(long_bases.end(), kw_bases + kw_off[w] + win, kw_bases + kw_off[w + 1]);
            entries.push_back(KwEntry{key, (int32_t)long_recs_h.size()});
            long_recs_h.push_back(rec);
        } else {
            entries.push_back(KwEntry{key, kw_vntr[w]});
        }
        if (std::find(lengths.begin(), lengths.end(), std::make_pair(win, tag)) == lengths.end()) lengths.emplace_back(win, tag);
    }
    if ((int)lengths.size() > KWF_MAX_LENGTHS) {
        fail(ADVNTR_ERR_UNSUPPORTED, "advntr_kwfilter_create: %zu distinct keyword lengths (max %d)", lengths.size(), KWF_MAX_LENGTHS);
        return nullptr;
    }
    std::sort(lengths.begin(), lengths.end());
    std::stable_sort(entries.begin(), entries.end(), [](const KwEntry &a, const KwEntry &b) { return a.key < b.key; });
    std::vector<KwGroup> groups;
    for (size_t i = 0; i < entries.size();) {
        size_t j = i + 1;
        while (j < entries.size() && entries[j].key == entries[i].key) ++j;
        groups.push_back(KwGroup{entries[i].key, KwSpan{entries.data() + i, entries.data() + j}});
        i = j;
    }
    size_t slots = 1024;
    while (slots < groups.size() * 4) slots <<= 1;
    std::vector<uint64_t> keys(slots, KWF_EMPTY);
    std::vector<uint32_t> vals(slots, 0), bitset(KWF_BITSET_BITS / 32, 0);
    // fingerprint table: buckets of eight 16-bit fingerprints, on average <= 2.5 keys per bucket at any keyword count (2 MiB,
    // L2-resident, for the ~320 k keywords of the 6 719-locus set; it simply grows beyond that).  A bucket that receives more
    // than seven keys gets the overflow marker in its last slot: every key that hashes there then goes to the exact table.
    size_t n_buckets = 512;
    while (n_buckets * 5 < groups.size() * 2) n_buckets <<= 1;
    std::vector<uint16_t> fps(n_buckets * 8, 0);
    std::vector<int32_t> ids;                                  // short keys: owners
    std::vector<KwfLongRec> long_recs;                         // long-prefix keys: their records, contiguous per key
    // one keyword length of at most 16 bases: keyword_filter_short_kernel with its own filters and exact table (below)
    bool short_ok = lengths.size() == 1 && lengths[0].second != KWF_LONG_TAG && lengths[0].first <= 16;
    for (auto &kv : groups)
        if (kv.second.size() > 127) short_ok = false;                 // (the short table's count field)
    std::vector<uint2> short_table;
    if (short_ok) short_table.assign(slots, make_uint2(0u, KWF_SHORT_EMPTY));
    for (auto &kv : groups) {
        const bool is_long = (kv.first >> 58) == KWF_LONG_TAG;
        if (kv.second.size() > 255) {
            fail(ADVNTR_ERR_UNSUPPORTED, "advntr_kwfilter_create: a keyword (or 29-base prefix) is shared by more than 255 VNTRs");
            return nullptr;
        }
        if ((is_long ? long_recs.size() : ids.size()) + kv.second.size() > 0xffffffu) {
            fail(ADVNTR_ERR_TOO_LARGE, "advntr_kwfilter_create: more than 2^24 (keyword, VNTR) pairs");
            return nullptr;
        }
        const uint64_t h = kwf_hash(kv.first);
        const unsigned b = (unsigned)(h >> 40) & (KWF_BITSET_BITS - 1);
        bitset[b >> 5] |= 1u << (b & 31);
        const unsigned b2 = (unsigned)(h >> 4) & (KWF_BITSET_BITS - 1);
        bitset[b2 >> 5] |= 1u << (b2 & 31);
        {
            uint16_t *bucket = fps.data() + (size_t)kwf_fp_bucket(h, (uint32_t)(n_buckets - 1)) * 8;
            int at = 0;
            while (at < 8 && bucket[at] != 0) ++at;
            if (at < 7) bucket[at] = kwf_fp(h);                         // (slot 7 is kept for the marker)
            else bucket[7] = KWF_FP_OVERFLOW;                           // eighth key (or later): the bucket answers "maybe" from now on
        }
        size_t s = (h & 0xffffffffull) & (slots - 1);
        while (keys[s] != KWF_EMPTY) s = (s + 1) & (slots - 1);         // terminates: at most a quarter of the slots are taken
        keys[s] = kv.first;
        if (is_long) {
            vals[s] = (uint32_t)long_recs.size() | ((uint32_t)kv.second.size() << 24);
            for (int32_t i : kv.second) long_recs.push_back(long_recs_h[i]);
        } else {
            vals[s] = (uint32_t)ids.size() | ((uint32_t)kv.second.size() << 24);
            if (short_ok) {
                const uint32_t k32 = (uint32_t)(kv.first & 0xffffffffull);
                if (kv.second.size() == 1 && (kv.second[0] < 0 || kv.second[0] > 0xffffff)) {
                    fail(ADVNTR_ERR_TOO_LARGE, "advntr_kwfilter_create: VNTR index %d", kv.second[0]);
                    return nullptr;
                }
                size_t t = kwf_short_slot(kwf_h24b(kwf_h24(k32))) & (uint32_t)(slots - 1);
                while (short_table[t].y != KWF_SHORT_EMPTY) t = (t + 1) & (slots - 1);
                short_table[t] = make_uint2(k32, kv.second.size() == 1 ? (KWF_SHORT_ONE | (uint32_t)kv.second[0]) : vals[s]);
            }
            for (int32_t owner : kv.second) ids.push_back(owner);
        }
    }
    // one keyword length of at most 16 bases: the two blocked Bloom filters of keyword_filter_short_kernel (LDS: 1 Mbit, two
    // bits per keyword; L2: a power of two of >= 16 bits per keyword, three bits per keyword)
    std::vector<uint32_t> bloom, bloom2;
    if (short_ok) {
        size_t words2 = KWF_BLOOM_WORDS;
        while (words2 * 32 < groups.size() * 16 && words2 < ((size_t)1 << 27)) words2 <<= 1;
        int word_bits = 0;
        while (((size_t)1 << word_bits) < words2) ++word_bits;
        bloom.assign(KWF_BLOOM_WORDS, 0u);
        bloom2.assign(words2, 0u);
        for (auto &kv : groups) {
            const uint32_t y = kwf_h24((uint32_t)(kv.first & 0xffffffffull));
            bloom[y & (KWF_BLOOM_WORDS - 1)] |= kwf_bloom_mask1(y);
            const uint32_t z = kwf_h24b(y);
            bloom2[z & (uint32_t)(words2 - 1)] |= kwf_bloom_mask2(z, word_bits);
        }
    }
    advntr_kwfilter *F = new advntr_kwfilter();
    F->n_keys = (int64_t)groups.size();
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        F->allocs.push_back(d);
        if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    };
    void *dk = up(keys.data(), keys.size() * 8), *dv = up(vals.data(), vals.size() * 4);
    void *di = up(ids.data(), ids.size() * 4), *db = up(bitset.data(), bitset.size() * 4);
    void *df = up(fps.data(), fps.size() * 2);
    void *dl = up(long_recs.data(), long_recs.size() * sizeof(KwfLongRec)), *dlb = up(long_bases.data(), long_bases.size());
    void *dbl = up(bloom.data(), bloom.size() * 4), *dbl2 = up(bloom2.data(), bloom2.size() * 4);
    void *dst = up(short_table.data(), short_table.size() * sizeof(uint2));
    if (!dk || !dv || !di || !db || !df || !dl || !dlb || !dbl || !dbl2 || !dst) {
        fail(ADVNTR_ERR_DEVICE, "advntr_kwfilter_create: device upload failed");
        advntr_kwfilter_destroy(F);
        return nullptr;
    }
    KwfDevice &D = F->dev;
    D.n_lengths = (int32_t)lengths.size();
    for (size_t i = 0; i < lengths.size(); ++i) {
        D.length[i] = lengths[i].first;
        D.tag[i] = lengths[i].second;
        D.mask[i] = (1ull << (2 * lengths[i].first)) - 1ull;
    }
    D.long_recs = (const KwfLongRec *)dl; D.long_bases = (const uint8_t *)dlb;
    D.table_mask = slots - 1;
    D.keys = (const uint64_t *)dk; D.vals = (const uint32_t *)dv; D.ids = (const int32_t *)di; D.bitset = (const uint32_t *)db;
    D.fp_buckets = (const uint4 *)df; D.bucket_mask = (uint32_t)(n_buckets - 1);
    D.short_ok = short_ok ? 1 : 0; D.short_bloom = (const uint32_t *)dbl; D.short_l2 = (const uint32_t *)dbl2;
    D.short_l2_mask = short_ok ? (uint32_t)(bloom2.size() - 1) : 0u;
    D.short_table = (const uint2 *)dst; D.short_table_mask = short_ok ? (uint32_t)(slots - 1) : 0u;
    return F;
}

// reads = spans [span_start[r], span_end[r]) of `bytes` (n_bytes long): base codes, or with `ascii` the text of a FASTA file
static int kwfilter_scan_spans(advntr_kwfilter *F, const uint8_t *bytes, int64_t n_bytes, const int64_t *span_start,
                               const int64_t *span_end, int32_t n_reads, bool ascii, int32_t *out_read, int32_t *out_vntr,
                               int32_t *out_count, int64_t capacity, int64_t *n_out, float *kernel_ms)
{
    *n_out = 0;
    if (n_reads == 0) return ADVNTR_OK;
    for (int r = 0; r < n_reads; ++r)
        if (span_start[r] < 0 || span_end[r] < span_start[r] || span_end[r] > n_bytes || span_end[r] - span_start[r] > 0x7fffffff)
            return fail(ADVNTR_ERR_ARG, "keyword prefilter: read %d is not a span of the %lld bytes given", r, (long long)n_bytes);
    uint8_t *d_bases = nullptr;
    int64_t *d_start = nullptr, *d_end = nullptr;
    int32_t *d_r = nullptr, *d_v = nullptr, *d_c = nullptr;
    unsigned long long *d_n = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // buffers, like a batch's, come from the per-device cache (a scan per FASTA chunk would otherwise spend more time in
    // hipMalloc / hipFree than in the kernel)
    const int dev = current_device();
    std::vector<std::pair<void *, size_t>> held;
    auto take = [&](size_t bytes_wanted) -> void * {
        size_t got = 0;
        void *p = g_cache.get(dev, bytes_wanted, &got);
        if (p) held.emplace_back(p, got);
        return p;
    };
    auto cleanup = [&]() {
        for (auto &h : held) g_cache.put(dev, h.first, h.second);
        if (e0) g_cache.put_event(dev, e0);
        if (e1) g_cache.put_event(dev, e1);
    };
    int rc = [&]() -> int {
        const size_t cap = (size_t)std::max<int64_t>(capacity, 1);
        const bool contiguous = span_end == span_start + 1;                 // read_off form: one array serves both
        d_bases = (uint8_t *)take((size_t)n_bytes + 80);                    // (+ slack: the kernel fetches whole 64-byte sectors
        d_start = (int64_t *)take(((size_t)n_reads + 1) * 8);               //  only when they lie inside a read)
        d_end = contiguous ? d_start + 1 : (int64_t *)take((size_t)n_reads * 8);
        d_r = (int32_t *)take(cap * 4); d_v = (int32_t *)take(cap * 4); d_c = (int32_t *)take(cap * 4);
        d_n = (unsigned long long *)take(8);
        if (!d_bases || !d_start || !d_end || !d_r || !d_v || !d_c || !d_n)
            return fail(ADVNTR_ERR_DEVICE, "keyword prefilter: device allocation failed");
        HIP_TRY(hipMemset(d_n, 0, 8));
        if (n_bytes) HIP_TRY(hipMemcpy(d_bases, bytes, (size_t)n_bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_start, span_start, ((size_t)n_reads + (contiguous ? 1 : 0)) * 8, hipMemcpyHostToDevice));
        if (!contiguous) HIP_TRY(hipMemcpy(d_end, span_end, (size_t)n_reads * 8, hipMemcpyHostToDevice));
        e0 = g_cache.get_event(dev);
        e1 = g_cache.get_event(dev);
        if (!e0 || !e1) return fail(ADVNTR_ERR_DEVICE, "keyword prefilter: event creation failed");
        KwfArgs a{};
        a.f = F->dev; a.bases = d_bases; a.span_start = d_start; a.span_end = d_end; a.n_reads = n_reads;
        a.out_read = d_r; a.out_vntr = d_v; a.out_count = d_c; a.n_out = d_n; a.capacity = capacity;
        const int grid = std::max(1, std::min((n_reads + KWF_BLOCK - 1) / KWF_BLOCK, device_cus()));
        const void *kernel = ascii ? (const void *)keyword_filter_kernel<true> : (const void *)keyword_filter_kernel<false>;
        if (F->dev.short_ok) {        // one keyword length of at most 16 bases
            const bool wide = F->dev.length[0] >= 15;
            kernel = ascii ? (wide ? (const void *)keyword_filter_short_kernel<true, true> : (const void *)keyword_filter_short_kernel<true, false>)
                           : (wide ? (const void *)keyword_filter_short_kernel<false, true> : (const void *)keyword_filter_short_kernel<false, false>);
        }
        // (the short-keyword kernel keeps a queue per wavefront behind its filter)
        const size_t lds = F->dev.short_ok ? (size_t)KWF_SHORT_LDS_BYTES : (size_t)(KWF_BITSET_BITS / 8);
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipEventRecord(e0, nullptr));
        void *kargs[] = {(void *)&a};
        HIP_TRY(hipLaunchKernel(kernel, dim3(grid), dim3(KWF_BLOCK), kargs, lds, nullptr));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, e0, e1));
        unsigned long long n = 0;
        HIP_TRY(hipMemcpy(&n, d_n, 8, hipMemcpyDeviceToHost));
        *n_out = (int64_t)n;
        if ((int64_t)n > capacity) return fail(ADVNTR_ERR_TOO_LARGE, "keyword prefilter: %llu records, capacity %lld", n, (long long)capacity);
        if (n) {
            HIP_TRY(hipMemcpy(out_read, d_r, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(out_vntr, d_v, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(out_count, d_c, n * 4, hipMemcpyDeviceToHost));
        }
        return ADVNTR_OK;
    }();
    std::string keep = g_err;
    cleanup();
    g_err = keep;
    return rc;
}

extern "C" int advntr_kwfilter_scan(advntr_kwfilter *F, const uint8_t *bases, const int64_t *read_off, int32_t n_reads,
                                    int32_t *out_read, int32_t *out_vntr, int32_t *out_count, int64_t capacity,
                                    int64_t *n_out, float *kernel_ms)
{
    if (!F || !read_off || n_reads < 0 || !n_out || capacity < 0 || (capacity && (!out_read || !out_vntr || !out_count)))
        return fail(ADVNTR_ERR_ARG, "advntr_kwfilter_scan: bad argument");
    if (n_reads && read_off[n_reads] > 0 && !bases) return fail(ADVNTR_ERR_ARG, "advntr_kwfilter_scan: null bases");
    return kwfilter_scan_spans(F, bases, n_reads ? read_off[n_reads] : 0, read_off, read_off + 1, n_reads, false, out_read, out_vntr,
                               out_count, capacity, n_out, kernel_ms);
}

extern "C" int advntr_kwfilter_scan_text(advntr_kwfilter *F, const char *text, int64_t n_bytes, const int64_t *span_start,
                                         const int64_t *span_end, int32_t n_reads, int32_t *out_read, int32_t *out_vntr,
                                         int32_t *out_count, int64_t capacity, int64_t *n_out, float *kernel_ms)
{
    if (!F || n_bytes < 0 || n_reads < 0 || !n_out || capacity < 0 || (capacity && (!out_read || !out_vntr || !out_count)) ||
        (n_reads && (!span_start || !span_end || !text)))
        return fail(ADVNTR_ERR_ARG, "advntr_kwfilter_scan_text: bad argument");
    return kwfilter_scan_spans(F, (const uint8_t *)text, n_bytes, span_start, span_end, n_reads, true, out_read, out_vntr,
                               out_count, capacity, n_out, kernel_ms);
}
