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
    // group keyword strings: packed key -> owners (in keyword order)
    std::map<uint64_t, std::vector<int32_t>> groups;
    std::vector<int> lengths;
    for (int w = 0; w < n_keywords; ++w) {
        const int64_t L = kw_off[w + 1] - kw_off[w];
        if (L < 1 || L > 29) {
            fail(ADVNTR_ERR_UNSUPPORTED, "advntr_kwfilter_create: keyword %d has length %lld (supported: 1..29)", w, (long long)L);
            return nullptr;
        }
        uint64_t key = 0;
        for (int64_t i = kw_off[w]; i < kw_off[w + 1]; ++i) {
            if (kw_bases[i] > 3) {
                fail(ADVNTR_ERR_SYMBOL, "advntr_kwfilter_create: keyword %d holds a non-ACGT symbol", w);
                return nullptr;
            }
            key = (key << 2) | kw_bases[i];
        }
        key |= (uint64_t)L << 58;
        groups[key].push_back(kw_vntr[w]);
        if (std::find(lengths.begin(), lengths.end(), (int)L) == lengths.end()) lengths.push_back((int)L);
    }
    if ((int)lengths.size() > KWF_MAX_LENGTHS) {
        fail(ADVNTR_ERR_UNSUPPORTED, "advntr_kwfilter_create: %zu distinct keyword lengths (max %d)", lengths.size(), KWF_MAX_LENGTHS);
        return nullptr;
    }
    std::sort(lengths.begin(), lengths.end());
    size_t slots = 1024;
    while (slots < groups.size() * 4) slots <<= 1;
    std::vector<uint64_t> keys(slots, KWF_EMPTY);
    std::vector<uint32_t> vals(slots, 0), bitset(KWF_BITSET_BITS / 32, 0);
    size_t fp_slots = 4096;
    while (fp_slots < groups.size() * 3 && fp_slots < (1u << 20)) fp_slots <<= 1;     // <= 2 MiB of uint16
    std::vector<uint16_t> fps(fp_slots, 0);
    std::vector<int32_t> ids;
    for (auto &kv : groups) {
        if (kv.second.size() > 255 || ids.size() > 0xffffffu) {
            fail(ADVNTR_ERR_UNSUPPORTED, "advntr_kwfilter_create: keyword shared by more than 255 VNTRs");
            return nullptr;
        }
        const uint64_t h = kwf_hash(kv.first);
        const unsigned b = (unsigned)(h >> 40) & (KWF_BITSET_BITS - 1);
        bitset[b >> 5] |= 1u << (b & 31);
        const unsigned b2 = (unsigned)(h >> 4) & (KWF_BITSET_BITS - 1);
        bitset[b2 >> 5] |= 1u << (b2 & 31);
        size_t fs = kwf_fp_slot(h, (uint32_t)(fp_slots - 1));
        while (fps[fs] != 0) fs = (fs + 1) & (fp_slots - 1);
        fps[fs] = kwf_fp(h);
        size_t s = (h & 0xffffffffull) & (slots - 1);
        while (keys[s] != KWF_EMPTY) s = (s + 1) & (slots - 1);
        keys[s] = kv.first;
        vals[s] = (uint32_t)ids.size() | ((uint32_t)kv.second.size() << 24);
        ids.insert(ids.end(), kv.second.begin(), kv.second.end());
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
    if (!dk || !dv || !di || !db || !df) {
        fail(ADVNTR_ERR_DEVICE, "advntr_kwfilter_create: device upload failed");
        advntr_kwfilter_destroy(F);
        return nullptr;
    }
    KwfDevice &D = F->dev;
    D.n_lengths = (int32_t)lengths.size();
    for (size_t i = 0; i < lengths.size(); ++i) {
        D.length[i] = lengths[i];
        D.mask[i] = lengths[i] >= 32 ? ~0ull : ((1ull << (2 * lengths[i])) - 1ull);
    }
    D.table_mask = slots - 1;
    D.keys = (const uint64_t *)dk; D.vals = (const uint32_t *)dv; D.ids = (const int32_t *)di; D.bitset = (const uint32_t *)db;
    D.fps = (const uint16_t *)df; D.fp_mask = (uint32_t)(fp_slots - 1);
    return F;
}

extern "C" int advntr_kwfilter_scan(advntr_kwfilter *F, const uint8_t *bases, const int64_t *read_off, int32_t n_reads,
                                    int32_t *out_read, int32_t *out_vntr, int32_t *out_count, int64_t capacity,
                                    int64_t *n_out, float *kernel_ms)
{
    if (!F || !read_off || n_reads < 0 || !n_out || capacity < 0 || (capacity && (!out_read || !out_vntr || !out_count)))
        return fail(ADVNTR_ERR_ARG, "advntr_kwfilter_scan: bad argument");
    *n_out = 0;
    if (n_reads == 0) return ADVNTR_OK;
    const int64_t total = read_off[n_reads];
    uint8_t *d_bases = nullptr;
    int64_t *d_off = nullptr;
    int32_t *d_r = nullptr, *d_v = nullptr, *d_c = nullptr;
    unsigned long long *d_n = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_bases); (void)hipFree(d_off); (void)hipFree(d_r); (void)hipFree(d_v); (void)hipFree(d_c); (void)hipFree(d_n);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc(&d_bases, (size_t)total + 16));
        HIP_TRY(hipMalloc(&d_off, ((size_t)n_reads + 1) * 8));
        HIP_TRY(hipMalloc(&d_r, std::max<int64_t>(capacity, 1) * 4));
        HIP_TRY(hipMalloc(&d_v, std::max<int64_t>(capacity, 1) * 4));
        HIP_TRY(hipMalloc(&d_c, std::max<int64_t>(capacity, 1) * 4));
        HIP_TRY(hipMalloc(&d_n, 8));
        HIP_TRY(hipMemset(d_n, 0, 8));
        if (total) HIP_TRY(hipMemcpy(d_bases, bases, (size_t)total, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_off, read_off, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        KwfArgs a{};
        a.f = F->dev; a.bases = d_bases; a.read_off = d_off; a.n_reads = n_reads;
        a.out_read = d_r; a.out_vntr = d_v; a.out_count = d_c; a.n_out = d_n; a.capacity = capacity;
        const int grid = std::max(1, std::min((n_reads + KWF_BLOCK - 1) / KWF_BLOCK, device_cus()));
        HIP_TRY(hipFuncSetAttribute((const void *)keyword_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KWF_BITSET_BITS / 8)));
        HIP_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(keyword_filter_kernel, dim3(grid), dim3(KWF_BLOCK), KWF_BITSET_BITS / 8, nullptr, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, e0, e1));
        unsigned long long n = 0;
        HIP_TRY(hipMemcpy(&n, d_n, 8, hipMemcpyDeviceToHost));
        *n_out = (int64_t)n;
        if ((int64_t)n > capacity) return fail(ADVNTR_ERR_TOO_LARGE, "advntr_kwfilter_scan: %llu records, capacity %lld", n, (long long)capacity);
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


