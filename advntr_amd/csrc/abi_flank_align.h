// abi_flank_align.h -- C-ABI entry point of the flank alignment (advntr_flank_align)
// Included by engine.hip (same translation unit: uses its error helpers, device caches and HIP_TRY).
#pragma once

// ------------------------------------------------------------------------------------------------
// Flank alignment (flank_align.h) behind the C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int advntr_flank_align(const uint8_t *bases, const int64_t *read_off, int32_t n_reads, const uint8_t *flank_bases,
                                  const int32_t *flank_off, int32_t n_flanks, const int32_t *pair_read, const int32_t *pair_flank,
                                  int32_t n_pairs, int32_t *out_score, int32_t *out_begin, int32_t *out_end, float *kernel_ms)
{
    if (n_reads < 0 || n_flanks < 0 || n_pairs < 0 || !read_off || !flank_off ||
        (n_pairs && (!pair_read || !pair_flank || !out_score || !out_begin || !out_end)))
        return fail(ADVNTR_ERR_ARG, "advntr_flank_align: bad argument");
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_pairs == 0) return ADVNTR_OK;
    for (int f = 0; f < n_flanks; ++f)
        if (flank_off[f + 1] - flank_off[f] < 0 || flank_off[f + 1] - flank_off[f] > 64 * FA_K)
            return fail(ADVNTR_ERR_TOO_LARGE, "advntr_flank_align: flank %d has %d bases (limit %d)", f,
                        flank_off[f + 1] - flank_off[f], 64 * FA_K);
    for (int p = 0; p < n_pairs; ++p)
        if (pair_read[p] < 0 || pair_read[p] >= 2 * (int64_t)n_reads || pair_flank[p] < 0 || pair_flank[p] >= n_flanks)
            return fail(ADVNTR_ERR_ARG, "advntr_flank_align: pair %d out of range", p);
    const int64_t total = read_off[n_reads];
    const int32_t ftotal = flank_off[n_flanks];
    const int dev = current_device();
    void *d_bases = nullptr, *d_off = nullptr, *d_fb = nullptr, *d_fo = nullptr, *d_pr = nullptr, *d_pf = nullptr, *d_out = nullptr;
    hipEvent_t e0 = g_cache.get_event(dev), e1 = g_cache.get_event(dev);
    std::vector<std::pair<void *, size_t>> held;          // buffers from the per-device cache, like a batch's
    auto take = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *q = g_cache.get(dev, bytes, &got);
        if (q) held.emplace_back(q, got);
        return q;
    };
    auto cleanup = [&]() {
        for (auto &h : held) g_cache.put(dev, h.first, h.second);
        if (e0) g_cache.put_event(dev, e0);
        if (e1) g_cache.put_event(dev, e1);
    };
    int rc = [&]() -> int {
        if (!e0 || !e1) return fail(ADVNTR_ERR_DEVICE, "advntr_flank_align: event creation failed");
        d_bases = take((size_t)total + 16); d_off = take(((size_t)n_reads + 1) * 8);
        d_fb = take((size_t)ftotal + 16); d_fo = take(((size_t)n_flanks + 1) * 4);
        d_pr = take((size_t)n_pairs * 4); d_pf = take((size_t)n_pairs * 4); d_out = take((size_t)n_pairs * 12);
        void *d_order = take((size_t)n_pairs * 4), *d_next = take(4);
        if (!d_bases || !d_off || !d_fb || !d_fo || !d_pr || !d_pf || !d_out || !d_order || !d_next)
            return fail(ADVNTR_ERR_DEVICE, "advntr_flank_align: device allocation failed");
        if (total) HIP_TRY(hipMemcpy(d_bases, bases, (size_t)total, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_off, read_off, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        if (ftotal) HIP_TRY(hipMemcpy(d_fb, flank_bases, (size_t)ftotal, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_fo, flank_off, ((size_t)n_flanks + 1) * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_pr, pair_read, (size_t)n_pairs * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_pf, pair_flank, (size_t)n_pairs * 4, hipMemcpyHostToDevice));
        {   // longest reads first (a counting sort would do; pair counts are modest), taken from a counter on the device
            std::vector<int32_t> order((size_t)n_pairs);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                const int32_t rx = pair_read[x] % n_reads, ry = pair_read[y] % n_reads;
                return read_off[rx + 1] - read_off[rx] > read_off[ry + 1] - read_off[ry];
            });
            HIP_TRY(hipMemcpy(d_order, order.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemset(d_next, 0, 4));
        }
        FaArgs a{};
        a.bases = (const uint8_t *)d_bases; a.read_off = (const int64_t *)d_off;
        a.flank_bases = (const uint8_t *)d_fb; a.flank_off = (const int32_t *)d_fo;
        a.pair_read = (const int32_t *)d_pr; a.pair_flank = (const int32_t *)d_pf; a.n_pairs = n_pairs; a.n_reads = n_reads;
        a.out_score = (int32_t *)d_out; a.out_begin = a.out_score + n_pairs; a.out_end = a.out_begin + n_pairs;
        a.order = (const int32_t *)d_order; a.next = (int32_t *)d_next;
        // eight wavefronts per SIMD (the sweep holds 43 registers): issue-bound work wants them all
        const int grid = std::max(1, std::min((n_pairs + FA_WAVES - 1) / FA_WAVES, device_cus() * 8));
        HIP_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(flank_align_kernel, dim3(grid), dim3(FA_WAVES * 64), 0, nullptr, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, e0, e1));
        HIP_TRY(hipMemcpy(out_score, a.out_score, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_begin, a.out_begin, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_end, a.out_end, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
        return ADVNTR_OK;
    }();
    std::string keep = g_err;
    cleanup();
    g_err = keep;
    return rc;
}

