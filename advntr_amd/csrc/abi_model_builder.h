// abi_model_builder.h -- C-ABI entry points of the native model builder and of the bulk model upload (advntr_build_read_matchers, advntr_built_*, advntr_align_repeats)
// Included by engine.hip (same translation unit: uses its error helpers, device caches and HIP_TRY).
#pragma once

// ------------------------------------------------------------------------------------------------
// Native model builder (model_builder.h) behind the C ABI
// ------------------------------------------------------------------------------------------------
struct advntr_built {
    // shared with the device models made from it (they read its CSR in place instead of copying it)
    std::shared_ptr<mb::Built> sp = std::make_shared<mb::Built>();
    mb::Built &b = *sp;
};

// Host threads for the per-locus jobs (model build, table preparation).  These jobs are allocation-heavy (thousands of
// small vectors and strings per locus); measured on a 256-thread host, 6 719 loci: 32 threads 0.31-0.37 s for the
// build and 0.30 s for the upload preparation, 256 threads 0.65 s and 1.3 s (allocator contention) -- hence the cap.
// host threads of the bulk calls when the caller says 0: up to 32 (measured on the host of the GPU box: 64 or 128 threads per
// call are no faster), never more than the CPUs this process may use (host_cpu_limit); ADVNTR_HOST_THREADS overrides
static int default_host_threads() { return std::min(32, host_cpu_limit()); }

extern "C" int advntr_build_read_matchers(int32_t n_loci, const char *const *left_flank, const char *const *right_flank,
                                          const char *const *repeats, const int32_t *repeat_off, const int32_t *copies,
                                          double max_error_rate, advntr_exp_fn exp_fn, void *user, int32_t n_threads,
                                          uint32_t flags, advntr_built **out)
{
    if (n_loci < 0 || (n_loci && (!left_flank || !right_flank || !repeats || !repeat_off || !copies || !out)))
        return fail(ADVNTR_ERR_ARG, "advntr_build_read_matchers: bad argument");
    for (int i = 0; i < n_loci; ++i) out[i] = nullptr;
    if (n_threads <= 0) n_threads = default_host_threads();
    n_threads = std::min<int>(n_threads, std::max(1, n_loci));

    struct Shared {
        std::mutex exp_mu, err_mu;
        advntr_exp_fn fn;
        void *user;
        int first_bad = -1;
        std::string msg;
    } sh;
    sh.fn = exp_fn;
    sh.user = user;
    // serialise the caller's exp (a Python callback holds the interpreter lock anyway)
    auto locked_exp = [](const double *in, double *o, int64_t n, void *u) {
        Shared *s = (Shared *)u;
        std::lock_guard<std::mutex> lk(s->exp_mu);
        s->fn(in, o, n, s->user);
    };
    // ... or call it as the thread-safe strided loop it was declared to be
    auto strided_exp = [](const double *in, double *o, int64_t n, void *u) {
        Shared *s = (Shared *)u;
        typedef void (*Loop)(char **, const intptr_t *, const intptr_t *, void *);
        char *args[2] = {(char *)const_cast<double *>(in), (char *)o};
        const intptr_t dims[1] = {(intptr_t)n}, steps[2] = {(intptr_t)sizeof(double), (intptr_t)sizeof(double)};
        ((Loop)(void *)s->fn)(args, dims, steps, s->user);
    };
    const mb::ExpFn exp_call = !exp_fn ? (mb::ExpFn) nullptr
                               : (flags & ADVNTR_BUILD_EXP_STRIDED_LOOP) ? (mb::ExpFn)strided_exp : (mb::ExpFn)locked_exp;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_loci) return;
            try {
                if (!left_flank[i] || !right_flank[i]) throw std::invalid_argument("null flanking region");
                std::vector<std::string> rows;
                for (int r = repeat_off[i]; r < repeat_off[i + 1]; ++r) rows.emplace_back(repeats[r] ? repeats[r] : "");
                if (flags & ADVNTR_BUILD_ALIGN_REPEATS) {
                    bool ragged = false;
                    for (const std::string &r : rows) ragged |= r.size() != rows[0].size();
                    if (ragged) rows = msa::align_units(rows);
                }
                advntr_built *B = new advntr_built;
                B->b = mb::build_read_matcher(left_flank[i], right_flank[i], rows, copies[i], max_error_rate,
                                              exp_call, &sh);
                out[i] = B;
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(sh.err_mu);
                if (sh.first_bad < 0 || i < sh.first_bad) { sh.first_bad = i; sh.msg = e.what(); }
            }
        }
    };
    if (n_threads == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (sh.first_bad >= 0) return fail(ADVNTR_ERR_ARG, "advntr_build_read_matchers: locus %d: %s", sh.first_bad, sh.msg.c_str());
    return ADVNTR_OK;
}

extern "C" int advntr_built_info(const advntr_built *B, int32_t *info)
{
    if (!B || !info) return fail(ADVNTR_ERR_ARG, "advntr_built_info: null argument");
    info[0] = B->b.m; info[1] = B->b.silent_start; info[2] = B->b.start_index; info[3] = B->b.end_index;
    info[4] = (int32_t)B->b.in_src.size(); info[5] = (int32_t)B->b.names.size();
    return ADVNTR_OK;
}

extern "C" int advntr_built_info_many(const advntr_built *const *built, int32_t n, int32_t *info)
{
    if (n < 0 || (n && (!built || !info))) return fail(ADVNTR_ERR_ARG, "advntr_built_info_many: bad argument");
    for (int i = 0; i < n; ++i) {
        if (built[i]) (void)advntr_built_info(built[i], info + 6 * (size_t)i);
        else
            for (int k = 0; k < 6; ++k) info[6 * (size_t)i + k] = -1;
    }
    return ADVNTR_OK;
}

extern "C" int advntr_built_export(const advntr_built *B, int32_t *in_ptr, int32_t *in_src, double *in_logp,
                                   double *emis_logp, uint16_t *state_class, char *names)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_built_export: null model");
    const mb::Built &b = B->b;
    if (in_ptr) memcpy(in_ptr, b.in_ptr.data(), b.in_ptr.size() * sizeof(int32_t));
    if (in_src) memcpy(in_src, b.in_src.data(), b.in_src.size() * sizeof(int32_t));
    if (in_logp) memcpy(in_logp, b.in_logp.data(), b.in_logp.size() * sizeof(double));
    if (emis_logp) memcpy(emis_logp, b.emis.data(), b.emis.size() * sizeof(double));
    if (state_class) memcpy(state_class, b.state_class.data(), b.state_class.size() * sizeof(uint16_t));
    if (names) memcpy(names, b.names.data(), b.names.size());
    return ADVNTR_OK;
}

extern "C" advntr_hmm *advntr_built_upload(const advntr_built *B)
{
    if (!B) { fail(ADVNTR_ERR_ARG, "advntr_built_upload: null model"); return nullptr; }
    advntr_hmm *H = nullptr;
    if (advntr_built_upload_many(&B, 1, 1, &H) != ADVNTR_OK) return nullptr;
    return H;
}

// Bulk upload.  Host threads turn every model into its device blob (validation, column-program compilation,
// serialization) and append it to staging segments shared by all threads (space is reserved under a lock, the copy happens
// outside it), so no model allocates a buffer of its own and nothing is staged twice; the segments then go to ONE device
// allocation with one copy each.  The models keep reading the builder's CSR arrays in place (shared ownership), so the only
// per-model host memory is the small handle.
// The segments are page-locked and kept across calls: a copy out of pageable memory goes through the runtime's own bounce
// buffers at a third of the link's rate, and fresh pages cost a fault each.  The pool only ever grows to kMax segments and
// nothing is unpinned while the process runs -- hipHostFree waits for the device, i.e. for whatever kernel another thread's
// batch is running (a pool that gave segments back made the uploads of a pipelined run five times slower); what does not fit
// the pool is staged in pageable memory as before.
struct PinnedSegments {
    static constexpr size_t kBytes = (size_t)8 << 20;
    static constexpr size_t kMax = 48;                   // 384 MiB
    std::mutex mu;
    std::vector<void *> free_list;
    size_t allocated = 0;
    void *get()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!free_list.empty()) { void *q = free_list.back(); free_list.pop_back(); return q; }
        if (allocated >= kMax) return nullptr;
        void *q = nullptr;
        if (hipHostMalloc(&q, kBytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); allocated = kMax; return nullptr; }
        ++allocated;
        return q;
    }
    void put(void *q) { std::lock_guard<std::mutex> lk(mu); free_list.push_back(q); }
};
static PinnedSegments g_pinned_segments;

extern "C" int advntr_built_upload_many(const advntr_built *const *built, int32_t n, int32_t n_threads, advntr_hmm **out)
{
    if (n < 0 || (n && (!built || !out))) return fail(ADVNTR_ERR_ARG, "advntr_built_upload_many: bad argument");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    if (n == 0) return ADVNTR_OK;
    if (n_threads <= 0) n_threads = default_host_threads();
    n_threads = std::max(1, std::min(n_threads, n));
    constexpr size_t kSegment = PinnedSegments::kBytes;
    struct Segment {
        uint8_t *mem = nullptr;
        bool pinned = false;
        size_t cap = 0, used = 0, device_off = 0;
    };
    const int dev = current_device();
    struct Placed { int segment = -1; size_t off = 0; };
    std::vector<Segment> segments;
    std::mutex seg_mu;
    std::vector<Placed> placed(n);
    std::mutex err_mu;
    int first_bad = -1;
    std::string msg;
    std::atomic<int> next(0);
    auto release = [&]() {
        for (Segment &sgm : segments) {
            if (sgm.pinned) g_pinned_segments.put(sgm.mem);
            else delete[] sgm.mem;
            sgm.mem = nullptr;
        }
    };
    // room for `need` bytes in the last segment, or in a new one (pinned while the pool has any; a blob larger than a
    // segment gets pageable memory of its own size)
    auto reserve = [&](size_t need, Placed &where) -> uint8_t * {
        std::lock_guard<std::mutex> lk(seg_mu);
        if (segments.empty() || segments.back().used + need > segments.back().cap) {
            Segment sgm;
            sgm.cap = std::max(kSegment, need);
            if (sgm.cap == kSegment && (sgm.mem = (uint8_t *)g_pinned_segments.get())) sgm.pinned = true;
            else sgm.mem = new uint8_t[sgm.cap];                                // uninitialised: pages are touched as they fill
            segments.push_back(sgm);
        }
        Segment &sgm = segments.back();
        where.segment = (int)segments.size() - 1;
        where.off = sgm.used;
        sgm.used += need;
        return sgm.mem + where.off;
    };
    auto work = [&](int) {
        (void)hipSetDevice(dev);                               // (host threads start on device 0)
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            std::string err = "null model";
            advntr_hmm *H = nullptr;
            if (built[i]) {
                const mb::Built &b = built[i]->b;
                H = hmm_prepare(b.m, b.silent_start, b.start_index, b.end_index, (int32_t)b.in_src.size(), b.in_ptr.data(),
                                b.in_src.data(), b.in_logp.data(), b.emis.data(), b.state_class.data(), err, built[i]->sp, false);
            }
            if (!H) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (first_bad < 0 || i < first_bad) { first_bad = i; msg = err; }
                continue;
            }
            const size_t need = (H->blob_bytes + 255) & ~size_t(255);           // 256-B aligned sub-blobs
            uint8_t *dst = reserve(need, placed[i]);
            memcpy(dst, tls_blob().bytes.data(), H->blob_bytes);
            memset(dst + H->blob_bytes, 0, need - H->blob_bytes);
            out[i] = H;
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (auto &t : pool) t.join();
    }
    auto drop_all = [&]() {
        for (int i = 0; i < n; ++i) { delete out[i]; out[i] = nullptr; }
    };
    if (first_bad >= 0) {
        drop_all();
        release();
        return fail(ADVNTR_ERR_ARG, "advntr_built_upload_many: model %d: %s", first_bad, msg.c_str());
    }
    size_t total = 0;
    for (Segment &sgm : segments) { sgm.device_off = total; total += sgm.used; }
    ModelSlab *slab = new ModelSlab;
    slab->device = dev;
    slab->d = g_cache.get(dev, total, &slab->bytes);
    bool ok = slab->d != nullptr;
    // all copies queued on a stream of their own (nothing here waits for, or holds up, kernels of other batches), one wait
    hipStream_t copy_stream = g_cache.get_stream(dev);
    ok = ok && copy_stream != nullptr;
    for (Segment &sgm : segments)
        ok = ok && hipMemcpyAsync((uint8_t *)slab->d + sgm.device_off, sgm.mem, sgm.used, hipMemcpyHostToDevice, copy_stream) == hipSuccess;
    if (copy_stream) {
        ok = (hipStreamSynchronize(copy_stream) == hipSuccess) && ok;
        g_cache.put_stream(dev, copy_stream);
    }
    release();
    if (!ok) {
        if (slab->d) g_cache.put(dev, slab->d, slab->bytes);
        delete slab;
        drop_all();
        return fail(ADVNTR_ERR_DEVICE, "advntr_built_upload_many: device upload failed (%zu B)", total);
    }
    slab->refs = n;
    for (int i = 0; i < n; ++i) {
        out[i]->slab = slab;
        out[i]->device = dev;
        hmm_bind(out[i], (const uint8_t *)slab->d + segments[placed[i].segment].device_off + placed[i].off);
    }
    return ADVNTR_OK;
}

extern "C" void advntr_built_destroy(advntr_built *B) { delete B; }

extern "C" int advntr_align_repeats(const char *const *units, int32_t n, char *out, int64_t capacity, int32_t *width)
{
    if (n < 0 || (n && !units) || !width) return fail(ADVNTR_ERR_ARG, "advntr_align_repeats: bad argument");
    try {
        std::vector<std::string> in;
        for (int i = 0; i < n; ++i) in.emplace_back(units[i] ? units[i] : "");
        const std::vector<std::string> rows = msa::align_units(in);
        *width = rows.empty() ? 0 : (int32_t)rows[0].size();
        if ((int64_t)n * *width > capacity || (n && !out))
            return fail(ADVNTR_ERR_TOO_LARGE, "advntr_align_repeats: need %lld bytes", (long long)n * *width);
        for (int i = 0; i < n; ++i) memcpy(out + (size_t)i * *width, rows[i].data(), (size_t)*width);
    } catch (const std::exception &e) {
        return fail(ADVNTR_ERR_ARG, "advntr_align_repeats: %s", e.what());
    }
    return ADVNTR_OK;
}


