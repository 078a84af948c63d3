// engine.hip -- host side of libadvntr_hip.so: the C ABI of include/advntr_hip.h, model upload,
// device-resident read batches and kernel launches.  gfx950 only; no CPU fallback anywhere: every
// entry point either runs the HIP kernels or returns an error code.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/advntr_hip.h"
#include "../../include/advntr_pyhost.h"
#include "device_model.h"
#include "column_program.h"
#include "viterbi_generic.h"
#include "viterbi_columns.h"
#include "column_launch.h"
#include "forward_generic.h"
#include "keyword_filter.h"
#include "model_builder.h"
#include "repeat_msa.h"
#include "flank_align.h"

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ADVNTR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                       \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Device-memory and stream caches.  A locus-sized call (a few hundred reads) is dominated by hipMalloc/hipFree and
// stream/event creation, not by the kernel; batches therefore draw their buffers, stream and events from small
// per-process caches (per device) and give them back on destroy.  advntr_trim() releases everything.
// ------------------------------------------------------------------------------------------------
namespace {

struct DeviceCaches {
    std::mutex mu;
    std::multimap<size_t, void *> blocks[16];          // per device: free blocks by size
    size_t cached[16] = {0};
    std::vector<hipStream_t> streams[16], streams_second[16];   // (second class: ADVNTR_FLAG_SECOND_QUEUE)
    std::vector<hipEvent_t> events[16];
    static constexpr size_t kMaxCached = (size_t)24 << 30;

    static size_t round_up(size_t b) { return b <= (1u << 20) ? ((b + 4095) & ~size_t(4095)) : ((b + (1u << 21) - 1) & ~size_t((1u << 21) - 1)); }

    void *get(int dev, size_t bytes, size_t *got)
    {
        bytes = round_up(std::max<size_t>(bytes, 16));
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = blocks[dev].lower_bound(bytes);
            if (it != blocks[dev].end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
                void *p = it->second;
                *got = it->first;
                cached[dev] -= it->first;
                blocks[dev].erase(it);
                return p;
            }
        }
        void *p = nullptr;
#ifdef ADVNTR_TRACE_ALLOC
        const auto t_alloc = std::chrono::steady_clock::now();
#endif
        if (hipMalloc(&p, bytes) != hipSuccess) {
            trim(dev);                                  // cached blocks may be what is missing
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        }
#ifdef ADVNTR_TRACE_ALLOC
        fprintf(stderr, "[alloc] hipMalloc %zu B: %.2f ms (at %.1f ms)\n", bytes,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count(),
                std::chrono::duration<double, std::milli>(t_alloc.time_since_epoch()).count());
#endif
        *got = bytes;
        return p;
    }

    void put(int dev, void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (cached[dev] + bytes > kMaxCached) { (void)hipFree(p); return; }
        blocks[dev].emplace(bytes, p);
        cached[dev] += bytes;
    }

    void trim(int dev)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : blocks[dev]) (void)hipFree(kv.second);
        blocks[dev].clear();
        cached[dev] = 0;
    }

    hipStream_t get_stream(int dev)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!streams[dev].empty()) { hipStream_t s = streams[dev].back(); streams[dev].pop_back(); return s; }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        return s;
    }
    void put_stream(int dev, hipStream_t s) { std::lock_guard<std::mutex> lk(mu); streams[dev].push_back(s); }

    // a stream of the second class: the highest stream priority the device offers (the runtime keeps the hardware queues of
    // different priorities apart), or a plain one where it offers none
    hipStream_t get_stream_second(int dev)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!streams_second[dev].empty()) { hipStream_t s = streams_second[dev].back(); streams_second[dev].pop_back(); return s; }
        }
        int least = 0, greatest = 0;
        hipStream_t s = nullptr;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
            hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) == hipSuccess)
            return s;
        (void)hipGetLastError();
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        return s;
    }
    void put_stream_second(int dev, hipStream_t s) { std::lock_guard<std::mutex> lk(mu); streams_second[dev].push_back(s); }

    hipEvent_t get_event(int dev)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!events[dev].empty()) { hipEvent_t e = events[dev].back(); events[dev].pop_back(); return e; }
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void put_event(int dev, hipEvent_t e) { std::lock_guard<std::mutex> lk(mu); events[dev].push_back(e); }
};

DeviceCaches g_cache;

// The device advntr_set_device chose is the process's: HIP keeps the current device per host thread, so a thread that has not
// chosen one itself (a host-side worker that prepares the next piece of a run) adopts it on its first call here.
std::atomic<int> g_process_device{-1};
thread_local bool t_device_chosen = false;

int current_device()
{
    if (!t_device_chosen) {
        t_device_chosen = true;
        const int want = g_process_device.load();
        if (want >= 0) (void)hipSetDevice(want);
    }
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) d = 0;
    return d;
}

}  // namespace

extern "C" const char *advntr_last_error(void) { return g_err.c_str(); }
extern "C" const char *advntr_version(void) { return "advntr_hip 0.1 (gfx950)"; }

extern "C" void advntr_trim(void)
{
    for (int d = 0; d < 16; ++d) g_cache.trim(d);
}

extern "C" int advntr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int advntr_set_device(int device)
{
    // ADVNTR_BLOCKING_SYNC=1: this process's waits on the device sleep instead of spinning (hipDeviceScheduleBlockingSync).  A rank
    // of a multi-GPU job sets it (advntr_amd/comm.py): N ranks spinning in hipStreamSynchronize beside RCCL's proxy threads burn N
    // cores of a CPU quota that the host stages need, and a process that exhausts its quota is stopped as a whole -- the thread
    // that launches kernels included -- for the rest of the accounting period (DESIGN.md sections 7 and 8).  Must precede the first use
    // of the device; a runtime that refuses the flag (device already active) is not an error.
    if (const char *e = getenv("ADVNTR_BLOCKING_SYNC"))
        if (atoi(e) > 0 && hipSetDevice(device) == hipSuccess && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess)
            (void)hipGetLastError();
    HIP_TRY(hipSetDevice(device));
    g_process_device.store(device);
    t_device_chosen = true;
    return ADVNTR_OK;
}

// ------------------------------------------------------------------------------------------------
// Model
// ------------------------------------------------------------------------------------------------
// read-only view of an array that is either owned by the model (advntr_hmm_create copies the caller's CSR) or shared
// with the native builder's result (advntr_built_upload*: no copy, the Built stays alive through a shared_ptr)
template <class T> struct ArrayView {
    const T *p = nullptr;
    size_t n = 0;
    const T &operator[](size_t i) const { return p[i]; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    void set(const std::vector<T> &v) { p = v.data(); n = v.size(); }
};

// what stays of a model's column program on the host once its tables are serialized into the device blob
struct ColProgramSummary {
    bool valid = false;
    int32_t n_cols = 0;
    std::string why;
};

struct advntr_hmm {
    int32_t m = 0, P = 0, start = 0, end = 0, finite = 0, n_edges = 0;
    int32_t bp_width = 1, max_indeg = 0;
    ArrayView<int32_t> in_ptr, in_src;
    ArrayView<double> in_logp, emis;
    ArrayView<uint16_t> sclass;
    std::vector<int32_t> own_in_ptr, own_in_src;          // storage behind the views when the model owns its CSR
    std::vector<double> own_in_logp, own_emis;
    std::vector<uint16_t> own_sclass;
    std::shared_ptr<const mb::Built> built;               // ... or the builder's result they point into
    bool has_class = false;
    ColProgramSummary colprog;    // valid = false when the model is not a recognised read matcher
    int32_t col_lds_bytes = 0;     // LDS-resident tables of the column program: classes, emissions, column info, states
    int32_t col_lds_core = 0;      // ... without the state table (only the traceback reads it)
    int32_t col_lds_min = 0;       // ... without the column-info table either (the sweep indexes a padded copy of it)
    bool generic_ok = true;
    void *d_blob = nullptr;        // own allocation, or nullptr when the model lives in a shared slab
    struct ModelSlab *slab = nullptr;
    size_t blob_bytes = 0;
    std::vector<uint8_t> host_blob;   // core blob (state classes + column program), filled by hmm_prepare, released once
                                      // the model is on the device; generic-only models carry their kernel tables in it too
    size_t off[14] = {0};             // table offsets: [0..11] generic-kernel tables, [12] classes, [13] column program
    bool gen_in_blob = false;         // the generic-kernel tables are part of host_blob / the core allocation
    void *d_gen = nullptr;            // generic-kernel tables uploaded on demand (hmm_ensure_generic)
    std::mutex gen_mu;            // guards d_gen and the generic-kernel pointers of `dev`
    int device = 0;               // the device that holds the blob
    DevModel dev{};
    int live_batches = 0;         // batches bound to this model (guarded by g_model_life): a model destroyed while one of them is
    bool doomed = false;          // alive is only marked and goes with the last of them (advntr_batch_destroy)
};
static std::mutex g_model_life;

// One device allocation shared by the models of a bulk upload (advntr_built_upload_many); freed with its last model.
// (a block of the per-device cache, like a batch's buffers: it goes back there, not to hipFree, which would wait for whatever
// kernels other threads' batches are running; the batches that used these models have synchronised their streams before their
// models may be destroyed -- the C-ABI contract, and what the Python wrappers enforce by holding references)
struct ModelSlab {
    void *d = nullptr;
    size_t bytes = 0;
    int device = 0;
    std::atomic<int> refs{0};
};

namespace {

struct BlobBuilder {
    std::vector<uint8_t> bytes;
    size_t add_raw(const void *p, size_t n_bytes, size_t elem)
    {
        size_t off = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(off + std::max<size_t>(n_bytes, elem) + 16, 0);
        if (n_bytes) memcpy(bytes.data() + off, p, n_bytes);
        return off;
    }
    template <class T> size_t add(const std::vector<T> &v) { return add_raw(v.data(), v.size() * sizeof(T), sizeof(T)); }
    template <class T> size_t add(const ArrayView<T> &v) { return add_raw(v.data(), v.size() * sizeof(T), sizeof(T)); }
};

}  // namespace

// the calling thread's serialization buffer: hmm_prepare leaves the model's blob in it (capacity kept across models)
static BlobBuilder &tls_blob()
{
    static thread_local BlobBuilder B;
    return B;
}

// Tables of the generic-CSR kernel, appended to `B`; o[0..11] receive their offsets.
static void generic_tables(const advntr_hmm &H, BlobBuilder &B, size_t *o)
{
    const int m = H.m, P = H.P, S = m - P;
    const int32_t *in_ptr = H.in_ptr.data(), *in_src = H.in_src.data();
    const double *in_logp = H.in_logp.data();
    // emitting CSR = the reference's, verbatim
    std::vector<int32_t> e_ptr(in_ptr, in_ptr + P + 1);
    const int eE = P ? in_ptr[P] : 0;
    std::vector<int32_t> e_src(in_src, in_src + eE);
    std::vector<double> e_logp(in_logp, in_logp + eE);
    int max_indeg = 0;
    for (int l = 0; l < P; ++l) max_indeg = std::max(max_indeg, in_ptr[l + 1] - in_ptr[l]);

    // silent states: reference evaluation order r = [emitting-sourced in list order (hmm.pyx:2044-2063)]
    // ++ [silent-sourced with ki < l in list order (hmm.pyx:2065-2083)]; edges from ki >= l never fire.
    std::vector<int32_t> r_ptr(S + 1, 0), r_src;
    std::vector<double> r_logp;
    std::vector<int32_t> s_ptr(S + 1, 0), s_mid(S, 0), s_src, s_ord;
    std::vector<double> s_logp;
    for (int l = P; l < m; ++l) {
        const int ls = l - P;
        const int base = P + (ls / ADV_WAVE) * ADV_WAVE;
        const int r0 = (int)r_src.size();
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (in_src[k] < P) { r_src.push_back(in_src[k]); r_logp.push_back(in_logp[k]); }
        for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k)
            if (in_src[k] >= P && in_src[k] < l) { r_src.push_back(in_src[k]); r_logp.push_back(in_logp[k]); }
        const int r1 = (int)r_src.size();
        r_ptr[ls + 1] = r1;
        max_indeg = std::max(max_indeg, r1 - r0);
        // A list: sources outside this state's chunk, reference order
        s_ptr[ls] = (int)s_src.size();
        for (int q = r0; q < r1; ++q)
            if (r_src[q] < base) { s_src.push_back(r_src[q]); s_logp.push_back(r_logp[q]); s_ord.push_back(q - r0); }
        s_mid[ls] = (int)s_src.size();
        // B list: in-chunk sources, ascending source index
        std::vector<int> idx;
        for (int q = r0; q < r1; ++q)
            if (r_src[q] >= base) idx.push_back(q);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return r_src[a] < r_src[b]; });
        for (int q : idx) { s_src.push_back(r_src[q]); s_logp.push_back(r_logp[q]); s_ord.push_back(q - r0); }
        s_ptr[ls + 1] = (int)s_src.size();
    }

    (void)max_indeg;
    o[0] = B.add(e_ptr); o[1] = B.add(e_src); o[2] = B.add(e_logp); o[3] = B.add(H.emis);
    o[4] = B.add(s_ptr); o[5] = B.add(s_mid); o[6] = B.add(s_src); o[7] = B.add(s_ord);
    o[8] = B.add(s_logp); o[9] = B.add(r_ptr); o[10] = B.add(r_src); o[11] = B.add(r_logp);
}

// Host half of advntr_hmm_create: validation, kernel-side tables, column program, the serialized blob.  No HIP call,
// no global state: safe to run on many threads (errors come back through `err`).
static advntr_hmm *hmm_prepare(int32_t m, int32_t silent_start, int32_t start_index, int32_t end_index, int32_t n_edges,
                               const int32_t *in_ptr, const int32_t *in_src, const double *in_logp,
                               const double *emis_logp, const uint16_t *state_class, std::string &err,
                               const std::shared_ptr<const mb::Built> &shared = nullptr, bool own_blob = true)
{
    auto fail = [&err](int, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return 0;
    };
    if (m <= 0 || silent_start < 0 || silent_start > m || start_index < 0 || start_index >= m ||
        end_index < 0 || end_index >= m || n_edges < 0 || !in_ptr || (n_edges && (!in_src || !in_logp)) ||
        (silent_start && !emis_logp)) {
        fail(ADVNTR_ERR_ARG, "advntr_hmm_create: bad argument");
        return nullptr;
    }
    if (in_ptr[0] != 0 || in_ptr[m] != n_edges) {
        fail(ADVNTR_ERR_ARG, "advntr_hmm_create: in_ptr does not span n_edges");
        return nullptr;
    }
    for (int l = 0; l < m; ++l)
        if (in_ptr[l + 1] < in_ptr[l]) {
            fail(ADVNTR_ERR_ARG, "advntr_hmm_create: in_ptr not monotone");
            return nullptr;
        }
    for (int k = 0; k < n_edges; ++k)
        if (in_src[k] < 0 || in_src[k] >= m) {
            fail(ADVNTR_ERR_ARG, "advntr_hmm_create: edge source out of range");
            return nullptr;
        }
    const int P = silent_start;
    advntr_hmm *H = new advntr_hmm();
    H->m = m; H->P = P; H->start = start_index; H->end = end_index; H->n_edges = n_edges;
    if (shared) {
        // the arrays live in the builder's result, which this model keeps alive: nothing is copied
        H->built = shared;
        H->in_ptr.p = in_ptr; H->in_ptr.n = (size_t)m + 1;
        H->in_src.p = in_src; H->in_src.n = (size_t)n_edges;
        H->in_logp.p = in_logp; H->in_logp.n = (size_t)n_edges;
        H->emis.p = emis_logp; H->emis.n = (size_t)P * 4;
        H->sclass.p = state_class; H->sclass.n = (size_t)m;
        H->has_class = state_class != nullptr;
    } else {
        H->own_in_ptr.assign(in_ptr, in_ptr + m + 1);
        H->own_in_src.assign(in_src, in_src + n_edges);
        H->own_in_logp.assign(in_logp, in_logp + n_edges);
        H->own_emis.assign(emis_logp, emis_logp + (size_t)P * 4);
        H->own_sclass.assign(m, 0);
        if (state_class) {
            H->own_sclass.assign(state_class, state_class + m);
            H->has_class = true;
        }
        H->in_ptr.set(H->own_in_ptr); H->in_src.set(H->own_in_src); H->in_logp.set(H->own_in_logp);
        H->emis.set(H->own_emis); H->sclass.set(H->own_sclass);
    }
    if (!H->sclass.p) {            // shared arrays without class words (not produced by the builder; kept for safety)
        H->own_sclass.assign(m, 0);
        H->sclass.set(H->own_sclass);
    }
    H->finite = (in_ptr[end_index + 1] - in_ptr[end_index]) != 0;   // hmm.pyx:977-980

    // widest in-edge list the generic kernel will see (decides its back-pointer width)
    {
        int max_indeg = 0;
        for (int l = 0; l < P; ++l) max_indeg = std::max(max_indeg, in_ptr[l + 1] - in_ptr[l]);
        for (int l = P; l < m; ++l) {
            int deg = 0;
            for (int k = in_ptr[l]; k < in_ptr[l + 1]; ++k) deg += in_src[k] < l;
            max_indeg = std::max(max_indeg, deg);
        }
        H->max_indeg = max_indeg;
        H->bp_width = max_indeg > 255 ? 2 : 1;
    }

    // column program for flank-repeats-flank read matchers (optional fast path); its tables are scratch of this thread
    // (they end up in the blob), only the summary stays with the model
    static thread_local ColProgramHost prog;
    build_column_program(*H, prog);
    H->colprog.valid = prog.valid;
    H->colprog.n_cols = prog.n_cols;
    H->colprog.why = prog.why;
    // the generic kernel keeps 2 fp64 trellis rows in the 160 KiB LDS of one CU; a bigger model is only
    // usable through its column program
    H->generic_ok = (size_t)m * 16 <= 160 * 1024;
    if (!H->generic_ok && !H->colprog.valid) {
        fail(ADVNTR_ERR_TOO_LARGE, "advntr_hmm_create: %d states need %zu B of LDS rows (> 160 KiB) and the model has "
             "no column program (%s)", m, (size_t)m * 16, H->colprog.why.c_str());
        delete H;
        return nullptr;
    }

    // the generic kernel's tables are three quarters of a read matcher's device data and are only needed when a read
    // cannot take the column kernel (forced, or longer than COL_MAX_LONG_READ): models with a column program upload them on
    // first use (hmm_ensure_generic), generic-only models carry them from the start
    BlobBuilder &B = tls_blob();
    B.bytes.clear();
    size_t offs[14] = {0};
    H->gen_in_blob = !H->colprog.valid;
    if (H->gen_in_blob) generic_tables(*H, B, offs);
    offs[12] = B.add(H->sclass);
    if (H->colprog.valid) {
        static thread_local std::vector<uint8_t> colblob;
        prog.serialize_into(colblob);
        H->col_lds_bytes = ((const ColProgram *)colblob.data())->lds_bytes;
        H->col_lds_core = (((const ColProgram *)colblob.data())->off_state - ((const ColProgram *)colblob.data())->off_class + 15) & ~15;
        H->col_lds_min = (((const ColProgram *)colblob.data())->off_info - ((const ColProgram *)colblob.data())->off_class + 15) & ~15;
        offs[13] = B.add(colblob);
    }
    H->blob_bytes = B.bytes.size();
    if (own_blob) H->host_blob.assign(B.bytes.begin(), B.bytes.end());      // (a bulk upload copies it out of tls_blob() itself)
    memcpy(H->off, offs, sizeof offs);
    return H;
}

static void bind_generic(DevModel &D, const uint8_t *d, const size_t *o)
{
    D.e_ptr = (const int32_t *)(d + o[0]); D.e_src = (const int32_t *)(d + o[1]);
    D.e_logp = (const double *)(d + o[2]); D.emis = (const double *)(d + o[3]);
    D.s_ptr = (const int32_t *)(d + o[4]); D.s_mid = (const int32_t *)(d + o[5]);
    D.s_src = (const int32_t *)(d + o[6]); D.s_ord = (const int32_t *)(d + o[7]);
    D.s_logp = (const double *)(d + o[8]); D.r_ptr = (const int32_t *)(d + o[9]);
    D.r_src = (const int32_t *)(d + o[10]); D.r_logp = (const double *)(d + o[11]);
}

// Upload the generic kernel's tables of a model that has so far only run on its column program.
static int hmm_ensure_generic(advntr_hmm *H)
{
    std::lock_guard<std::mutex> lk(H->gen_mu);
    if (H->gen_in_blob || H->d_gen) return ADVNTR_OK;
    if (current_device() != H->device)
        return fail(ADVNTR_ERR_DEVICE, "model lives on device %d, the current device is %d", H->device, current_device());
    BlobBuilder B;
    size_t o[12];
    generic_tables(*H, B, o);
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, B.bytes.size()));
    const hipError_t e = hipMemcpy(d, B.bytes.data(), B.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(ADVNTR_ERR_DEVICE, "upload of the generic-kernel tables failed: %s", hipGetErrorString(e));
    }
    bind_generic(H->dev, (const uint8_t *)d, o);
    H->d_gen = d;              // published last: a later call returns early only when the tables are bound
    return ADVNTR_OK;
}

// Point the model's device view at its blob, uploaded at device address `d`.
static void hmm_bind(advntr_hmm *H, const void *dptr)
{
    const uint8_t *d = (const uint8_t *)dptr;
    const size_t *o = H->off;
    DevModel &D = H->dev;
    D.m = H->m; D.P = H->P; D.start = H->start; D.end = H->end; D.finite = H->finite;
    D.bp_width = H->bp_width; D.n_chunks = (H->m - H->P + ADV_WAVE - 1) / ADV_WAVE; D.max_indeg = H->max_indeg;
    if (H->gen_in_blob) bind_generic(D, d, o);
    D.sclass = (const uint16_t *)(d + o[12]);
    D.cols = H->colprog.valid ? (const ColProgram *)(d + o[13]) : nullptr;
    std::vector<uint8_t>().swap(H->host_blob);
}

extern "C" advntr_hmm *advntr_hmm_create(int32_t m, int32_t silent_start, int32_t start_index,
                                         int32_t end_index, int32_t n_edges, const int32_t *in_ptr,
                                         const int32_t *in_src, const double *in_logp,
                                         const double *emis_logp, const uint16_t *state_class)
{
    std::string err;
    advntr_hmm *H = hmm_prepare(m, silent_start, start_index, end_index, n_edges, in_ptr, in_src, in_logp, emis_logp,
                                state_class, err);
    if (!H) {
        fail(err.find("LDS rows") != std::string::npos ? ADVNTR_ERR_TOO_LARGE : ADVNTR_ERR_ARG, "%s", err.c_str());
        return nullptr;
    }
    if (hipMalloc(&H->d_blob, H->blob_bytes) != hipSuccess ||
        hipMemcpy(H->d_blob, H->host_blob.data(), H->blob_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fail(ADVNTR_ERR_DEVICE, "advntr_hmm_create: device upload failed (%zu B)", H->blob_bytes);
        if (H->d_blob) (void)hipFree(H->d_blob);
        delete H;
        return nullptr;
    }
    H->device = current_device();
    hmm_bind(H, H->d_blob);
    return H;
}

static void hmm_release(advntr_hmm *H);

// The contract says a model outlives its batches.  A caller that breaks it (destroys a model while an asynchronous
// advntr_batch_run may still read it, or before advntr_batch_destroy) used to get its memory reissued at once -- wrong
// scores, silently.  The destruction is deferred instead: the last batch bound to the model releases it after its stream
// has been synchronised.
extern "C" void advntr_hmm_destroy(advntr_hmm *H)
{
    if (!H) return;
    {
        std::lock_guard<std::mutex> lock(g_model_life);
        if (H->live_batches > 0) { H->doomed = true; return; }
    }
    hmm_release(H);
}

static void hmm_release(advntr_hmm *H)
{
    if (H->d_blob) (void)hipFree(H->d_blob);
    if (H->d_gen) (void)hipFree(H->d_gen);
    if (H->slab && H->slab->refs.fetch_sub(1) == 1) {
        g_cache.put(H->slab->device, H->slab->d, H->slab->bytes);
        delete H->slab;
    }
    delete H;
}

extern "C" int advntr_hmm_has_column_program(const advntr_hmm *H) { return H && H->colprog.valid ? 1 : 0; }

extern "C" int advntr_hmm_info(const advntr_hmm *H, int32_t *m, int32_t *silent_start, int32_t *n_edges,
                               int32_t *n_columns)
{
    if (!H) return fail(ADVNTR_ERR_ARG, "advntr_hmm_info: null model");
    if (m) *m = H->m;
    if (silent_start) *silent_start = H->P;
    if (n_edges) *n_edges = H->n_edges;
    if (n_columns) *n_columns = H->colprog.valid ? H->colprog.n_cols : 0;
    return ADVNTR_OK;
}

// ------------------------------------------------------------------------------------------------
// Batch
// ------------------------------------------------------------------------------------------------
struct advntr_batch {
    std::vector<advntr_hmm *> models;
    int32_t n_reads = 0;
    uint32_t flags = 0;
    int n_max = 0, m_max = 0;
    // generic-kernel launch (reads without a column program, empty reads, reads longer than COL_MAX_READ)
    int n_gen = 0, grid_gen = 0, m_max_gen = 0;
    size_t lds_gen = 0;
    int64_t bp_stride_gen = 0;
    // anti-diagonal kernel launch
    int n_col = 0;
    ColumnLaunch col{};
    int64_t device_bytes = 0;
    std::vector<int64_t> path_off;      // internal capacities (n + m + 2 per read)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevModel *d_models = nullptr;
    uint8_t *d_bases = nullptr;
    int64_t *d_read_off = nullptr;
    int32_t *d_read_model = nullptr, *d_order = nullptr;   // d_order: [column reads | generic reads]
    double *d_logp = nullptr;
    int32_t *d_summary = nullptr, *d_path = nullptr, *d_path_len = nullptr;
    int64_t *d_path_off = nullptr;
    uint8_t *d_bp_gen = nullptr;
    int32_t *d_pathbuf_gen = nullptr, *d_pathbuf_col = nullptr;
    // advntr_batch_recruit (abi_recruit.h): per-read choice, per-wavefront counts, the survivors' records
    uint8_t *d_rec_choice = nullptr, *d_rec_reversed = nullptr;
    int32_t *d_rec_count = nullptr, *d_rec_index = nullptr, *d_rec_summary = nullptr;
    double *d_rec_scaled = nullptr, *d_rec_logp = nullptr;
    int32_t n_recruited = 0;
    bool d_rec_ready = false;           // every d_rec_* buffer is there
    int32_t *d_counter = nullptr;       // [0]: generic dequeue head, [3..8]: tile heads of the column kernels' lists
    int32_t path_cap = 0;
    int device = 0;
    std::vector<std::pair<void *, size_t>> allocs;

    template <class T> int dmalloc(T **p, size_t count)
    {
        size_t bytes = std::max<size_t>(count, 1) * sizeof(T), got = 0;
        void *q = g_cache.get(device, bytes, &got);
        if (!q) return fail(ADVNTR_ERR_DEVICE, "device allocation of %zu bytes failed", bytes);
        allocs.emplace_back(q, got);
        device_bytes += (int64_t)bytes;
        *p = (T *)q;
        return ADVNTR_OK;
    }
};

static int device_cus()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

// second strand of a both-strands batch: call n_fwd + i = reverse complement of read i (one wavefront per read)
__global__ void __launch_bounds__(256) reverse_complement_kernel(uint8_t *bases, const int64_t *read_off, const int n_fwd)
{
    const int lane = threadIdx.x & 63;
    const int64_t total = read_off[n_fwd];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_fwd; r += gridDim.x * 4) {
        const int64_t o = read_off[r], n = read_off[r + 1] - o;
        for (int64_t j = lane; j < n; j += 64) bases[total + o + j] = (uint8_t)(3 - bases[o + n - 1 - j]);
    }
}

static int batch_build(advntr_batch *B, advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                       const int64_t *read_off_in, const int32_t *read_model_in, int32_t n_reads_in, uint32_t flags)
{
    if (!models || n_models <= 0 || !read_off_in || !read_model_in || n_reads_in < 0 ||
        (n_reads_in && !bases && read_off_in[n_reads_in] > 0))
        return fail(ADVNTR_ERR_ARG, "batch: bad argument");
    for (int i = 0; i < n_models; ++i)
        if (!models[i]) return fail(ADVNTR_ERR_ARG, "batch: null model %d", i);
    // ADVNTR_FLAG_BOTH_STRANDS: the caller passes the forward reads; calls n .. 2n-1 are their reverse complements
    // (process_unmapped_read scores both, vntr_finder.py:239-242), made on the device from the uploaded forward bases
    const bool both = (flags & ADVNTR_FLAG_BOTH_STRANDS) != 0 && n_reads_in > 0;
    std::vector<int64_t> off2;
    std::vector<int32_t> model2;
    if (both) {
        if (n_reads_in > 0x3fffffff) return fail(ADVNTR_ERR_TOO_LARGE, "batch: too many reads for a both-strands batch");
        if (read_off_in[0] != 0) return fail(ADVNTR_ERR_ARG, "batch: read_off[0] must be 0");
        off2.assign(read_off_in, read_off_in + n_reads_in + 1);
        off2.resize((size_t)2 * n_reads_in + 1);
        model2.assign(read_model_in, read_model_in + n_reads_in);
        model2.insert(model2.end(), read_model_in, read_model_in + n_reads_in);
        const int64_t fwd = read_off_in[n_reads_in];
        for (int r = 0; r < n_reads_in; ++r) off2[(size_t)n_reads_in + 1 + r] = fwd + read_off_in[r + 1];
    }
    const int64_t *read_off = both ? off2.data() : read_off_in;
    const int32_t *read_model = both ? model2.data() : read_model_in;
    const int32_t n_reads = both ? 2 * n_reads_in : n_reads_in;
    const int64_t total_given = read_off_in[n_reads_in];          // bytes the caller's `bases` holds
    B->models.assign(models, models + n_models);
    {
        std::lock_guard<std::mutex> lock(g_model_life);
        for (advntr_hmm *H : B->models) H->live_batches++;
    }
    B->n_reads = n_reads;
    B->flags = flags;
    if (read_off[0] != 0) return fail(ADVNTR_ERR_ARG, "batch: read_off[0] must be 0");
    const int64_t total = read_off[n_reads];
    for (int r = 0; r < n_reads; ++r) {
        if (read_off[r + 1] < read_off[r]) return fail(ADVNTR_ERR_ARG, "batch: read_off not monotone at %d", r);
        if (read_model[r] < 0 || read_model[r] >= n_models)
            return fail(ADVNTR_ERR_ARG, "batch: read_model[%d]=%d out of range", r, read_model[r]);
        B->n_max = std::max<int>(B->n_max, (int)(read_off[r + 1] - read_off[r]));
    }
    {   // symbol check, eight codes per test (a valid code has no bit above the low two)
        uint64_t bad = 0;
        int64_t i = 0;
        for (; i + 8 <= total_given; i += 8) {
            uint64_t w;
            memcpy(&w, bases + i, 8);
            bad |= w & 0xFCFCFCFCFCFCFCFCull;
        }
        for (; i < total_given; ++i) bad |= (uint64_t)(bases[i] & 0xFC);
        if (bad)
            for (i = 0; i < total_given; ++i)
                if (bases[i] > 3)   // the reference raises ValueError("Symbol ... not defined") (hmm.pyx:72,79)
                    return fail(ADVNTR_ERR_SYMBOL, "batch: base code %d at offset %lld is not one of A,C,G,T", (int)bases[i],
                                (long long)i);
    }
    for (auto *H : B->models) B->m_max = std::max(B->m_max, H->m);

    B->device = current_device();
    B->stream = (flags & ADVNTR_FLAG_SECOND_QUEUE) ? g_cache.get_stream_second(B->device) : g_cache.get_stream(B->device);
    B->ev0 = g_cache.get_event(B->device);
    B->ev1 = g_cache.get_event(B->device);
    if (!B->stream || !B->ev0 || !B->ev1) return fail(ADVNTR_ERR_DEVICE, "stream/event creation failed");

    int rc;
    if ((rc = B->dmalloc(&B->d_bases, (size_t)total + 16))) return rc;
    if (total_given) HIP_TRY(hipMemcpy(B->d_bases, bases, (size_t)total_given, hipMemcpyHostToDevice));
    if ((rc = B->dmalloc(&B->d_read_off, (size_t)n_reads + 1))) return rc;
    HIP_TRY(hipMemcpy(B->d_read_off, read_off, ((size_t)n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (both && total_given) {
        const int grid = (int)std::min<int64_t>(((int64_t)n_reads_in + 3) / 4, 65536);
        hipLaunchKernelGGL(reverse_complement_kernel, dim3(grid), dim3(256), 0, B->stream, B->d_bases, B->d_read_off, n_reads_in);
        HIP_TRY(hipGetLastError());
    }
    if ((rc = B->dmalloc(&B->d_read_model, (size_t)n_reads))) return rc;
    if (n_reads) HIP_TRY(hipMemcpy(B->d_read_model, read_model, (size_t)n_reads * sizeof(int32_t), hipMemcpyHostToDevice));
    if ((rc = B->dmalloc(&B->d_logp, (size_t)n_reads))) return rc;
    if ((rc = B->dmalloc(&B->d_summary, (size_t)n_reads * ADVNTR_SUMMARY_INTS))) return rc;
    if ((rc = B->dmalloc(&B->d_counter, 12))) return rc;
    // split: reads the anti-diagonal kernel can take vs. the generic kernel
    std::vector<int32_t> col_reads, gen_reads;
    for (int r = 0; r < n_reads; ++r) {
        const int64_t n = read_off[r + 1] - read_off[r];
        const advntr_hmm *H = B->models[read_model[r]];
        if (!(flags & ADVNTR_FLAG_FORCE_GENERIC) && H->colprog.valid && n >= 1 && n <= COL_MAX_LONG_READ) col_reads.push_back(r);
        else {
            if (!H->generic_ok)
                return fail(ADVNTR_ERR_TOO_LARGE, "read %d (%lld bases) needs the generic kernel but its model has %d states "
                            "(> 10240, LDS rows)", r, (long long)n, H->m);
            gen_reads.push_back(r);
        }
    }
    // column reads: grouped by chunk count K=ceil(n/64), then model, then longest first
    // bucket 1..4 = chunk count of a single-tile read, 5 = row-tiled long read
    // buckets 0..2 = short reads: row-blocked kernels (viterbi_rows.h: rows_configs[bucket], 2 or 4 reads per wavefront);
    // 3 = longer reads: the tiled row-blocked kernel; 3 + K = the anti-diagonal kernel with K chunks (K = 5: row-tiled),
    // which runs when ADVNTR_FLAG_ANTIDIAGONAL asks for it or a model's tables do not fit 16-bit LDS addresses.  (The
    // row-blocked kernels are the faster ones at every batch size: 16 reads 0.41 against 0.50 ms, 4 000 reads 0.62
    // against 0.81 ms, scripts/small_batch_bench.py.)
    auto rows_len = [&](int64_t n) { return n >= 1 && n <= ROWS_MAX_READ; };
    const bool use_rows = !(flags & (ADVNTR_FLAG_STREAM | ADVNTR_FLAG_ANTIDIAGONAL));
    // bucket 3 = longer reads (156 bases and up): the row-tiled row-blocked kernel (one read per wavefront like the
    // anti-diagonal one, so batches of any size go there)
    const bool use_rows_long = use_rows;
    auto kof = [&](int r) {
        const int64_t n = read_off[r + 1] - read_off[r];
        // (the row-blocked sweep reaches class and emission records through 16-bit LDS addresses: a model whose two tables
        // pass 64 KiB keeps its reads on the anti-diagonal kernel, which has a range-checked sweep for that case)
        // (... and keeps the reversed path of a read as 16-bit state indices in LDS: path_summary.h)
        if (use_rows && rows_len(n) && B->models[read_model[r]]->col_lds_min + 64 <= 0x10000 && B->models[read_model[r]]->m <= 0xffff) {
            int cfg = 0;
            while (cfg + 1 < ROWS_CONFIGS && n <= rows_configs[cfg + 1].max_read) ++cfg;       // the tightest fit
            return cfg;
        }
        if (use_rows_long && n > ROWS_MAX_READ && B->models[read_model[r]]->col_lds_min + 64 <= 0x10000) return 3;
        return 3 + (int)std::min<int64_t>(5, (n + 63) / 64);
    };
    {   // order: bucket, then model, then longest first, then read index.  A counting sort over (bucket, model) -- both
        // small ranges -- followed by a stable sort by length inside the bins that are not already ordered (reads of one
        // locus mostly share a length): O(n) instead of the comparison sort that took 100 ms per 1.6 M reads
        const size_t n_col = col_reads.size();
        const size_t n_bins = (size_t)9 * (size_t)n_models;
        std::vector<uint32_t> bin_of(n_col);
        std::vector<int64_t> start(n_bins + 1, 0);
        for (size_t i = 0; i < n_col; ++i) {
            const int r = col_reads[i];
            bin_of[i] = (uint32_t)((size_t)kof(r) * (size_t)n_models + (size_t)read_model[r]);
            start[bin_of[i] + 1]++;
        }
        for (size_t b = 0; b < n_bins; ++b) start[b + 1] += start[b];
        std::vector<int32_t> sorted(n_col);
        {
            std::vector<int64_t> at(start.begin(), start.end() - 1);
            for (size_t i = 0; i < n_col; ++i) sorted[(size_t)at[bin_of[i]]++] = col_reads[i];      // stable: index order kept
        }
        auto len_of = [&](int r) { return read_off[r + 1] - read_off[r]; };
        for (size_t b = 0; b < n_bins; ++b) {
            const int64_t lo = start[b], hi = start[b + 1];
            bool ordered = true;
            for (int64_t i = lo + 1; i < hi && ordered; ++i) ordered = len_of(sorted[i - 1]) >= len_of(sorted[i]);
            if (!ordered)
                std::stable_sort(sorted.begin() + lo, sorted.begin() + hi, [&](int a, int c) { return len_of(a) > len_of(c); });
        }
        col_reads.swap(sorted);
    }
    // generic reads: heaviest (n+1)*E first (dynamic dequeue in-kernel)
    std::stable_sort(gen_reads.begin(), gen_reads.end(), [&](int a, int b) {
        const int64_t wa = (read_off[a + 1] - read_off[a] + 1) * (int64_t)B->models[read_model[a]]->n_edges;
        const int64_t wb = (read_off[b + 1] - read_off[b] + 1) * (int64_t)B->models[read_model[b]]->n_edges;
        if (wa != wb) return wa > wb;
        return read_model[a] < read_model[b];
    });
    // models of reads that go to the generic kernel need its tables on the device (uploaded on first use)
    for (int r : gen_reads)
        if ((rc = hmm_ensure_generic(B->models[read_model[r]]))) return rc;
    {
        std::vector<DevModel> dm;
        for (auto *H : B->models) {
            if (H->device != B->device)
                return fail(ADVNTR_ERR_DEVICE, "batch on device %d holds a model that lives on device %d", B->device, H->device);
            std::lock_guard<std::mutex> lk(H->gen_mu);          // another thread may be binding the generic tables
            dm.push_back(H->dev);
        }
        if ((rc = B->dmalloc(&B->d_models, dm.size()))) return rc;
        HIP_TRY(hipMemcpy(B->d_models, dm.data(), dm.size() * sizeof(DevModel), hipMemcpyHostToDevice));
    }
    B->n_col = (int)col_reads.size();
    B->n_gen = (int)gen_reads.size();
    std::vector<int32_t> order(col_reads);
    order.insert(order.end(), gen_reads.begin(), gen_reads.end());
    if ((rc = B->dmalloc(&B->d_order, (size_t)n_reads))) return rc;
    if (n_reads) HIP_TRY(hipMemcpy(B->d_order, order.data(), (size_t)n_reads * sizeof(int32_t), hipMemcpyHostToDevice));

    if (flags & ADVNTR_FLAG_PATH) {
        B->path_off.assign((size_t)n_reads + 1, 0);
        for (int r = 0; r < n_reads; ++r)
            B->path_off[r + 1] = B->path_off[r] + (read_off[r + 1] - read_off[r]) + B->models[read_model[r]]->m + 2;
        if ((rc = B->dmalloc(&B->d_path, (size_t)B->path_off[n_reads]))) return rc;
        if ((rc = B->dmalloc(&B->d_path_off, (size_t)n_reads + 1))) return rc;
        HIP_TRY(hipMemcpy(B->d_path_off, B->path_off.data(), ((size_t)n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        if ((rc = B->dmalloc(&B->d_path_len, (size_t)n_reads))) return rc;
    }
    const int cus = device_cus();
    // reversed-path scratch of a wave: the longest path any kernel accepts is the reference's own buffer, n + m entries
    // (hmm.pyx:1953; longer paths are refused per read, col_emit_outputs); the tracebacks write up to 64 states at a time and
    // stop 66 short of the end, hence the margin
    B->path_cap = B->n_max + B->m_max + 2 + 66;

    if (B->n_col) {
        ColumnLaunch &C = B->col;
        C.stream = (flags & ADVNTR_FLAG_STREAM) != 0;
        int n_max_col = 0, rows_groups = 1;
        size_t lds_core = 0, lds_min = 0;
        for (int r : col_reads) {
            const advntr_hmm *H = B->models[read_model[r]];
            C.nc_max = std::max(C.nc_max, H->colprog.n_cols);
            const size_t pad = (size_t)(H->colprog.n_cols + 128 * 4) * sizeof(ColInfo);
            C.lds_bytes = std::max(C.lds_bytes, (size_t)H->col_lds_bytes + pad);
            lds_core = std::max(lds_core, (size_t)H->col_lds_core + pad);
            lds_min = std::max(lds_min, (size_t)H->col_lds_min + pad);
            n_max_col = std::max<int>(n_max_col, (int)(read_off[r + 1] - read_off[r]));
        }
        // workgroups per CU are bounded by LDS (4 waves each; 1 workgroup per CU = 1 wave per SIMD): wide models (PacBio:
        // > 1000 columns) leave the traceback's state table, then the unpadded column-info table, in HBM/L2 when
        // that buys another resident workgroup
        // (the LDS behind the tables that only viterbi_rows_kernel uses -- read stash, reversed paths, tail values -- counts only when
        // some read goes there: a PacBio batch, long reads only, is not charged for it)
        bool any_short = false;
        for (int r : col_reads) if (kof(r) < 3) { any_short = true; break; }
        const size_t rows_extra = any_short ? (size_t)(ROWS_STASH_BYTES + ROWS_REV_BYTES + ROWS_TAIL_LDS_BYTES) : 0;
        auto wgs = [rows_extra](size_t lds) { return (int)std::max<size_t>(1, std::min<size_t>(4, (150 * 1024) / (lds + 16 + rows_extra + 1024))); };
        C.lds_core_bytes = lds_core; C.lds_min_bytes = lds_min;
        if (wgs(lds_core) > wgs(C.lds_bytes)) { C.lds_bytes = lds_core; C.lds_level = 1; }
        if (wgs(lds_min) > wgs(C.lds_bytes)) { C.lds_bytes = lds_min; C.lds_level = 0; }
        // (one workgroup per CU either way: still take the level that fits at all)
        if (C.lds_bytes + 16 + rows_extra > 160 * 1024 && C.lds_level > 1 && lds_core + 16 + rows_extra <= 160 * 1024) { C.lds_bytes = lds_core; C.lds_level = 1; }
        if (C.lds_bytes + 16 + rows_extra > 160 * 1024 && C.lds_level > 0) { C.lds_bytes = lds_min; C.lds_level = 0; }
        if (C.lds_bytes + 16 + rows_extra > 160 * 1024)
            return fail(ADVNTR_ERR_TOO_LARGE, "batch: the class / emission tables of a %d-column model take %zu B of LDS "
                        "(> 160 KiB per CU)", C.nc_max, C.lds_bytes + 16);
        int per_cu = wgs(C.lds_bytes);
        if (C.stream) {
            // stream kernel: reads of one model, any length, packed back to back by each wavefront
            std::stable_sort(col_reads.begin(), col_reads.end(), [&](int a, int b) { return read_model[a] < read_model[b]; });
            std::copy(col_reads.begin(), col_reads.end(), order.begin());
            HIP_TRY(hipMemcpy(B->d_order, order.data(), (size_t)n_reads * sizeof(int32_t), hipMemcpyHostToDevice));
            const int per_tile = COL_WAVES * COL_STREAM_READS;
            for (int i = 0; i < B->n_col;) {
                const int mod = read_model[col_reads[i]];
                int j = i;
                while (j < B->n_col && j - i < per_tile && read_model[col_reads[j]] == mod) ++j;
                C.tiles[0].push_back(ColTile{mod, i, j - i, 0});
                i = j;
            }
            const int TP = 64 * COL_STREAM_K;
            C.ring = (n_max_col + TP - 1) / TP + 1;
            C.bp_stride = (((int64_t)C.ring * (TP + C.nc_max) * TP) + 255) & ~int64_t(255);
            C.sink_stride = C.ring * TP + 2;
            C.rown_stride = 6 * (int64_t)C.nc_max + (int64_t)COL_STREAM_CAPS * (3 * (int64_t)C.nc_max + COL_MAX_TAIL);
        } else {
            // tiles of up to 16 reads of one model and one chunk count; the last ~15 % of the reads go out in
            // smaller tiles (8, then 4 = one read per wave) so the dynamic dequeue ends evenly across the CUs
            for (int i = 0; i < B->n_col;) {
                const int r0 = col_reads[i], bucket = kof(r0), mod = read_model[r0];
                const int left = B->n_col - i;
                // small batches (a locus-sized call): one read per wavefront so the reads spread over the CUs
                const int full = std::max<int>(COL_WAVES, std::min<int>(COL_TILE_READS, B->n_col / std::max(1, cus * per_cu)));
                int cap = left > B->n_col * 15 / 100 ? full : (left > B->n_col * 5 / 100 ? std::max<int>(COL_WAVES, full / 2) : COL_WAVES);
                // long reads: fewer reads per tile so that a modest batch still spreads over all CUs
                const int64_t nlen = read_off[r0 + 1] - read_off[r0];
                if (nlen > 192) cap = std::max<int>(COL_WAVES, std::min<int64_t>(cap, COL_TILE_READS * 192 / nlen));
                // (the tiled row-blocked kernel: one read per wavefront and tile -- a tile of ten 300-base reads is three reads in a
                // row for two of its wavefronts, and with the cheapest tiles going out last those were the 2 ms at the end of a launch)
                if (bucket == 3) cap = COL_WAVES;
                if (bucket < 3) {
                    // G reads per wavefront side by side and up to ROWS_DEPTH one behind the other (back-to-back sweeps,
                    // viterbi_rows.h): full-depth tiles while most of the batch is still ahead, then half depth, then
                    // single sweeps, so that the dynamic dequeue still ends evenly across the CUs
                    const int G = rows_configs[bucket].G, round = COL_WAVES * G;
                    const int nc = B->models[mod]->colprog.n_cols;
                    int depth = nc >= ROWS_STREAM_MIN_COLS ? ROWS_DEPTH : 1;
                    if (!(flags & ADVNTR_FLAG_DEEP_TILES)) {
                        depth = std::max(1, std::min(depth, B->n_col / std::max(1, cus * per_cu * round)));
                        if (left <= B->n_col * ROWS_TAIL_HALF_PCT / 100) depth = std::min(depth, std::max(1, ROWS_DEPTH / 2));
                        if (left <= B->n_col * ROWS_TAIL_SINGLE_PCT / 100) depth = 1;
                    }
                    cap = round * depth;
                    rows_groups = std::max(rows_groups, G);
                    C.rows_depth = std::max(C.rows_depth, depth);
                }
                int j = i;
                while (j < B->n_col && j - i < cap && kof(col_reads[j]) == bucket && read_model[col_reads[j]] == mod) ++j;
                C.tiles[bucket <= 3 ? 5 + bucket : bucket - 4].push_back(ColTile{mod, i, j - i, 0});
                i = j;
            }
            // long reads, one per wavefront: the tiles of a launch differ by an order of magnitude in work (columns x rows of
            // their longest read, which is the first one) -- heaviest first, so that what the dynamic dequeue hands out last is
            // small (in model order the launch ended on whatever came last: 4 ms of 200 on BASELINE config 4)
            // The order is by what a tile COSTS, not by its trellis cells: a read is swept in row tiles of 64 x 5 rows, each of
            // NC + 63 steps whatever it holds, the last one with as few rows per lane as cover it; a step costs a fixed part (about
            // 0.7 of a row's) plus its rows.  By cells a 321-row read ranks next to a 320-row one and takes a third longer.
            // (Round 5, per-workgroup clocks of a 22 400-call share: with tiles of up to ten short reads its workgroups ended over
            // 2 ms, mean idle 4.4 % of the launch; one read per wavefront and tile + this order: over 0.6 ms, 1.3 %.  The short
            // reads' sweep tiles stay in model order: a set's models are all within 450-550 columns, and the sorted order measured
            // 2.5 % slower -- neighbouring tiles no longer share a model in L2.)
            auto tile_cost = [&](const ColTile &t, const int slot) -> double {
                const int r = col_reads[(size_t)t.first];
                const int64_t n = read_off[r + 1] - read_off[r];
                const double nc = (double)B->models[t.model]->colprog.n_cols;
                if (slot != 8) return nc * (double)n;
                const int64_t RT = 64 * ROWS_LONG_R, full = (n - 1) / RT, last_rows = n - full * RT;
                const double rl_last = (double)((last_rows + 63) / 64);
                return (nc + 63.0) * ((double)full * (0.7 + ROWS_LONG_R) + 0.7 + rl_last) + 0.02 * (double)n * 64.0;      // (+ the traceback: ~ n rounds)
            };
            for (int k : {4, 8})
                std::stable_sort(C.tiles[k].begin(), C.tiles[k].end(),
                                 [&](const ColTile &x, const ColTile &y) { return tile_cost(x, k) > tile_cost(y, k); });
            const int kmax = std::min(4, (n_max_col + 63) / 64);
            const int TL = 64 * COL_LONG_K;
            // (the anti-diagonal kernel's row-tiled slabs only when some read really goes there: by default longer reads take
            // the tiled row-blocked kernel, whose slabs are sized below)
            const int64_t row_tiles = (n_max_col > COL_MAX_READ && !C.tiles[4].empty()) ? (n_max_col + TL - 1) / TL : 0;
            const int64_t short_bp = (int64_t)(64 * kmax + C.nc_max) * (64 * kmax), long_bp = row_tiles * (int64_t)(TL + C.nc_max) * TL;
            C.bp_stride = (std::max(short_bp, long_bp) + 255) & ~int64_t(255);
            C.rown_stride = 2 * (3 * (int64_t)C.nc_max + COL_MAX_TAIL);
            C.sink_stride = std::max(COL_MAX_READ, n_max_col) + 1;
            if (!C.tiles[8].empty()) {        // tiled row-blocked kernel: one slab of (NC + 64) x 64 dwords per 64 R rows
                const int RT = 64 * ROWS_LONG_R;
                const int64_t tiled_bp = (int64_t)((n_max_col + RT - 1) / RT) * (C.nc_max + 64) * 64 * 4;
                C.bp_stride = (std::max(C.bp_stride, tiled_bp) + 255) & ~int64_t(255);
                C.rown_stride = std::max<int64_t>(C.rown_stride, 2 * (3 * ((int64_t)C.nc_max + 128) + COL_MAX_TAIL));
            }
            if (rows_groups > 1) {            // row-blocked kernels: G reads per wave side by side, ROWS_DEPTH one behind the other; six 64-bit lane masks per cell of a step (256 B per step at R <= 5)
                const int64_t rows_bp = ((int64_t)C.rows_depth * C.nc_max + 33) * 64 * 4;
                C.bp_stride = (std::max(C.bp_stride, rows_bp) + 255) & ~int64_t(255);
                C.rown_stride = std::max<int64_t>(C.rown_stride, ROWS_MAX_GROUPS * 3 * ((int64_t)C.rows_depth * C.nc_max + 64) + COL_MAX_TAIL);
            }
        }
        // useful share of the row-blocked sweeps' lane-steps: a sweep of `depth` reads per lane group costs depth * NC + (rows of
        // the last read - 1) / R steps of 64 lanes x R cells (viterbi_rows_kernel), a row tile of a long read NC + (rows - 1) / R
        // (viterbi_rows_long_kernel; R = ceil(rows / 64) in a read's last tile); what the reads need is length x NC cells each
        if (!C.stream) {
            for (int k = 5; k <= 8; ++k)
                for (const ColTile &t : C.tiles[k]) {
                    const int nc = B->models[t.model]->colprog.n_cols;
                    auto len_at = [&](int idx) { const int r = col_reads[(size_t)t.first + idx]; return (int)(read_off[r + 1] - read_off[r]); };
                    for (int idx = 0; idx < t.count; ++idx) C.useful_cells[k] += (double)len_at(idx) * nc;
                    if (k == 8) {
                        const int R = ROWS_LONG_R, RT = 64 * R;
                        for (int idx = 0; idx < t.count; ++idx)
                            for (int n = len_at(idx), row0 = 0; row0 < n; row0 += RT) {
                                // (a read's last tile runs with as few rows per lane as cover it)
                                const int rows = std::min(RT, n - row0), rl = row0 + RT < n ? R : (rows + 63) / 64;
                                C.swept_cells[k] += (double)(nc + (rows - 1) / rl) * 64 * rl;
                            }
                        continue;
                    }
                    const int R = rows_configs[k - 5].R, G = rows_configs[k - 5].G, round = COL_WAVES * G;
                    const int dmax = nc >= ROWS_STREAM_MIN_COLS ? C.rows_depth : 1;
                    for (int j0 = 0; j0 < t.count; j0 += round * dmax)
                        for (int w = 0; w < COL_WAVES; ++w) {
                            const int jw = j0 + w * G;
                            if (jw >= t.count) break;
                            const int depth = std::min(dmax, (t.count - jw + round - 1) / round);
                            int nlast = 1;
                            for (int g = 0; g < G; ++g)
                                if (jw + (depth - 1) * round + g < t.count) nlast = std::max(nlast, len_at(jw + (depth - 1) * round + g));
                            C.swept_cells[k] += (double)(depth * nc + (nlast - 1) / R) * 64 * R;
                        }
                }
        }
        size_t n_tiles = 0;
        for (int k = 0; k < 9; ++k) n_tiles = std::max(n_tiles, C.tiles[k].size());
        C.grid = (int)std::max<size_t>(1, std::min<size_t>(n_tiles, (size_t)cus * per_cu));
        // back-pointer scratch of all resident waves: at most 60 % of the free HBM (288 GB per MI355X: 4 096 resident
        // waves x 25 MB for 15-kb reads on a 1 440-column model still fit); beyond that, fewer resident waves
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)48 << 30;
        const int64_t bp_budget = (int64_t)(free_b / 10 * 6) + (int64_t)g_cache.cached[B->device];
        while (C.grid > 1 && (int64_t)C.grid * COL_WAVES * C.bp_stride > bp_budget) C.grid = (C.grid + 1) / 2;
        if ((int64_t)COL_WAVES * C.bp_stride > bp_budget)
            return fail(ADVNTR_ERR_TOO_LARGE, "batch: the back-pointers of its longest read (%d bases on a %d-column model) take "
                        "%lld B per wavefront, %lld B for one workgroup; %lld B of device memory can be had", n_max_col, C.nc_max,
                        (long long)C.bp_stride, (long long)COL_WAVES * C.bp_stride, (long long)bp_budget);
        C.aux_stride = COL_MAX_TAIL + (int64_t)rows_groups * (rows_groups > 1 ? C.rows_depth + 1 : 1) * COL_MAX_SINKS * C.sink_stride;
        const size_t waves = (size_t)C.grid * COL_WAVES;
        if ((rc = B->dmalloc(&C.d_bp, waves * C.bp_stride))) return rc;
        if ((rc = B->dmalloc(&C.d_rown, waves * C.rown_stride))) return rc;
        if ((rc = B->dmalloc(&C.d_aux, waves * C.aux_stride))) return rc;
        if ((rc = B->dmalloc(&B->d_pathbuf_col, waves * B->path_cap))) return rc;
        for (int k = 0; k < 9; ++k) {
            if (C.tiles[k].empty()) continue;
            if ((rc = B->dmalloc(&C.d_tiles[k], C.tiles[k].size()))) return rc;
            HIP_TRY(hipMemcpy(C.d_tiles[k], C.tiles[k].data(), C.tiles[k].size() * sizeof(ColTile), hipMemcpyHostToDevice));
        }
        C.d_tile_counters = B->d_counter + 3;
    }
    if (B->n_gen) {
        int bpw_max = 1;
        for (int r : gen_reads) {
            const advntr_hmm *H = B->models[read_model[r]];
            B->m_max_gen = std::max(B->m_max_gen, H->m);
            bpw_max = std::max(bpw_max, H->bp_width);
        }
        int n_max_gen = 0;
        for (int r : gen_reads) n_max_gen = std::max<int>(n_max_gen, (int)(read_off[r + 1] - read_off[r]));
        B->lds_gen = (size_t)B->m_max_gen * 16;
        int per_cu = (int)std::min<size_t>(16, (160 * 1024) / std::max<size_t>(B->lds_gen, 1));
        per_cu = std::max(per_cu, 1);
        B->grid_gen = std::max(1, std::min(B->n_gen, cus * per_cu));
        B->bp_stride_gen = (((int64_t)(n_max_gen + 1) * B->m_max_gen * bpw_max) + 255) & ~int64_t(255);
        if ((rc = B->dmalloc(&B->d_bp_gen, (size_t)B->grid_gen * B->bp_stride_gen))) return rc;
        if ((rc = B->dmalloc(&B->d_pathbuf_gen, (size_t)B->grid_gen * B->path_cap))) return rc;
        if (B->lds_gen > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void *)viterbi_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)B->lds_gen));
            HIP_TRY(hipFuncSetAttribute((const void *)forward_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)B->lds_gen));
        }
    }
    return ADVNTR_OK;
}

extern "C" advntr_batch *advntr_batch_create(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                                             const int64_t *read_off, const int32_t *read_model,
                                             int32_t n_reads, uint32_t flags)
{
    advntr_batch *B = new advntr_batch();
    if (batch_build(B, models, n_models, bases, read_off, read_model, n_reads, flags) != ADVNTR_OK) {
        std::string keep = g_err;
        advntr_batch_destroy(B);
        g_err = keep;
        return nullptr;
    }
    return B;
}

extern "C" void advntr_batch_destroy(advntr_batch *B)
{
    if (!B) return;
    if (B->stream) (void)hipStreamSynchronize(B->stream);        // nothing of this batch may still be running
    for (auto &a : B->allocs) g_cache.put(B->device, a.first, a.second);
    if (B->ev0) g_cache.put_event(B->device, B->ev0);
    if (B->ev1) g_cache.put_event(B->device, B->ev1);
    if (B->stream) {
        if (B->flags & ADVNTR_FLAG_SECOND_QUEUE) g_cache.put_stream_second(B->device, B->stream);
        else g_cache.put_stream(B->device, B->stream);
    }
    std::vector<advntr_hmm *> last;                               // models destroyed by their owner while this batch lived
    {
        std::lock_guard<std::mutex> lock(g_model_life);
        for (advntr_hmm *H : B->models)
            if (--H->live_batches == 0 && H->doomed) last.push_back(H);
    }
    for (advntr_hmm *H : last) hmm_release(H);
    delete B;
}

extern "C" int advntr_batch_result_ptrs(advntr_batch *B, void **d_logp, void **d_summary)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_result_ptrs: null batch");
    if (d_logp) *d_logp = B->d_logp;
    if (d_summary) *d_summary = B->d_summary;
    return ADVNTR_OK;
}

extern "C" int64_t advntr_batch_device_bytes(const advntr_batch *B) { return B ? B->device_bytes : 0; }

// Which kernels advntr_batch_run launches for this batch: one line "kernel reads tiles" per non-empty launch, in launch
// order (the routing is decided in batch_build from read lengths, model tables and flags only).
#define ADV_STR_(x) #x
#define ADV_STR(x) ADV_STR_(x)
extern "C" int advntr_batch_info(const advntr_batch *B, char *buf, int32_t capacity)
{
    if (!B || !buf || capacity <= 0) return fail(ADVNTR_ERR_ARG, "advntr_batch_info: bad argument");
    static const char *const names[9] = {
        "viterbi_columns_kernel<1, false>", "viterbi_columns_kernel<2, false>", "viterbi_columns_kernel<3, false>",
        "viterbi_columns_kernel<4, false>", "viterbi_columns_kernel<3, true>", "viterbi_rows_kernel<5, 2>",
        "viterbi_rows_kernel<4, 2>", "viterbi_rows_kernel<4, 4>", "viterbi_rows_long_kernel<" ADV_STR(ROWS_LONG_R) ">"};
    static_assert(COL_LONG_K == 3, "kernel names above");
    std::string out;
    char line[160];
    if (B->n_col) {
        static const int launch_order[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
        for (int k : launch_order) {
            const auto &tiles = B->col.tiles[k];
            if (tiles.empty()) continue;
            int64_t reads = 0;
            for (const ColTile &t : tiles) reads += t.count;
            // (fourth field: useful share of the sweeps' lane-steps in per mille, -1 where it is not accounted)
            const int useful = B->col.swept_cells[k] > 0 ? (int)(1000.0 * B->col.useful_cells[k] / B->col.swept_cells[k] + 0.5) : -1;
            snprintf(line, sizeof line, "%s %lld %zu %d\n", B->col.stream ? "viterbi_columns_stream_kernel<3>" : names[k],
                     (long long)reads, tiles.size(), useful);
            out += line;
        }
    }
    if (B->n_gen) {
        snprintf(line, sizeof line, "viterbi_generic_kernel %d %d -1\n", B->n_gen, B->grid_gen);
        out += line;
    }
    if ((int64_t)out.size() + 1 > capacity) return fail(ADVNTR_ERR_TOO_LARGE, "advntr_batch_info: need %zu bytes", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return ADVNTR_OK;
}

static BatchArgs make_args(advntr_batch *B)
{
    BatchArgs a{};
    a.models = B->d_models; a.bases = B->d_bases; a.read_off = B->d_read_off; a.read_model = B->d_read_model;
    a.out_logp = B->d_logp;
    a.out_summary = (B->flags & ADVNTR_FLAG_NO_SUMMARY) ? nullptr : B->d_summary;
    a.out_path = B->d_path; a.out_path_off = B->d_path_off; a.out_path_len = B->d_path_len;
    a.path_cap = B->path_cap;
    return a;
}

static BatchArgs generic_args(advntr_batch *B)
{
    BatchArgs a = make_args(B);
    a.n_reads = B->n_gen;
    a.order = B->d_order + B->n_col;
    a.counter = B->d_counter;
    a.bp_scratch = B->d_bp_gen; a.bp_stride = B->bp_stride_gen;
    a.path_scratch = B->d_pathbuf_gen;
    a.m_max = B->m_max_gen;
    return a;
}

extern "C" int advntr_batch_run(advntr_batch *B)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_run: null batch");
    if (B->n_reads == 0) return ADVNTR_OK;
    HIP_TRY(hipMemsetAsync(B->d_counter, 0, 12 * sizeof(int32_t), B->stream));
    if (B->n_col) {
        BatchArgs a = make_args(B);
        a.n_reads = B->n_col;
        a.order = B->d_order;
        a.path_scratch = B->d_pathbuf_col;
        if (B->col.stream) column_launch_stream<COL_STREAM_K>(B->col, a, B->flags, B->stream);
        else
        column_launch_k<1, false>(B->col, a, B->flags, B->stream);
        if (!B->col.stream) {
            column_launch_k<2, false>(B->col, a, B->flags, B->stream);
            column_launch_k<3, false>(B->col, a, B->flags, B->stream);
            column_launch_k<4, false>(B->col, a, B->flags, B->stream);
            column_launch_k<COL_LONG_K, true>(B->col, a, B->flags, B->stream);
            column_launch_rows<5, 2>(B->col, a, B->flags, B->stream, 0);
            column_launch_rows<4, 2>(B->col, a, B->flags, B->stream, 1);
            column_launch_rows<4, 4>(B->col, a, B->flags, B->stream, 2);
            column_launch_rows_long(B->col, a, B->flags, B->stream);
        }
    }
    if (B->n_gen) {
        BatchArgs a = generic_args(B);
        hipLaunchKernelGGL(viterbi_generic_kernel, dim3(B->grid_gen), dim3(ADV_WAVE), B->lds_gen, B->stream, a, B->flags);
    }
    B->col.reserve_workgroups = 0;              // a gather's request covers one pass: the one just queued
    HIP_TRY(hipGetLastError());
    return ADVNTR_OK;
}

extern "C" int advntr_batch_reserve_next(advntr_batch *B, int32_t n_workgroups)
{
    if (!B || n_workgroups < 0) return fail(ADVNTR_ERR_ARG, "advntr_batch_reserve_next: bad argument");
    B->col.reserve_workgroups = n_workgroups;
    return ADVNTR_OK;
}

#ifdef ADVNTR_WG_CLOCKS
// measurement build only (viterbi_rows.h, WG_CLOCKS_*): the four clocks of every workgroup of the last row-blocked launch
extern "C" int advntr_debug_wg_clocks(advntr_batch *B, unsigned long long *out, int cap)
{
    HIP_TRY(hipStreamSynchronize(B->stream));
    const int n = std::min(cap, B->col.grid);
    for (int w = 0; w < n; ++w)
        HIP_TRY(hipMemcpy(out + 4 * w, B->col.d_rown + (int64_t)w * COL_WAVES * B->col.rown_stride, 32, hipMemcpyDeviceToHost));
    return n;
}
#endif

extern "C" int advntr_batch_sync(advntr_batch *B)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_sync: null batch");
    HIP_TRY(hipStreamSynchronize(B->stream));
    return ADVNTR_OK;
}

extern "C" int advntr_batch_run_timed(advntr_batch *B, int32_t iters, float *ms_per_run)
{
    if (!B || iters <= 0 || !ms_per_run) return fail(ADVNTR_ERR_ARG, "advntr_batch_run_timed: bad argument");
    HIP_TRY(hipEventRecord(B->ev0, B->stream));
    for (int i = 0; i < iters; ++i) {
        int rc = advntr_batch_run(B);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(B->ev1, B->stream));
    HIP_TRY(hipEventSynchronize(B->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, B->ev0, B->ev1));
    *ms_per_run = ms / (float)iters;
    return ADVNTR_OK;
}

extern "C" int advntr_batch_fetch(advntr_batch *B, double *out_logp, int32_t *out_summary)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_fetch: null batch");
    HIP_TRY(hipStreamSynchronize(B->stream));
    if (out_logp && B->n_reads)
        HIP_TRY(hipMemcpy(out_logp, B->d_logp, (size_t)B->n_reads * sizeof(double), hipMemcpyDeviceToHost));
    if (out_summary && B->n_reads && !(B->flags & ADVNTR_FLAG_NO_SUMMARY))
        HIP_TRY(hipMemcpy(out_summary, B->d_summary, (size_t)B->n_reads * ADVNTR_SUMMARY_INTS * sizeof(int32_t),
                          hipMemcpyDeviceToHost));
    return ADVNTR_OK;
}

extern "C" int advntr_batch_fetch_paths(advntr_batch *B, int32_t *out_path, const int64_t *out_path_off,
                                        int32_t *out_path_len)
{
    if (!B || !out_path || !out_path_off || !out_path_len)
        return fail(ADVNTR_ERR_ARG, "advntr_batch_fetch_paths: bad argument");
    if (!(B->flags & ADVNTR_FLAG_PATH)) return fail(ADVNTR_ERR_ARG, "batch was created without ADVNTR_FLAG_PATH");
    HIP_TRY(hipStreamSynchronize(B->stream));
    const int n = B->n_reads;
    if (!n) return ADVNTR_OK;
    std::vector<int32_t> lens(n), all((size_t)B->path_off[n]);
    HIP_TRY(hipMemcpy(lens.data(), B->d_path_len, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(all.data(), B->d_path, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int r = 0; r < n; ++r) {
        const int64_t cap = out_path_off[r + 1] - out_path_off[r];
        int len = lens[r];
        if (len > 0 && len > cap) len = -2;
        if (len > 0) memcpy(out_path + out_path_off[r], all.data() + B->path_off[r], (size_t)len * sizeof(int32_t));
        out_path_len[r] = len;
    }
    return ADVNTR_OK;
}

// One-shot call on a large batch: the host side of a batch (validation, routing sort, tile lists, uploads: 1.7 ms per
// 100 000 reads) would sit in front of its 10 ms of kernels.  The reads are cut into up to four contiguous chunks, each a
// batch with its own stream: chunk i+1 is prepared and uploaded while chunk i computes, and the results of a chunk are
// downloaded while the next one still runs.  Reads are independent, so the records are the ones a single batch gives.
static const int32_t kOneShotChunk = 16384;
static int viterbi_batch_chunked(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases, const int64_t *read_off,
                                 const int32_t *read_model, int32_t n_reads, double *out_logp, int32_t *out_summary,
                                 uint32_t flags)
{
    if (read_off[0] != 0) return fail(ADVNTR_ERR_ARG, "batch: read_off[0] must be 0");
    const int n_chunks = (int)std::min<int64_t>(4, n_reads / kOneShotChunk);
    const bool both = (flags & ADVNTR_FLAG_BOTH_STRANDS) != 0;
    std::vector<advntr_batch *> batches(n_chunks, nullptr);
    std::vector<int32_t> first(n_chunks + 1);
    for (int c = 0; c <= n_chunks; ++c) first[c] = (int32_t)((int64_t)n_reads * c / n_chunks);
    int rc = ADVNTR_OK;
    std::vector<int64_t> off;
    for (int c = 0; c < n_chunks && rc == ADVNTR_OK; ++c) {
        const int32_t r0 = first[c], m = first[c + 1] - r0;
        off.resize((size_t)m + 1);
        for (int32_t i = 0; i <= m; ++i) off[i] = read_off[r0 + i] - read_off[r0];
        if (off[m] < 0) { rc = fail(ADVNTR_ERR_ARG, "batch: read_off not monotone"); break; }
        batches[c] = new advntr_batch();
        // (neighbouring chunks on streams of different classes: the next chunk's kernels start while the last workgroups of
        // this one drain -- two streams of one class may share a hardware queue and then never overlap)
        rc = batch_build(batches[c], models, n_models, bases ? bases + read_off[r0] : nullptr, off.data(), read_model + r0, m,
                         flags | ((c & 1) ? ADVNTR_FLAG_SECOND_QUEUE : 0u));
        if (rc == ADVNTR_OK) rc = advntr_batch_run(batches[c]);
    }
    std::vector<double> lp;
    std::vector<int32_t> sm;
    for (int c = 0; c < n_chunks && rc == ADVNTR_OK; ++c) {
        const int32_t r0 = first[c], m = first[c + 1] - r0;
        if (!both) {
            rc = advntr_batch_fetch(batches[c], out_logp + r0, out_summary ? out_summary + (size_t)r0 * ADVNTR_SUMMARY_INTS : nullptr);
        } else {            // the chunk's batch holds its forward calls, then their reverse complements
            lp.resize((size_t)2 * m);
            if (out_summary) sm.resize((size_t)2 * m * ADVNTR_SUMMARY_INTS);
            rc = advntr_batch_fetch(batches[c], lp.data(), out_summary ? sm.data() : nullptr);
            if (rc != ADVNTR_OK) break;
            memcpy(out_logp + r0, lp.data(), (size_t)m * sizeof(double));
            memcpy(out_logp + n_reads + r0, lp.data() + m, (size_t)m * sizeof(double));
            if (out_summary) {
                const size_t row = ADVNTR_SUMMARY_INTS * sizeof(int32_t);
                memcpy(out_summary + (size_t)r0 * ADVNTR_SUMMARY_INTS, sm.data(), (size_t)m * row);
                memcpy(out_summary + ((size_t)n_reads + r0) * ADVNTR_SUMMARY_INTS, sm.data() + (size_t)m * ADVNTR_SUMMARY_INTS, (size_t)m * row);
            }
        }
    }
    std::string keep = g_err;
    for (advntr_batch *B : batches)
        if (B) advntr_batch_destroy(B);
    g_err = keep;
    return rc;
}

extern "C" int advntr_viterbi_batch(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                                    const int64_t *read_off, const int32_t *read_model, int32_t n_reads,
                                    double *out_logp, int32_t *out_summary, int32_t *out_path,
                                    const int64_t *out_path_off, int32_t *out_path_len, uint32_t flags)
{
    if (!out_logp && n_reads) return fail(ADVNTR_ERR_ARG, "advntr_viterbi_batch: out_logp is null");
    if ((flags & ADVNTR_FLAG_PATH) && (!out_path || !out_path_off || !out_path_len))
        return fail(ADVNTR_ERR_ARG, "advntr_viterbi_batch: ADVNTR_FLAG_PATH needs out_path/out_path_off/out_path_len");
    if (!out_summary) flags |= ADVNTR_FLAG_NO_SUMMARY;
    if (n_reads >= 2 * kOneShotChunk && !(flags & ADVNTR_FLAG_PATH) && read_off && read_model)
        return viterbi_batch_chunked(models, n_models, bases, read_off, read_model, n_reads, out_logp, out_summary, flags);
    advntr_batch *B = new advntr_batch();
    int rc = batch_build(B, models, n_models, bases, read_off, read_model, n_reads, flags);
    if (rc == ADVNTR_OK) rc = advntr_batch_run(B);
    if (rc == ADVNTR_OK) rc = advntr_batch_fetch(B, out_logp, out_summary);
    if (rc == ADVNTR_OK && (flags & ADVNTR_FLAG_PATH)) rc = advntr_batch_fetch_paths(B, out_path, out_path_off, out_path_len);
    std::string keep = g_err;
    advntr_batch_destroy(B);
    g_err = keep;
    return rc;
}

// The sum-product launches of a batch (the lists batch_build made serve both recurrences): results go to d_logp.
static int forward_launch(advntr_batch *B)
{
    HIP_TRY(hipMemsetAsync(B->d_counter, 0, 12 * sizeof(int32_t), B->stream));
    if (B->n_col) {               // reads of models with a column program: sum-product on the anti-diagonal sweep
        BatchArgs a = make_args(B);
        a.n_reads = B->n_col;
        a.order = B->d_order;
        // batch_build sized the LDS staging level for the Viterbi kernels; the sum-product kernels add a linear
        // row-0 table (16 B per column): step the level down until the launch fits the 160 KiB of a CU
        ColumnLaunch C = B->col;                  // (a copy: the Viterbi launches of the same batch keep their level)
        const size_t kLds = 160 * 1024;
        if (forward_lds_bytes(C.lds_bytes, C.nc_max) > kLds && C.lds_level > 1) { C.lds_bytes = C.lds_core_bytes; C.lds_level = 1; }
        if (forward_lds_bytes(C.lds_bytes, C.nc_max) > kLds && C.lds_level > 0) { C.lds_bytes = C.lds_min_bytes; C.lds_level = 0; }
        if (forward_lds_bytes(C.lds_bytes, C.nc_max) > kLds)
            return fail(ADVNTR_ERR_TOO_LARGE, "log_probability: a model of %d columns needs %zu B of LDS for the "
                        "sum-product sweep (> 160 KiB)", C.nc_max, forward_lds_bytes(C.lds_bytes, C.nc_max));
        if (C.stream) return fail(ADVNTR_ERR_UNSUPPORTED, "log_probability: the batch was created with ADVNTR_FLAG_STREAM");
        HIP_TRY((column_launch_fwd<1, false>(C, a, B->stream, 0)));
        HIP_TRY((column_launch_fwd<2, false>(C, a, B->stream, 1)));
        HIP_TRY((column_launch_fwd<3, false>(C, a, B->stream, 2)));
        HIP_TRY((column_launch_fwd<COL_LONG_K, true>(C, a, B->stream, 3)));      // 193-256 rows: two row tiles
        HIP_TRY((column_launch_fwd<COL_LONG_K, true>(C, a, B->stream, 4)));
        HIP_TRY((column_launch_fwd_rows<5, 2>(C, a, B->stream, 0)));              // the row-blocked kernels' lists
        HIP_TRY((column_launch_fwd_rows<4, 2>(C, a, B->stream, 1)));
        HIP_TRY((column_launch_fwd_rows<4, 4>(C, a, B->stream, 2)));
        HIP_TRY((column_launch_fwd<COL_LONG_K, true>(C, a, B->stream, 8)));
    }
    if (B->n_gen) {
        BatchArgs a = generic_args(B);
        hipLaunchKernelGGL(forward_generic_kernel, dim3(B->grid_gen), dim3(ADV_WAVE), B->lds_gen, B->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return ADVNTR_OK;
}

extern "C" int advntr_forward_batch(advntr_hmm *const *models, int32_t n_models, const uint8_t *bases,
                                    const int64_t *read_off, const int32_t *read_model, int32_t n_reads,
                                    double *out_logp, uint32_t flags)
{
    if (!out_logp && n_reads) return fail(ADVNTR_ERR_ARG, "advntr_forward_batch: out_logp is null");
    advntr_batch *B = new advntr_batch();
    int rc = batch_build(B, models, n_models, bases, read_off, read_model, n_reads,
                         (flags | ADVNTR_FLAG_NO_SUMMARY) & ~(ADVNTR_FLAG_PATH | ADVNTR_FLAG_STREAM));
    if (rc == ADVNTR_OK && n_reads) rc = forward_launch(B);
    if (rc == ADVNTR_OK) rc = advntr_batch_fetch(B, out_logp, nullptr);
    std::string keep = g_err;
    advntr_batch_destroy(B);
    g_err = keep;
    return rc;
}

// log_probability on a device-resident batch: the reads stay where advntr_batch_create put them, the sum-product kernels
// write d_logp (the summaries of an earlier Viterbi run are left alone)
extern "C" int advntr_batch_forward(advntr_batch *B)
{
    if (!B) return fail(ADVNTR_ERR_ARG, "advntr_batch_forward: null batch");
    if (B->n_reads == 0) return ADVNTR_OK;
    return forward_launch(B);
}

extern "C" int advntr_batch_forward_timed(advntr_batch *B, int32_t iters, float *ms_per_run)
{
    if (!B || iters <= 0 || !ms_per_run) return fail(ADVNTR_ERR_ARG, "advntr_batch_forward_timed: bad argument");
    HIP_TRY(hipEventRecord(B->ev0, B->stream));
    for (int i = 0; i < iters; ++i) {
        int rc = advntr_batch_forward(B);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(B->ev1, B->stream));
    HIP_TRY(hipEventSynchronize(B->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, B->ev0, B->ev1));
    *ms_per_run = ms / (float)iters;
    return ADVNTR_OK;
}

// CPUs this process may really use: the hardware threads, cut down to the CPU-time quota of its control group (a container on
// a 256-thread host is typically given a few cores: 16 on the GPU boxes of this pool).  A bulk call that starts more worker
// threads than that has its whole group stopped by the scheduler for the rest of every accounting period -- seen as pauses of
// tens of milliseconds in ALL threads of a pipelined run (scripts/e2e_timeline.py).  ADVNTR_HOST_THREADS overrides.
static int host_cpu_limit()
{
    // (ADVNTR_HOST_THREADS is read at every call: a caller may change it between bulk calls; the probe of the machine is made once)
    if (const char *e = getenv("ADVNTR_HOST_THREADS")) {
        const int v = atoi(e);
        if (v > 0) return std::min(v, 1024);
    }
    static const int limit = [] {
        int n = (int)std::max(1u, std::thread::hardware_concurrency());
        // the CPUs this process may be scheduled on (taskset, numactl, a cpuset-limited container): hardware_concurrency counts
        // the online ones
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n, (int)CPU_COUNT(&set));
        // CPU-time quota: the tightest cpu.max on the way from the process's own control group (/proc/self/cgroup, "0::<path>";
        // "/" inside a cgroup namespace) up to the root of the mounted hierarchy
        auto quota_cores = [](const std::string &file) -> long long {
            long long quota = -1, period = 100000;
            if (FILE *f = fopen(file.c_str(), "r")) {                                // cgroup v2: "<quota|max> <period>"
                char q[32] = {0};
                if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
                fclose(f);
            }
            return (quota > 0 && period > 0) ? std::max<long long>(1, (quota + period - 1) / period) : -1;
        };
        std::string rel = "/";
        if (FILE *f = fopen("/proc/self/cgroup", "r")) {
            char line[4096];
            while (fgets(line, sizeof line, f))
                if (strncmp(line, "0::", 3) == 0) {
                    rel = line + 3;
                    while (!rel.empty() && (rel.back() == '\n' || rel.back() == '\r')) rel.pop_back();
                }
            fclose(f);
        }
        if (rel.empty() || rel[0] != '/') rel = "/";
        for (std::string path = rel;;) {
            const long long c = quota_cores("/sys/fs/cgroup" + (path == "/" ? std::string() : path) + "/cpu.max");
            if (c > 0) n = (int)std::min<long long>(n, c);
            if (path == "/") break;
            const size_t cut = path.find_last_of('/');
            path = cut == 0 ? "/" : path.substr(0, cut);
        }
        {   // cgroup v1
            long long quota = -1, period = 100000;
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
                if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                    if (fscanf(h, "%lld", &period) != 1) period = 100000;
                    fclose(h);
                }
            }
            if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        }
        return std::max(1, n);
    }();
    return limit;
}
extern "C" int advntr_host_threads(void) { return host_cpu_limit(); }

#include "abi_keyword_filter.h"
#include "abi_model_builder.h"
#include "abi_flank_align.h"
#include "abi_recruit.h"
#include "abi_comm.h"
#include "abi_genotype.h"
