// abi_comm.h -- RCCL behind the C ABI (included by engine.hip): the ONE exchange step of the scoring path, the gather of
// the per-read result records (fp64 log-probability + 8 x int32 summary) of every rank's batch to a root rank over
// xGMI, plus the small collectives a multi-GPU driver needs around it (counts, barrier, max of a timing).
//
// The reference has no counterpart: it scores loci serially in one process
// (/root/reference/advntr/genome_analyzer.py:280-297) and collects results of its optional worker processes through a
// multiprocessing.Manager().list() (/root/reference/advntr/vntr_finder.py:425-427).  Here every rank owns whole loci
// (advntr_amd/sharding.py), nothing is exchanged during scoring, and the records travel device to device:
//   counts differ per rank (ragged)  ->  grouped ncclSend / ncclRecv (gather-v); the root's own share is a device copy.
// librccl.so is 0.5 GB and only multi-GPU runs need it, so it is dlopen'ed on first use instead of linked.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <type_traits>

namespace {

struct RcclApi {
    void *handle = nullptr;
    // every pointer has the type of the prototype in <rccl/rccl.h> itself (no hand-written copy that could drift from the
    // header the library was built with: a changed argument list shows up as a compile error at the call sites below)
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    std::string error;
};

// what the calls below assume of those prototypes, checked against the header at compile time
static_assert(std::is_same_v<decltype(&::ncclSend), ncclResult_t (*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>,
              "ncclSend(sendbuff, count, datatype, peer, comm, stream)");
static_assert(std::is_same_v<decltype(&::ncclRecv), ncclResult_t (*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>,
              "ncclRecv(recvbuff, count, datatype, peer, comm, stream)");
static_assert(std::is_same_v<decltype(&::ncclAllGather), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>,
              "ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream)");
static_assert(std::is_same_v<decltype(&::ncclAllReduce),
                             ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>,
              "ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream)");
static_assert(std::is_same_v<decltype(&::ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>,
              "ncclCommInitRank(comm, nranks, commId BY VALUE, rank)");
static_assert(std::is_same_v<decltype(&::ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)> &&
              std::is_same_v<decltype(&::ncclCommDestroy), ncclResult_t (*)(ncclComm_t)> &&
              std::is_same_v<decltype(&::ncclGetErrorString), const char *(*)(ncclResult_t)> &&
              std::is_same_v<decltype(&::ncclGroupStart), ncclResult_t (*)()> && std::is_same_v<decltype(&::ncclGroupEnd), ncclResult_t (*)()>,
              "ncclGetUniqueId / ncclCommDestroy / ncclGetErrorString / ncclGroupStart / ncclGroupEnd");

RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // ADVNTR_RCCL_LIB names the library to load (an RCCL build outside the loader's path); by default the usual names
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char *given = getenv("ADVNTR_RCCL_LIB");
        if (given && *given) api.handle = dlopen(given, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char *n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!api.handle) {
            const char *why = dlerror();
            api.error = std::string("dlopen(") + (given && *given ? given : "librccl.so") + ") failed: " + (why ? why : "?");
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    });
    return &api;
}

}  // namespace

#define RCCL_TRY(expr)                                                                                                \
    do {                                                                                                              \
        ncclResult_t r_ = (expr);                                                                                     \
        if (r_ != ncclSuccess)                                                                                        \
            return fail(ADVNTR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct advntr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t staged = nullptr;            // the records of the pass to gather sit in the staging buffers
    hipEvent_t g0 = nullptr, g1 = nullptr;  // around the gather on `stream` (timing: advntr_comm_last_gather_ms)
    float last_gather_ms = -1.f;
    double *d_stage_logp = nullptr;         // this rank's records, copied out of the batch so that the next pass may
    int32_t *d_stage_sum = nullptr;         // overwrite the batch's arrays while the gather is still in flight
    double *d_all_logp = nullptr;           // root: every rank's records, rank after rank
    int32_t *d_all_sum = nullptr;
    size_t cap_stage_logp = 0, cap_stage_sum = 0, cap_all_logp = 0, cap_all_sum = 0;      // bytes
    uint8_t *d_bytes = nullptr;             // gather_bytes: send buffer, and on the root the receive buffer behind it
    size_t bytes_cap = 0;
    int64_t *d_small = nullptr;             // world + 2 words for the small collectives
    std::vector<int64_t> counts;            // of the gather in flight
    int root = 0;
    bool in_flight = false;

    // grow a device buffer (never while a collective that uses it may be running: the stream is drained first)
    int reserve(void **p, size_t *cap, size_t bytes)
    {
        if (*cap >= bytes && *p) return ADVNTR_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        if (*p) HIP_TRY(hipFree(*p));
        *p = nullptr; *cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
        HIP_TRY(hipMalloc(p, want));
        *cap = want;
        return ADVNTR_OK;
    }
};

// Can this process use RCCL at all (library found, every entry point resolved)?  No GPU call, no collective: what the ranks
// tell each other BEFORE any of them enters ncclCommInitRank.
extern "C" int advntr_comm_available(void)
{
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(ADVNTR_ERR_DEVICE, "%s", api->error.c_str());
    return ADVNTR_OK;
}

extern "C" int advntr_comm_unique_id(uint8_t *id128)
{
    if (!id128) return fail(ADVNTR_ERR_ARG, "advntr_comm_unique_id: null buffer");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(ADVNTR_ERR_DEVICE, "%s", api->error.c_str());
    ncclUniqueId id;
    RCCL_TRY(api->GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes (NCCL_UNIQUE_ID_BYTES)");
    memcpy(id128, &id, sizeof id);
    return ADVNTR_OK;
}

extern "C" void advntr_comm_destroy(advntr_comm *C)
{
    if (!C) return;
    if (C->stream) (void)hipStreamSynchronize(C->stream);
    if (C->comm) (void)rccl_api()->CommDestroy(C->comm);
    for (void *p : {(void *)C->d_stage_logp, (void *)C->d_stage_sum, (void *)C->d_all_logp, (void *)C->d_all_sum,
                    (void *)C->d_bytes, (void *)C->d_small})
        if (p) (void)hipFree(p);
    if (C->staged) (void)hipEventDestroy(C->staged);
    if (C->g0) (void)hipEventDestroy(C->g0);
    if (C->g1) (void)hipEventDestroy(C->g1);
    if (C->stream) (void)hipStreamDestroy(C->stream);
    delete C;
}

extern "C" advntr_comm *advntr_comm_create(int32_t rank, int32_t world, const uint8_t *id128)
{
    if (world < 1 || rank < 0 || rank >= world || !id128) {
        fail(ADVNTR_ERR_ARG, "advntr_comm_create: bad argument (rank %d of %d)", rank, world);
        return nullptr;
    }
    RcclApi *api = rccl_api();
    if (!api->error.empty()) { fail(ADVNTR_ERR_DEVICE, "%s", api->error.c_str()); return nullptr; }
    advntr_comm *C = new advntr_comm();
    C->rank = rank; C->world = world; C->device = current_device();
    const int rc = [&]() -> int {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        RCCL_TRY(api->CommInitRank(&C->comm, world, id, rank));
        HIP_TRY(hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&C->staged, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&C->g0));
        HIP_TRY(hipEventCreate(&C->g1));
        HIP_TRY(hipMalloc((void **)&C->d_small, ((size_t)world + 2) * sizeof(int64_t)));
        return ADVNTR_OK;
    }();
    if (rc != ADVNTR_OK) {
        std::string keep = g_err;
        advntr_comm_destroy(C);
        g_err = keep;
        return nullptr;
    }
    return C;
}

extern "C" int advntr_comm_info(const advntr_comm *C, int32_t *rank, int32_t *world)
{
    if (!C) return fail(ADVNTR_ERR_ARG, "advntr_comm_info: null communicator");
    if (rank) *rank = C->rank;
    if (world) *world = C->world;
    return ADVNTR_OK;
}

extern "C" int advntr_comm_allgather_i64(advntr_comm *C, int64_t mine, int64_t *out_world)
{
    if (!C || !out_world) return fail(ADVNTR_ERR_ARG, "advntr_comm_allgather_i64: bad argument");
    RcclApi *api = rccl_api();
    HIP_TRY(hipMemcpyAsync(C->d_small + C->world, &mine, sizeof mine, hipMemcpyHostToDevice, C->stream));
    RCCL_TRY(api->AllGather(C->d_small + C->world, C->d_small, 1, ncclInt64, C->comm, C->stream));
    HIP_TRY(hipMemcpyAsync(out_world, C->d_small, (size_t)C->world * sizeof(int64_t), hipMemcpyDeviceToHost, C->stream));
    HIP_TRY(hipStreamSynchronize(C->stream));
    return ADVNTR_OK;
}

extern "C" int advntr_comm_allreduce_max_f64(advntr_comm *C, double *inout)
{
    if (!C || !inout) return fail(ADVNTR_ERR_ARG, "advntr_comm_allreduce_max_f64: bad argument");
    RcclApi *api = rccl_api();
    double *d = (double *)C->d_small;
    HIP_TRY(hipMemcpyAsync(d, inout, sizeof(double), hipMemcpyHostToDevice, C->stream));
    RCCL_TRY(api->AllReduce(d, d + 1, 1, ncclFloat64, ncclMax, C->comm, C->stream));
    HIP_TRY(hipMemcpyAsync(inout, d + 1, sizeof(double), hipMemcpyDeviceToHost, C->stream));
    HIP_TRY(hipStreamSynchronize(C->stream));
    return ADVNTR_OK;
}

extern "C" int advntr_comm_barrier(advntr_comm *C)
{
    double x = 0.0;
    return advntr_comm_allreduce_max_f64(C, &x);
}

// gather-v of `count_bytes[r]` bytes from every rank r to `root`, all on C->stream (no sync): rank r's bytes start at
// dst + sum(count_bytes[0..r)).  comm_gatherv_post only posts the sends / receives (inside the caller's ncclGroup, so that
// several arrays travel in ONE group = one RCCL launch per rank); comm_gatherv_own copies the root's own share.
static int comm_gatherv_post(advntr_comm *C, int root, const void *src, void *dst, const int64_t *count_bytes)
{
    RcclApi *api = rccl_api();
    if (C->rank == root) {
        size_t at = 0;
        for (int r = 0; r < C->world; ++r) {
            const size_t nb = (size_t)count_bytes[r];
            if (r != root && nb) RCCL_TRY(api->Recv((uint8_t *)dst + at, nb, ncclUint8, r, C->comm, C->stream));
            at += nb;
        }
    } else if (count_bytes[C->rank]) {
        RCCL_TRY(api->Send(src, (size_t)count_bytes[C->rank], ncclUint8, root, C->comm, C->stream));
    }
    return ADVNTR_OK;
}

static int comm_gatherv_own(advntr_comm *C, int root, const void *src, void *dst, const int64_t *count_bytes)
{
    if (C->rank != root || !count_bytes[root]) return ADVNTR_OK;
    size_t mine_at = 0;
    for (int r = 0; r < root; ++r) mine_at += (size_t)count_bytes[r];
    HIP_TRY(hipMemcpyAsync((uint8_t *)dst + mine_at, src, (size_t)count_bytes[root], hipMemcpyDeviceToDevice, C->stream));
    return ADVNTR_OK;
}

static int comm_gatherv(advntr_comm *C, int root, const void *src, void *dst, const int64_t *count_bytes)
{
    RcclApi *api = rccl_api();
    RCCL_TRY(api->GroupStart());
    const int rc = comm_gatherv_post(C, root, src, dst, count_bytes);
    const ncclResult_t e = api->GroupEnd();
    if (rc != ADVNTR_OK) return rc;
    RCCL_TRY(e);
    return comm_gatherv_own(C, root, src, dst, count_bytes);
}

// Start the gather of the batch's result records (as they are after the kernels queued on the batch's stream so far)
// to `root`.  counts[r] = reads of rank r's batch (advntr_comm_allgather_i64, or known from the shared plan).  Returns at
// once: the records are first copied to staging buffers ON THE BATCH'S STREAM, so the next advntr_batch_run may follow
// immediately and computes while the gather is in flight on the communicator's stream.
extern "C" int advntr_comm_gather_results_start(advntr_comm *C, advntr_batch *B, int32_t root, const int64_t *counts)
{
    if (!C || !B || !counts || root < 0 || root >= C->world)
        return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_start: bad argument");
    if (C->in_flight) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_start: the previous gather was not finished");
    if (counts[C->rank] != B->n_reads)
        return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_start: counts[%d] = %lld, the batch holds %d reads", C->rank,
                    (long long)counts[C->rank], B->n_reads);
    if (B->device != C->device) return fail(ADVNTR_ERR_DEVICE, "batch and communicator live on different devices");
    const size_t n = (size_t)B->n_reads;
    size_t total = 0;
    for (int r = 0; r < C->world; ++r) {
        if (counts[r] < 0) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_start: negative count");
        total += (size_t)counts[r];
    }
    int rc;
    if ((rc = C->reserve((void **)&C->d_stage_logp, &C->cap_stage_logp, std::max<size_t>(n, 1) * sizeof(double)))) return rc;
    if ((rc = C->reserve((void **)&C->d_stage_sum, &C->cap_stage_sum, std::max<size_t>(n, 1) * 8 * sizeof(int32_t)))) return rc;
    if (C->rank == root) {
        if ((rc = C->reserve((void **)&C->d_all_logp, &C->cap_all_logp, std::max<size_t>(total, 1) * sizeof(double)))) return rc;
        if ((rc = C->reserve((void **)&C->d_all_sum, &C->cap_all_sum, std::max<size_t>(total, 1) * 8 * sizeof(int32_t)))) return rc;
    }
    const bool with_summary = !(B->flags & ADVNTR_FLAG_NO_SUMMARY);
    if (n) {
        HIP_TRY(hipMemcpyAsync(C->d_stage_logp, B->d_logp, n * sizeof(double), hipMemcpyDeviceToDevice, B->stream));
        if (with_summary)
            HIP_TRY(hipMemcpyAsync(C->d_stage_sum, B->d_summary, n * 8 * sizeof(int32_t), hipMemcpyDeviceToDevice, B->stream));
        else
            HIP_TRY(hipMemsetAsync(C->d_stage_sum, 0, n * 8 * sizeof(int32_t), B->stream));
    }
    HIP_TRY(hipEventRecord(C->staged, B->stream));
    HIP_TRY(hipStreamWaitEvent(C->stream, C->staged, 0));
    HIP_TRY(hipEventRecord(C->g0, C->stream));
    // RCCL's send / receive kernels need a few workgroup slots.  The scoring kernels are persistent -- their workgroups stay
    // until the pass is done and fill every compute unit's registers --, so with peers the passes that follow leave a few
    // slots unclaimed: the gather of pass i then runs beside pass i + 1 instead of behind it (1 % of the slots).  The request
    // covers the next advntr_batch_run only and only launches that fill the device honour it (column_launch.h: launch_grid)
    if (C->world > 1) B->col.reserve_workgroups = 8;
    // both arrays in one group: one RCCL launch per rank and pass
    std::vector<int64_t> bytes_l(C->world), bytes_s(C->world);
    for (int r = 0; r < C->world; ++r) {
        bytes_l[r] = counts[r] * (int64_t)sizeof(double);
        bytes_s[r] = counts[r] * (int64_t)(8 * sizeof(int32_t));
    }
    RcclApi *api = rccl_api();
    RCCL_TRY(api->GroupStart());
    rc = comm_gatherv_post(C, root, C->d_stage_logp, C->d_all_logp, bytes_l.data());
    if (rc == ADVNTR_OK) rc = comm_gatherv_post(C, root, C->d_stage_sum, C->d_all_sum, bytes_s.data());
    const ncclResult_t ge = api->GroupEnd();
    if (rc != ADVNTR_OK) return rc;
    RCCL_TRY(ge);
    if ((rc = comm_gatherv_own(C, root, C->d_stage_logp, C->d_all_logp, bytes_l.data()))) return rc;
    if ((rc = comm_gatherv_own(C, root, C->d_stage_sum, C->d_all_sum, bytes_s.data()))) return rc;
    HIP_TRY(hipEventRecord(C->g1, C->stream));
    C->counts.assign(counts, counts + C->world);
    C->root = root;
    C->in_flight = true;
    return ADVNTR_OK;
}

// Wait for the gather started last.  On the root, out_logp / out_summary (host, sum(counts) records in rank order;
// either may be NULL) receive the gathered records; other ranks pass NULL.
extern "C" int advntr_comm_gather_results_finish(advntr_comm *C, double *out_logp, int32_t *out_summary)
{
    if (!C) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_finish: null communicator");
    if (!C->in_flight) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_results_finish: no gather in flight");
    C->in_flight = false;
    if (C->rank == C->root) {
        size_t total = 0;
        for (int64_t c : C->counts) total += (size_t)c;
        if (out_logp && total)
            HIP_TRY(hipMemcpyAsync(out_logp, C->d_all_logp, total * sizeof(double), hipMemcpyDeviceToHost, C->stream));
        if (out_summary && total)
            HIP_TRY(hipMemcpyAsync(out_summary, C->d_all_sum, total * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, C->stream));
    }
    HIP_TRY(hipStreamSynchronize(C->stream));
    if (hipEventElapsedTime(&C->last_gather_ms, C->g0, C->g1) != hipSuccess) C->last_gather_ms = -1.f;
    return ADVNTR_OK;
}

// Milliseconds the gather finished last spent on the communicator's stream, from the moment its pass's records were
// staged to the arrival of the last record (HIP events): the transfer alone when it ran beside the next pass's kernels,
// about a whole pass when it had to wait for them.  -1 before the first gather.
extern "C" int advntr_comm_last_gather_ms(const advntr_comm *C, float *ms)
{
    if (!C || !ms) return fail(ADVNTR_ERR_ARG, "advntr_comm_last_gather_ms: bad argument");
    *ms = C->last_gather_ms;
    return ADVNTR_OK;
}

// Ragged gather of host byte strings through the devices (the per-locus result rows of the genotype driver):
// counts[r] = bytes of rank r (every rank passes the same array); dst (root only) receives them rank after rank.
extern "C" int advntr_comm_gather_bytes(advntr_comm *C, int32_t root, const void *src, const int64_t *counts, void *dst)
{
    if (!C || !counts || root < 0 || root >= C->world) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_bytes: bad argument");
    if (C->in_flight) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_bytes: a result gather is in flight");
    size_t total = 0;
    for (int r = 0; r < C->world; ++r) {
        if (counts[r] < 0) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_bytes: negative count");
        total += (size_t)counts[r];
    }
    const size_t mine = (size_t)counts[C->rank];
    if ((mine && !src) || (C->rank == root && total && !dst)) return fail(ADVNTR_ERR_ARG, "advntr_comm_gather_bytes: null buffer");
    const size_t send_cap = (mine + 255) & ~size_t(255);
    int rc;
    if ((rc = C->reserve((void **)&C->d_bytes, &C->bytes_cap, send_cap + (C->rank == root ? total : 0) + 256))) return rc;
    uint8_t *d_recv = C->d_bytes + send_cap;
    if (mine) HIP_TRY(hipMemcpyAsync(C->d_bytes, src, mine, hipMemcpyHostToDevice, C->stream));
    if ((rc = comm_gatherv(C, root, C->d_bytes, d_recv, counts))) return rc;
    if (C->rank == root && total) HIP_TRY(hipMemcpyAsync(dst, d_recv, total, hipMemcpyDeviceToHost, C->stream));
    HIP_TRY(hipStreamSynchronize(C->stream));
    return ADVNTR_OK;
}
