// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that lets the ranks of ONE process on
// ONE GPU talk to each other, so that csrc/abi_comm.h's peer paths (the Send branch, a Recv per peer, a zero-count peer, a
// root other than 0) execute on a 1-GPU box.  Loaded through ADVNTR_RCCL_LIB (abi_comm.h: rccl_api); every "rank" is a host
// THREAD with its own advntr_comm.  Semantics kept from RCCL: ncclCommInitRank returns only when all ranks have entered it;
// a collective completes only when every rank has called it; ncclSend / ncclRecv are matched per (source, destination) in
// posting order and carried out at ncclGroupEnd; sizes of a matched pair must agree.  The transfers are hipMemcpyAsync device
// to device, ordered against the ranks' streams with events.  The prototypes come from <rccl/rccl.h> itself, so this file
// fails to compile if they drift.  Every wait gives up after 60 s with ncclInternalError instead of hanging a test.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {

std::mutex g_mu;
std::condition_variable g_cv;
const auto kPatience = std::chrono::seconds(60);

struct SendPost {
    const void *src; size_t bytes; hipEvent_t ready; hipStream_t stream;
    bool consumed = false; hipEvent_t done = nullptr; bool size_mismatch = false;
};

struct Collective {               // one slot per job: ranks deposit, the last one in carries it out
    int arrived = 0, generation = 0;
    std::vector<const void *> send; std::vector<void *> recv; std::vector<hipStream_t> stream;
};

struct Job {
    int nranks = 0, joined = 0, destroyed = 0;
    std::map<std::pair<int, int>, std::deque<SendPost *>> mail;        // (source, destination) -> posted sends, in order
    Collective coll;
};

struct FakeComm { Job *job; int rank, nranks; };

std::map<uint64_t, Job *> g_jobs;
uint64_t g_next_id = 1;
int64_t g_stats[8];               // 0 sends, 1 recvs, 2 groups with operations, 3 bytes sent, 4 all-gathers, 5 all-reduces, 6 comms

struct Op { bool is_send; const void *src; void *dst; size_t bytes; int peer; FakeComm *comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

ncclResult_t flush_ops()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (ops.empty()) return ncclSuccess;
    std::vector<SendPost *> mine;
    ncclResult_t rc = ncclSuccess;
    {   // 1. post every send (never blocks)
        std::unique_lock<std::mutex> lk(g_mu);
        g_stats[2] += 1;
        for (const Op &o : ops) {
            if (!o.is_send) continue;
            SendPost *p = new SendPost{o.src, o.bytes, nullptr, o.stream};
            if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(p->ready, o.stream) != hipSuccess)
                return ncclUnhandledCudaError;
            o.comm->job->mail[{o.comm->rank, o.peer}].push_back(p);
            mine.push_back(p);
            g_stats[0] += 1;
            g_stats[3] += (int64_t)o.bytes;
        }
        g_cv.notify_all();
    }
    // 2. every receive takes the oldest send its peer posted for this rank
    for (const Op &o : ops) {
        if (o.is_send) continue;
        SendPost *p = nullptr;
        {
            std::unique_lock<std::mutex> lk(g_mu);
            auto &q = o.comm->job->mail[{o.peer, o.comm->rank}];
            if (!g_cv.wait_for(lk, kPatience, [&] { return !q.empty(); })) return ncclInternalError;
            p = q.front();
            q.pop_front();
            g_stats[1] += 1;
        }
        if (p->bytes != o.bytes) { p->size_mismatch = true; rc = ncclInvalidArgument; }
        else if (hipStreamWaitEvent(o.stream, p->ready, 0) != hipSuccess ||
                 hipMemcpyAsync(o.dst, p->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
        hipEvent_t done = nullptr;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, o.stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(g_mu);
        p->done = done;
        p->consumed = true;
        g_cv.notify_all();
    }
    // 3. a sender's stream may not run ahead of the copy out of its buffer
    for (SendPost *p : mine) {
        {
            std::unique_lock<std::mutex> lk(g_mu);
            if (!g_cv.wait_for(lk, kPatience, [&] { return p->consumed; })) return ncclInternalError;
        }
        if (p->size_mismatch) rc = ncclInvalidArgument;
        if (p->done) { (void)hipStreamWaitEvent(p->stream, p->done, 0); (void)hipEventDestroy(p->done); }
        (void)hipEventDestroy(p->ready);
        delete p;
    }
    return rc;
}

// deposit this rank's buffers; the last rank to arrive runs `carry_out` for everybody; all leave together
template <class F> ncclResult_t collective(FakeComm *c, const void *send, void *recv, hipStream_t stream, F carry_out)
{
    std::unique_lock<std::mutex> lk(g_mu);
    Collective &k = c->job->coll;
    if (k.arrived == 0) { k.send.assign(c->nranks, nullptr); k.recv.assign(c->nranks, nullptr); k.stream.assign(c->nranks, nullptr); }
    k.send[c->rank] = send; k.recv[c->rank] = recv; k.stream[c->rank] = stream;
    const int gen = k.generation;
    if (++k.arrived == c->nranks) {
        ncclResult_t rc = ncclSuccess;
        for (int r = 0; r < c->nranks; ++r)
            if (hipStreamSynchronize(k.stream[r]) != hipSuccess) rc = ncclUnhandledCudaError;
        if (rc == ncclSuccess) rc = carry_out(k);
        k.arrived = 0;
        k.generation += 1;
        g_cv.notify_all();
        return rc;
    }
    if (!g_cv.wait_for(lk, kPatience, [&] { return k.generation != gen; })) return ncclInternalError;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::unique_lock<std::mutex> lk(g_mu);
    memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, sizeof v);
    memcpy(id->internal + 8, "fake-rccl", 9);
    g_jobs[v] = new Job();
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    uint64_t v;
    memcpy(&v, id.internal, sizeof v);
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_jobs.find(v);
    if (it == g_jobs.end() || memcmp(id.internal + 8, "fake-rccl", 9) != 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Job *j = it->second;
    if (j->nranks == 0) j->nranks = nranks;
    if (j->nranks != nranks) return ncclInvalidArgument;
    j->joined += 1;
    g_cv.notify_all();
    if (!g_cv.wait_for(lk, kPatience, [&] { return j->joined >= j->nranks; })) return ncclInternalError;     // a collective
    *comm = reinterpret_cast<ncclComm_t>(new FakeComm{j, rank, nranks});
    g_stats[6] += 1;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<FakeComm *>(comm);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (fake RCCL: mismatched sizes or ranks)";
    case ncclInternalError: return "internal error (fake RCCL: a peer did not show up within 60 s)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    default: return "error (fake RCCL)";
    }
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t nb = count * type_bytes(t);
    { std::unique_lock<std::mutex> lk(g_mu); g_stats[4] += 1; }
    return collective(c, sendbuff, recvbuff, stream, [&](Collective &k) {
        for (int dst = 0; dst < c->nranks; ++dst)
            for (int src = 0; src < c->nranks; ++src)
                if (hipMemcpy((uint8_t *)k.recv[dst] + (size_t)src * nb, k.send[src], nb, hipMemcpyDeviceToDevice) != hipSuccess)
                    return ncclUnhandledCudaError;
        return ncclSuccess;
    });
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (t != ncclFloat64 || op != ncclMax) return ncclInvalidArgument;           // all abi_comm.h asks for
    { std::unique_lock<std::mutex> lk(g_mu); g_stats[5] += 1; }
    return collective(c, sendbuff, recvbuff, stream, [&](Collective &k) {
        std::vector<double> best(count), one(count);
        for (int r = 0; r < c->nranks; ++r) {
            if (hipMemcpy(one.data(), k.send[r], count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            for (size_t i = 0; i < count; ++i) best[i] = (r == 0 || one[i] > best[i]) ? one[i] : best[i];
        }
        for (int r = 0; r < c->nranks; ++r)
            if (hipMemcpy(k.recv[r], best.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    });
}

ncclResult_t ncclGroupStart() { t_depth += 1; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth) return ncclSuccess;
    return flush_ops();
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (peer < 0 || peer >= c->nranks || peer == c->rank || !type_bytes(t)) return ncclInvalidArgument;
    t_ops.push_back(Op{true, sendbuff, nullptr, count * type_bytes(t), peer, c, stream});
    return t_depth ? ncclSuccess : flush_ops();
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (peer < 0 || peer >= c->nranks || peer == c->rank || !type_bytes(t)) return ncclInvalidArgument;
    t_ops.push_back(Op{false, nullptr, recvbuff, count * type_bytes(t), peer, c, stream});
    return t_depth ? ncclSuccess : flush_ops();
}

// what the test reads: {sends, receives, groups with operations, bytes sent, all-gathers, all-reduces, communicators, 0}
void fake_rccl_stats(int64_t out[8])
{
    std::unique_lock<std::mutex> lk(g_mu);
    memcpy(out, g_stats, sizeof g_stats);
}

}  // extern "C"
