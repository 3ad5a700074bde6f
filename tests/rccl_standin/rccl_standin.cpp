// rccl_standin.cpp — TEST INFRASTRUCTURE, not part of the product and never loaded by it on its own.
//
// A stand-in for librccl that connects the ranks of ONE PROCESS (one thread per rank, every rank with an engine of
// its own on the same GPU).  RCCL itself refuses two ranks of one communicator on one device and the test boxes have
// one GPU, so the multi-rank paths of the router behind the C ABI (csrc/dj_router.cpp: bucket offsets, pieces,
// rounds) could otherwise only run with one rank.  The library exports the nine entry points the router binds and is
// built as tests/rccl_standin/librccl.so.1; only tests/rccl_standin/router_ranks (run with that directory on the
// dynamic linker's search path) ever sees it.
//
// Semantics kept: ncclAllGather is a collective in call order per communicator; ncclSend / ncclRecv inside one
// ncclGroupStart/End pair match by (source, destination) in issue order, a group with no operation is a no-op, and
// nothing blocks before ncclGroupEnd.  Everything is done synchronously at the call (the caller's stream is drained
// first, the bytes are moved with device-to-device copies on that stream, which are waited for), which is a legal if slow
// execution of calls that RCCL would merely queue on the stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Post { const void* buf; size_t bytes; bool taken; };
struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Post*>> p2p;          // (from, to) -> posted sends, oldest first
    std::map<long, std::vector<const void*>> gathers;                // call number -> every rank's send buffer
    std::map<long, int> gathers_read;                                // call number -> ranks that have copied
};
struct Comm { World* w; int rank; long gather_calls; };
struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; Comm* c; hipStream_t stream; };

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
long g_ids = 0;
thread_local std::vector<Op> tl_ops;
thread_local int tl_depth = 0;

size_t width(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

ncclResult_t run(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    for (Op& o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<Post> mine(ops.size());
    World* w = ops[0].c->w;
    {   // every send is posted before any receive waits: no order of ranks can deadlock
        std::lock_guard<std::mutex> g(w->mu);
        for (size_t k = 0; k < ops.size(); ++k)
            if (ops[k].send) {
                mine[k] = Post{ops[k].sbuf, ops[k].bytes, false};
                w->p2p[{ops[k].c->rank, ops[k].peer}].push_back(&mine[k]);
            }
    }
    w->cv.notify_all();
    for (Op& o : ops) {
        if (o.send) continue;
        Post* p = nullptr;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            auto& q = w->p2p[{o.peer, o.c->rank}];
            w->cv.wait(lk, [&] { return !q.empty(); });
            p = q.front();
            q.pop_front();
        }
        if (p->bytes != o.bytes) return ncclInvalidArgument;          // the two sides planned different pieces
        // on the caller's stream, then waited for: a device-to-device hipMemcpy may return before the bytes have moved, and
        // it runs on the null stream, which the router's and the engine's non-blocking streams do not wait for
        if (hipMemcpyAsync(o.rbuf, p->buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
            hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> g(w->mu);
            p->taken = true;
        }
        w->cv.notify_all();
    }
    std::unique_lock<std::mutex> lk(w->mu);
    for (size_t k = 0; k < ops.size(); ++k)
        if (ops[k].send) w->cv.wait(lk, [&] { return mine[k].taken; });   // the buffer may be reused after the group
    return ncclSuccess;
}

ncclResult_t queue_or_run(const Op& o) {
    tl_ops.push_back(o);
    if (tl_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run(ops);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> g(g_mu);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "standin-%ld", ++g_ids);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId id, int rank) {
    std::lock_guard<std::mutex> g(g_mu);
    World*& w = g_worlds[std::string(id.internal)];
    if (!w) { w = new World(); w->n = n; }
    if (w->n != n || rank < 0 || rank >= n) return ncclInvalidArgument;
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank, 0});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++tl_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    return queue_or_run(Op{true, buf, nullptr, count * width(t), peer, reinterpret_cast<Comm*>(comm), stream});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    return queue_or_run(Op{false, nullptr, buf, count * width(t), peer, reinterpret_cast<Comm*>(comm), stream});
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    World* w = c->w;
    const size_t bytes = count * width(t);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const long call = c->gather_calls++;
    std::vector<const void*> all;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        auto& v = w->gathers[call];
        if (v.empty()) v.assign(w->n, nullptr);
        v[c->rank] = send;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { for (const void* p : w->gathers[call]) if (!p) return false; return true; });
        all = w->gathers[call];
    }
    for (int r = 0; r < w->n; ++r)
        if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)r * bytes, all[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(w->mu);
    ++w->gathers_read[call];
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->gathers_read[call] == w->n; });       // nobody's send buffer changes before all have read it
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "error in the stand-in"; }
}
