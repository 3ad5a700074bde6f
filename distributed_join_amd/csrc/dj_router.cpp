// dj_router.cpp — multi-GPU routing behind the C ABI (include/dj_engine.h, "routing").
//
// Barriers shard by fingerprint over the GPUs of one node (dj_owner_of); a batch of join or fork
// records reaches its owners with ONE exchange step: stable partition by owner (dj_partition_joins)
// -> all-gather of the bucket sizes -> grouped send/recv of the buckets over RCCL (xGMI).  The
// reference has no counterpart: its replicas share one Redis (reference src/distributed_join/redis.clj:13-15).
//
// A rank's OWN bucket never goes through the exchange: the partition writes it straight into the receive buffers
// (dj_partition_joins_self), at the front, and every receiver's stream is "its own rows first, then the other sources in
// rank order"; what arrives from the peers is placed right behind the own rows, so a piece is one contiguous run of
// rows for one engine call.  With one rank nothing is copied at all.
//
// Safety under skew: every rank sees the whole matrix of bucket sizes, so every rank computes the
// same plan (dj_route_plan): a receiver's incoming rows, in source-rank order, are cut into pieces
// of at most max_recv rows and exchanged piece by piece.  No rank ever decides alone.
//
// RCCL is bound at run time (dlopen), so the engine library loads on hosts without it; only
// dj_route_unique_id / dj_router_create need it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "dj_engine.h"

// ---- the plan ------------------------------------------------------------------------------
extern "C" uint32_t dj_route_plan(uint32_t world, uint32_t rank, const uint32_t* matrix, uint64_t max_recv,
                                  uint32_t round, uint32_t* send_off, uint32_t* send_cnt, uint32_t* recv_cnt) {
    if (world == 0 || rank >= world || !matrix || max_recv == 0) return 0;
    uint32_t rounds = 1;
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t total = 0;
        for (uint32_t s = 0; s < world; ++s) total += matrix[(size_t)s * world + r];
        const uint64_t k = (total + max_recv - 1) / max_recv;
        if (k > rounds) rounds = (uint32_t)k;
    }
    const uint64_t lo = (uint64_t)round * max_recv, hi = lo + max_recv;      // this round's window of every receiver's stream
    // A receiver's stream: its OWN rows first (they never cross the exchange: the partition puts them where the
    // receiver's engine reads them), then the other sources in rank order.
    auto piece = [&](uint32_t s, uint32_t r, uint32_t* off, uint32_t* cnt) {
        uint64_t before = s == r ? 0 : matrix[(size_t)r * world + r];       // rows ahead of s's in r's stream
        if (s != r)
            for (uint32_t q = 0; q < s; ++q)
                if (q != r) before += matrix[(size_t)q * world + r];
        const uint64_t end = before + matrix[(size_t)s * world + r];
        const uint64_t a = before > lo ? before : lo, b = end < hi ? end : hi;
        *off = a < b ? (uint32_t)(a - before) : 0;
        *cnt = a < b ? (uint32_t)(b - a) : 0;
    };
    for (uint32_t p = 0; p < world; ++p) {
        uint32_t off, cnt;
        if (send_off || send_cnt) {
            piece(rank, p, &off, &cnt);
            if (send_off) send_off[p] = off;
            if (send_cnt) send_cnt[p] = cnt;
        }
        if (recv_cnt) {
            piece(p, rank, &off, &cnt);
            recv_cnt[p] = cnt;
        }
    }
    return rounds;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local std::string g_router_err;

bool rccl_load(std::string* why) {
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.lib) return true;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { *why = std::string("librccl not found: ") + dlerror(); return false; }
    Rccl r;
    r.lib = lib;
#define BIND(field, name) \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, name)); \
    if (!r.field) { *why = std::string("librccl lacks ") + name; return false; }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(AllGather, "ncclAllGather")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    g_rccl = r;
    return true;
}

constexpr int MAX_RANKS = 64;     // dj_partition_joins' limit
constexpr int MAX_DEPTH = 8;

struct Set {
    dj_join_rec* recs_sorted = nullptr;
    uint8_t* pay_sorted = nullptr;
    uint32_t* origin = nullptr;
    uint32_t* counts = nullptr;       // [world] this rank's bucket sizes (device)
    uint32_t* matrix_d = nullptr;     // [world * world] everybody's
    uint32_t* gathered_h = nullptr;   // pinned copy of matrix_d: per rank its world bucket sizes and one word "has messages"
    uint32_t* matrix_h = nullptr;     // the world x world bucket sizes alone, as dj_route_plan reads them
    dj_join_rec* recv_recs = nullptr;
    uint8_t* recv_pay = nullptr;
    // the way back: the engine writes a piece's statuses to status_piece (piece order), they return to the ranks
    // their records came from and land in ret_sorted at the place of their record in recs_sorted
    int32_t* status_piece = nullptr;  // [max_recv]
    int32_t* ret_sorted = nullptr;    // [max_send]
    hipEvent_t parted = nullptr, counts_ev = nullptr, ready = nullptr, freed = nullptr;
    hipEvent_t applied = nullptr;     // engine stream: the call that wrote status_piece has been queued
    hipEvent_t returned = nullptr;    // communication stream: the latest piece's statuses have left status_piece
    hipEvent_t home = nullptr;        // communication stream: dj_route_statuses' scatter is done
    bool free_pending = false, has_pay = false, active = false, planned = false;
    bool unplanned_release = false;   // the set was released before its bucket sizes were ever waited for
    bool return_pending = false;      // `returned` has been recorded and nobody has waited for it yet
    bool home_pending = false;        // `home` likewise: the next partition into this set must not overwrite origin[] before
    uint32_t n = 0, rounds = 0;
    uint32_t round = 0;               // pieces handed to the caller
    uint32_t returned_rounds = 0;     // pieces whose statuses have been sent back
    uint32_t issued = 0;              // pieces whose exchange has been queued (round or round + 1)
    uint32_t n_in = 0;                // rows of the piece queued last
    uint32_t piece_at = 0;            // where that piece starts in the receive buffers (rows)
    uint32_t bucket_off[MAX_RANKS];   // where each destination's bucket starts in recs_sorted
};
}  // namespace

struct dj_router {
    dj_engine* e = nullptr;
    dj_config cfg;
    int world = 0, rank = 0, depth = 0, turn = 0;
    uint32_t max_send = 0, max_recv = 0;
    uint64_t pitch = 0;
    ncclComm_t comm = nullptr;
    // Three stages on three streams: the partition kernels (stream), everything that goes through RCCL
    // (comm_stream: its kernels hold a few compute units for as long as the links need, and would only add to
    // the partition's time on one stream with it), and the engine's own stream that applies what arrived.
    hipStream_t stream = nullptr, comm_stream = nullptr, engine_stream = nullptr;
    Set sets[MAX_DEPTH];
    uint32_t* agree_d = nullptr;      // [1 + MAX_RANKS] this rank's word, then everybody's (dj_route_agree)
    uint32_t* agree_h = nullptr;      // pinned copy of everybody's
    hipEvent_t agree_ev = nullptr;
    std::string err;
    std::mutex mu;
};

static int rfail(dj_router* r, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (r) r->err = buf; else g_router_err = buf;
    return code;
}

#define R_HIP(r, call)                                                                                  \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) return rfail(r, DJ_E_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)
#define R_NCCL(r, call)                                                                                      \
    do {                                                                                                     \
        ncclResult_t _e = (call);                                                                            \
        if (_e != ncclSuccess) return rfail(r, DJ_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(_e)); \
    } while (0)

extern "C" const char* dj_router_last_error(const dj_router* r) { return r ? r->err.c_str() : g_router_err.c_str(); }

extern "C" int dj_route_unique_id(uint8_t* id128) {
    if (!id128) return rfail(nullptr, DJ_E_INVALID, "null argument");
    std::string why;
    if (!rccl_load(&why)) return rfail(nullptr, DJ_E_INVALID, "%s", why.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    ncclUniqueId id;
    R_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return DJ_OK;
}

extern "C" void dj_router_destroy(dj_router* r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
    for (int k = 0; k < r->depth; ++k) {
        Set& s = r->sets[k];
        (void)hipFree(s.recs_sorted); (void)hipFree(s.pay_sorted); (void)hipFree(s.origin); (void)hipFree(s.counts);
        (void)hipFree(s.matrix_d); (void)hipFree(s.recv_recs); (void)hipFree(s.recv_pay);
        (void)hipFree(s.status_piece); (void)hipFree(s.ret_sorted);
        if (s.applied) (void)hipEventDestroy(s.applied);
        if (s.returned) (void)hipEventDestroy(s.returned);
        if (s.home) (void)hipEventDestroy(s.home);
        if (s.matrix_h) (void)hipHostFree(s.matrix_h);
        if (s.gathered_h) (void)hipHostFree(s.gathered_h);
        if (s.parted) (void)hipEventDestroy(s.parted);
        if (s.counts_ev) (void)hipEventDestroy(s.counts_ev);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.freed) (void)hipEventDestroy(s.freed);
    }
    (void)hipFree(r->agree_d);
    if (r->agree_h) (void)hipHostFree(r->agree_h);
    if (r->agree_ev) (void)hipEventDestroy(r->agree_ev);
    if (r->comm) (void)g_rccl.CommDestroy(r->comm);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    if (r->comm_stream) (void)hipStreamDestroy(r->comm_stream);
    delete r;
}

extern "C" int dj_router_create(dj_engine* e, uint32_t n_ranks, uint32_t rank, const uint8_t* id128, uint32_t max_send,
                                uint32_t max_recv, uint64_t pitch, uint32_t depth, dj_router** out) {
    if (!e || !id128 || !out) return rfail(nullptr, DJ_E_INVALID, "null argument");
    if (n_ranks < 1 || n_ranks > MAX_RANKS || rank >= n_ranks) return rfail(nullptr, DJ_E_INVALID, "n_ranks must be 1..%d, rank below it", MAX_RANKS);
    if (depth < 1 || depth > MAX_DEPTH) return rfail(nullptr, DJ_E_INVALID, "depth must be 1..%d", MAX_DEPTH);
    if (max_send < 1 || max_recv < 1 || pitch == 0 || pitch % 16) return rfail(nullptr, DJ_E_INVALID, "max_send, max_recv >= 1, pitch a positive multiple of 16");
    std::string why;
    if (!rccl_load(&why)) return rfail(nullptr, DJ_E_INVALID, "%s", why.c_str());
    dj_router* r = new (std::nothrow) dj_router();
    if (!r) return rfail(nullptr, DJ_E_NOMEM, "out of host memory");
    r->e = e;
    r->world = (int)n_ranks; r->rank = (int)rank; r->depth = (int)depth;
    r->max_send = max_send; r->max_recv = max_recv; r->pitch = pitch;
    int rc = dj_get_config(e, &r->cfg);
    if (rc == DJ_OK && max_send > r->cfg.max_batch) rc = rfail(nullptr, DJ_E_TOO_LARGE, "max_send exceeds the engine's max_batch");
    auto body = [&]() -> int {
        R_HIP(nullptr, hipSetDevice(r->cfg.device));
        R_HIP(nullptr, hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        R_HIP(nullptr, hipStreamCreateWithFlags(&r->comm_stream, hipStreamNonBlocking));
        r->engine_stream = (hipStream_t)dj_stream(e);
        for (int k = 0; k < r->depth; ++k) {
            Set& s = r->sets[k];
            R_HIP(nullptr, hipMalloc((void**)&s.recs_sorted, (size_t)max_send * 32));
            R_HIP(nullptr, hipMalloc((void**)&s.pay_sorted, (size_t)max_send * pitch));
            R_HIP(nullptr, hipMalloc((void**)&s.origin, (size_t)max_send * 4));
            R_HIP(nullptr, hipMalloc((void**)&s.counts, (MAX_RANKS + 1) * 4));
            R_HIP(nullptr, hipMalloc((void**)&s.matrix_d, (size_t)MAX_RANKS * (MAX_RANKS + 1) * 4));
            R_HIP(nullptr, hipHostMalloc((void**)&s.gathered_h, (size_t)MAX_RANKS * (MAX_RANKS + 1) * 4, hipHostMallocDefault));
            R_HIP(nullptr, hipHostMalloc((void**)&s.matrix_h, (size_t)MAX_RANKS * MAX_RANKS * 4, hipHostMallocDefault));
            // receive buffers: this rank's own bucket (up to max_send rows, written by the partition) with a round's
            // arrivals (up to max_recv rows) right behind it
            R_HIP(nullptr, hipMalloc((void**)&s.recv_recs, ((size_t)max_send + max_recv) * 32));
            R_HIP(nullptr, hipMalloc((void**)&s.recv_pay, ((size_t)max_send + max_recv) * pitch));
            R_HIP(nullptr, hipMalloc((void**)&s.status_piece, (size_t)max_recv * 4));
            R_HIP(nullptr, hipMalloc((void**)&s.ret_sorted, (size_t)max_send * 4));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.applied, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.returned, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.home, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.parted, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.counts_ev, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
            R_HIP(nullptr, hipEventCreateWithFlags(&s.freed, hipEventDisableTiming));
        }
        R_HIP(nullptr, hipMalloc((void**)&r->agree_d, (1 + MAX_RANKS) * 4));
        R_HIP(nullptr, hipHostMalloc((void**)&r->agree_h, (1 + MAX_RANKS) * 4, hipHostMallocDefault));
        R_HIP(nullptr, hipEventCreateWithFlags(&r->agree_ev, hipEventDisableTiming));
        ncclUniqueId id;
        memcpy(&id, id128, 128);
        R_NCCL(nullptr, g_rccl.CommInitRank(&r->comm, r->world, id, r->rank));
        return DJ_OK;
    };
    if (rc == DJ_OK) rc = body();
    if (rc != DJ_OK) {
        dj_router_destroy(r);
        return rc;
    }
    *out = r;
    return DJ_OK;
}

extern "C" int dj_route_start(dj_router* r, uint32_t n, const dj_join_rec* d_recs, const uint8_t* d_payloads, int* handle) {
    if (!r || !handle) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    if (n > r->max_send) return rfail(r, DJ_E_TOO_LARGE, "route: n=%u exceeds max_send=%u", n, r->max_send);
    if (n && !d_recs) return rfail(r, DJ_E_INVALID, "route: null records");
    Set& s = r->sets[r->turn];
    if (s.active) return rfail(r, DJ_E_INVALID, "route: all %d buffer sets are in use (dj_route_release the oldest)", r->depth);
    R_HIP(r, hipSetDevice(r->cfg.device));
    if (s.free_pending) {                 // the engine may still be reading this set's receive buffers
        R_HIP(r, hipStreamWaitEvent(r->stream, s.freed, 0));
        s.free_pending = false;
    }
    if (s.home_pending) {                 // the scatter of the set's previous batch may still be reading origin[]
        R_HIP(r, hipStreamWaitEvent(r->stream, s.home, 0));
        s.home_pending = false;
    }
    if (s.unplanned_release) {            // released without a piece taken: its all-gather may still be reading counts[]
        R_HIP(r, hipStreamWaitEvent(r->stream, s.counts_ev, 0));
        s.unplanned_release = false;
    }
    s.has_pay = d_payloads != nullptr;
    s.n = n;
    s.planned = false;
    s.return_pending = false;
    s.round = s.rounds = s.issued = s.n_in = s.returned_rounds = 0;
    int rc = dj_partition_joins_self(r->e, n, (uint32_t)r->world, (uint32_t)r->rank, d_recs, d_payloads, r->pitch, s.recs_sorted,
                                     s.has_pay ? s.pay_sorted : nullptr, s.recv_recs, s.has_pay ? s.recv_pay : nullptr,
                                     s.origin, s.counts, r->stream);
    if (rc != DJ_OK) return rfail(r, rc, "route: partition failed: %s", dj_last_error(r->e));
    // one more word behind the bucket sizes: does this rank's batch come with messages?  Every rank must agree (the
    // exchange pairs a send with a receive), and with the word in the gathered matrix every rank can check that
    R_HIP(r, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s.counts + r->world), s.has_pay ? 1 : 0, 1, r->stream));
    R_HIP(r, hipEventRecord(s.parted, r->stream));
    R_HIP(r, hipStreamWaitEvent(r->comm_stream, s.parted, 0));
    // everybody's bucket sizes to everybody (world * (world + 1) words), then to this host without waiting
    R_NCCL(r, g_rccl.AllGather(s.counts, s.matrix_d, (size_t)r->world + 1, ncclUint32, r->comm, r->comm_stream));
    R_HIP(r, hipMemcpyAsync(s.gathered_h, s.matrix_d, (size_t)r->world * (r->world + 1) * 4, hipMemcpyDeviceToHost, r->comm_stream));
    R_HIP(r, hipEventRecord(s.counts_ev, r->comm_stream));
    s.active = true;
    *handle = r->turn;
    r->turn = (r->turn + 1) % r->depth;
    return DJ_OK;
}

// Queues the exchange of piece s.issued on the communication stream (planning first, which waits for the bucket
// sizes on the host) and records s.ready behind it.
static int issue_piece(dj_router* r, Set& s) {
    const uint32_t w = (uint32_t)r->world;
    if (!s.planned) {
        R_HIP(r, hipEventSynchronize(s.counts_ev));      // already there when the start was issued a batch ahead
        for (uint32_t q = 0; q < w; ++q) {
            memcpy(s.matrix_h + (size_t)q * w, s.gathered_h + (size_t)q * (w + 1), (size_t)w * 4);
            // the same verdict on every rank (they all read the same matrix), before any rank has queued a transfer
            if ((s.gathered_h[(size_t)q * (w + 1) + w] != 0) != s.has_pay)
                return rfail(r, DJ_E_INVALID, "route: rank %u started this batch %s messages, rank %d %s them", q,
                             s.has_pay ? "without" : "with", r->rank, s.has_pay ? "with" : "without");
        }
        uint32_t off = 0;
        for (uint32_t p = 0; p < w; ++p) { s.bucket_off[p] = off; off += s.matrix_h[(size_t)r->rank * w + p]; }
        s.rounds = dj_route_plan(w, (uint32_t)r->rank, s.matrix_h, r->max_recv, 0, nullptr, nullptr, nullptr);
        s.planned = true;
    } else {
        if (s.issued >= s.rounds) return rfail(r, DJ_E_INVALID, "route: no piece left");
        // the receive buffers are reused: wait for what the engine's stream holds so far (the caller has
        // queued the kernels that read the previous piece)
        R_HIP(r, hipEventRecord(s.freed, r->engine_stream));
        R_HIP(r, hipStreamWaitEvent(r->comm_stream, s.freed, 0));
    }
    uint32_t send_off[MAX_RANKS], send_cnt[MAX_RANKS], recv_cnt[MAX_RANKS];
    dj_route_plan(w, (uint32_t)r->rank, s.matrix_h, r->max_recv, s.issued, send_off, send_cnt, recv_cnt);
    // The piece: this round's part of the own bucket, rows [own_off, own_off + own_cnt) of it, which lie at the same
    // rows of the receive buffers (the partition put the bucket at their front), and what the peers send, which goes
    // right behind the own bucket's end while own rows are part of the piece (the bucket ends inside this round's
    // window then) and to the front of the buffers otherwise (the own rows have all been consumed by earlier pieces).
    const uint32_t me = (uint32_t)r->rank, own = s.matrix_h[(size_t)me * w + me];
    const uint32_t own_off = send_off[me], own_cnt = send_cnt[me];
    const bool window_has_own = (uint64_t)s.issued * r->max_recv < own;
    const uint32_t piece_at = window_has_own ? own_off : 0;
    uint32_t n_in = own_cnt, at[MAX_RANKS];
    for (uint32_t p = 0; p < w; ++p) {
        at[p] = piece_at + n_in;
        if (p != me) n_in += recv_cnt[p];
    }
    // records and messages of all peers in ONE group: one RCCL launch per piece (both sides issue records first,
    // then messages, so the two transfers between a pair of ranks match up in that order)
    R_NCCL(r, g_rccl.GroupStart());
    auto queue_transfers = [&]() -> int {
        for (int pass = 0; pass < (s.has_pay ? 2 : 1); ++pass) {
            const size_t row = pass == 0 ? 32 : (size_t)r->pitch;
            const uint8_t* src = pass == 0 ? reinterpret_cast<const uint8_t*>(s.recs_sorted) : s.pay_sorted;
            uint8_t* dst = pass == 0 ? reinterpret_cast<uint8_t*>(s.recv_recs) : s.recv_pay;
            for (uint32_t p = 0; p < w; ++p) {
                if (p == me) continue;                   // in place already
                const uint8_t* from = src + (size_t)(s.bucket_off[p] + send_off[p]) * row;
                uint8_t* to = dst + (size_t)at[p] * row;
                if (send_cnt[p]) R_NCCL(r, g_rccl.Send(from, (size_t)send_cnt[p] * row, ncclUint8, (int)p, r->comm, r->comm_stream));
                if (recv_cnt[p]) R_NCCL(r, g_rccl.Recv(to, (size_t)recv_cnt[p] * row, ncclUint8, (int)p, r->comm, r->comm_stream));
            }
        }
        return DJ_OK;
    };
    const int queued = queue_transfers();
    const ncclResult_t ended = g_rccl.GroupEnd();          // the group is closed whatever happened inside it
    if (queued != DJ_OK) return queued;
    R_NCCL(r, ended);
    R_HIP(r, hipEventRecord(s.ready, r->comm_stream));
    s.n_in = n_in;
    s.piece_at = piece_at;
    ++s.issued;
    return DJ_OK;
}

extern "C" int dj_route_advance(dj_router* r, int handle) {
    if (!r || handle < 0 || handle >= r->depth) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active) return rfail(r, DJ_E_INVALID, "route: handle %d is not in flight", handle);
    if (s.issued > 0) return DJ_OK;        // only the first piece can cross ahead: later ones reuse its buffers
    R_HIP(r, hipSetDevice(r->cfg.device));
    return issue_piece(r, s);
}

extern "C" int dj_route_next(dj_router* r, int handle, const dj_join_rec** d_recs_out, const uint8_t** d_payloads_out,
                             uint32_t* n_out, int* more) {
    if (!r || handle < 0 || handle >= r->depth || !d_recs_out || !n_out || !more) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active) return rfail(r, DJ_E_INVALID, "route: handle %d is not in flight", handle);
    R_HIP(r, hipSetDevice(r->cfg.device));
    if (s.planned && s.round >= s.rounds) return rfail(r, DJ_E_INVALID, "route: handle %d has no piece left", handle);
    if (s.issued == s.round) {             // not sent ahead by dj_route_advance
        const int rc = issue_piece(r, s);
        if (rc != DJ_OK) return rc;
    }
    R_HIP(r, hipStreamWaitEvent(r->engine_stream, s.ready, 0));      // the engine's stream waits, the host does not
    if (s.return_pending) {                // the previous piece's statuses must have left status_piece
        R_HIP(r, hipStreamWaitEvent(r->engine_stream, s.returned, 0));
        s.return_pending = false;
    }
    ++s.round;
    *d_recs_out = s.recv_recs + s.piece_at;
    if (d_payloads_out) *d_payloads_out = s.has_pay ? s.recv_pay + (size_t)s.piece_at * r->pitch : nullptr;
    *n_out = s.n_in;
    *more = s.round < s.rounds ? 1 : 0;
    return DJ_OK;
}

extern "C" int dj_route_release(dj_router* r, int handle) {
    if (!r || handle < 0 || handle >= r->depth) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active) return rfail(r, DJ_E_INVALID, "route: handle %d is not in flight", handle);
    if (s.planned && s.round < s.rounds) return rfail(r, DJ_E_INVALID, "route: handle %d still has pieces to exchange (a collective: every rank must take them)", handle);
    R_HIP(r, hipSetDevice(r->cfg.device));
    R_HIP(r, hipEventRecord(s.freed, r->engine_stream));
    s.free_pending = true;
    s.unplanned_release = !s.planned;
    s.active = false;
    return DJ_OK;
}

// ---- the way back: per-record answers to the rank that submitted the record ----------------------------------------
extern "C" int dj_route_status_buffer(dj_router* r, int handle, int32_t** d_status_piece) {
    if (!r || handle < 0 || handle >= r->depth || !d_status_piece) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active) return rfail(r, DJ_E_INVALID, "route: handle %d is not in flight", handle);
    *d_status_piece = s.status_piece;
    return DJ_OK;
}

extern "C" int dj_route_return(dj_router* r, int handle) {
    if (!r || handle < 0 || handle >= r->depth) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active || !s.planned || s.round == 0) return rfail(r, DJ_E_INVALID, "route: handle %d has no piece whose statuses could return", handle);
    if (s.returned_rounds + 1 != s.round) return rfail(r, DJ_E_INVALID, "route: the statuses of piece %u of handle %d have %s", s.round - 1, handle,
                                                        s.returned_rounds == s.round ? "been returned already" : "to follow those of the piece before it");
    R_HIP(r, hipSetDevice(r->cfg.device));
    const uint32_t w = (uint32_t)r->world;
    // the piece's plan once more, read backwards: what this rank received from p it answers to p, what it sent to p it
    // gets answers for, at the same places
    uint32_t send_off[MAX_RANKS], send_cnt[MAX_RANKS], recv_cnt[MAX_RANKS], at[MAX_RANKS];
    dj_route_plan(w, (uint32_t)r->rank, s.matrix_h, r->max_recv, s.round - 1, send_off, send_cnt, recv_cnt);
    const uint32_t me = (uint32_t)r->rank;
    uint32_t n_in = send_cnt[me];                  // the piece's rows in order: own rows, then the peers' in rank order
    for (uint32_t p = 0; p < w; ++p) {
        at[p] = p == me ? 0 : n_in;
        if (p != me) n_in += recv_cnt[p];
    }
    // behind the engine call that writes the statuses (the caller has queued it), on the communication stream
    R_HIP(r, hipEventRecord(s.applied, r->engine_stream));
    R_HIP(r, hipStreamWaitEvent(r->comm_stream, s.applied, 0));
    if (send_cnt[me])      // the answers to this rank's own rows stay here
        R_HIP(r, hipMemcpyAsync(s.ret_sorted + s.bucket_off[me] + send_off[me], s.status_piece, (size_t)send_cnt[me] * 4,
                                hipMemcpyDeviceToDevice, r->comm_stream));
    R_NCCL(r, g_rccl.GroupStart());
    auto queue_transfers = [&]() -> int {
        for (uint32_t p = 0; p < w; ++p) {
            if (p == me) continue;
            const int32_t* from = s.status_piece + at[p];
            int32_t* to = s.ret_sorted + s.bucket_off[p] + send_off[p];
            if (recv_cnt[p]) R_NCCL(r, g_rccl.Send(from, (size_t)recv_cnt[p], ncclInt32, (int)p, r->comm, r->comm_stream));
            if (send_cnt[p]) R_NCCL(r, g_rccl.Recv(to, (size_t)send_cnt[p], ncclInt32, (int)p, r->comm, r->comm_stream));
        }
        return DJ_OK;
    };
    const int queued = queue_transfers();
    const ncclResult_t ended = g_rccl.GroupEnd();
    if (queued != DJ_OK) return queued;
    R_NCCL(r, ended);
    R_HIP(r, hipEventRecord(s.returned, r->comm_stream));
    s.return_pending = true;
    ++s.returned_rounds;
    return DJ_OK;
}

extern "C" int dj_route_statuses(dj_router* r, int handle, int32_t* d_status_out, void* stream) {
    if (!r || handle < 0 || handle >= r->depth) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    Set& s = r->sets[handle];
    if (!s.active || !s.planned) return rfail(r, DJ_E_INVALID, "route: handle %d is not in flight", handle);
    if (s.returned_rounds != s.rounds) return rfail(r, DJ_E_INVALID, "route: %u of the %u pieces of handle %d have returned their statuses", s.returned_rounds, s.rounds, handle);
    if (s.n && !d_status_out) return rfail(r, DJ_E_INVALID, "route: null status buffer");
    R_HIP(r, hipSetDevice(r->cfg.device));
    // ret_sorted[j] answers record j of recs_sorted, which was record origin[j] of the caller's batch
    const int rc = dj_scatter_statuses(r->e, s.n, s.origin, s.ret_sorted, d_status_out, r->comm_stream);
    if (rc != DJ_OK) return rfail(r, rc, "route: scatter failed: %s", dj_last_error(r->e));
    R_HIP(r, hipEventRecord(s.home, r->comm_stream));
    s.home_pending = true;
    if (stream) R_HIP(r, hipStreamWaitEvent((hipStream_t)stream, s.home, 0));
    else R_HIP(r, hipEventSynchronize(s.home));
    return DJ_OK;
}

// ---- small agreements between the ranks' hosts ---------------------------------------------------------------------
extern "C" int dj_route_agree(dj_router* r, uint32_t word, uint32_t* h_words) {
    if (!r || !h_words) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(r->mu);
    R_HIP(r, hipSetDevice(r->cfg.device));
    r->agree_h[0] = word;
    R_HIP(r, hipMemcpyAsync(r->agree_d, r->agree_h, 4, hipMemcpyHostToDevice, r->comm_stream));
    R_NCCL(r, g_rccl.AllGather(r->agree_d, r->agree_d + 1, 1, ncclUint32, r->comm, r->comm_stream));
    R_HIP(r, hipMemcpyAsync(r->agree_h + 1, r->agree_d + 1, (size_t)r->world * 4, hipMemcpyDeviceToHost, r->comm_stream));
    R_HIP(r, hipEventRecord(r->agree_ev, r->comm_stream));
    R_HIP(r, hipEventSynchronize(r->agree_ev));
    memcpy(h_words, r->agree_h + 1, (size_t)r->world * 4);
    return DJ_OK;
}

extern "C" int dj_router_get_limits(const dj_router* r, uint32_t* n_ranks, uint32_t* rank, uint32_t* max_send, uint32_t* max_recv, uint64_t* pitch) {
    if (!r) return DJ_E_INVALID;
    if (n_ranks) *n_ranks = (uint32_t)r->world;
    if (rank) *rank = (uint32_t)r->rank;
    if (max_send) *max_send = r->max_send;
    if (max_recv) *max_recv = r->max_recv;
    if (pitch) *pitch = r->pitch;
    return DJ_OK;
}
