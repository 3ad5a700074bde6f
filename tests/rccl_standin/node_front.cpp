// node_front.cpp — TEST INFRASTRUCTURE.  The per-request front of a node whose barriers are sharded over several engines
// (dj_batcher_create_routed, include/dj_batcher.h): N ranks as threads of this program, each with its own engine on GPU 0,
// its router (over the stand-in for librccl of this directory) and its routed batcher; on every rank T request threads
// that call dj_batcher_fork_wait / dj_batcher_join_wait one request at a time, as the reference's handlers do
// (ring.clj:54-69), for barriers owned by ANY rank.
//
// Every barrier's requests come from one thread, one after the other, each answered before the next is sent, so their
// order is fixed and the answers must be exactly what the CPU oracle (oracle/dj_oracle.h, the checker) answers for the
// same sequence on one store: 404 before the fork (redis.clj:152), 201, WAITING / COMPLETE per join (core.clj:55,58), 400
// for a position beyond the count (redis.clj:153), replays, re-forks that reset the barrier (redis.clj:90-104).  One more
// barrier is shared: every request thread of every rank joins one position of it at the same time; exactly the last
// arrival is answered COMPLETE.  At the end every rank's publisher pops what is due on its own engine (dj_batcher_run);
// together they pop what the one store pops.
//
//   usage: node_front <ranks> <request threads per rank> <barriers>
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dj_batcher.h"
#include "dj_oracle.h"

static const uint32_t COUNT = 5, MSG = 24;
static const uint64_t T0 = 1700000000000ull;

static uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Req { bool fork; uint32_t position; uint32_t salt; };

// the request sequence of barrier b
static std::vector<Req> sequence(uint32_t b) {
    std::vector<Req> q;
    const uint64_t h = mix(b * 77ull + 5);
    if (h & 1) q.push_back({false, 0, 1});                       // before the fork: 404
    q.push_back({true, 0, 0});
    uint32_t order[COUNT];
    for (uint32_t p = 0; p < COUNT; ++p) order[p] = p;
    for (uint32_t p = COUNT - 1; p > 0; --p) { const uint32_t j = (uint32_t)(mix(h + p) % (p + 1)); const uint32_t t = order[p]; order[p] = order[j]; order[j] = t; }
    const bool leave_open = (h >> 1) % 5 == 0, refork = (h >> 4) % 7 == 0;
    const uint32_t upto = leave_open ? COUNT - 1 : COUNT;
    for (uint32_t k = 0; k < upto; ++k) {
        q.push_back({false, order[k], 2 + k});
        if (k == 1) q.push_back({false, order[0], 20});           // a replay with other bytes: the last writer wins
        if (k == 2) q.push_back({false, COUNT + (uint32_t)((h >> 8) % 3), 21});      // 400
        if (k == 2 && refork) {                                   // a second /v1/fork of the same id resets the barrier
            q.push_back({true, 0, 0});
            for (uint32_t j = 0; j <= k; ++j) q.push_back({false, order[j], 30 + j});
        }
    }
    if (!leave_open) q.push_back({false, order[0], 40});          // a replay after completion: COMPLETE again (HLEN == count)
    return q;
}

static void message_of(uint8_t* out, uint32_t b, uint32_t position, uint32_t salt) {
    uint64_t x = mix(((uint64_t)b << 32) ^ ((uint64_t)position << 16) ^ salt);
    for (uint32_t k = 0; k < MSG; k += 8) { x = mix(x); memcpy(out + k, &x, 8); }
}

struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    uint8_t id[128];
    std::atomic<int> failures{0};
    std::atomic<long> popped{0}, shared_complete{0}, shared_waiting{0};
};

static void barrier(Shared& s, int n) {
    std::unique_lock<std::mutex> lk(s.mu);
    const long my = s.gen;
    if (++s.arrived == n) { s.arrived = 0; ++s.gen; s.cv.notify_all(); }
    else s.cv.wait(lk, [&] { return s.gen != my; });
}

static uint64_t fixed_clock(void*) { return T0 + 5; }

struct PopTask { uint32_t cap; dj_cont* d_conts; uint32_t n; };
static int pop_task(void* user, dj_engine* e, uint64_t) {
    PopTask* t = static_cast<PopTask*>(user);
    dj_sweep_result res;
    const int rc = dj_pop_due(e, T0 + 1000, t->cap, t->d_conts, nullptr, &res);
    t->n = res.n_conts;
    return rc;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2;
    const int T = argc > 2 ? atoi(argv[2]) : 4;
    const uint32_t NB = argc > 3 ? (uint32_t)atoi(argv[3]) : 600;
    const uint32_t SHARED = NB, n_ids = NB + 1;                   // id NB is the shared barrier, count N * T
    std::vector<uint64_t> fp(2 * (size_t)n_ids);
    std::vector<uint8_t> id_bytes;
    std::vector<uint64_t> id_off(n_ids + 1, 0);
    for (uint32_t k = 0; k < n_ids; ++k) {
        char buf[64];
        snprintf(buf, sizeof buf, "n%x~t%x", k, k * 3);
        dj_fingerprint(buf, strlen(buf), &fp[2 * k], &fp[2 * k + 1]);
        id_bytes.insert(id_bytes.end(), buf, buf + strlen(buf));
        id_off[k + 1] = id_bytes.size();
    }
    Shared sh;
    std::vector<std::vector<int32_t>> answers(NB);                 // per barrier, in request order
    std::vector<std::thread> ranks;
    for (int r = 0; r < N; ++r)
        ranks.emplace_back([&, r] {
            if (hipSetDevice(0) != hipSuccess) exit(2);
            dj_config ec;
            memset(&ec, 0, sizeof ec);
            ec.struct_size = sizeof ec;
            ec.capacity_barriers = NB + 64;
            ec.arena_bytes = (uint64_t)(NB + 64) * 4096 + (4u << 20);
            ec.max_count = 64;
            ec.payload_stride = 64;
            ec.max_batch = 8192;
            ec.check_publish_delay_ms = 10000;
            ec.max_retry_publish = 3;
            ec.max_retry_timeout_publish = 3;
            dj_engine* e = nullptr;
            if (dj_create(&ec, &e) != DJ_OK) { fprintf(stderr, "rank %d: create: %s\n", r, dj_last_error(nullptr)); exit(2); }
            if (r == 0 && dj_route_unique_id(sh.id) != DJ_OK) { fprintf(stderr, "id: %s\n", dj_router_last_error(nullptr)); exit(2); }
            barrier(sh, N);
            dj_router* rt = nullptr;
            if (dj_router_create(e, (uint32_t)N, (uint32_t)r, sh.id, 4096, 8192, 32, 2, &rt) != DJ_OK) {
                fprintf(stderr, "rank %d: router: %s\n", r, dj_router_last_error(nullptr));
                exit(2);
            }
            dj_batcher_config bc;
            bc.struct_size = sizeof bc;
            bc.max_batch = 4096;
            bc.max_wait_us = 50;
            bc.max_message = MSG;                                    // staging rows of 32 bytes = the router's pitch
            dj_batcher* bt = nullptr;
            if (dj_batcher_create_routed(e, rt, &bc, &bt) != DJ_OK) { fprintf(stderr, "rank %d: routed batcher refused\n", r); exit(2); }
            dj_batcher_set_clock(bt, fixed_clock, nullptr);
            if (r == 0) {                                            // the shared barrier exists before anybody joins it
                dj_fork_rec f;
                memset(&f, 0, sizeof f);
                f.fp_lo = fp[2 * SHARED]; f.fp_hi = fp[2 * SHARED + 1]; f.count = (uint32_t)(N * T); f.deadline_ms = T0 + 3600000;
                int32_t st = 0;
                if (dj_batcher_fork_wait(bt, &f, &st) != DJ_OK || st != DJ_FORK_CREATED) { fprintf(stderr, "shared fork answered %d\n", st); ++sh.failures; }
            }
            barrier(sh, N);
            std::vector<std::thread> req;
            for (int t = 0; t < T; ++t)
                req.emplace_back([&, t] {
                    const uint32_t me = (uint32_t)(r * T + t), all = (uint32_t)(N * T);
                    uint8_t msg[MSG];
                    {   // the shared barrier: this thread's one position
                        dj_join_rec j;
                        memset(&j, 0, sizeof j);
                        j.fp_lo = fp[2 * SHARED]; j.fp_hi = fp[2 * SHARED + 1]; j.position = me; j.len = MSG;
                        message_of(msg, SHARED, me, 0);
                        int32_t st = 0;
                        if (dj_batcher_join_wait(bt, &j, msg, &st) != DJ_OK) { ++sh.failures; return; }
                        if (st == DJ_JOIN_COMPLETE) ++sh.shared_complete;
                        else if (st == DJ_JOIN_WAITING) ++sh.shared_waiting;
                        else { fprintf(stderr, "shared join answered %d\n", st); ++sh.failures; }
                    }
                    for (uint32_t b = me; b < NB; b += all) {
                        for (const Req& q : sequence(b)) {
                            int32_t st = 0;
                            int rc;
                            if (q.fork) {
                                dj_fork_rec f;
                                memset(&f, 0, sizeof f);
                                f.fp_lo = fp[2 * b]; f.fp_hi = fp[2 * b + 1]; f.count = COUNT; f.deadline_ms = T0 + 3600000;
                                rc = dj_batcher_fork_wait(bt, &f, &st);
                            } else {
                                dj_join_rec j;
                                memset(&j, 0, sizeof j);
                                j.fp_lo = fp[2 * b]; j.fp_hi = fp[2 * b + 1]; j.position = q.position; j.len = MSG;
                                message_of(msg, b, q.position, q.salt);
                                rc = dj_batcher_join_wait(bt, &j, msg, &st);
                            }
                            if (rc != DJ_OK) { fprintf(stderr, "rank %d: request refused (%d)\n", r, rc); ++sh.failures; return; }
                            answers[b].push_back(st);
                        }
                    }
                });
            for (auto& x : req) x.join();
            barrier(sh, N);                                          // every request of every rank has been answered
            PopTask pt;
            pt.cap = NB + 64;
            pt.n = 0;
            if (hipMalloc((void**)&pt.d_conts, (size_t)pt.cap * sizeof(dj_cont)) != hipSuccess) exit(2);
            if (dj_batcher_run(bt, pop_task, &pt) != DJ_OK) { fprintf(stderr, "rank %d: pop failed: %s\n", r, dj_last_error(e)); ++sh.failures; }
            sh.popped += pt.n;
            dj_batcher_stats bs;
            dj_batcher_get_stats(bt, &bs);
            if (r == 0) printf("rank 0: %llu forks and %llu joins staged here, %llu routed batches\n", (unsigned long long)bs.forks, (unsigned long long)bs.joins, (unsigned long long)bs.batches);
            dj_batcher_destroy(bt);                                  // a collective: every rank's drain thread leaves in the same cycle
            dj_router_destroy(rt);
            (void)hipFree(pt.d_conts);
            dj_destroy(e);
        });
    for (auto& t : ranks) t.join();

    // one store, barrier by barrier (barriers do not interact)
    djo* o = djo_create(10000, 3, 3, 64, 1u << 20);
    djo_set_ids(o, id_bytes.data(), id_off.data(), n_ids);
    long seen[8] = {0};
    for (uint32_t b = 0; b < NB; ++b) {
        const std::vector<Req> q = sequence(b);
        if (answers[b].size() != q.size()) { fprintf(stderr, "barrier %u: %zu of %zu requests answered\n", b, answers[b].size(), q.size()); ++sh.failures; continue; }
        for (size_t k = 0; k < q.size(); ++k) {
            int32_t want = 0;
            if (q[k].fork) {
                const uint32_t cnt = COUNT;
                const uint64_t dl = T0 + 3600000;
                djo_fork_batch(o, 1, &b, &cnt, &dl, &want);
            } else {
                uint8_t msg[MSG];
                message_of(msg, b, q[k].position, q[k].salt);
                const uint32_t len = MSG, mw = 0;
                djo_join_batch(o, 1, &b, &q[k].position, &len, &mw, msg, MSG, T0 + 5, &want);
                if (want >= 0 && want < 8) ++seen[want];
            }
            if (answers[b][k] != want) {
                fprintf(stderr, "barrier %u request %zu (%s): answered %d, one store answers %d\n", b, k, q[k].fork ? "fork" : "join", answers[b][k], want);
                ++sh.failures;
                break;
            }
        }
    }
    {   // the shared barrier on the one store: all positions, in any order
        const uint32_t cnt = (uint32_t)(N * T);
        const uint64_t dl = T0 + 3600000;
        int32_t st = 0;
        djo_fork_batch(o, 1, &SHARED, &cnt, &dl, &st);
        for (uint32_t p = 0; p < cnt; ++p) {
            uint8_t msg[MSG];
            message_of(msg, SHARED, p, 0);
            const uint32_t len = MSG, mw = 0;
            djo_join_batch(o, 1, &SHARED, &p, &len, &mw, msg, MSG, T0 + 5, &st);
        }
    }
    if (sh.shared_complete.load() != 1 || sh.shared_waiting.load() != (long)N * T - 1) {
        fprintf(stderr, "the shared barrier: %ld COMPLETE and %ld WAITING answers (want 1 and %d)\n", sh.shared_complete.load(), sh.shared_waiting.load(), N * T - 1);
        ++sh.failures;
    }
    std::vector<djo_cont> oc(NB + 64);
    const long want_popped = (long)djo_pop_due(o, T0 + 1000, NB + 64, oc.data());
    if (want_popped != sh.popped.load()) { fprintf(stderr, "the engines popped %ld barriers, one store pops %ld\n", sh.popped.load(), want_popped); ++sh.failures; }
    if (!seen[DJ_JOIN_WAITING] || !seen[DJ_JOIN_COMPLETE] || !seen[DJ_JOIN_NOT_FOUND] || !seen[DJ_JOIN_BAD_POSITION]) { fprintf(stderr, "the streams lack one of the four answers\n"); ++sh.failures; }
    djo_destroy(o);
    printf("ranks %d x %d request threads, %u barriers: joins answered waiting %ld complete %ld 404 %ld 400 %ld, %ld barriers due at the end, %s\n", N, T, NB,
           seen[1], seen[2], seen[3], seen[4], want_popped, sh.failures.load() ? "FAILED" : "every caller got one store's answer");
    return sh.failures.load() ? 1 : 0;
}
