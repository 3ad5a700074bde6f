// router_ranks.cpp — TEST INFRASTRUCTURE.  The router behind the C ABI (dj_router_*, include/dj_engine.h) with several
// ranks: one thread per rank, each with an engine of its own on GPU 0, connected by the stand-in for librccl in this
// directory (run with this directory on LD_LIBRARY_PATH; see rccl_standin.cpp for why).
//
// Every rank routes B batches of n records with their messages; a share of all records names one hot barrier, so its
// owner's incoming rows exceed max_recv and every batch takes several rounds.  Checked per rank and batch: the pieces,
// put one behind the other, are exactly the records this rank owns (dj_owner_of): from its own batch first (they never
// cross the exchange), then from rank 0's batch in array order, rank 1's, ..., each with its message; and the answers to them (one word per record, here a function of the
// record) come back to the rank that submitted the record, in its own submission order.  Batches are started ahead (two in flight) and every other batch's
// first piece is sent ahead of its use (dj_route_advance), as a pipelined host would.
//
//   usage: router_ranks <ranks> <records per rank and batch> <max_recv> <hot percent> [batches]
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "dj_engine.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static const uint32_t PITCH = 64;

static uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// record i of rank r's batch b; the position / len fields carry (i, r) so that a received row names its origin
static dj_join_rec make_rec(int r, int b, uint32_t i, int hot_percent) {
    dj_join_rec j;
    memset(&j, 0, sizeof j);
    const uint64_t h = mix(((uint64_t)r << 40) ^ ((uint64_t)b << 32) ^ i);
    const bool hot = (int)(h % 100) < hot_percent;
    j.fp_lo = hot ? 0x1234567812345678ull : (mix(h) | 2);
    j.fp_hi = hot ? 0x8765432187654321ull : mix(h ^ 0x55);
    j.position = i;
    j.len = (uint32_t)r;
    j.max_wait_s = (uint32_t)b;
    return j;
}

// what the stand-in for the engine call answers to record i of rank r's batch b
static int32_t answer_of(const dj_join_rec& j) { return (int32_t)(mix(((uint64_t)j.len << 40) ^ ((uint64_t)j.max_wait_s << 32) ^ j.position) & 0x7FFFFFFF); }

static void make_msg(uint8_t* row, int r, int b, uint32_t i) {
    uint64_t x = mix(((uint64_t)r << 48) ^ ((uint64_t)b << 40) ^ i ^ 0xABCDEF);
    for (uint32_t k = 0; k < PITCH; k += 8) { x = mix(x); memcpy(row + k, &x, 8); }
}

struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    uint8_t id[128];
    std::atomic<int> failures{0};
};

static void barrier(Shared& s, int n) {
    std::unique_lock<std::mutex> lk(s.mu);
    const long my = s.gen;
    if (++s.arrived == n) { s.arrived = 0; ++s.gen; s.cv.notify_all(); }
    else s.cv.wait(lk, [&] { return s.gen != my; });
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4;
    const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 20000;
    const uint32_t max_recv = argc > 3 ? (uint32_t)atoi(argv[3]) : 8192;
    const int hot = argc > 4 ? atoi(argv[4]) : 40;
    const int B = argc > 5 ? atoi(argv[5]) : 4;
    Shared sh;
    std::vector<std::thread> th;
    std::vector<uint32_t> rounds_seen(N, 0);
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] {
            HIP_OK(hipSetDevice(0));
            dj_config ec;
            memset(&ec, 0, sizeof ec);
            ec.struct_size = sizeof ec;
            ec.capacity_barriers = 1024;
            ec.arena_bytes = 4u << 20;
            ec.max_count = 8;
            ec.payload_stride = PITCH;
            ec.max_batch = n > 4096 ? n : 4096;
            ec.check_publish_delay_ms = 10000;
            ec.max_retry_publish = 3;
            ec.max_retry_timeout_publish = 3;
            dj_engine* e = nullptr;
            if (dj_create(&ec, &e) != DJ_OK) { fprintf(stderr, "rank %d: create: %s\n", r, dj_last_error(nullptr)); exit(2); }
            if (r == 0 && dj_route_unique_id(sh.id) != DJ_OK) { fprintf(stderr, "id: %s\n", dj_router_last_error(nullptr)); exit(2); }
            barrier(sh, N);
            dj_router* rt = nullptr;
            if (dj_router_create(e, (uint32_t)N, (uint32_t)r, sh.id, n, max_recv, PITCH, 3, &rt) != DJ_OK) {
                fprintf(stderr, "rank %d: router: %s\n", r, dj_router_last_error(nullptr));
                exit(2);
            }
            // the batches of this rank, on the device
            std::vector<dj_join_rec*> d_recs(B);
            std::vector<uint8_t*> d_pay(B);
            std::vector<dj_join_rec> h(n);
            std::vector<uint8_t> hp((size_t)n * PITCH);
            for (int b = 0; b < B; ++b) {
                for (uint32_t i = 0; i < n; ++i) { h[i] = make_rec(r, b, i, hot); make_msg(&hp[(size_t)i * PITCH], r, b, i); }
                HIP_OK(hipMalloc((void**)&d_recs[b], (size_t)n * sizeof(dj_join_rec)));
                HIP_OK(hipMalloc((void**)&d_pay[b], (size_t)n * PITCH));
                HIP_OK(hipMemcpy(d_recs[b], h.data(), (size_t)n * sizeof(dj_join_rec), hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(d_pay[b], hp.data(), (size_t)n * PITCH, hipMemcpyHostToDevice));
            }
            int32_t* d_answers = nullptr;
            HIP_OK(hipMalloc((void**)&d_answers, (size_t)n * 4));
            HIP_OK(hipMemset(d_answers, 0xFF, (size_t)n * 4));
            std::vector<int> handle(B, -1);
            auto start = [&](int b) {
                if (dj_route_start(rt, n, d_recs[b], d_pay[b], &handle[b]) != DJ_OK) {
                    fprintf(stderr, "rank %d: start %d: %s\n", r, b, dj_router_last_error(rt));
                    exit(2);
                }
            };
            start(0);
            for (int b = 0; b < B; ++b) {
                if (b % 2 == 0 && dj_route_advance(rt, handle[b]) != DJ_OK) {     // before the next start, as the header says
                    fprintf(stderr, "rank %d: advance: %s\n", r, dj_router_last_error(rt));
                    exit(2);
                }
                if (b + 1 < B) start(b + 1);                 // the next batch's partition and sizes are under way
                std::vector<dj_join_rec> got;
                std::vector<uint8_t> got_pay;
                int more = 1;
                uint32_t rounds = 0;
                while (more) {
                    const dj_join_rec* pr = nullptr;
                    const uint8_t* pp = nullptr;
                    uint32_t pn = 0;
                    if (dj_route_next(rt, handle[b], &pr, &pp, &pn, &more) != DJ_OK) {
                        fprintf(stderr, "rank %d: next: %s\n", r, dj_router_last_error(rt));
                        exit(2);
                    }
                    if (pn > max_recv) { fprintf(stderr, "rank %d: a piece of %u rows exceeds max_recv\n", r, pn); ++sh.failures; }
                    // the engine's stream waits for the piece; stand in for the engine call that would read it
                    HIP_OK(hipStreamSynchronize((hipStream_t)dj_stream(e)));
                    const size_t at = got.size();
                    got.resize(at + pn);
                    got_pay.resize((at + pn) * (size_t)PITCH);
                    if (pn) {
                        HIP_OK(hipMemcpy(&got[at], pr, (size_t)pn * sizeof(dj_join_rec), hipMemcpyDeviceToHost));
                        HIP_OK(hipMemcpy(&got_pay[at * PITCH], pp, (size_t)pn * PITCH, hipMemcpyDeviceToHost));
                    }
                    // the way back: the "engine call" writes one answer per received record, the router returns them
                    int32_t* d_st = nullptr;
                    if (dj_route_status_buffer(rt, handle[b], &d_st) != DJ_OK) { fprintf(stderr, "rank %d: status buffer: %s\n", r, dj_router_last_error(rt)); exit(2); }
                    std::vector<int32_t> ans(pn);
                    for (uint32_t k = 0; k < pn; ++k) ans[k] = answer_of(got[at + k]);
                    if (pn) HIP_OK(hipMemcpyAsync(d_st, ans.data(), (size_t)pn * 4, hipMemcpyHostToDevice, (hipStream_t)dj_stream(e)));
                    HIP_OK(hipStreamSynchronize((hipStream_t)dj_stream(e)));      // (ans lives on this thread's stack frame)
                    if (dj_route_return(rt, handle[b]) != DJ_OK) { fprintf(stderr, "rank %d: return: %s\n", r, dj_router_last_error(rt)); exit(2); }
                    ++rounds;
                }
                {   // every record of this rank's own batch has its answer, in the order the batch was submitted
                    if (dj_route_statuses(rt, handle[b], d_answers, nullptr) != DJ_OK) { fprintf(stderr, "rank %d: statuses: %s\n", r, dj_router_last_error(rt)); exit(2); }
                    std::vector<int32_t> back(n);
                    HIP_OK(hipMemcpy(back.data(), d_answers, (size_t)n * 4, hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < n; ++i)
                        if (back[i] != answer_of(make_rec(r, b, i, hot))) {
                            fprintf(stderr, "rank %d batch %d: the answer to record %u is %d, not %d\n", r, b, i, back[i], answer_of(make_rec(r, b, i, hot)));
                            ++sh.failures;
                            break;
                        }
                    HIP_OK(hipMemset(d_answers, 0xFF, (size_t)n * 4));
                }
                if (dj_route_release(rt, handle[b]) != DJ_OK) { fprintf(stderr, "rank %d: release: %s\n", r, dj_router_last_error(rt)); exit(2); }
                if (rounds > rounds_seen[r]) rounds_seen[r] = rounds;
                // what this rank must have received: source ranks in order, each one's records in array order
                size_t k = 0;
                bool ok = true;
                uint8_t row[PITCH];
                for (int q = 0; q < N && ok; ++q) {
                    const int s = q == 0 ? r : (q <= r ? q - 1 : q);       // this rank's own records first, then the others in rank order
                    for (uint32_t i = 0; i < n; ++i) {
                        const dj_join_rec want = make_rec(s, b, i, hot);
                        if ((int)dj_owner_of(want.fp_lo, want.fp_hi, (uint32_t)N) != r) continue;
                        make_msg(row, s, b, i);
                        if (k >= got.size() || memcmp(&got[k], &want, sizeof want) != 0 || memcmp(&got_pay[k * PITCH], row, PITCH) != 0) {
                            fprintf(stderr, "rank %d batch %d: row %zu differs (want record %u of rank %d; got %zu rows)\n", r, b, k, i, s, got.size());
                            ok = false;
                            break;
                        }
                        ++k;
                    }
                }
                if (ok && k != got.size()) { fprintf(stderr, "rank %d batch %d: %zu rows too many\n", r, b, got.size() - k); ok = false; }
                if (!ok) ++sh.failures;
            }
            barrier(sh, N);
            dj_router_destroy(rt);
            for (int b = 0; b < B; ++b) { (void)hipFree(d_recs[b]); (void)hipFree(d_pay[b]); }
            (void)hipFree(d_answers);
            dj_destroy(e);
        });
    for (auto& t : th) t.join();
    uint32_t most = 0;
    for (uint32_t x : rounds_seen) if (x > most) most = x;
    for (uint32_t x : rounds_seen)
        if (x != most) { fprintf(stderr, "ranks took different numbers of rounds\n"); ++sh.failures; }
    printf("ranks %d, %u records per rank and batch, %d batches, max_recv %u, hot %d %%: %u rounds per batch at most, %s\n", N, n, B,
           max_recv, hot, most, sh.failures.load() ? "FAILED" : "every rank received its records in order and every record's answer came back");
    return sh.failures.load() ? 1 : 0;
}
