// router_answers.cpp — TEST INFRASTRUCTURE.  Routed mode end to end behind the C ABI: several ranks (one thread each,
// each with its own engine on GPU 0, connected by the stand-in for librccl of this directory) fork and join through
// dj_router_*, the owner's engine answers every record (dj_fork_batch / dj_join_batch on the piece), and the answers
// return to the rank that submitted the record (dj_route_status_buffer / dj_route_return / dj_route_statuses).
//
// Checked against the CPU oracle (oracle/dj_oracle.h, the checker) on ONE store: the concatenated stream — batch by
// batch, and for every owner its own rank's records first, then the other ranks' (the router's order) — must give, for every rank, exactly
// the per-record statuses that came home to it, in its own submission order: WAITING / COMPLETE (core.clj:55,58),
// 404 for ids nobody forked and 400 for positions beyond the count (redis.clj:152-153).  A hot barrier makes its
// owner take every batch in several rounds.  At the end every engine pops what is due; together they must pop as many
// barriers as the one store.
//
//   usage: router_answers <ranks> <barriers per rank> <events per rank and batch> <max_recv> <hot percent> [batches]
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

#include "dj_engine.h"
#include "dj_oracle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define DJ_OK_(x, who) do { if ((x) != DJ_OK) { fprintf(stderr, "rank %d: %s: %s / %s\n", r, #x, dj_router_last_error(who), dj_last_error(e)); exit(2); } } while (0)

static const uint32_t PITCH = 64, COUNT = 6, GHOSTS = 5;
static const uint64_t T0 = 1700000000000ull;

static uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Event { uint32_t id, position; };

// event i of rank r's batch b over `total` forked barriers (index total = the hot one, behind it the ghosts)
static Event make_event(int r, int b, uint32_t i, uint32_t total, int hot_percent) {
    const uint64_t h = mix(((uint64_t)r << 40) ^ ((uint64_t)b << 32) ^ i);
    Event ev;
    const uint32_t roll = (uint32_t)(h % 100);
    if ((int)roll < hot_percent) {
        ev.id = total;                                        // the hot barrier: positions 0..COUNT-2, it never completes
        ev.position = (uint32_t)((h >> 8) % (COUNT - 1));
    } else if (roll >= 98) {
        ev.id = total + 1 + (uint32_t)((h >> 8) % GHOSTS);    // nobody forked it: 404
        ev.position = (uint32_t)((h >> 20) % COUNT);
    } else {
        ev.id = (uint32_t)((h >> 8) % total);
        ev.position = (uint32_t)((h >> 40) % (COUNT + 1));    // position COUNT: 400
    }
    return ev;
}

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
    std::atomic<long> popped{0};
};

static void barrier(Shared& s, int n) {
    std::unique_lock<std::mutex> lk(s.mu);
    const long my = s.gen;
    if (++s.arrived == n) { s.arrived = 0; ++s.gen; s.cv.notify_all(); }
    else s.cv.wait(lk, [&] { return s.gen != my; });
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2;
    const uint32_t NB = argc > 2 ? (uint32_t)atoi(argv[2]) : 2000;
    const uint32_t n = argc > 3 ? (uint32_t)atoi(argv[3]) : 8000;
    const uint32_t max_recv = argc > 4 ? (uint32_t)atoi(argv[4]) : 4096;
    const int hot = argc > 5 ? atoi(argv[5]) : 40;
    const int B = argc > 6 ? atoi(argv[6]) : 4;
    const uint32_t total = (uint32_t)N * NB, n_ids = total + 1 + GHOSTS;
    // ids and fingerprints
    std::vector<std::string> ids(n_ids);
    std::vector<uint64_t> fp(2 * (size_t)n_ids);
    std::vector<uint8_t> id_bytes;
    std::vector<uint64_t> id_off(n_ids + 1, 0);
    for (uint32_t k = 0; k < n_ids; ++k) {
        char buf[64];
        snprintf(buf, sizeof buf, "j%x~t%x", k, k);
        ids[k] = buf;
        dj_fingerprint(buf, strlen(buf), &fp[2 * k], &fp[2 * k + 1]);
        id_bytes.insert(id_bytes.end(), buf, buf + strlen(buf));
        id_off[k + 1] = id_bytes.size();
    }
    Shared sh;
    std::vector<std::vector<std::vector<int32_t>>> answers(N, std::vector<std::vector<int32_t>>(B));
    std::vector<uint32_t> rounds_seen(N, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] {
            HIP_OK(hipSetDevice(0));
            dj_config ec;
            memset(&ec, 0, sizeof ec);
            ec.struct_size = sizeof ec;
            ec.capacity_barriers = total + 64;
            ec.arena_bytes = (uint64_t)(total + 64) * 1024 + (4u << 20);
            ec.max_count = 8;
            ec.payload_stride = PITCH;
            // a piece holds at most max_recv rows, and never more than all ranks submit together
            uint32_t piece = (uint64_t)N * (n > NB + 1 ? n : NB + 1) < max_recv ? (uint32_t)N * (n > NB + 1 ? n : NB + 1) : max_recv;
            ec.max_batch = piece > 4096 ? piece : 4096;
            if (ec.max_batch < n) ec.max_batch = n;
            if (ec.max_batch < NB + 1) ec.max_batch = NB + 1;
            ec.check_publish_delay_ms = 10000;
            ec.max_retry_publish = 3;
            ec.max_retry_timeout_publish = 3;
            dj_engine* e = nullptr;
            if (dj_create(&ec, &e) != DJ_OK) { fprintf(stderr, "rank %d: create: %s\n", r, dj_last_error(nullptr)); exit(2); }
            if (r == 0 && dj_route_unique_id(sh.id) != DJ_OK) { fprintf(stderr, "id: %s\n", dj_router_last_error(nullptr)); exit(2); }
            barrier(sh, N);
            dj_router* rt = nullptr;
            if (dj_router_create(e, (uint32_t)N, (uint32_t)r, sh.id, n > NB + 1 ? n : NB + 1, max_recv, PITCH, 3, &rt) != DJ_OK) {
                fprintf(stderr, "rank %d: router: %s\n", r, dj_router_last_error(nullptr));
                exit(2);
            }
            hipStream_t es = (hipStream_t)dj_stream(e);
            int32_t* d_answers = nullptr;
            HIP_OK(hipMalloc((void**)&d_answers, (size_t)ec.max_batch * 4));

            // one routed batch: records (and messages) in, answers out; fork = true applies pieces with dj_fork_batch
            auto routed = [&](uint32_t m, const void* d_recs, const uint8_t* d_pay, bool fork, uint64_t now, std::vector<int32_t>* out) {
                int h = -1;
                DJ_OK_(dj_route_start(rt, m, (const dj_join_rec*)d_recs, d_pay, &h), rt);
                int more = 1;
                uint32_t rounds = 0;
                while (more) {
                    const dj_join_rec* pr = nullptr;
                    const uint8_t* pp = nullptr;
                    uint32_t pn = 0;
                    int32_t* d_st = nullptr;
                    DJ_OK_(dj_route_next(rt, h, &pr, &pp, &pn, &more), rt);
                    DJ_OK_(dj_route_status_buffer(rt, h, &d_st), rt);
                    if (fork) DJ_OK_(dj_fork_batch(e, pn, (const dj_fork_rec*)pr, d_st), rt);
                    else DJ_OK_(dj_join_batch(e, pn, pr, pp, PITCH, now, d_st), rt);
                    if (getenv("DJ_TEST_DEBUG") && fork) {          // what did the owner's engine answer, and to which record
                        HIP_OK(hipStreamSynchronize(es));
                        std::vector<int32_t> st(pn);
                        std::vector<dj_fork_rec> rr(pn);
                        if (pn) { HIP_OK(hipMemcpy(st.data(), d_st, (size_t)pn * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(rr.data(), pr, (size_t)pn * 32, hipMemcpyDeviceToHost)); }
                        for (uint32_t k = 0; k < pn; ++k)
                            if (st[k] != DJ_FORK_CREATED) { fprintf(stderr, "owner %d: fork row %u of %u answered %d: count %u fp %llx\n", r, k, pn, st[k], rr[k].count, (unsigned long long)rr[k].fp_lo); break; }
                    }
                    DJ_OK_(dj_route_return(rt, h), rt);
                    ++rounds;
                }
                HIP_OK(hipMemsetAsync(d_answers, 0xFF, (size_t)m * 4, es));
                HIP_OK(hipStreamSynchronize(es));
                DJ_OK_(dj_route_statuses(rt, h, d_answers, nullptr), rt);
                DJ_OK_(dj_route_release(rt, h), rt);
                out->resize(m);
                if (m) HIP_OK(hipMemcpy(out->data(), d_answers, (size_t)m * 4, hipMemcpyDeviceToHost));
                if (!fork && rounds > rounds_seen[r]) rounds_seen[r] = rounds;
            };

            // forks: this rank submits its own range of barriers (the last rank also the hot one)
            const uint32_t f0 = (uint32_t)r * NB, f1 = f0 + NB + (r == N - 1 ? 1u : 0u);
            std::vector<dj_fork_rec> fr(f1 - f0);
            for (uint32_t k = f0; k < f1; ++k) {
                dj_fork_rec& x = fr[k - f0];
                memset(&x, 0, sizeof x);
                x.fp_lo = fp[2 * k]; x.fp_hi = fp[2 * k + 1]; x.count = COUNT; x.deadline_ms = T0 + 3600000;
            }
            void* d_fr = nullptr;
            HIP_OK(hipMalloc(&d_fr, fr.size() * sizeof(dj_fork_rec)));
            HIP_OK(hipMemcpy(d_fr, fr.data(), fr.size() * sizeof(dj_fork_rec), hipMemcpyHostToDevice));
            std::vector<int32_t> fork_answers;
            routed((uint32_t)fr.size(), d_fr, nullptr, true, 0, &fork_answers);
            for (int32_t a : fork_answers)
                if (a != DJ_FORK_CREATED) { fprintf(stderr, "rank %d: a fork was answered %d\n", r, a); ++sh.failures; break; }
            barrier(sh, N);        // every barrier exists before the first join (as in the one store's stream below)

            std::vector<dj_join_rec> jr(n);
            std::vector<uint8_t> jp((size_t)n * PITCH);
            dj_join_rec* d_jr = nullptr;
            uint8_t* d_jp = nullptr;
            HIP_OK(hipMalloc((void**)&d_jr, (size_t)n * sizeof(dj_join_rec)));
            HIP_OK(hipMalloc((void**)&d_jp, (size_t)n * PITCH));
            for (int b = 0; b < B; ++b) {
                for (uint32_t i = 0; i < n; ++i) {
                    const Event ev = make_event(r, b, i, total, hot);
                    dj_join_rec& x = jr[i];
                    memset(&x, 0, sizeof x);
                    x.fp_lo = fp[2 * ev.id]; x.fp_hi = fp[2 * ev.id + 1]; x.position = ev.position; x.len = PITCH;
                    make_msg(&jp[(size_t)i * PITCH], r, b, i);
                }
                HIP_OK(hipMemcpy(d_jr, jr.data(), (size_t)n * sizeof(dj_join_rec), hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(d_jp, jp.data(), (size_t)n * PITCH, hipMemcpyHostToDevice));
                routed(n, d_jr, d_jp, false, T0 + 5 + (uint64_t)b, &answers[r][b]);
            }
            // what is due at the end, per engine
            dj_cont* d_conts = nullptr;
            HIP_OK(hipMalloc((void**)&d_conts, (size_t)(total + 64) * sizeof(dj_cont)));
            dj_sweep_result res;
            if (dj_pop_due(e, T0 + 1000, total + 64, d_conts, nullptr, &res) != DJ_OK) { fprintf(stderr, "rank %d: pop: %s\n", r, dj_last_error(e)); exit(2); }
            sh.popped += res.n_conts;
            barrier(sh, N);
            dj_router_destroy(rt);
            (void)hipFree(d_fr); (void)hipFree(d_jr); (void)hipFree(d_jp); (void)hipFree(d_answers); (void)hipFree(d_conts);
            dj_destroy(e);
        });
    for (auto& t : th) t.join();

    // the one store: the same requests, batch by batch, rank by rank
    djo* o = djo_create(10000, 3, 3, 8, 1u << 20);
    djo_set_ids(o, id_bytes.data(), id_off.data(), n_ids);
    {
        std::vector<uint32_t> idx(total + 1), cnt(total + 1, COUNT);
        std::vector<uint64_t> dl(total + 1, T0 + 3600000);
        std::vector<int32_t> st(total + 1);
        for (uint32_t k = 0; k <= total; ++k) idx[k] = k;
        djo_fork_batch(o, total + 1, idx.data(), cnt.data(), dl.data(), st.data());
    }
    // batch by batch; inside a batch owner by owner (barriers of different owners do not interact), and for an owner its
    // own rank's records first, then the other ranks' in rank order: the order in which the router delivers them
    long seen[8] = {0};
    for (int b = 0; b < B; ++b) {
        std::vector<std::vector<int32_t>> want(N, std::vector<int32_t>(n, 0));
        for (int owner = 0; owner < N; ++owner)
            for (int q = 0; q < N; ++q) {
                const int r = q == 0 ? owner : (q <= owner ? q - 1 : q);
                std::vector<uint32_t> idx, pos, len, mw, which;
                std::vector<uint8_t> pay;
                for (uint32_t i = 0; i < n; ++i) {
                    const Event ev = make_event(r, b, i, total, hot);
                    if ((int)dj_owner_of(fp[2 * ev.id], fp[2 * ev.id + 1], (uint32_t)N) != owner) continue;
                    idx.push_back(ev.id); pos.push_back(ev.position); len.push_back(PITCH); mw.push_back(0); which.push_back(i);
                    pay.resize(pay.size() + PITCH);
                    make_msg(&pay[pay.size() - PITCH], r, b, i);
                }
                std::vector<int32_t> st(idx.size());
                djo_join_batch(o, (uint32_t)idx.size(), idx.data(), pos.data(), len.data(), mw.data(), pay.data(), PITCH, T0 + 5 + (uint64_t)b, st.data());
                for (size_t k = 0; k < st.size(); ++k) want[r][which[k]] = st[k];
            }
        for (int r = 0; r < N; ++r)
            for (uint32_t i = 0; i < n; ++i) {
                if (want[r][i] >= 0 && want[r][i] < 8) ++seen[want[r][i]];
                if (answers[r][b].size() != n || answers[r][b][i] != want[r][i]) {
                    fprintf(stderr, "rank %d batch %d: record %u was answered %d, one store answers %d\n", r, b, i,
                            answers[r][b].size() == n ? answers[r][b][i] : -999, want[r][i]);
                    ++sh.failures;
                    break;
                }
            }
    }
    std::vector<djo_cont> oc(total + 64);
    const long want_popped = (long)djo_pop_due(o, T0 + 1000, total + 64, oc.data());
    if (want_popped != sh.popped.load()) { fprintf(stderr, "the engines popped %ld barriers, one store pops %ld\n", sh.popped.load(), want_popped); ++sh.failures; }
    if (!seen[DJ_JOIN_WAITING] || !seen[DJ_JOIN_COMPLETE] || !seen[DJ_JOIN_NOT_FOUND] || !seen[DJ_JOIN_BAD_POSITION]) {
        fprintf(stderr, "the stream lacks one of the four answers\n");
        ++sh.failures;
    }
    djo_destroy(o);
    uint32_t most = 0;
    for (uint32_t x : rounds_seen) if (x > most) most = x;
    for (uint32_t x : rounds_seen)
        if (x != most) { fprintf(stderr, "ranks took different numbers of rounds\n"); ++sh.failures; }
    printf("ranks %d, %u barriers and %u events per rank and batch, %d batches, max_recv %u, hot %d %%: answers waiting %ld complete %ld 404 %ld 400 %ld, "
           "%ld barriers due at the end, %u rounds per batch at most, %s\n", N, NB, n, B, max_recv, hot, seen[1], seen[2], seen[3], seen[4], want_popped, most,
           sh.failures.load() ? "FAILED" : "every rank got one store's answers");
    return sh.failures.load() ? 1 : 0;
}
