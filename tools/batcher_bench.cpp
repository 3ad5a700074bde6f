// Requests per second through the per-request front (include/dj_batcher.h), host buffers in,
// statuses out: the PCIe-inclusive rate of the engine behind a server's request threads.
//   T producer threads; each owns barriers of count C and sends fork, then C joins of P bytes.
//   async mode: producers never wait for answers (a completion counter ends the run);
//   sync mode:  every request waits for its answer (one request in flight per thread).
//   usage: batcher_bench [threads [barriers per thread [mode [linger_us]]]]
//   open mode (2): all forks first, then every thread streams its joins without waiting: the rate
//                  the staging path itself sustains (mutex + copy into pinned memory + H2D + kernels).
// Build (from the repo root):
//   hipcc -O2 -std=c++17 -Iinclude -o tools/_build/batcher_bench tools/batcher_bench.cpp \
//         -Ldistributed_join_amd -ldj_engine -Wl,-rpath,'$ORIGIN/../../distributed_join_amd' -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "dj_batcher.h"

static std::atomic<uint64_t> g_done{0}, g_complete{0}, g_bad{0};

static void on_status(void*, int32_t st) {
    if (st == DJ_JOIN_COMPLETE) g_complete.fetch_add(1, std::memory_order_relaxed);
    else if (st != DJ_JOIN_WAITING && st != DJ_FORK_CREATED) g_bad.fetch_add(1, std::memory_order_relaxed);
    g_done.fetch_add(1, std::memory_order_relaxed);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t per_thread = argc > 2 ? (uint32_t)atoi(argv[2]) : 20000;   // barriers per thread
    const int sync_mode = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t linger_us = argc > 4 ? (uint32_t)atoi(argv[4]) : 100;      // 0: a window closes as soon as the drain thread is free
    const uint32_t C = 8, P = 64;
    dj_config ec;
    memset(&ec, 0, sizeof ec);
    ec.struct_size = sizeof ec;
    ec.capacity_barriers = (uint64_t)T * per_thread;
    ec.arena_bytes = ec.capacity_barriers * 704 + (64u << 20);
    ec.max_count = C;
    ec.payload_stride = P;
    ec.max_batch = 1u << 16;
    ec.check_publish_delay_ms = 10000;
    ec.max_retry_publish = 3;
    ec.max_retry_timeout_publish = 3;
    dj_engine* e = nullptr;
    if (dj_create(&ec, &e) != DJ_OK) { fprintf(stderr, "create: %s\n", dj_last_error(nullptr)); return 1; }
    dj_batcher_config bc = {sizeof bc, 1u << 16, linger_us, P};
    dj_batcher* b = nullptr;
    if (dj_batcher_create(e, &bc, &b) != DJ_OK) { fprintf(stderr, "batcher create failed\n"); return 1; }

    const uint64_t total = (uint64_t)T * per_thread * (C + 1);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    if (sync_mode == 2) {
        // phase 1 (not timed): every barrier forked, answers awaited
        for (int t = 0; t < T; ++t)
            th.emplace_back([=] {
                for (uint32_t k = 0; k < per_thread; ++k) {
                    const std::string id = "t" + std::to_string(t) + "~b" + std::to_string(k);
                    dj_fork_rec f;
                    memset(&f, 0, sizeof f);
                    dj_fingerprint(id.data(), id.size(), &f.fp_lo, &f.fp_hi);
                    f.count = C;
                    f.deadline_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                                        std::chrono::system_clock::now().time_since_epoch()).count() + 3600000;
                    dj_batcher_fork(b, &f, on_status, nullptr);
                }
            });
        for (auto& x : th) x.join();
        th.clear();
        while (g_done.load() < (uint64_t)T * per_thread) std::this_thread::sleep_for(std::chrono::microseconds(50));
        t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t)
            th.emplace_back([=] {
                uint8_t msg[P];
                memset(msg, t, P);
                for (uint32_t p = 0; p < C; ++p)
                    for (uint32_t k = 0; k < per_thread; ++k) {
                        const std::string id = "t" + std::to_string(t) + "~b" + std::to_string(k);
                        dj_join_rec j;
                        memset(&j, 0, sizeof j);
                        dj_fingerprint(id.data(), id.size(), &j.fp_lo, &j.fp_hi);
                        j.position = p; j.len = P;
                        dj_batcher_join(b, &j, msg, on_status, nullptr);
                    }
            });
        for (auto& x : th) x.join();
        th.clear();
    } else
    for (int t = 0; t < T; ++t)
        th.emplace_back([=] {
            uint8_t msg[P];
            memset(msg, t, P);
            for (uint32_t k = 0; k < per_thread; ++k) {
                const std::string id = "t" + std::to_string(t) + "~b" + std::to_string(k);
                dj_fork_rec f;
                memset(&f, 0, sizeof f);
                dj_fingerprint(id.data(), id.size(), &f.fp_lo, &f.fp_hi);
                f.count = C;
                f.deadline_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                                    std::chrono::system_clock::now().time_since_epoch()).count() + 3600000;
                int32_t st = 0;
                // the fork is always awaited: a client joins only after its fork was answered
                if (dj_batcher_fork_wait(b, &f, &st) != DJ_OK || st != DJ_FORK_CREATED) g_bad++;
                g_done++;
                for (uint32_t p = 0; p < C; ++p) {
                    dj_join_rec j;
                    memset(&j, 0, sizeof j);
                    j.fp_lo = f.fp_lo; j.fp_hi = f.fp_hi; j.position = p; j.len = P;
                    if (sync_mode) {
                        dj_batcher_join_wait(b, &j, msg, &st);
                        on_status(nullptr, st);
                    } else {
                        dj_batcher_join(b, &j, msg, on_status, nullptr);
                    }
                }
            }
        });
    for (auto& x : th) x.join();
    while (g_done.load() < total) std::this_thread::sleep_for(std::chrono::microseconds(50));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    dj_batcher_stats s;
    dj_batcher_get_stats(b, &s);
    const uint64_t timed = sync_mode == 2 ? (uint64_t)T * per_thread * C : total;
    printf("threads %d linger %u us mode %s: %llu requests in %.3f s = %.3f M requests/s (joins %.3f M/s); batches %llu, avg %.0f, largest %llu; "
           "complete %llu (want %llu), unexpected %llu\n",
           T, linger_us, sync_mode == 2 ? "open (joins only timed)" : sync_mode ? "sync" : "async-joins", (unsigned long long)timed, dt, timed / dt / 1e6,
           (double)T * per_thread * C / dt / 1e6, (unsigned long long)s.batches,
           s.batches ? (double)(s.forks + s.joins) / s.batches : 0.0, (unsigned long long)s.largest_batch,
           (unsigned long long)g_complete.load(), (unsigned long long)T * per_thread, (unsigned long long)g_bad.load());
    dj_batcher_destroy(b);
    dj_destroy(e);
    return g_bad.load() || g_complete.load() != (uint64_t)T * per_thread;
}
