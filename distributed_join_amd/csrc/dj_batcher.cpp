// dj_batcher.cpp — per-request front of the engine (include/dj_batcher.h).
//
// Producers (request threads) fill one of two staging sets under a mutex; the drain thread swaps
// the sets, ships the closed one to the device from pinned memory, applies it and answers.  The
// drain thread is the only caller of the engine.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "dj_batcher.h"

namespace {

struct Waiter {              // a caller blocked in *_wait / run
    int32_t status = 0;
    bool done = false;
};

struct Task {
    int (*fn)(void*, dj_engine*, uint64_t);
    void* user;
    Waiter* w;
};

struct Answer {
    dj_status_cb cb;
    void* user;
};

struct Side {
    dj_join_rec* jrecs = nullptr;   // pinned
    uint8_t* jpay = nullptr;        // pinned, max_batch * pitch
    dj_fork_rec* frecs = nullptr;   // pinned
    std::vector<Answer> jans, fans;
    uint32_t nj = 0, nf = 0;
    std::vector<Task> tasks;
};

uint64_t wall_clock_ms(void*) {
    using namespace std::chrono;
    return (uint64_t)duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
}

}  // namespace

struct dj_batcher {
    dj_engine* e = nullptr;
    dj_batcher_config cfg{};
    dj_config ecfg{};
    uint64_t pitch = 0;
    Side side[2];
    int fill = 0;                    // producers write side[fill]
    bool first_set = false;
    std::chrono::steady_clock::time_point first_arrival;
    std::mutex mu;
    std::condition_variable cv_work, cv_space, cv_done;
    std::thread th;
    bool stop = false;
    uint64_t (*clock_fn)(void*) = wall_clock_ms;
    void* clock_user = nullptr;
    // device side
    dj_join_rec* d_jrecs = nullptr;
    uint8_t* d_jpay = nullptr;
    dj_fork_rec* d_frecs = nullptr;
    int32_t* d_status = nullptr;
    int32_t* h_status = nullptr;     // pinned, 2 * max_batch (forks, joins)
    dj_batcher_stats stats{};
    int fatal = 0;                   // first engine error seen by the drain thread; answers carry it as status
};

namespace {

void wait_cb(void* user, int32_t status) {
    // runs on the drain thread with the batcher's mutex NOT held; the waiter sleeps on cv_done
    Waiter* w = static_cast<Waiter*>(user);
    w->status = status;
}

bool hip_ok(hipError_t r) { return r == hipSuccess; }

// one drain cycle on a closed side
void apply(dj_batcher* b, Side& s, uint64_t (*clock_fn)(void*), void* clock_user) {
    hipStream_t st = static_cast<hipStream_t>(dj_stream(b->e));
    const uint64_t now = clock_fn(clock_user);
    const uint32_t mb = b->cfg.max_batch;
    int rc = DJ_OK;
    if (s.nf && rc == DJ_OK) {
        if (!hip_ok(hipMemcpyAsync(b->d_frecs, s.frecs, (size_t)s.nf * sizeof(dj_fork_rec), hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
        if (rc == DJ_OK) rc = dj_fork_batch(b->e, s.nf, b->d_frecs, b->d_status);
        if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->h_status, b->d_status, (size_t)s.nf * 4, hipMemcpyDeviceToHost, st))) rc = DJ_E_HIP;
    }
    if (s.nj && rc == DJ_OK) {
        if (!hip_ok(hipMemcpyAsync(b->d_jrecs, s.jrecs, (size_t)s.nj * sizeof(dj_join_rec), hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
        if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->d_jpay, s.jpay, (size_t)s.nj * b->pitch, hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
        if (rc == DJ_OK) rc = dj_join_batch(b->e, s.nj, b->d_jrecs, b->d_jpay, b->pitch, now, b->d_status + mb);
        if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->h_status + mb, b->d_status + mb, (size_t)s.nj * 4, hipMemcpyDeviceToHost, st))) rc = DJ_E_HIP;
    }
    if (rc == DJ_OK && (s.nf || s.nj)) rc = dj_sync(b->e);
    if (rc != DJ_OK && !b->fatal) b->fatal = rc;
    // answers: forks first (they were applied first), then joins; an engine failure is reported as
    // the (negative) error code in place of a status
    for (uint32_t i = 0; i < s.nf; ++i) s.fans[i].cb(s.fans[i].user, rc == DJ_OK ? b->h_status[i] : rc);
    for (uint32_t i = 0; i < s.nj; ++i) s.jans[i].cb(s.jans[i].user, rc == DJ_OK ? b->h_status[mb + i] : rc);
    for (Task& t : s.tasks) {
        const int r = t.fn(t.user, b->e, now);
        t.w->status = r;
    }
    {
        std::lock_guard<std::mutex> g(b->mu);
        if (s.nf || s.nj) {
            ++b->stats.batches;
            b->stats.forks += s.nf;
            b->stats.joins += s.nj;
            const uint64_t big = s.nf > s.nj ? s.nf : s.nj;
            if (big > b->stats.largest_batch) b->stats.largest_batch = big;
        }
        b->stats.tasks += s.tasks.size();
        // every waiter of this cycle (wait_cb users and tasks) may go now
        for (uint32_t i = 0; i < s.nf; ++i)
            if (s.fans[i].cb == wait_cb) static_cast<Waiter*>(s.fans[i].user)->done = true;
        for (uint32_t i = 0; i < s.nj; ++i)
            if (s.jans[i].cb == wait_cb) static_cast<Waiter*>(s.jans[i].user)->done = true;
        for (Task& t : s.tasks) t.w->done = true;
        s.nf = s.nj = 0;
        s.tasks.clear();
    }
    b->cv_done.notify_all();
}

void drain_loop(dj_batcher* b) {
    (void)hipSetDevice(b->ecfg.device);
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        Side* s = &b->side[b->fill];
        b->cv_work.wait(lk, [&] { s = &b->side[b->fill]; return b->stop || s->nf || s->nj || !s->tasks.empty(); });
        if (!(s->nf || s->nj || !s->tasks.empty())) break;      // stop requested and nothing queued
        // linger for more requests, but never longer than max_wait_us after the first one; tasks do not linger
        if (s->tasks.empty() && !b->stop && b->cfg.max_wait_us) {
            const auto deadline = b->first_arrival + std::chrono::microseconds(b->cfg.max_wait_us);
            b->cv_work.wait_until(lk, deadline, [&] {
                return b->stop || s->nf >= b->cfg.max_batch || s->nj >= b->cfg.max_batch || !s->tasks.empty();
            });
        }
        const int cur = b->fill;
        b->fill ^= 1;
        b->first_set = false;
        uint64_t (*const clock_fn)(void*) = b->clock_fn;     // as set under the mutex; called without it
        void* const clock_user = b->clock_user;
        lk.unlock();
        b->cv_space.notify_all();
        apply(b, b->side[cur], clock_fn, clock_user);
        lk.lock();
    }
}

void note_first(dj_batcher* b) {
    if (!b->first_set) {
        b->first_set = true;
        b->first_arrival = std::chrono::steady_clock::now();
    }
}

}  // namespace

extern "C" int dj_batcher_create(dj_engine* e, const dj_batcher_config* cfg, dj_batcher** out) {
    if (!e || !cfg || !out || cfg->struct_size != sizeof(dj_batcher_config)) return DJ_E_INVALID;
    dj_config ec;
    if (dj_get_config(e, &ec) != DJ_OK) return DJ_E_INVALID;
    const uint32_t longest = ec.max_message_bytes > ec.payload_stride ? ec.max_message_bytes : ec.payload_stride;
    if (cfg->max_batch < 1 || cfg->max_batch > ec.max_batch || cfg->max_message > longest) return DJ_E_INVALID;
    dj_batcher* b = new (std::nothrow) dj_batcher();
    if (!b) return DJ_E_NOMEM;
    b->e = e;
    b->cfg = *cfg;
    b->ecfg = ec;
    b->pitch = ((uint64_t)(cfg->max_message ? cfg->max_message : 1) + 15) / 16 * 16;
    const size_t mb = cfg->max_batch;
    bool ok = hip_ok(hipSetDevice(ec.device));
    for (Side& s : b->side) {
        ok = ok && hip_ok(hipHostMalloc((void**)&s.jrecs, mb * sizeof(dj_join_rec), hipHostMallocDefault));
        ok = ok && hip_ok(hipHostMalloc((void**)&s.jpay, mb * b->pitch, hipHostMallocDefault));
        ok = ok && hip_ok(hipHostMalloc((void**)&s.frecs, mb * sizeof(dj_fork_rec), hipHostMallocDefault));
        s.jans.resize(mb);
        s.fans.resize(mb);
    }
    ok = ok && hip_ok(hipMalloc((void**)&b->d_jrecs, mb * sizeof(dj_join_rec)));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_jpay, mb * b->pitch));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_frecs, mb * sizeof(dj_fork_rec)));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_status, 2 * mb * 4));
    ok = ok && hip_ok(hipHostMalloc((void**)&b->h_status, 2 * mb * 4, hipHostMallocDefault));
    if (!ok) {
        dj_batcher_destroy(b);
        return DJ_E_NOMEM;
    }
    b->th = std::thread(drain_loop, b);
    *out = b;
    return DJ_OK;
}

extern "C" void dj_batcher_destroy(dj_batcher* b) {
    if (!b) return;
    if (b->th.joinable()) {
        {
            std::lock_guard<std::mutex> g(b->mu);
            b->stop = true;
        }
        b->cv_work.notify_all();
        b->th.join();
    }
    (void)hipSetDevice(b->ecfg.device);
    for (Side& s : b->side) {
        if (s.jrecs) (void)hipHostFree(s.jrecs);
        if (s.jpay) (void)hipHostFree(s.jpay);
        if (s.frecs) (void)hipHostFree(s.frecs);
    }
    if (b->d_jrecs) (void)hipFree(b->d_jrecs);
    if (b->d_jpay) (void)hipFree(b->d_jpay);
    if (b->d_frecs) (void)hipFree(b->d_frecs);
    if (b->d_status) (void)hipFree(b->d_status);
    if (b->h_status) (void)hipHostFree(b->h_status);
    delete b;
}

extern "C" int dj_batcher_set_clock(dj_batcher* b, uint64_t (*now_ms)(void*), void* user) {
    if (!b) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(b->mu);
    b->clock_fn = now_ms ? now_ms : wall_clock_ms;
    b->clock_user = user;
    return DJ_OK;
}

extern "C" int dj_batcher_join(dj_batcher* b, const dj_join_rec* rec, const void* message, dj_status_cb cb, void* user) {
    if (!b || !rec || !cb || (rec->len && !message)) return DJ_E_INVALID;
    if (rec->len > b->cfg.max_message) {          // would not fit the staging row; the engine's answer for it
        cb(user, DJ_JOIN_TOO_LONG);
        if (cb == wait_cb) {
            std::lock_guard<std::mutex> g(b->mu);
            static_cast<Waiter*>(user)->done = true;
        }
        return DJ_OK;
    }
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) return DJ_E_INVALID;
    b->cv_space.wait(lk, [&] { return b->side[b->fill].nj < b->cfg.max_batch; });
    Side& s = b->side[b->fill];
    const uint32_t i = s.nj++;
    s.jrecs[i] = *rec;
    if (rec->len) memcpy(s.jpay + (size_t)i * b->pitch, message, rec->len);
    s.jans[i] = Answer{cb, user};
    note_first(b);
    const bool wake = i == 0 || s.nj >= b->cfg.max_batch;
    lk.unlock();
    if (wake) b->cv_work.notify_one();
    return DJ_OK;
}

extern "C" int dj_batcher_fork(dj_batcher* b, const dj_fork_rec* rec, dj_status_cb cb, void* user) {
    if (!b || !rec || !cb) return DJ_E_INVALID;
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) return DJ_E_INVALID;
    b->cv_space.wait(lk, [&] { return b->side[b->fill].nf < b->cfg.max_batch; });
    Side& s = b->side[b->fill];
    const uint32_t i = s.nf++;
    s.frecs[i] = *rec;
    s.fans[i] = Answer{cb, user};
    note_first(b);
    const bool wake = i == 0 || s.nf >= b->cfg.max_batch;
    lk.unlock();
    if (wake) b->cv_work.notify_one();
    return DJ_OK;
}

static int wait_for(dj_batcher* b, Waiter& w, int32_t* status) {
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv_done.wait(lk, [&] { return w.done; });
    if (status) *status = w.status;
    return DJ_OK;
}

extern "C" int dj_batcher_join_wait(dj_batcher* b, const dj_join_rec* rec, const void* message, int32_t* status) {
    Waiter w;
    const int rc = dj_batcher_join(b, rec, message, wait_cb, &w);
    if (rc != DJ_OK) return rc;
    return wait_for(b, w, status);
}

extern "C" int dj_batcher_fork_wait(dj_batcher* b, const dj_fork_rec* rec, int32_t* status) {
    Waiter w;
    const int rc = dj_batcher_fork(b, rec, wait_cb, &w);
    if (rc != DJ_OK) return rc;
    return wait_for(b, w, status);
}

extern "C" int dj_batcher_run(dj_batcher* b, int (*fn)(void*, dj_engine*, uint64_t), void* user) {
    if (!b || !fn) return DJ_E_INVALID;
    Waiter w;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        if (b->stop) return DJ_E_INVALID;
        b->side[b->fill].tasks.push_back(Task{fn, user, &w});
        note_first(b);
    }
    b->cv_work.notify_one();
    int32_t r = 0;
    wait_for(b, w, &r);
    return r;
}

extern "C" int dj_batcher_get_stats(dj_batcher* b, dj_batcher_stats* out) {
    if (!b || !out) return DJ_E_INVALID;
    std::lock_guard<std::mutex> g(b->mu);
    *out = b->stats;
    return DJ_OK;
}
