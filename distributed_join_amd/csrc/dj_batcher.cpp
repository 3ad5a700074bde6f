// dj_batcher.cpp — per-request front of the engine (include/dj_batcher.h).
//
// Producers (request threads) stage their requests in LANES: a thread is given a lane on its first call
// and only ever takes that lane's (uncontended) lock, so staging costs no shared cache line.  A lane has
// two sides; the drain thread closes every lane's filling side at once (a window), gathers the closed
// sides into pinned buffers, ships them to the device, applies them and answers.  The drain thread is the
// only caller of the engine.
//
// Order.  Requests of one lane are applied in the order they were staged: a window is cut into phases
// (forks of phase p, then joins of phase p, then phase p + 1 ...) and a lane starts a new phase whenever it
// stages a fork of a barrier that has a join waiting in its current phase, so "join, then fork of the same
// id" keeps that order although the engine applies forks and joins as separate batches.  Requests of
// different lanes that were staged in the same window are concurrent requests: any interleaving is a valid
// serialisation.  A request staged after another one was ANSWERED is applied in a later window.
//
// Routed form (dj_batcher_create_routed): the batcher of one rank of a node whose barriers are sharded over several
// engines.  Its windows are exchanged through the rank's dj_router: every rank's drain thread runs the same CYCLES —
// agree (one word per rank: batches of forks, batches of joins, "I want to stop"), then that many routed fork batches,
// then that many routed join batches, each applied piece by piece on the owners' engines with the statuses returned
// to the submitting rank (dj_route_return / dj_route_statuses) and handed to the callers that wait for them.  A window
// has ONE phase there (forks, then joins): a fork that would have to follow a staged join of its barrier waits for the
// next window instead of cutting a phase, so that every rank makes the same number of exchanges per cycle.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "dj_batcher.h"

namespace {

constexpr int LANES = 64;

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

struct LaneSide {
    std::vector<dj_join_rec> jrecs;
    std::vector<uint64_t> joff;            // where each join's message starts in bytes
    std::vector<uint8_t> bytes;
    std::vector<Answer> jans;
    std::vector<dj_fork_rec> frecs;
    std::vector<Answer> fans;
    std::vector<uint32_t> cut_f, cut_j;    // forks / joins staged when each later phase began
    // low fingerprint halves of the joins of the current phase (open addressing, 0 = free): a fork that hits one
    // starts a new phase.  A false hit (two ids with the same low half) only costs a phase.
    std::vector<uint64_t> jset;
    std::vector<uint32_t> jset_cells;      // the cells in use, so that clearing costs what was inserted, not the table
    uint32_t jset_n = 0;

    bool empty() const { return jrecs.empty() && frecs.empty(); }
    void clear() {
        jrecs.clear(); joff.clear(); bytes.clear(); jans.clear(); frecs.clear(); fans.clear(); cut_f.clear(); cut_j.clear();
        clear_set();
    }
    void clear_set() {
        for (uint32_t c : jset_cells) jset[c] = 0;
        jset_cells.clear();
        jset_n = 0;
    }
    bool set_has(uint64_t k) const {
        if (!jset_n) return false;
        const size_t m = jset.size() - 1;
        for (size_t h = (k * 0x9E3779B97F4A7C15ull) >> 20 & m;; h = (h + 1) & m) {
            if (jset[h] == k) return true;
            if (jset[h] == 0) return false;
        }
    }
    void set_add(uint64_t k) {
        if (jset.size() < 2 * (size_t)(jset_n + 1)) {
            std::vector<uint64_t> old;
            old.swap(jset);
            jset.assign(old.empty() ? 64 : old.size() * 2, 0ull);
            jset_n = 0;
            jset_cells.clear();
            for (uint64_t v : old) if (v) set_add(v);
        }
        const size_t m = jset.size() - 1;
        for (size_t h = (k * 0x9E3779B97F4A7C15ull) >> 20 & m;; h = (h + 1) & m) {
            if (jset[h] == k) return;
            if (jset[h] == 0) { jset[h] = k; jset_cells.push_back((uint32_t)h); ++jset_n; return; }
        }
    }
};

struct alignas(128) Lane {
    std::mutex mu;
    std::condition_variable cv_space;
    LaneSide side[2];
    int fill = 0;
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
    Lane lanes[LANES];
    std::atomic<uint32_t> next_lane{0};
    std::atomic<bool> sleeping{false};
    // shared state (tasks, waiters, statistics, stop): off the staging path
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Task> tasks;
    std::thread th;
    bool stop = false;
    bool poked = false, lane_full = false;   // set by producers (under mu) when they wake the drain thread
    uint64_t (*clock_fn)(void*) = wall_clock_ms;
    void* clock_user = nullptr;
    // pinned staging of one engine call, and the device side
    dj_join_rec* h_jrecs = nullptr;
    uint8_t* h_jpay = nullptr;
    dj_fork_rec* h_frecs = nullptr;
    std::vector<Answer> ans;
    dj_join_rec* d_jrecs = nullptr;
    uint8_t* d_jpay = nullptr;
    dj_fork_rec* d_frecs = nullptr;
    int32_t* d_status = nullptr;
    int32_t* h_status = nullptr;     // pinned, 2 x max_batch (fork answers, join answers)
    dj_batcher_stats stats{};
    int fatal = 0;                   // first engine error seen by the drain thread; answers carry it as status
    dj_router* router = nullptr;     // routed form: this rank's router (d_status then holds the answers that came home)
    uint32_t world = 1;
};

namespace {

thread_local const dj_batcher* tl_owner = nullptr;
thread_local int tl_lane = 0;

Lane& my_lane(dj_batcher* b) {
    if (tl_owner != b) {
        tl_owner = b;
        tl_lane = (int)(b->next_lane.fetch_add(1, std::memory_order_relaxed) % LANES);
    }
    return b->lanes[tl_lane];
}

void wait_cb(void* user, int32_t status) {
    // runs on the drain thread with the batcher's mutex NOT held; the waiter sleeps on cv_done
    static_cast<Waiter*>(user)->status = status;
}

bool hip_ok(hipError_t r) { return r == hipSuccess; }

// Wake the drain thread if it sleeps (called after a lane went from empty to non-empty, or when a lane is full).
// The drain thread raises `sleeping` BEFORE it looks at the lanes (under their locks), so a producer that
// staged after that look finds the flag raised; one that staged before it is found by the look.
void poke(dj_batcher* b, bool full) {
    if (b->sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> g(b->mu);
        b->poked = true;
        if (full) b->lane_full = true;
        b->cv_work.notify_one();
    }
}

struct Window {
    LaneSide* side[LANES];
    int n = 0;
    uint64_t forks = 0, joins = 0;
    uint32_t phases = 1;
};

// rows [from, to) of one closed side's forks / joins in phase p
void phase_range(const std::vector<uint32_t>& cuts, size_t total, uint32_t p, uint32_t* from, uint32_t* to) {
    *from = p == 0 ? 0 : (p - 1 < cuts.size() ? cuts[p - 1] : (uint32_t)total);
    *to = p < cuts.size() ? cuts[p] : (uint32_t)total;
}

void answer_all(dj_batcher* b, uint32_t n, int rc) {
    for (uint32_t i = 0; i < n; ++i) b->ans[i].cb(b->ans[i].user, rc == DJ_OK ? b->h_status[i] : rc);
}

// Batches of at most this many records are read by the kernels where the drain thread staged them (pinned host memory
// that the device maps) and answered into pinned memory as well: three small copies per batch cost more than a few
// hundred kilobytes read over the host link.  Larger batches go through device buffers as before.
constexpr uint32_t ZERO_COPY_MAX = 2048;

// one window: phases of (fork batch, join batch), each in engine calls of at most max_batch records.  The forks of a
// phase are answered together with its joins, after ONE wait for the stream (fork statuses in h_status[0, mb), join
// statuses in h_status[mb, 2 mb)).
void apply(dj_batcher* b, Window& w, uint64_t now, std::vector<Waiter*>& waiters) {
    hipStream_t st = static_cast<hipStream_t>(dj_stream(b->e));
    const uint32_t mb = b->cfg.max_batch;
    int rc = b->fatal;
    uint64_t largest = 0, calls = 0;
    uint32_t forks_pending = 0;            // fork records launched whose answers wait for the next stream wait
    auto wait_and_answer = [&](uint32_t nf, uint32_t nj) {
        if (rc == DJ_OK) rc = dj_sync(b->e);
        for (uint32_t i = 0; i < nf; ++i) b->ans[i].cb(b->ans[i].user, rc == DJ_OK ? b->h_status[i] : rc);
        for (uint32_t i = 0; i < nj; ++i) b->ans[mb + i].cb(b->ans[mb + i].user, rc == DJ_OK ? b->h_status[mb + i] : rc);
        forks_pending = 0;
    };
    auto flush_forks = [&](uint32_t n, bool joins_follow) {
        if (!n) return;
        if (n <= ZERO_COPY_MAX) {
            if (rc == DJ_OK) rc = dj_fork_batch(b->e, n, b->h_frecs, b->h_status);
        } else {
            if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->d_frecs, b->h_frecs, (size_t)n * sizeof(dj_fork_rec), hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
            if (rc == DJ_OK) rc = dj_fork_batch(b->e, n, b->d_frecs, b->d_status);
            if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->h_status, b->d_status, (size_t)n * 4, hipMemcpyDeviceToHost, st))) rc = DJ_E_HIP;
        }
        forks_pending = n;
        if (!joins_follow) wait_and_answer(n, 0);
        if (n > largest) largest = n;
        ++calls;
    };
    auto flush_joins = [&](uint32_t n) {
        if (!n) return;
        if (n <= ZERO_COPY_MAX) {
            if (rc == DJ_OK) rc = dj_join_batch(b->e, n, b->h_jrecs, b->h_jpay, b->pitch, now, b->h_status + mb);
        } else {
            if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->d_jrecs, b->h_jrecs, (size_t)n * sizeof(dj_join_rec), hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
            if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->d_jpay, b->h_jpay, (size_t)n * b->pitch, hipMemcpyHostToDevice, st))) rc = DJ_E_HIP;
            if (rc == DJ_OK) rc = dj_join_batch(b->e, n, b->d_jrecs, b->d_jpay, b->pitch, now, b->d_status);
            if (rc == DJ_OK && !hip_ok(hipMemcpyAsync(b->h_status + mb, b->d_status, (size_t)n * 4, hipMemcpyDeviceToHost, st))) rc = DJ_E_HIP;
        }
        wait_and_answer(forks_pending, n);
        if (n > largest) largest = n;
        ++calls;
    };
    for (uint32_t p = 0; p < w.phases; ++p) {
        // does this phase hold joins?  (then its last fork batch is answered with them)
        bool phase_joins = false;
        for (int k = 0; k < w.n && !phase_joins; ++k) {
            uint32_t from, to;
            phase_range(w.side[k]->cut_j, w.side[k]->jrecs.size(), p, &from, &to);
            phase_joins = to > from;
        }
        uint32_t n = 0;
        for (int k = 0; k < w.n; ++k) {
            LaneSide& s = *w.side[k];
            uint32_t from, to;
            phase_range(s.cut_f, s.frecs.size(), p, &from, &to);
            for (uint32_t i = from; i < to; ++i) {
                b->h_frecs[n] = s.frecs[i];
                b->ans[n] = s.fans[i];
                if (s.fans[i].cb == wait_cb) waiters.push_back(static_cast<Waiter*>(s.fans[i].user));
                if (++n == mb) { flush_forks(n, false); n = 0; }
            }
        }
        flush_forks(n, phase_joins);
        n = 0;
        for (int k = 0; k < w.n; ++k) {
            LaneSide& s = *w.side[k];
            uint32_t from, to;
            phase_range(s.cut_j, s.jrecs.size(), p, &from, &to);
            for (uint32_t i = from; i < to; ++i) {
                b->h_jrecs[n] = s.jrecs[i];
                if (s.jrecs[i].len) memcpy(b->h_jpay + (size_t)n * b->pitch, s.bytes.data() + s.joff[i], s.jrecs[i].len);
                b->ans[mb + n] = s.jans[i];
                if (s.jans[i].cb == wait_cb) waiters.push_back(static_cast<Waiter*>(s.jans[i].user));
                if (++n == mb) { flush_joins(n); n = 0; }
            }
        }
        flush_joins(n);
        if (forks_pending) wait_and_answer(forks_pending, 0);      // (cannot be: a phase with joins flushes them last)
    }
    if (rc != DJ_OK && !b->fatal) b->fatal = rc;
    std::lock_guard<std::mutex> g(b->mu);
    if (w.forks || w.joins) {
        b->stats.batches += calls;
        b->stats.forks += w.forks;
        b->stats.joins += w.joins;
        if (largest > b->stats.largest_batch) b->stats.largest_batch = largest;
    }
}

void drain_loop(dj_batcher* b) {
    (void)hipSetDevice(b->ecfg.device);
    std::vector<Waiter*> waiters;
    std::vector<Task> tasks;
    for (;;) {
        // ---- wait for work ----------------------------------------------------------------------------------
        auto any_staged = [&] {
            for (Lane& l : b->lanes) {
                std::lock_guard<std::mutex> g(l.mu);
                if (!l.side[l.fill].empty()) return true;
            }
            return false;
        };
        bool staged = any_staged();
        {
            std::unique_lock<std::mutex> lk(b->mu);
            if (!staged && b->tasks.empty()) {
                if (b->stop) return;                  // stop requested and nothing queued
                b->sleeping.store(true, std::memory_order_seq_cst);
                lk.unlock();
                staged = any_staged();
                lk.lock();
                if (!staged) b->cv_work.wait_for(lk, std::chrono::milliseconds(20), [&] { return b->poked || !b->tasks.empty() || b->stop; });
                b->sleeping.store(false, std::memory_order_seq_cst);
                b->poked = b->lane_full = false;
                continue;                             // look again
            }
            // linger for more requests, but never longer than max_wait_us after the window's first one was
            // noticed; tasks do not linger.  A producer that finds its lane full ends the linger.
            if (staged && b->tasks.empty() && !b->stop && b->cfg.max_wait_us) {
                b->sleeping.store(true, std::memory_order_seq_cst);
                b->cv_work.wait_for(lk, std::chrono::microseconds(b->cfg.max_wait_us), [&] { return b->lane_full || !b->tasks.empty() || b->stop; });
                b->sleeping.store(false, std::memory_order_seq_cst);
            }
            b->poked = b->lane_full = false;
            tasks.swap(b->tasks);
        }
        // ---- close the window: every lane's filling side at once ----------------------------------------------
        Window w;
        for (Lane& l : b->lanes) {
            std::unique_lock<std::mutex> g(l.mu);
            LaneSide& s = l.side[l.fill];
            if (s.empty()) continue;
            l.fill ^= 1;
            g.unlock();
            l.cv_space.notify_all();
            w.side[w.n++] = &s;
            w.forks += s.frecs.size();
            w.joins += s.jrecs.size();
            const uint32_t ph = (uint32_t)s.cut_f.size() + 1;
            if (ph > w.phases) w.phases = ph;
        }
        uint64_t (*clock_fn)(void*);
        void* clock_user;
        {
            std::lock_guard<std::mutex> g(b->mu);
            clock_fn = b->clock_fn;
            clock_user = b->clock_user;
        }
        const uint64_t now = clock_fn(clock_user);
        waiters.clear();
        if (w.n) apply(b, w, now, waiters);
        for (int k = 0; k < w.n; ++k) {
            // the closed side is the lane's spare now; nobody touches it until the next swap, which this thread does
            w.side[k]->clear();
        }
        for (Task& t : tasks) t.w->status = t.fn(t.user, b->e, now);
        if (!waiters.empty() || !tasks.empty()) {
            std::lock_guard<std::mutex> g(b->mu);
            for (Waiter* x : waiters) x->done = true;
            for (Task& t : tasks) t.w->done = true;
            b->stats.tasks += tasks.size();
        }
        if (!waiters.empty() || !tasks.empty()) b->cv_done.notify_all();
        tasks.clear();
    }
}

// ---- routed form ---------------------------------------------------------------------------------------------------
constexpr uint32_t AGREE_STOP = 1u << 31, AGREE_MASK = 0xFFFu;      // word: fork batches | join batches << 12 | stop

// One routed batch of this cycle (a collective: every rank calls it, with n = 0 when it has nothing of that kind):
// the first n rows of the pinned staging go to their owners, this rank's engine applies what it owns, and the first n
// answers in b->h_status are this rank's own requests' statuses.
int routed_exchange(dj_batcher* b, bool forks, uint32_t n, uint64_t now) {
    hipStream_t st = static_cast<hipStream_t>(dj_stream(b->e));
    if (n) {
        if (forks) {
            if (!hip_ok(hipMemcpyAsync(b->d_frecs, b->h_frecs, (size_t)n * sizeof(dj_fork_rec), hipMemcpyHostToDevice, st))) return DJ_E_HIP;
        } else {
            if (!hip_ok(hipMemcpyAsync(b->d_jrecs, b->h_jrecs, (size_t)n * sizeof(dj_join_rec), hipMemcpyHostToDevice, st))) return DJ_E_HIP;
            if (!hip_ok(hipMemcpyAsync(b->d_jpay, b->h_jpay, (size_t)n * b->pitch, hipMemcpyHostToDevice, st))) return DJ_E_HIP;
        }
        if (!hip_ok(hipStreamSynchronize(st))) return DJ_E_HIP;      // the partition reads them on the router's stream
    }
    int h = -1;
    int rc = dj_route_start(b->router, n, forks ? reinterpret_cast<const dj_join_rec*>(b->d_frecs) : b->d_jrecs, forks ? nullptr : b->d_jpay, &h);
    int more = 1;
    while (rc == DJ_OK && more) {
        const dj_join_rec* pr = nullptr;
        const uint8_t* pp = nullptr;
        uint32_t pn = 0;
        int32_t* d_st = nullptr;
        rc = dj_route_next(b->router, h, &pr, &pp, &pn, &more);
        if (rc == DJ_OK) rc = dj_route_status_buffer(b->router, h, &d_st);
        if (rc == DJ_OK) rc = forks ? dj_fork_batch(b->e, pn, reinterpret_cast<const dj_fork_rec*>(pr), d_st)
                                    : dj_join_batch(b->e, pn, pr, pp, b->pitch, now, d_st);
        if (rc == DJ_OK) rc = dj_route_return(b->router, h);
    }
    if (rc == DJ_OK) rc = dj_route_statuses(b->router, h, b->d_status, st);
    if (rc == DJ_OK) rc = dj_route_release(b->router, h);
    if (rc == DJ_OK && n && !hip_ok(hipMemcpyAsync(b->h_status, b->d_status, (size_t)n * 4, hipMemcpyDeviceToHost, st))) rc = DJ_E_HIP;
    if (rc == DJ_OK) rc = dj_sync(b->e);
    return rc;
}

void drain_loop_routed(dj_batcher* b) {
    (void)hipSetDevice(b->ecfg.device);
    std::vector<Waiter*> waiters;
    std::vector<Task> tasks;
    std::vector<uint32_t> words(b->world);
    const uint32_t mb = b->cfg.max_batch;
    // how long a rank with nothing to do sleeps before it joins the next agreement: its peers' requests wait that long
    const uint32_t idle_us = b->cfg.max_wait_us ? b->cfg.max_wait_us : 100;
    auto any_staged = [&] {
        for (Lane& l : b->lanes) {
            std::lock_guard<std::mutex> g(l.mu);
            if (!l.side[l.fill].empty()) return true;
        }
        return false;
    };
    for (;;) {
        bool staged = any_staged(), want_stop = false;
        {
            std::unique_lock<std::mutex> lk(b->mu);
            if (!staged && b->tasks.empty() && !b->stop) {
                b->sleeping.store(true, std::memory_order_seq_cst);
                lk.unlock();
                staged = any_staged();
                lk.lock();
                if (!staged) b->cv_work.wait_for(lk, std::chrono::microseconds(idle_us), [&] { return b->poked || !b->tasks.empty() || b->stop; });
                b->sleeping.store(false, std::memory_order_seq_cst);
            } else if (staged && b->tasks.empty() && !b->stop && b->cfg.max_wait_us) {
                b->sleeping.store(true, std::memory_order_seq_cst);
                b->cv_work.wait_for(lk, std::chrono::microseconds(b->cfg.max_wait_us), [&] { return b->lane_full || !b->tasks.empty() || b->stop; });
                b->sleeping.store(false, std::memory_order_seq_cst);
            }
            b->poked = b->lane_full = false;
            tasks.swap(b->tasks);
            want_stop = b->stop;
        }
        Window w;
        for (Lane& l : b->lanes) {
            std::unique_lock<std::mutex> g(l.mu);
            LaneSide& s = l.side[l.fill];
            if (s.empty()) continue;
            l.fill ^= 1;
            g.unlock();
            l.cv_space.notify_all();
            w.side[w.n++] = &s;
            w.forks += s.frecs.size();
            w.joins += s.jrecs.size();
        }
        uint64_t (*clock_fn)(void*);
        void* clock_user;
        {
            std::lock_guard<std::mutex> g(b->mu);
            clock_fn = b->clock_fn;
            clock_user = b->clock_user;
        }
        const uint64_t now = clock_fn(clock_user);
        // the agreement: what every rank brings to this cycle
        const uint32_t fc = (uint32_t)((w.forks + mb - 1) / mb), jc = (uint32_t)((w.joins + mb - 1) / mb);
        const bool idle = w.n == 0 && tasks.empty();
        int rc = b->fatal;
        if (rc == DJ_OK) rc = dj_route_agree(b->router, (fc & AGREE_MASK) | ((jc & AGREE_MASK) << 12) | (want_stop && idle ? AGREE_STOP : 0u), words.data());
        uint32_t FC = 0, JC = 0;
        bool all_stop = rc == DJ_OK;
        if (rc == DJ_OK)
            for (uint32_t x : words) {
                if ((x & AGREE_MASK) > FC) FC = x & AGREE_MASK;
                if (((x >> 12) & AGREE_MASK) > JC) JC = (x >> 12) & AGREE_MASK;
                if (!(x & AGREE_STOP)) all_stop = false;
            }
        waiters.clear();
        uint64_t largest = 0, calls = 0;
        // forks, then joins: batch c takes this rank's next mb rows of that kind in lane order (none when it has run out:
        // the exchange is a collective all the same)
        for (int kind = 0; kind < 2; ++kind) {
            const bool forks = kind == 0;
            int k = 0;
            size_t i = 0;
            const uint32_t batches = forks ? FC : JC, mine = forks ? fc : jc;
            for (uint32_t c = 0; c < (rc == DJ_OK ? batches : mine); ++c) {
                uint32_t n = 0;
                for (; k < w.n && n < mb; ++k, i = 0) {
                    LaneSide& s = *w.side[k];
                    const size_t total = forks ? s.frecs.size() : s.jrecs.size();
                    for (; i < total && n < mb; ++i, ++n) {
                        if (forks) {
                            b->h_frecs[n] = s.frecs[i];
                            b->ans[n] = s.fans[i];
                        } else {
                            b->h_jrecs[n] = s.jrecs[i];
                            if (s.jrecs[i].len) memcpy(b->h_jpay + (size_t)n * b->pitch, s.bytes.data() + s.joff[i], s.jrecs[i].len);
                            b->ans[n] = s.jans[i];
                        }
                        if (b->ans[n].cb == wait_cb) waiters.push_back(static_cast<Waiter*>(b->ans[n].user));
                    }
                    if (i < total) break;          // the batch is full in the middle of this lane
                }
                if (rc == DJ_OK) rc = routed_exchange(b, forks, n, now);
                answer_all(b, n, rc);
                if (n > largest) largest = n;
                if (n) ++calls;
            }
        }
        if (rc != DJ_OK && !b->fatal) b->fatal = rc;
        for (int k = 0; k < w.n; ++k) w.side[k]->clear();
        for (Task& t : tasks) t.w->status = t.fn(t.user, b->e, now);
        {
            std::lock_guard<std::mutex> g(b->mu);
            if (w.forks || w.joins) {
                b->stats.batches += calls;
                b->stats.forks += w.forks;
                b->stats.joins += w.joins;
                if (largest > b->stats.largest_batch) b->stats.largest_batch = largest;
            }
            for (Waiter* x : waiters) x->done = true;
            for (Task& t : tasks) t.w->done = true;
            b->stats.tasks += tasks.size();
        }
        if (!waiters.empty() || !tasks.empty()) b->cv_done.notify_all();
        tasks.clear();
        if (all_stop) return;                          // every rank read the same words: they all leave after this cycle
        if (rc != DJ_OK && want_stop && idle) return;  // the exchange is broken: nothing left to agree on
    }
}

}  // namespace

static int batcher_create(dj_engine* e, dj_router* router, const dj_batcher_config* cfg, dj_batcher** out);

extern "C" int dj_batcher_create(dj_engine* e, const dj_batcher_config* cfg, dj_batcher** out) {
    return batcher_create(e, nullptr, cfg, out);
}

extern "C" int dj_batcher_create_routed(dj_engine* e, dj_router* router, const dj_batcher_config* cfg, dj_batcher** out) {
    if (!router) return DJ_E_INVALID;
    return batcher_create(e, router, cfg, out);
}

static int batcher_create(dj_engine* e, dj_router* router, const dj_batcher_config* cfg, dj_batcher** out) {
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
    if (router) {
        // the router moves rows of its own pitch and hands out pieces of up to max_recv rows, which the engine must take
        uint32_t world = 0, max_send = 0, max_recv = 0;
        uint64_t rpitch = 0;
        if (dj_router_get_limits(router, &world, nullptr, &max_send, &max_recv, &rpitch) != DJ_OK || rpitch != b->pitch ||
            max_send < cfg->max_batch || max_recv > ec.max_batch || world < 1) {
            delete b;
            return DJ_E_INVALID;
        }
        b->router = router;
        b->world = world;
    }
    const size_t mb = cfg->max_batch;
    bool ok = hip_ok(hipSetDevice(ec.device));
    ok = ok && hip_ok(hipHostMalloc((void**)&b->h_jrecs, mb * sizeof(dj_join_rec), hipHostMallocDefault));
    ok = ok && hip_ok(hipHostMalloc((void**)&b->h_jpay, mb * b->pitch, hipHostMallocDefault));
    ok = ok && hip_ok(hipHostMalloc((void**)&b->h_frecs, mb * sizeof(dj_fork_rec), hipHostMallocDefault));
    b->ans.resize(2 * (size_t)mb);           // forks [0, mb), joins [mb, 2 mb)
    ok = ok && hip_ok(hipMalloc((void**)&b->d_jrecs, mb * sizeof(dj_join_rec)));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_jpay, mb * b->pitch));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_frecs, mb * sizeof(dj_fork_rec)));
    ok = ok && hip_ok(hipMalloc((void**)&b->d_status, mb * 4));
    ok = ok && hip_ok(hipHostMalloc((void**)&b->h_status, 2 * (size_t)mb * 4, hipHostMallocDefault));
    if (!ok) {
        dj_batcher_destroy(b);
        return DJ_E_NOMEM;
    }
    b->th = router ? std::thread(drain_loop_routed, b) : std::thread(drain_loop, b);
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
    if (b->h_jrecs) (void)hipHostFree(b->h_jrecs);
    if (b->h_jpay) (void)hipHostFree(b->h_jpay);
    if (b->h_frecs) (void)hipHostFree(b->h_frecs);
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

static bool stopping(dj_batcher* b) {
    std::lock_guard<std::mutex> g(b->mu);
    return b->stop;
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
    Lane& l = my_lane(b);
    std::unique_lock<std::mutex> lk(l.mu);
    while (l.side[l.fill].jrecs.size() >= b->cfg.max_batch) {
        lk.unlock();
        if (stopping(b)) return DJ_E_INVALID;
        poke(b, true);                            // a full lane ends the drain thread's linger
        lk.lock();
        l.cv_space.wait_for(lk, std::chrono::milliseconds(1), [&] { return l.side[l.fill].jrecs.size() < b->cfg.max_batch; });
    }
    LaneSide& s = l.side[l.fill];
    const bool was_empty = s.empty();
    s.jrecs.push_back(*rec);
    s.joff.push_back(s.bytes.size());
    if (rec->len) s.bytes.insert(s.bytes.end(), static_cast<const uint8_t*>(message), static_cast<const uint8_t*>(message) + rec->len);
    s.jans.push_back(Answer{cb, user});
    s.set_add(rec->fp_lo);
    lk.unlock();
    if (was_empty) poke(b, false);
    return DJ_OK;
}

extern "C" int dj_batcher_fork(dj_batcher* b, const dj_fork_rec* rec, dj_status_cb cb, void* user) {
    if (!b || !rec || !cb) return DJ_E_INVALID;
    Lane& l = my_lane(b);
    std::unique_lock<std::mutex> lk(l.mu);
    // (routed form: a window has one phase, so a fork behind a staged join of its barrier waits for the next window)
    auto must_wait = [&] {
        const LaneSide& f = l.side[l.fill];
        return f.frecs.size() >= b->cfg.max_batch || (b->router && f.set_has(rec->fp_lo));
    };
    while (must_wait()) {
        lk.unlock();
        if (stopping(b)) return DJ_E_INVALID;
        poke(b, true);
        lk.lock();
        l.cv_space.wait_for(lk, std::chrono::milliseconds(1), [&] { return !must_wait(); });
    }
    LaneSide& s = l.side[l.fill];
    const bool was_empty = s.empty();
    if (s.set_has(rec->fp_lo)) {
        // a join of this barrier waits in the current phase: the fork must not overtake it
        s.cut_f.push_back((uint32_t)s.frecs.size());
        s.cut_j.push_back((uint32_t)s.jrecs.size());
        s.clear_set();
    }
    s.frecs.push_back(*rec);
    s.fans.push_back(Answer{cb, user});
    lk.unlock();
    if (was_empty) poke(b, false);
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
        b->tasks.push_back(Task{fn, user, &w});
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
