// dj_kernels.hpp — HIP kernels of the barrier engine (gfx950, wave64).
//
// Nothing here is a dense contraction: every kernel is integer / byte work bound by HBM
// latency and bandwidth.  Citations are to the reference tree (src/distributed_join/).
#pragma once
#include "dj_device.hpp"

namespace dj {

// =====================================================================================
// fork: redis/fork (redis.clj:90-104) + store-in-publish-queue at the deadline (core.clj:30)
// =====================================================================================

// pass 1: find or claim the slot of every record.  The highest record index per slot wins (a later /v1/fork
// of the same barrier fully resets an earlier one, redis.clj:97-103): a record that finds its barrier in the
// table raises the slot's winner word to its index + 1 and marks the batch contested.  A record whose claim
// made the slot does neither: in a batch that nobody contests it owns the slot, and the apply pass then neither
// needs the winner word nor reads the slot at all (an atomic and a random read less per barrier).
__global__ void __launch_bounds__(256) fork_claim_kernel(Table t, Work w, const dj_fork_rec* recs, uint32_t n,
                                                         int32_t* status) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // ring pops of the apply pass stop at the tails as they are now (nothing is pushed or popped
    // during this pass), see wave_slab_alloc
    if (blockIdx.x == 0 && threadIdx.x < NUM_CLASSES) t.ring_tail_snap[threadIdx.x] = t.ring_tail[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) *w.fork_contested_next = 0;
    if (i >= n) return;
    const ulonglong2 a = reinterpret_cast<const ulonglong2*>(recs)[2 * (size_t)i];
    const uint32_t count = recs[i].count;
    if (count < 1 || count > t.max_count) {   // store.clj:32-34
        status[i] = DJ_FORK_BAD_COUNT;
        w.slot_of[i] = 0xFFFFFFFFu;
        return;
    }
    bool created;
    int64_t s = find_or_claim(t, a.x, a.y, &created);
    if (s < 0) {
        status[i] = DJ_FORK_TABLE_FULL;
        w.slot_of[i] = 0xFFFFFFFFu;
        return;
    }
    w.slot_of[i] = (uint32_t)s | (created ? SLOT_CREATED : 0u);
    if (!created) {
        a_max(&t.slots[s].chain, i + 1);
        *w.fork_contested = 1;
    }
    status[i] = DJ_FORK_CREATED;
}

// pass 1b, only in a contested batch: the records that made their slots enter the contest too (another record of
// the batch may name the same barrier, before or after them).
__global__ void __launch_bounds__(256) fork_contest_kernel(Table t, Work w, uint32_t n) {
    if (*w.fork_contested == 0) return;        // the usual case: a few hundred blocks that leave at once
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t so = w.slot_of[i];
        if (so != 0xFFFFFFFFu && (so & SLOT_CREATED)) a_max(&t.slots[so & ~SLOT_CREATED].chain, (uint32_t)i + 1);
    }
}

// pass 2: the winning record of each slot resets the barrier.
constexpr int FORK_APPLY_BLOCK = 1024;

__global__ void __launch_bounds__(FORK_APPLY_BLOCK) fork_apply_kernel(Table t, Work w, const dj_fork_rec* recs, uint32_t n,
                                                                      int32_t* status) {
    __shared__ uint32_t lds[BLOCK_SLAB_LDS];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // phase A (per lane): is this record the winner of its slot, and does the slot need a new slab
    bool active = false, need = false, had_long = false;
    uint32_t s = 0xFFFFFFFFu, k_new = 0, k_old = 0, old_slab = 0, old_count = 0, old_meta = 0;
    const bool long_on = t.max_message > t.stride;
    dj_fork_rec r = {};
    Slot* sl = nullptr;
    const bool contested = *w.fork_contested != 0;
    if (i < n) s = w.slot_of[i];
    if (s != 0xFFFFFFFFu) {
        s &= ~SLOT_CREATED;
        sl = &t.slots[s];
        active = !contested || sl->chain == i + 1;   // an equal barrier-id later in the batch supersedes this record
    }
    if (active) {
        r = recs[i];
        k_new = class_of(r.count);
        if (contested) {            // otherwise the slot is this record's own, claimed empty: nothing to read
            old_slab = sl->slab;
            old_count = sl->count;
            old_meta = sl->meta;
            had_long = long_on && (old_meta & META_OVF) != 0;
        }
        k_old = slab_class(old_count, old_meta);
        need = old_slab == 0 || k_old != k_new;
    }
    // phase B (block-cooperative): one ring pop + one bump add per block
    const uint32_t fresh = block_slab_alloc(t, need, k_new, lds);
    // a barrier that is reset gives its long-message blocks back (not when the reset is refused below)
    if (long_on) wave_release_long_each(t, had_long && !(need && fresh == 0), old_slab, k_old, 0,
                                        (old_meta & META_KX_MASK) ? class_positions(k_old) : old_count);
    block_slab_free(t, need && fresh != 0, k_old, old_slab, lds);
    // phase C (per lane)
    __shared__ unsigned long long due_lo;
    if (threadIdx.x == 0) due_lo = NQ;
    __syncthreads();
    bool ok = active;
    uint32_t slab = old_slab;
    if (active && need) {
        if (fresh == 0) {
            // engine limit, no counterpart in the reference: report it and leave the table as it was
            status[i] = DJ_FORK_NO_SPACE;
            sl->chain = 0;
            if (!(sl->meta & META_HAS) && t.q[s] == Q_NONE) clear_slot(t, s);
            ok = false;
        } else {
            slab = fresh;
        }
    }
    // pre-filled words are what a barrier without META_PMD relies on; with long messages enabled the words of
    // absent positions are read as well (none may look like a long message)
    wave_fill_posmeta_each(t, ok && (!t.fork_pmd || long_on), slab, k_new);
    if (ok) {
        if (r.count > 64) {
            uint64_t* bm = slab_bitmap(t, slab);
            for (uint32_t j = 0; j < (class_positions(k_new) >> 6); ++j) bm[j] = 0;
        }
        sl->count = r.count;                                  // SET <id>:metadata
        sl->meta = meta_pack(true, RETRY_NIL, RETRY_NIL) | (t.fork_pmd ? META_PMD : 0u);
        sl->slab = slab;
        sl->chain = 0;
        sl->deadline = r.deadline_ms;
        sl->join_ts = 0;                                      // DEL <id>:join-ts
        sl->join_ts_read = 0;                                 // DEL <id>:join-ts-read
        sl->bits = 0;                                         // DEL <id>
        t.q[s] = r.deadline_ms;                               // ZREM both, then ZADD both (redis.clj:257-258)
        note_region_date(t, s, r.deadline_ms);
    }
    // the sweep's scan bound must not be above any deadline queued here: one atomic per block at most
    unsigned long long lo = ok ? r.deadline_ms : NQ;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(lo >> 32), off) << 32) | __shfl_xor((uint32_t)lo, off);
        if (o < lo) lo = o;
    }
    if (t.ready_on && (threadIdx.x & 63) == 0 && lo != NQ) atomicMin(&due_lo, lo);
    __syncthreads();
    if (threadIdx.x == 0 && due_lo != NQ) note_global_date(t, due_lo);
}

// =====================================================================================
// join: redis/join (redis.clj:143-159), all-there? (core.clj:36-38), queue update (core.clj:52-58)
//
// A batch is applied with the results of processing its events one by one in array order.
// Pass 1 links every event into a per-barrier chain with one atomic exchange on the slot;
// pass 2 gives each touched barrier to ONE thread, which replays its chain in array order
// with plain loads and stores; pass 3 copies the messages that survive (last writer per
// position, the HSET semantics of redis.clj:135).
// =====================================================================================

// HOT = false: independent threads, no barrier anywhere (a thread with a long probe sequence delays nobody);
// blocks of 512 measured 0.5 % ahead of 256 and level with 1024 on the step.  HOT = true (the engine saw a barrier with hundreds of events in the
// previous batch): blocks of 1024 with block-level pre-aggregation.
constexpr int LINK_BLOCK = 512, LINK_BLOCK_HOT = 1024;

// Copies the message of one event into its row, G lanes per message (wave-cooperative: the event's lane hands
// its source and destination to the G lanes that move it).  The row is written in whole 64-byte lines
// (zeros behind the message): a partly written line costs a read-modify-write in memory.
template <int G, int STEP_FROM = 0>
__device__ inline void wave_copy_rows(const Table& t, const uint8_t* payloads, uint64_t pitch, uint64_t ev0, uint64_t row_off,
                                      uint32_t len) {
    constexpr int EPS = 64 / G;                     // events per step
    const uint32_t lane = threadIdx.x & 63, sub = lane / G, j0 = lane % G;
    const uint32_t rlo = (uint32_t)row_off, rhi = (uint32_t)(row_off >> 32);
    // A row is at most G chunks (one per lane of its group: G * 16 bytes cover the stride).  Four steps' loads are
    // issued before the first store, so a wave waits for memory once per four rows instead of once per row.
    constexpr int FLIGHT = 4;
#pragma unroll 1
    for (int step0 = STEP_FROM; step0 < G; step0 += FLIGHT) {
        uint4 v[FLIGHT];
        uint4* to[FLIGHT];
        bool on[FLIGHT];
#pragma unroll
        for (int f = 0; f < FLIGHT; ++f) {
            const int src = (step0 + f) * EPS + (int)sub;
            const uint64_t row = ((uint64_t)__shfl(rhi, src) << 32) | __shfl(rlo, src);
            const uint32_t ln = __shfl(len, src);
            const uint32_t chunks = (ln + 15) >> 4, full = (chunks + 3) & ~3u;
            on[f] = row != ~0ull && j0 < full;
            to[f] = reinterpret_cast<uint4*>(t.arena + (on[f] ? row : 0)) + j0;
            v[f] = make_uint4(0, 0, 0, 0);
            if (on[f] && j0 < chunks) v[f] = nt_load(&reinterpret_cast<const uint4*>(payloads + (ev0 + (uint64_t)src) * pitch)[j0]);
        }
#pragma unroll
        for (int f = 0; f < FLIGHT; ++f)
            if (on[f]) nt_store(to[f], v[f]);
    }
}

// The same copy in two halves, for a caller that knows where a message comes from before it knows where it goes: the
// loads of steps step0 .. step0 + 3 (len_pre: the message's length where it may be read at all, else 0), and their
// stores once the rows are known (row_off = ~0: no row; len as in wave_copy_rows).  What is in flight between the two
// calls overlaps whatever the caller does there (the table probe).
template <int G>
__device__ inline void wave_rows_load(const uint8_t* payloads, uint64_t pitch, uint64_t ev0, uint32_t len_pre, int step0, uint4 (&v)[4]) {
    constexpr int EPS = 64 / G;
    const uint32_t lane = threadIdx.x & 63, sub = lane / G, j0 = lane % G;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int src = (step0 + f) * EPS + (int)sub;
        const uint32_t ln = __shfl(len_pre, src);
        v[f] = make_uint4(0, 0, 0, 0);
        if (j0 < ((ln + 15) >> 4)) v[f] = nt_load(&reinterpret_cast<const uint4*>(payloads + (ev0 + (uint64_t)src) * pitch)[j0]);
    }
}
template <int G>
__device__ inline void wave_rows_store(const Table& t, uint64_t row_off, uint32_t len, int step0, const uint4 (&v)[4]) {
    constexpr int EPS = 64 / G;
    const uint32_t lane = threadIdx.x & 63, sub = lane / G, j0 = lane % G;
    const uint32_t rlo = (uint32_t)row_off, rhi = (uint32_t)(row_off >> 32);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int src = (step0 + f) * EPS + (int)sub;
        const uint64_t row = ((uint64_t)__shfl(rhi, src) << 32) | __shfl(rlo, src);
        const uint32_t ln = __shfl(len, src);
        const uint32_t chunks = (ln + 15) >> 4, full = (chunks + 3) & ~3u;
        if (row != ~0ull && j0 < full)
            nt_store(reinterpret_cast<uint4*>(t.arena + row) + j0, j0 < chunks ? v[f] : make_uint4(0, 0, 0, 0));
    }
}

// G > 0: fused form, the pass also copies the messages (G lanes each) and stores the {len, max-wait} words;
// payloads may then not be null.  G = 0: the copy pass does that later.
template <bool HOT, int G>
__global__ void __launch_bounds__(HOT ? LINK_BLOCK_HOT : LINK_BLOCK)
join_link_kernel(Table t, Work w, const dj_join_rec* recs, uint32_t n, uint64_t pitch, int32_t* status, const uint8_t* payloads) {
    // Block-level pre-aggregation for barriers flagged hot: their events are chained inside the block
    // through LDS and the block issues ONE exchange per hot barrier (a queued returning atomic costs
    // ~40 ns, ten thousand of them on one word would serialise the whole pass).  Up to HOT_KEYS hot
    // barriers per block: the events count themselves per barrier, a prefix sum gives every barrier its
    // range in one shared list, the last entry of a range goes into the slot's chain word.
    constexpr int HOT_KEYS = 64;
    __shared__ uint32_t hot_key[HOT ? HOT_KEYS : 1];               // slot + 1, 0 = free
    __shared__ uint32_t hot_cnt[HOT ? HOT_KEYS : 1];
    __shared__ uint32_t hot_off[HOT ? HOT_KEYS : 1];
    __shared__ uint32_t hot_prev[HOT ? HOT_KEYS : 1];
    __shared__ uint32_t hot_ev[HOT ? LINK_BLOCK_HOT : 1];          // event index + 1, grouped by barrier
    if constexpr (HOT) {
        if (threadIdx.x < HOT_KEYS) { hot_key[threadIdx.x] = 0; hot_cnt[threadIdx.x] = 0; }
        __syncthreads();
    }
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the previous batch's counters (the bank this batch does not use) are final: to the host's memory, then to zero
    // for the next batch
    if (blockIdx.x == 0 && threadIdx.x < C_COUNT) {
        if (w.publish_hints) {
            if (threadIdx.x == C_COMPLEX) w.hint[0] = w.counters_next[C_COMPLEX];
            if (threadIdx.x == C_MAXCHAIN) w.hint[1] = w.counters_next[C_MAXCHAIN];
            if (threadIdx.x == C_MID) w.hint[8] = w.counters_next[C_MID];
        }
        w.counters_next[threadIdx.x] = 0;
    }
    uint32_t slot = 0;
    uint64_t row_off = ~0ull;      // fused form: byte offset of the event's row in the arena
    uint32_t row_len = 0;
    bool stores_meta = false;      // this event writes its position's {len, max-wait} word
    int agg_key = -1;
    uint32_t agg_rank = 0, agg_pos = 0, agg_mw = 0;
    bool need_x = false;           // this event still has to put itself at the head of its barrier's chain
    uint32_t x_pos = 0, x_mw = 0;
    if (i < n) {
        const ulonglong2 a = reinterpret_cast<const ulonglong2*>(recs)[2 * (size_t)i];
        const uint4 b = reinterpret_cast<const uint4*>(recs)[2 * (size_t)i + 1];  // position, len, max_wait, -
        int32_t st = 0;
        int64_t s = -1;
        uint4 info = make_uint4(0, 0, 0, 0);               // count, meta, slab, chain of the slot
        if (b.y > t.max_message || b.y > pitch) {
            st = DJ_JOIN_TOO_LONG;                         // store.clj:24-26
        } else if (b.z >= NODE_MW_LIMIT) {
            st = DJ_JOIN_BAD_MAX_WAIT;                     // engine limit (store.clj:27 caps it far lower)
        } else {
            s = find_slot(t, a.x, a.y, &info);
            if (s < 0 || !(info.y & META_HAS) || info.z == 0) st = DJ_JOIN_NOT_FOUND;   // redis.clj:152
            else if (b.x >= info.x) st = DJ_JOIN_BAD_POSITION;                          // redis.clj:153
        }
        if (st != 0) {
            status[i] = st;
            if (G == 0) w.dst[i] = NODST;
            w.ev_slot[i] = 0xFFFFFFFFu;
        } else {
            // the common answers are written here, coalesced; the resolve pass only overwrites the
            // exceptions (COMPLETE / NEAREST_NOT_QUEUED statuses, superseded messages)
            status[i] = DJ_JOIN_WAITING;
            // {len, max_wait} of a position is stored only when it differs from a full row without max-wait,
            // or when the barrier has seen such a position before (META_PMD, sticky until the next fork)
            const bool pm = (info.y & META_PMD) || b.y != t.stride || b.z != 0;
            stores_meta = pm;
            const uint32_t k = slab_class(info.x, info.y);
            if (G == 0)
                w.dst[i] = dst_pack(info.z, k, b.x) | (pm ? DST_PM : 0ull) | (b.y == t.stride ? DST_FULL : 0ull) |
                           ((info.y & META_OVF) ? DST_OVF : 0ull);
            if (pm && !(info.y & META_PMD)) atomicOr(&t.slots[s].meta, META_PMD);
            if (G > 0) {
                row_off = (uint64_t)info.z * 64 + slab_payload_off(k) + (uint64_t)b.x * t.stride;
                row_len = b.y;
                if (pm) slab_posmeta(t, info.z, k)[b.x] = make_uint2(b.y, b.z);
            }
            slot = (uint32_t)s;
            w.ev_slot[i] = slot;
            if (HOT && (info.y & META_HOT)) {
                // open addressing over the block's key table; a full table leaves the event to the direct exchange
                uint32_t j = (slot * 2654435761u) >> 26;            // 6 bits: HOT_KEYS = 64
                for (int probe = 0; probe < HOT_KEYS && agg_key < 0; ++probe, j = (j + 1) & (HOT_KEYS - 1)) {
                    const uint32_t k = atomicCAS(&hot_key[j], 0u, slot + 1);
                    if (k == 0 || k == slot + 1) {
                        agg_key = (int)j;
                        agg_rank = atomicAdd(&hot_cnt[j], 1u);
                        agg_pos = b.x; agg_mw = b.z;
                    }
                }
            }
            if (agg_key < 0) { need_x = true; x_pos = b.x; x_mw = b.z; }
        }
    }
    // Many lanes of one wave on the same barrier (a barrier turning hot before anyone has flagged it): they
    // chain among themselves and issue ONE exchange, instead of queueing up on the chain word at 20-40 ns each.
    // Two candidates per wave, the barriers of its first and of its last such lane; a batch without a hot
    // barrier never gets past the popcount.
    {
        const uint32_t lane = threadIdx.x & 63;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint64_t rem = __ballot(need_x);
            if (rem == 0) break;
            const int l0 = r == 0 ? __ffsll((unsigned long long)rem) - 1 : 63 - __clzll((unsigned long long)rem);
            const uint32_t s0 = __shfl(slot, l0);
            const uint64_t grp = __ballot(need_x && slot == s0);
            if (__popcll(grp) < 4) continue;
            const int hi = 63 - __clzll((unsigned long long)grp);
            uint32_t prev = 0;
            if ((int)lane == hi) prev = a_xchg(&t.slots[slot].chain, i + 1);
            prev = __shfl(prev, hi);
            if (need_x && slot == s0) {
                const uint64_t below = grp & ((1ull << lane) - 1);
                // the next lower lane of the group (its event index is i minus the lane distance), or the chain as it was
                const uint32_t next = below ? i - (lane - (uint32_t)(63 - __clzll((unsigned long long)below))) + 1 : prev;
                w.node[i] = node_pack(next, x_pos, x_mw);
                need_x = false;
            }
        }
    }
    if (need_x) {
        const uint32_t prev = a_xchg(&t.slots[slot].chain, i + 1);
        w.node[i] = node_pack(prev, x_pos, x_mw);
    }
    if constexpr (G > 0) wave_copy_rows<G>(t, payloads, pitch, (uint64_t)i - (threadIdx.x & 63), row_off, row_len);
    // a sample for the host: how ragged is the traffic (it decides how the next barriers are forked)
    if (blockIdx.x == 0 && threadIdx.x < 256 && w.publish_hints) {
        const uint64_t m = __ballot(stores_meta);
        if ((threadIdx.x & 63) == 0) w.hint[4 + (threadIdx.x >> 6)] = (unsigned int)__popcll(m);
    }
    if constexpr (!HOT) return;
    __syncthreads();
    if (threadIdx.x < 64) {                                     // HOT_KEYS = 64: one wave scans the counts
        const uint32_t c = hot_cnt[threadIdx.x];
        hot_off[threadIdx.x] = wave_incl_scan(c) - c;
    }
    __syncthreads();
    if (agg_key >= 0) hot_ev[hot_off[agg_key] + agg_rank] = i + 1;
    __syncthreads();
    if (threadIdx.x < HOT_KEYS && hot_cnt[threadIdx.x] != 0)
        hot_prev[threadIdx.x] = a_xchg(&t.slots[hot_key[threadIdx.x] - 1].chain,
                                       hot_ev[hot_off[threadIdx.x] + hot_cnt[threadIdx.x] - 1]);
    __syncthreads();
    if (agg_key >= 0) {
        // element r links to element r-1 of its barrier's range, element 0 to the chain as it was
        const uint32_t next = agg_rank == 0 ? hot_prev[agg_key] : hot_ev[hot_off[agg_key] + agg_rank - 1];
        w.node[i] = node_pack(next, agg_pos, agg_mw);
    }
}

// Sequential replay of one barrier's events e[0..m) (sorted by array index) — the body of
// core/join per event.  `bits`/`arrived` are the presence state, pa the primary-queue score.
struct ReplayState {
    uint64_t bits;      // count <= 64
    uint32_t arrived;
    uint32_t count;
    uint64_t pa;        // primary score (NQ = not queued); valid when pa_loaded
    bool pa_loaded, pa_dirty, enq;
};

__device__ inline int32_t replay_one(const Table& t, uint64_t s, uint32_t slab, ReplayState& r, uint32_t pos,
                                     uint32_t mw, uint64_t now) {
    if (r.count > 64) {
        uint64_t* bm = slab_bitmap(t, slab);
        uint64_t word = bm[pos >> 6], bit = 1ull << (pos & 63);
        if (!(word & bit)) { bm[pos >> 6] = word | bit; ++r.arrived; }      // HSET + HLEN
    } else {
        uint64_t bit = 1ull << pos;
        if (!(r.bits & bit)) { r.bits |= bit; ++r.arrived; }
    }
    if (r.arrived == r.count) {              // all-there?, core.clj:36-38
        r.enq = true;                        // store-in-publish-queue at now, core.clj:54
        return DJ_JOIN_COMPLETE;
    }
    if (mw) {                                // join-with-max-wait, core.clj:40-42
        if (!r.pa_loaded) { r.pa = q_primary_at(t.q[s]); r.pa_loaded = true; }
        if (r.pa == NQ) return DJ_JOIN_NEAREST_NOT_QUEUED;   // ZSCORE nil: the Lua compare raises (redis.clj:268-269)
        uint64_t at = now + 1000ull * mw;
        if (r.pa > at) { r.pa = at; r.pa_dirty = true; }     // redis.clj:269-271
    }
    return DJ_JOIN_WAITING;
}

// Chains up to CR events are replayed from registers.  CR = 8 is the common form; the host picks CR = 16
// (more registers, fewer resident waves) while many barriers receive 9..16 events per batch, which would
// otherwise each take a wave of the long-chain kernels.
constexpr int CHAIN_REG = 8, CHAIN_REG_WIDE = 16;
constexpr int RESOLVE_BLOCK = 512;

// The register replay of one barrier's m <= CR events of this batch (idx[k] their array indices, nd[k] their link
// words: position and max-wait; unused entries carry index 0xFFFFFFFF), in array order, with plain loads and stores by
// the one thread that owns the barrier for this batch.  Returns whether the barrier completed.
template <int CR>
__device__ __forceinline__ bool replay_short_chain(const Table& t, const Work& w, uint32_t s, const uint4& info, uint64_t bits0,
                                          const uint32_t (&idx)[CR], const uint64_t (&nd)[CR], int m, uint64_t now,
                                          int32_t* status) {
    Slot* sl = &t.slots[s];
    const uint32_t count = info.x, slab = info.z;
    bool completed = false;
    {
        {
            ReplayState r;
            r.count = count;
            r.bits = count > 64 ? 0 : bits0;
            r.arrived = count > 64 ? (uint32_t)bits0 : (uint32_t)__popcll(bits0);
            r.pa = NQ; r.pa_loaded = false; r.pa_dirty = false; r.enq = false;
            // Replay in array order: every round picks the smallest event index not yet replayed (unused
            // entries carry 0xFFFFFFFF).  The body exists once (a real loop), the scans over the register
            // arrays are unrolled with constant indices: few live registers, no scratch.
            uint32_t done = 0;
#pragma unroll 1
            for (int round = 0; round < m; ++round) {
                uint32_t best = 0xFFFFFFFFu, bk = 0;
                uint64_t bn = 0;
#pragma unroll
                for (int k = 0; k < CR; ++k)
                    if (!(done & (1u << k)) && idx[k] < best) { best = idx[k]; bk = (uint32_t)k; bn = nd[k]; }
                done |= 1u << bk;
                const uint32_t pos = node_pos(bn);
                const int32_t st = replay_one(t, s, slab, r, pos, node_mw(bn), now);
                if (w.hlen) status[best] = st | (int32_t)(r.arrived << DJ_JOIN_HLEN_SHIFT);     // HLEN after this event's HSET (redis.clj:135-137)
                else if (st != DJ_JOIN_WAITING) status[best] = st;
                bool last = true;   // last writer of its position inside the batch: no later event has it
#pragma unroll
                for (int k = 0; k < CR; ++k)
                    if (!(done & (1u << k)) && idx[k] != 0xFFFFFFFFu && node_pos(nd[k]) == pos) last = false;
                if (!w.fused) {
                    if (!last) w.dst[best] = NODST;
                } else if (last) {
                    // every event has copied its message already, in no order: where earlier events of the batch
                    // wrote this position too, the last writer copies again (HSET: the last write stays)
                    bool again = false;
#pragma unroll
                    for (int k = 0; k < CR; ++k)
                        if ((done & (1u << k)) && (uint32_t)k != bk && node_pos(nd[k]) == pos) again = true;
                    // (rare: the event goes on a list that the pass for long chains works off, a wave per message)
                    if (again) w.dst[atomicAdd(&w.counters[C_RECOPY], 1u)] = best;
                }
            }
            store_ts_bits(sl, now, count > 64 ? (uint64_t)r.arrived : r.bits);   // SET <id>:join-ts, redis.clj:136
            sl->chain = 0;
            if (info.y & META_HOT) sl->meta = info.y & ~META_HOT; // a short chain: not hot any more
            if (r.enq) { t.q[s] = now; completed = true; }               // ZADD both at now, redis.clj:257-258
            else if (r.pa_dirty) { t.q[s] = r.pa; note_queue_date(t, s, r.pa); }   // ZADD both at the nearer date, redis.clj:270-271
        }
    }
    return completed;
}

template <int CR>
__global__ void __launch_bounds__(RESOLVE_BLOCK) __attribute__((amdgpu_waves_per_eu(CR <= 8 ? 8 : 4)))
join_resolve_kernel(Table t, Work w, uint32_t n, uint64_t now, int32_t* status) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    // block pops of the copy pass stop at the tails as they are now (nothing touches these rings here)
    if (blockIdx.x == 0 && threadIdx.x < OVF_CLASSES) t.oring_tail_snap[threadIdx.x] = t.oring_tail[threadIdx.x];
    bool is_complex = false, mid = false, completed = false;
    uint32_t s = tid < n ? w.ev_slot[tid] : 0xFFFFFFFFu;
    // The event that linked FIRST (it found the chain empty) resolves the barrier: it knows that from
    // its own link word, so every other event leaves without touching the table again.
    const uint64_t own = s != 0xFFFFFFFFu ? w.node[tid] : 0;
    if (s != 0xFFFFFFFFu && node_next(own) != 0) s = 0xFFFFFFFFu;
    uint4 info = make_uint4(0, 0, 0, 0);
    uint64_t bits0 = 0;
    if (s != 0xFFFFFFFFu) {
        info = reinterpret_cast<const uint4*>(&t.slots[s])[1];   // count, meta, slab, chain
        bits0 = t.slots[s].bits;                                 // requested with it: one round trip
    }
    if (s != 0xFFFFFFFFu) {
        Slot* sl = &t.slots[s];
        const uint32_t count = info.x, slab = info.z;
        uint32_t cur = info.w;
        // walk the chain (newest link first)
        uint32_t idx[CR];
        uint64_t nd[CR];
        int m = 0;
#pragma unroll
        for (int k = 0; k < CR; ++k) {
            idx[k] = 0xFFFFFFFFu;
            nd[k] = 0;
            if (cur != 0) {
                idx[k] = cur - 1;
                nd[k] = cur - 1 == tid ? own : w.node[cur - 1];      // every chain ends at this thread's own event
                cur = node_next(nd[k]);
                ++m;
            }
        }
        if (cur != 0 || (info.y & META_KX_MASK)) {
            is_complex = true;      // a long chain, or a barrier whose count went back (only the wave kernel replays those)
        } else {
            mid = m > CHAIN_REG;
            completed = replay_short_chain<CR>(t, w, s, info, bits0, idx, nd, m, now, status);
        }
    }
    ready_append_wave(t, completed, s, tid >> 6);
    uint32_t at = wave_append(&w.counters[C_COMPLEX], is_complex);
    if (is_complex) w.complex_[at] = s;
    if (CR > CHAIN_REG && ((tid >> 6) & 7) == 0) {      // one wave in eight reports how many such chains it replayed
        const uint32_t k = (uint32_t)__popcll(__ballot(mid));
        if ((threadIdx.x & 63) == 0 && k) atomicAdd(&w.counters[C_MID], 8 * k);
    }
}

// =====================================================================================
// Bucketed form of the two join passes (same results as join_link_kernel<false, G> + join_resolve_kernel<8>).
//
// The atomic exchange per event is what the chain form pays most for (about 35 us per 2^20 events wherever the word
// lives).  Here the first pass does everything else the link pass does, in the same order (records and messages in
// array order, probe, answers, message into its row), and instead of linking the event on its slot it files an 8-byte
// key {slot, event, position} under the bucket of the slot's range: a block takes a tile of 4096 events, counts its
// keys per bucket in LDS, reserves their places with ONE atomic per bucket and tile, and writes them out of LDS in
// runs.  The second pass has one block per bucket: it chains its keys per slot in LDS (hash + exchange, both in LDS),
// and the first key of every slot replays that barrier's events from LDS with the code of the register replay, so a
// touched barrier's slot is read and written once and no link word is ever chased through memory.
// Barriers the register replay does not take (more than CHAIN_REG events, or a count that went back) get their chain
// written to the slot and the link words after all and go on the list of the long-chain pass, as in the chain form.
// A bucket that would overflow its LDS (one range with more than RB_CAP events: a hot barrier nobody has flagged yet)
// raises C_BOVF: the second pass then links every event the old way instead and puts every touched barrier on the
// long-chain pass's list (slow, correct, and the host leaves the bucketed form from the next batch on, whose hints
// show the long chain).
// =====================================================================================
constexpr int BK_TILE = 4096, BK_BLOCK = 1024, BK_MAX_BUCKETS = 1024;      // BK_TILE: the largest tile (four events per thread)
constexpr uint32_t BK_MEAN_MAX = 1280;     // events per bucket on average at most (a bucket holds RB_CAP; a batch of random keys spreads with a deviation of about 36)
constexpr int RB_T = 512, RB_EPT = 4, RB_CAP = RB_T * RB_EPT, RB_H = RB_CAP;    // (as many cells as keys at most: a free cell is always left for a new slot)
// key: [max-wait, M bits] slot inside its bucket's range << 35 | event index (24 bits) << 11 | has-max-wait << 10 | position (10 bits)
// M = Work::bmwbits: the bits above the slot field that the bucket's range leaves free (0 to 14).  A max-wait below 2^M
// travels in the key; a longer one (or M = 0) leaves the field 0 and the second pass reads it from the event's record.
__device__ inline uint64_t bkey_pack(uint32_t slot_local, uint32_t i, uint32_t pos, uint32_t mw, uint32_t M) {
    const uint64_t inl = (M != 0 && mw < (1u << M)) ? ((uint64_t)mw << (64 - M)) : 0ull;     // (mw = 0: nothing to carry)
    return inl | ((uint64_t)slot_local << 35) | ((uint64_t)i << 11) | (mw ? 1024u : 0u) | pos;
}
__device__ inline uint32_t bkey_slot(uint64_t k, uint32_t M) { return (uint32_t)((k << M) >> (35 + M)); }
__device__ inline uint32_t bkey_mw(uint64_t k, uint32_t M, const dj_join_rec* recs, uint32_t i) {
    if (!(k & 1024u)) return 0u;
    const uint32_t inl = M ? (uint32_t)(k >> (64 - M)) : 0u;
    return inl ? inl : recs[i].max_wait_s;
}
__device__ inline uint32_t bkey_event(uint64_t k) { return (uint32_t)(k >> 11) & 0xFFFFFFu; }

// A tile is tile_events consecutive events (a multiple of 64, at most 1024 * EPT): the host cuts a batch into equal
// tiles, one or two per compute unit (a thread works through its events one after the other, so that a tile more than
// the GPU holds at once, or one longer than the others, costs a whole extra round), and a tile still holds two to four
// keys per bucket.  Round k of a tile takes its events k * 1024 .. k * 1024 + 1023.
template <int G, int EPT>
__global__ void __launch_bounds__(BK_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
join_link_bucket_kernel(Table t, Work w, const dj_join_rec* recs, uint32_t n, uint32_t tile_events, uint64_t pitch, int32_t* status,
                        const uint8_t* payloads) {
    const uint32_t tile_base = blockIdx.x * tile_events;
    const uint32_t tile_end = n - tile_base < tile_events ? n : tile_base + tile_events;
    __shared__ uint32_t hist[BK_MAX_BUCKETS], off[BK_MAX_BUCKETS], base[BK_MAX_BUCKETS];
    __shared__ uint64_t stage[BK_TILE];
    __shared__ uint16_t stage_b[BK_TILE];
    __shared__ uint32_t wsum[BK_BLOCK / 64];
    const uint32_t tix = threadIdx.x, lane = tix & 63, wave = tix >> 6;
    const uint32_t B = w.bbuckets, shift = w.bshift;
    for (uint32_t b = tix; b < B; b += BK_BLOCK) hist[b] = 0;
    // the previous batch's counters (the bank this batch does not use) are final: to the host's memory, then to zero
    if (blockIdx.x == 0 && tix < C_COUNT) {
        if (w.publish_hints) {
            if (tix == C_COMPLEX) w.hint[0] = w.counters_next[C_COMPLEX];
            if (tix == C_MAXCHAIN) w.hint[1] = w.counters_next[C_MAXCHAIN];
            if (tix == C_MID) w.hint[8] = w.counters_next[C_MID];
        }
        w.counters_next[tix] = 0;
    }
    __syncthreads();
    uint64_t key[EPT];
    uint32_t bk[EPT], rank[EPT];
    bool linked[EPT];
    // all records of the thread first (they are where the array says), then per event: its message's loads, the table
    // probe while those are in flight, the stores
    ulonglong2 ra[EPT];
    uint4 rb[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t i = tile_base + k * BK_BLOCK + tix;
        ra[k] = make_ulonglong2(0, 0);
        rb[k] = make_uint4(0, 0, 0, 0);
        if (i < tile_end) {
            const uint4 ra4 = nt_load(&reinterpret_cast<const uint4*>(recs)[2 * (size_t)i]);
            ra[k] = make_ulonglong2(((uint64_t)ra4.y << 32) | ra4.x, ((uint64_t)ra4.w << 32) | ra4.z);
            rb[k] = nt_load(&reinterpret_cast<const uint4*>(recs)[2 * (size_t)i + 1]);      // position, len, max_wait, -
        }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t i = tile_base + k * BK_BLOCK + tix;
        uint64_t row_off = ~0ull;
        uint32_t row_len = 0;
        bool stores_meta = false;
        key[k] = 0; bk[k] = 0; rank[k] = 0; linked[k] = false;
        const ulonglong2 a = ra[k];
        const uint4 b = rb[k];
        uint4 pv[4];
        wave_rows_load<G>(payloads, pitch, (uint64_t)i - lane, (i < tile_end && b.y <= t.max_message && b.y <= pitch) ? b.y : 0u, 0, pv);
        if (i < tile_end) {
            int32_t st = 0;
            int64_t s = -1;
            uint4 info = make_uint4(0, 0, 0, 0);
            if (b.y > t.max_message || b.y > pitch) {
                st = DJ_JOIN_TOO_LONG;                         // store.clj:24-26
            } else if (b.z >= NODE_MW_LIMIT) {
                st = DJ_JOIN_BAD_MAX_WAIT;
            } else {
                s = find_slot(t, a.x, a.y, &info);
                if (s < 0 || !(info.y & META_HAS) || info.z == 0) st = DJ_JOIN_NOT_FOUND;   // redis.clj:152
                else if (b.x >= info.x) st = DJ_JOIN_BAD_POSITION;                          // redis.clj:153
            }
            if (st != 0) {
                status[i] = st;
                w.ev_slot[i] = 0xFFFFFFFFu;
            } else {
                status[i] = DJ_JOIN_WAITING;
                const bool pm = (info.y & META_PMD) || b.y != t.stride || b.z != 0;
                stores_meta = pm;
                const uint32_t kc = slab_class(info.x, info.y);
                if (pm && !(info.y & META_PMD)) atomicOr(&t.slots[s].meta, META_PMD);
                row_off = (uint64_t)info.z * 64 + slab_payload_off(kc) + (uint64_t)b.x * t.stride;
                row_len = b.y;
                if (pm) slab_posmeta(t, info.z, kc)[b.x] = make_uint2(b.y, b.z);
                const uint32_t slot = (uint32_t)s;
                w.ev_slot[i] = slot;
                key[k] = bkey_pack(slot & ((1u << shift) - 1), i, b.x, b.z, w.bmwbits);
                bk[k] = slot >> shift;
                linked[k] = true;
                rank[k] = atomicAdd(&hist[bk[k]], 1u);
            }
        }
        wave_rows_store<G>(t, row_off, row_len, 0, pv);
        if constexpr (G > 4) wave_copy_rows<G, 4>(t, payloads, pitch, (uint64_t)i - lane, row_off, row_len);
        // a sample for the host: how ragged is the traffic (the batch's first 4 x 64 events)
        if (blockIdx.x == 0 && k == 0 && wave < 4 && w.publish_hints) {
            const uint64_t m = __ballot(stores_meta);
            if (lane == 0) w.hint[4 + wave] = (unsigned int)__popcll(m);
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts (one per thread), one cursor atomic per bucket that has keys
    {
        const uint32_t c = tix < B ? hist[tix] : 0;
        const uint32_t incl = wave_incl_scan(c);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t bw = 0;
        for (uint32_t v = 0; v < wave; ++v) bw += wsum[v];
        if (tix < B) {
            off[tix] = bw + incl - c;
            base[tix] = c ? a_add(&w.bcursor[tix], c) : 0u;
            if (c && base[tix] + c > w.bcap) a_store(&w.counters[C_BOVF], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (linked[k]) {
            stage[off[bk[k]] + rank[k]] = key[k];
            stage_b[off[bk[k]] + rank[k]] = (uint16_t)bk[k];
        }
    __syncthreads();
    // the keys leave LDS in bucket order: neighbouring threads write neighbouring keys of one bucket
    const uint32_t total = off[B - 1] + hist[B - 1];
    for (uint32_t e = tix; e < total; e += BK_BLOCK) {
        const uint32_t bq = stage_b[e];
        const uint32_t at = base[bq] + (e - off[bq]);
        if (at < w.bcap) w.bkeys[(size_t)bq * w.bcap + at] = stage[e];
    }
}

// FAT: a batch of millions of events files more keys per bucket than the LDS below holds (w.bfat times RB_CAP at most): the
// bucket's slot range is then cut into w.bfat sub-ranges with a block each (block = sub-range x buckets + bucket: the
// blocks of one bucket run on one XCD and find its keys in that L2 after the first has read them); a block reads all
// keys of its bucket and keeps those of its sub-range.  A sub-range that still holds more than RB_CAP keys (a hot barrier
// nobody has flagged) chains its events on their slots the way join_link_kernel does and hands its barriers to the
// long-chain pass.  The host clears the cursors after a FAT launch (no block knows when its bucket's others have read theirs).
template <bool FAT>
__global__ void __launch_bounds__(RB_T) __attribute__((amdgpu_waves_per_eu(8)))
join_resolve_bucket_kernel(Table t, Work w, const dj_join_rec* recs, uint32_t n, uint64_t now, int32_t* status) {
    __shared__ uint32_t hkey[RB_H];        // slot inside the range + 1, 0 = free
    __shared__ uint32_t hhead[RB_H];       // newest key of the slot + 1
    __shared__ uint16_t nxt[RB_CAP];       // the key filed before it on the same slot + 1, 0 = none: the slot's first key
    __shared__ uint16_t cellof[RB_CAP];
    __shared__ uint64_t kbuf[RB_CAP];
    __shared__ uint32_t sub_n;
    const uint32_t tix = threadIdx.x, blk = blockIdx.x;
    const uint32_t b = FAT ? blk & (w.bbuckets - 1) : blk, sub = FAT ? blk / w.bbuckets : 0u;
    uint32_t nb = w.bcursor[b];
    if (!FAT) {
        __syncthreads();
        if (tix == 0) w.bcursor[b] = 0;    // for the next batch
    }
    if (a_load(&w.counters[C_BOVF]) != 0) {
        // A bucket overflowed (the same answer in every thread of every block): every linked event enters its barrier's
        // chain the way join_link_kernel does it, and every touched barrier goes on the long-chain pass's list.
        for (uint64_t i0 = (uint64_t)blk * RB_T; i0 < n; i0 += (uint64_t)gridDim.x * RB_T) {
            const uint64_t i = i0 + tix;
            const uint32_t s = i < n ? w.ev_slot[i] : 0xFFFFFFFFu;
            bool first = false;
            if (s != 0xFFFFFFFFu) {
                const uint4 rb = reinterpret_cast<const uint4*>(recs)[2 * (size_t)i + 1];      // position, len, max_wait, -
                const uint32_t prev = a_xchg(&t.slots[s].chain, (uint32_t)i + 1);
                w.node[i] = node_pack(prev, rb.x, rb.z);
                first = prev == 0;
            }
            const uint32_t at = wave_append(&w.counters[C_COMPLEX], first);
            if (first) w.complex_[at] = s;
        }
        return;
    }
    if (nb == 0) return;                   // the same in every thread
    if (nb > w.bcap) nb = w.bcap;                                   // (cannot be: C_BOVF would be set)
    const uint64_t* bucket_keys = w.bkeys + (size_t)b * w.bcap;
    if (FAT) {
        // this sub-range's keys out of the bucket's (slot inside the bucket's range >> bsubshift names the sub-range)
        const uint32_t nb_all = nb;
        if (tix == 0) sub_n = 0;
        __syncthreads();
        for (uint32_t e = tix; e < nb_all; e += RB_T) {
            const uint64_t kv = bucket_keys[e];
            if (bkey_slot(kv, w.bmwbits) >> w.bsubshift == sub) {
                const uint32_t at = atomicAdd(&sub_n, 1u);
                if (at < (uint32_t)RB_CAP) kbuf[at] = kv;
            }
        }
        __syncthreads();
        nb = sub_n;
        if (nb == 0) return;               // the same in every thread
        if (nb > (uint32_t)RB_CAP) {
            // too many for the LDS: chain them on their slots instead; the long-chain pass takes every touched barrier
            for (uint32_t e0 = 0; e0 < nb_all; e0 += RB_T) {          // uniform: wave_append below
                const uint32_t e = e0 + tix;
                bool first = false;
                uint32_t s = 0;
                if (e < nb_all) {
                    const uint64_t kv = bucket_keys[e];
                    if (bkey_slot(kv, w.bmwbits) >> w.bsubshift == sub) {
                        const uint32_t i = bkey_event(kv);
                        s = (b << w.bshift) | bkey_slot(kv, w.bmwbits);
                        const uint32_t prev = a_xchg(&t.slots[s].chain, i + 1);
                        w.node[i] = node_pack(prev, (uint32_t)kv & 1023u, bkey_mw(kv, w.bmwbits, recs, i));
                        first = prev == 0;
                    }
                }
                const uint32_t at = wave_append(&w.counters[C_COMPLEX], first);
                if (first) w.complex_[at] = s;
            }
            return;
        }
    }
    for (uint32_t c = tix; c < (uint32_t)RB_H; c += RB_T) { hkey[c] = 0; hhead[c] = 0; }
    if (!FAT)
        for (uint32_t e = tix; e < nb; e += RB_T) kbuf[e] = bucket_keys[e];
    __syncthreads();
    for (uint32_t e = tix; e < nb; e += RB_T) {
        const uint32_t sl = bkey_slot(kbuf[e], w.bmwbits);
        uint32_t c = ((sl * 2654435761u) >> 19) & (RB_H - 1);
        for (;; c = (c + 1) & (RB_H - 1)) {
            const uint32_t old = atomicCAS(&hkey[c], 0u, sl + 1);
            if (old == 0 || old == sl + 1) break;
        }
        cellof[e] = (uint16_t)c;
        nxt[e] = (uint16_t)atomicExch(&hhead[c], e + 1);
    }
    __syncthreads();
    for (uint32_t e0 = 0; e0 < nb; e0 += RB_T) {          // uniform: the list appends below are wave-cooperative
        const uint32_t e = e0 + tix;
        const bool first = e < nb && nxt[e] == 0;
        bool is_complex = false, completed = false;
        uint32_t s = 0xFFFFFFFFu;
        if (first) {
            const uint32_t c = cellof[e];
            s = (b << w.bshift) | (hkey[c] - 1);
            const uint4 info = reinterpret_cast<const uint4*>(&t.slots[s])[1];   // count, meta, slab, chain
            const uint64_t bits0 = t.slots[s].bits;
            uint32_t cur = hhead[c];
            uint32_t idx[CHAIN_REG];
            uint64_t nd[CHAIN_REG];
            int m = 0;
#pragma unroll
            for (int k = 0; k < CHAIN_REG; ++k) {
                idx[k] = 0xFFFFFFFFu;
                nd[k] = 0;
                if (cur != 0) {
                    const uint64_t kv = kbuf[cur - 1];
                    idx[k] = bkey_event(kv);
                    nd[k] = node_pack(0, (uint32_t)kv & 1023u, bkey_mw(kv, w.bmwbits, recs, idx[k]));
                    cur = nxt[cur - 1];
                    ++m;
                }
            }
            if (cur != 0 || (info.y & META_KX_MASK)) {
                // not for the register replay: the chain goes where the long-chain pass looks for it
                is_complex = true;
                uint32_t c2 = hhead[c];
                t.slots[s].chain = bkey_event(kbuf[c2 - 1]) + 1;
                while (c2 != 0) {
                    const uint64_t kv = kbuf[c2 - 1];
                    const uint32_t i = bkey_event(kv), nx = nxt[c2 - 1];
                    w.node[i] = node_pack(nx != 0 ? bkey_event(kbuf[nx - 1]) + 1 : 0u, (uint32_t)kv & 1023u,
                                          bkey_mw(kv, w.bmwbits, recs, i));
                    c2 = nx;
                }
            } else {
                completed = replay_short_chain<CHAIN_REG>(t, w, s, info, bits0, idx, nd, m, now, status);
            }
        }
        ready_append_wave(t, completed, s, blk * (RB_T / 64) + (tix >> 6) + (e0 / RB_T));
        const uint32_t at = wave_append(&w.counters[C_COMPLEX], is_complex);
        if (is_complex) w.complex_[at] = s;
    }
}

// One wave (a block of its own) calls f(i) for every event i of the batch that was linked to slot s (ev_slot
// names the slot of a linked event).  The batch is read 768 events at a time, the next piece already on its
// way; the lanes put the indices that match into the queue q (768 entries in LDS) and then take them from it
// 64 at a time, so that whatever f loads is requested by all lanes together and not match by match.
template <class F>
__device__ inline void for_events_of_slot(const Work& w, uint32_t n, uint32_t s, uint32_t lane, uint32_t* q, F f) {
    constexpr int U = 3;
    const uint64_t quads = n >> 2;
    const uint4* v = reinterpret_cast<const uint4*>(w.ev_slot);
    const uint4 none = make_uint4(IDX_NONE, IDX_NONE, IDX_NONE, IDX_NONE);
    uint4 r[U], rn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t g = (uint64_t)u * 64 + lane;
        r[u] = g < quads ? v[g] : none;
    }
    for (uint64_t g0 = 0; g0 < quads; g0 += 64 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t g = g0 + 64 * U + (uint64_t)u * 64 + lane;
            rn[u] = g < quads ? v[g] : none;
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) cnt += (r[u].x == s) + (r[u].y == s) + (r[u].z == s) + (r[u].w == s);
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = __shfl(incl, 63);
        if (total) {                                  // the same in every lane
            uint32_t off = incl - cnt;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = (uint32_t)((g0 + (uint64_t)u * 64 + lane) << 2);
                if (r[u].x == s) q[off++] = i;
                if (r[u].y == s) q[off++] = i + 1;
                if (r[u].z == s) q[off++] = i + 2;
                if (r[u].w == s) q[off++] = i + 3;
            }
            __syncthreads();
            for (uint32_t k = lane; k < total; k += 64) f((uint64_t)q[k]);
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = rn[u];
    }
    for (uint64_t i = (quads << 2) + lane; i < n; i += 64)
        if (w.ev_slot[i] == s) f(i);
}

// Barriers with more events in the batch than the register replay takes: one wave each.  The result of the
// sequential replay is computed without sorting, from the first and last event index per
// position: the barrier completes at T = max over newly present positions of the first
// index touching it; events at or after T answer COMPLETE, the message of a position comes
// from its last index.
// This kernel is the serial form: lane 0 walks the chain.  It handles every long chain when the
// parallel kernels below are off, and the barriers they could not register when they are on.
__global__ void __launch_bounds__(64) join_resolve_complex_kernel(Table t, Work w, uint32_t n, uint64_t now, int32_t* status,
                                                                  const dj_join_rec* recs, const uint8_t* payloads, uint64_t pitch) {
    __shared__ uint32_t fmin[1024];
    __shared__ uint32_t fmax[1024];
    // the chain as lane 0 found it (link words and event indices): the answers are then written by all lanes
    constexpr uint32_t CHAIN_LDS = 256;
    // Links after which the walk is given up for a scan of the batch: a link costs about 0.6 us, the two scans
    // about 2.9 ns per event of the batch, so the walk of n / 512 links costs less than half of what follows.
    const uint32_t long_walk = max(256u, n >> 9);
    __shared__ uint64_t cbuf[CHAIN_LDS + CHAIN_LDS / 2];
    uint64_t* cn = cbuf;                                                   // link words
    uint32_t* ci = reinterpret_cast<uint32_t*>(cbuf + CHAIN_LDS);           // event indices
    uint32_t* scan_q = reinterpret_cast<uint32_t*>(cbuf);                   // the same 3 KB as a queue of 768 indices
    __shared__ uint32_t s_len;
    __shared__ uint16_t kx_order[CHAIN_LDS];
    const uint32_t lane = threadIdx.x;
    const uint32_t nc = w.counters[C_COMPLEX];
    uint32_t n_mid = 0;        // chains the wide register replay would have taken (lane 0 counts)
    if (w.fused) {
        // messages the register replay wants copied again (the last of several writers of one position)
        const uint32_t nre = w.counters[C_RECOPY];
        for (uint32_t k = blockIdx.x; k < nre; k += gridDim.x) {
            const uint32_t ev = (uint32_t)w.dst[k];
            const Slot* sl = &t.slots[w.ev_slot[ev]];
            const uint32_t kcls = slab_class(sl->count, sl->meta), slab = sl->slab;
            const uint4 b = reinterpret_cast<const uint4*>(recs)[2 * (size_t)ev + 1];     // position, len, max_wait
            const uint4* from = reinterpret_cast<const uint4*>(payloads + (uint64_t)ev * pitch);
            uint4* to = reinterpret_cast<uint4*>(slab_payload(t, slab, kcls) + (uint64_t)b.x * t.stride);
            const uint32_t chunks = (b.y + 15) >> 4, full = (chunks + 3) & ~3u;
            for (uint32_t c = lane; c < full; c += 64) to[c] = c < chunks ? from[c] : make_uint4(0, 0, 0, 0);
            if (lane == 0 && ((sl->meta & META_PMD) || b.y != t.stride || b.z != 0)) slab_posmeta(t, slab, kcls)[b.x] = make_uint2(b.y, b.z);
        }
    }
    for (uint32_t c = blockIdx.x; c < nc; c += gridDim.x) {
        if (w.cx_on && c < CX_CAP && w.cx_off[c] != IDX_NONE) continue;       // the parallel kernels have it
        const uint32_t s = w.complex_[c];
        Slot* sl = &t.slots[s];
        const uint32_t count = sl->count, slab = sl->slab, meta0 = sl->meta;
        const uint32_t kcls = slab_class(count, meta0);
        const bool wide = class_has_bitmap(kcls);
        const uint32_t head = sl->chain;
        for (uint32_t p = lane; p < count; p += 64) { fmin[p] = IDX_NONE; fmax[p] = 0; }    // count <= 999
        __syncthreads();
        uint64_t at_min = NQ;      // smallest now + 1000*max-wait over the batch's waiting events
        if (lane == 0) {
            uint32_t len = 0, cur = head;
            for (; cur != 0 && len < long_walk; ++len) {
                const uint64_t x = w.node[cur - 1];
                const uint32_t i = cur - 1, pos = node_pos(x);
                if (i < fmin[pos]) fmin[pos] = i;
                if (i + 1 > fmax[pos]) fmax[pos] = i + 1;
                if (len < CHAIN_LDS) { cn[len] = x; ci[len] = i; }
                cur = node_next(x);
            }
            s_len = cur != 0 ? IDX_NONE : len;       // IDX_NONE: gave up, the wave scans the batch instead
            // the host picks the parallel kernels for the next batch from this; thousands of barriers report, so
            // the atomic (they queue up on the one word) is only issued by a chain longer than the longest so far
            if (len > a_load(&w.counters[C_MAXCHAIN])) a_max(&w.counters[C_MAXCHAIN], len);
            if (len >= 256) sl->meta |= META_HOT; else sl->meta &= ~META_HOT;
            if (len <= CHAIN_REG_WIDE) ++n_mid;
        }
        __syncthreads();
        const bool scan = s_len == IDX_NONE;
        if (scan) {
            // A chain of thousands of links on a barrier nobody knew to be hot (the first batch of a replay storm):
            // one lane would walk it at about 0.6 us per link.  The events of a barrier are exactly those whose
            // ev_slot names its slot, so the wave reads that array once instead, four events per lane and step.
            uint32_t mine = 0;
            for_events_of_slot(w, n, s, lane, scan_q, [&](uint64_t i) {
                const uint32_t pos = node_pos(w.node[i]);
                atomicMin(&fmin[pos], (uint32_t)i);
                atomicMax(&fmax[pos], (uint32_t)i + 1);
                ++mine;
            });
            for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
            if (lane == 0) {
                if (mine > a_load(&w.counters[C_MAXCHAIN])) a_max(&w.counters[C_MAXCHAIN], mine);
                sl->meta |= META_HOT;
            }
            __syncthreads();
        }
        const bool one_by_one = (meta0 & META_KX_MASK) != 0 || w.hlen != 0;
        if (one_by_one) {
            // The barrier holds messages beyond its count (a failed publish wrote an older, smaller count back): HLEN
            // can pass the count and come back, so there is no single first COMPLETE index.  Lane 0 replays the events
            // one by one in array order, as core/join would (core.clj:44-58).  The same replay serves callers that want
            // HLEN with every answer: it is the one form that has the hash's size after each single event.
            const uint32_t len = s_len;
            if (!scan && len <= CHAIN_LDS) {
                // order of the chain's entries by event index: entry k has as many entries before it as are smaller
                for (uint32_t k = lane; k < len; k += 64) {
                    uint32_t r = 0;
                    for (uint32_t j = 0; j < len; ++j) r += ci[j] < ci[k] ? 1u : 0u;
                    kx_order[r] = (uint16_t)k;
                }
            }
            __syncthreads();
            uint64_t bits1 = sl->bits, pa = q_primary_at(t.q[s]);
            uint32_t arrived = wide ? (uint32_t)bits1 : (uint32_t)__popcll(bits1);
            bool enq = false, dirty = false;
            auto step = [&](uint64_t x, uint32_t i) {           // lane 0 only
                const uint32_t pos = node_pos(x), mw = node_mw(x);
                if (wide) {
                    uint64_t* bm = slab_bitmap(t, slab);
                    const uint64_t word = bm[pos >> 6], bit = 1ull << (pos & 63);
                    if (!(word & bit)) { bm[pos >> 6] = word | bit; ++arrived; }
                } else {
                    const uint64_t bit = 1ull << pos;
                    if (!(bits1 & bit)) { bits1 |= bit; ++arrived; }
                }
                int32_t st = DJ_JOIN_WAITING;
                if (arrived == count) {                          // all-there?, then queued at now (core.clj:52-54)
                    st = DJ_JOIN_COMPLETE;
                    enq = true;
                    pa = now;
                } else if (mw) {                                 // join-with-max-wait (core.clj:40-42, redis.clj:262-275)
                    if (pa == NQ) st = DJ_JOIN_NEAREST_NOT_QUEUED;
                    else if (pa > now + 1000ull * mw) { pa = now + 1000ull * mw; dirty = true; }
                }
                if (w.hlen) status[i] = st | (int32_t)(arrived << DJ_JOIN_HLEN_SHIFT);
                else if (st != DJ_JOIN_WAITING) status[i] = st;
                if (!w.fused && fmax[pos] != i + 1) w.dst[i] = NODST;
            };
            if (!scan && len <= CHAIN_LDS) {
                if (lane == 0)
                    for (uint32_t r = 0; r < len; ++r) step(cn[kx_order[r]], ci[kx_order[r]]);
            } else {
                // a chain too long for that: the batch in order, 64 events at a time (slow, and as rare as a replay
                // storm on a barrier in this state)
                for (uint64_t base = 0; base < n; base += 64) {
                    const uint64_t i = base + lane;
                    uint64_t m = __ballot(i < n && w.ev_slot[i] == s);
                    while (m) {
                        const int b = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        if (lane == 0) step(w.node[base + b], (uint32_t)(base + b));
                    }
                }
            }
            if (lane == 0) {
                sl->bits = wide ? (uint64_t)arrived : bits1;
                sl->join_ts = now;
                sl->chain = 0;
                if (enq) { t.q[s] = now; ready_append_one(t, s, blockIdx.x); }
                else if (dirty) { t.q[s] = pa; note_queue_date(t, s, pa); }
            }
        }
        const bool plain = !one_by_one;
        // newly present positions: 64 positions (one bitmap word) per step, one lane per position
        uint32_t t_complete = 0, n_new = 0;
        const uint32_t words = plain ? (count + 63) >> 6 : 0;          // (the replay above has done all of this otherwise)
        const uint64_t bits_pre = sl->bits;
        uint64_t bits0 = bits_pre;
        for (uint32_t wd = 0; wd < words; ++wd) {
            const uint32_t p = wd * 64 + lane;
            const uint64_t word = wide ? slab_bitmap(t, slab)[wd] : bits_pre;
            const uint32_t f = p < count ? fmin[p] : IDX_NONE;
            const bool fresh = f != IDX_NONE && !((word >> lane) & 1);
            if (fresh && f > t_complete) t_complete = f;
            const uint64_t add = __ballot(fresh);
            n_new += (uint32_t)__popcll(add);          // the same in every lane
            if (add) {
                if (wide) { if (lane == 0) slab_bitmap(t, slab)[wd] = word | add; }
                else bits0 = word | add;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(t_complete, off);
            if (o > t_complete) t_complete = o;
        }
        const uint32_t arrived_pre = wide ? (uint32_t)bits_pre : (uint32_t)__popcll(bits_pre);
        const uint32_t arrived = arrived_pre + n_new;
        // first event index that answers COMPLETE
        uint32_t T = IDX_NONE;
        if (arrived_pre == count) T = 0;
        else if (arrived == count) T = t_complete;
        const uint64_t pa_pre = q_primary_at(t.q[s]);
        const uint32_t len = s_len;
        // one event's answer; returns its queue date when it waits with a max-wait, else NQ
        auto answer = [&](uint64_t x, uint32_t i) -> uint64_t {
            const uint32_t pos = node_pos(x), mw = node_mw(x);
            int32_t st = DJ_JOIN_WAITING;
            uint64_t at = NQ;
            if (i >= T && T != IDX_NONE) st = DJ_JOIN_COMPLETE;
            else if (mw && pa_pre == NQ) st = DJ_JOIN_NEAREST_NOT_QUEUED;
            else if (mw) at = now + 1000ull * mw;
            if (st != DJ_JOIN_WAITING) status[i] = st;
            if (!w.fused && fmax[pos] != i + 1) w.dst[i] = NODST;
            return at;
        };
        if (!plain) {
        } else if (scan) {                        // all lanes, over the batch once more
            for_events_of_slot(w, n, s, lane, scan_q, [&](uint64_t i) {
                const uint64_t at = answer(w.node[i], (uint32_t)i);
                if (at < at_min) at_min = at;
            });
        } else if (len <= CHAIN_LDS) {            // all lanes, from the copy in LDS
            for (uint32_t k = lane; k < len; k += 64) {
                const uint64_t at = answer(cn[k], ci[k]);
                if (at < at_min) at_min = at;
            }
        } else if (lane == 0) {                   // longer than the copy: walk again
            for (uint32_t cur = head; cur != 0;) {
                const uint64_t x = w.node[cur - 1];
                const uint64_t at = answer(x, cur - 1);
                if (at < at_min) at_min = at;
                cur = node_next(x);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(at_min >> 32), off) << 32) | __shfl_xor((uint32_t)at_min, off);
            if (o < at_min) at_min = o;
        }
        if (w.fused) {
            // every event has copied its message already, in no order: a position that several events of the batch
            // wrote gets its last writer's message again (HSET: the last write stays), by the whole wave
            const bool stored = (sl->meta & META_PMD) != 0;
            for (uint32_t p = 0; p < count; ++p) {
                const uint32_t last = fmax[p];
                if (last == 0 || fmin[p] + 1 == last) continue;           // untouched, or one writer
                const uint4 b = reinterpret_cast<const uint4*>(recs)[2 * (size_t)(last - 1) + 1];     // position, len, max_wait
                const uint4* from = reinterpret_cast<const uint4*>(payloads + (uint64_t)(last - 1) * pitch);
                uint4* to = reinterpret_cast<uint4*>(slab_payload(t, slab, kcls) + (uint64_t)p * t.stride);
                const uint32_t chunks = (b.y + 15) >> 4, full = (chunks + 3) & ~3u;
                for (uint32_t c = lane; c < full; c += 64) to[c] = c < chunks ? from[c] : make_uint4(0, 0, 0, 0);
                if (lane == 0 && (stored || b.y != t.stride || b.z != 0)) slab_posmeta(t, slab, kcls)[p] = make_uint2(b.y, b.z);
            }
        }
        if (lane == 0 && plain) {
            sl->bits = wide ? (uint64_t)arrived : bits0;
            sl->join_ts = now;
            sl->chain = 0;
            if (T != IDX_NONE) { t.q[s] = now; ready_append_one(t, s, blockIdx.x); }
            else if (pa_pre != NQ && at_min < pa_pre) { t.q[s] = at_min; note_queue_date(t, s, at_min); }
        }
        __syncthreads();
    }
    if (lane == 0 && n_mid) atomicAdd(&w.counters[C_MID], n_mid);            // one atomic per block
}

// ---- parallel form for hot barriers ----------------------------------------------------------
// A chain of thousands of events (one hot barrier, at-least-once replays) is too long to walk with
// one lane.  Instead: cx_init registers the hot barriers in a small hash keyed by slab;
// cx_scan lets EVERY event of the batch look its barrier up there and fold its index into the
// per-position first / last tables with atomics; cx_finalize (one wave per hot barrier) derives
// the completion index T and updates the barrier; the copy pass then answers the hot barriers'
// events from T and the last-writer table.  Same results as the serial kernel.

__device__ inline uint32_t cx_hash(uint32_t slab) { return (slab * 2654435761u) & (CX_HASH - 1); }

__device__ inline uint32_t cx_lookup(const Work& w, uint32_t slab) {
    for (uint32_t h = cx_hash(slab);; h = (h + 1) & (CX_HASH - 1)) {
        const uint32_t k = w.cx_keys[h];
        if (k == slab) return w.cx_val[h];
        if (k == 0) return IDX_NONE;
    }
}

constexpr int CX_BPB = 16;      // barriers per block in cx_init

// 16 barriers per block: their shares of the pool come from ONE atomic per block (a wave scan of the
// counts), then all threads clear the shares.
__global__ void __launch_bounds__(256) cx_init_kernel(Table t, Work w) {
    __shared__ uint32_t s_off[CX_BPB], s_cnt[CX_BPB];
    const uint32_t nc = min(w.counters[C_COMPLEX], CX_CAP);
    const uint32_t c0 = blockIdx.x * CX_BPB;
    if (c0 >= nc) return;
    const uint32_t tid = threadIdx.x;
    if (tid < 64) {
        const uint32_t c = c0 + tid;
        const bool has = tid < CX_BPB && c < nc;
        uint32_t count = 0, slab = 0, hot = 0;
        bool serial_only = false;                  // a barrier whose count went back: replayed event by event, see the wave kernel
        if (has) {
            const Slot* sl = &t.slots[w.complex_[c]];
            count = sl->count;
            slab = sl->slab;                       // unique per barrier, never 0 for a joinable barrier
            hot = (sl->meta & META_HOT) ? 1u : 0u;
            serial_only = (sl->meta & META_KX_MASK) != 0;
            if (serial_only) count = 0;            // no share of the pool
        }
        // a barrier already known to be hot spreads its positions over one 64-byte line each: thousands
        // of atomics on ONE line would queue up
        const uint32_t span = hot ? cx_hot_span(count) : 0u;
        const uint32_t share = hot ? count * span : count;
        const uint32_t incl = wave_incl_scan(share);
        const uint32_t total = __shfl(incl, 63);
        uint32_t base = 0;
        if (tid == 63) base = atomicAdd(&w.counters[C_CXPOOL], total);
        base = __shfl(base, 63);
        if (tid < CX_BPB) {
            uint32_t off = IDX_NONE;
            if (has) {
                off = base + incl - share;
                if (serial_only || off > CX_POOL - share) off = IDX_NONE;          // (or pool exhausted:) left to the serial kernel
                w.cx_off[c] = off;
                if (off != IDX_NONE) {
                    w.cx_atmin[c] = NQ;
                    w.cx_cnt[c] = 0;
                    w.cx_hot[c] = span;
                    for (uint32_t h = cx_hash(slab);; h = (h + 1) & (CX_HASH - 1)) {
                        if (atomicCAS(&w.cx_keys[h], 0u, slab) == 0u) { w.cx_val[h] = c; break; }
                    }
                }
            }
            s_off[tid] = off;
            s_cnt[tid] = share;
        }
    }
    __syncthreads();
    for (int j = 0; j < CX_BPB; ++j) {
        const uint32_t off = s_off[j];
        if (off == IDX_NONE) continue;
        for (uint32_t p = tid; p < s_cnt[j]; p += 256) w.cx_mm[off + p] = make_uint2(IDX_NONE, 0);
    }
}

__global__ void __launch_bounds__(256) cx_scan_kernel(Work w, uint32_t n, uint64_t now) {
    if (w.counters[C_COMPLEX] == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) - lane;
    // every wave walks whole groups of 64 consecutive events (uniform trip count: the ballots below need it)
    for (uint64_t base = wave0; base < n; base += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = base + lane;
        uint32_t c = IDX_NONE, key = 0, mw = 0;
        if (i < n) {
            const uint64_t d = w.dst[i];
            if (d != NODST) c = cx_lookup(w, (uint32_t)d);
            if (c != IDX_NONE) {
                w.dst[i] = d | DST_CX;             // the copy pass looks the barrier up again only for these
                const uint64_t x = w.node[i];
                // first and last index share 8 bytes; a hot barrier's position may have several lines, picked by wave
                const uint32_t span = w.cx_hot[c];
                key = span ? w.cx_off[c] + node_pos(x) * span + (uint32_t)((base >> 6) & (cx_span_replicas(span) - 1)) * CX_HOT_STRIDE
                           : w.cx_off[c] + node_pos(x);
                mw = node_mw(x);
            }
        }
        const bool live = c != IDX_NONE;
        const bool hot = live && w.cx_hot[c] != 0;
        if (live && !hot) {
            atomicMin(&w.cx_mm[key].x, (uint32_t)i);
            atomicMax(&w.cx_mm[key].y, (uint32_t)i + 1);
            atomicAdd(&w.cx_cnt[c], 1u);
        }
        // a barrier known to be hot: the lanes that hit the same position fold into one set of atomics
        // (their indices grow with the lane, so the first lane has the smallest, the last the largest)
        uint64_t todo = __ballot(hot);
        while (todo) {
            const int l0 = __ffsll((unsigned long long)todo) - 1;
            const uint32_t key0 = __shfl(key, l0);
            const uint64_t grp = __ballot(hot && key == key0);
            if ((int)lane == l0) {
                const int last = 63 - __clzll((unsigned long long)grp);
                atomicMin(&w.cx_mm[key].x, (uint32_t)i);
                atomicMax(&w.cx_mm[key].y, (uint32_t)(base + last) + 1);
                // the event count only steers heuristics (stay hot? which kernels next batch?): one wave
                // in 16 reports, scaled, instead of every wave hammering one word
                if (((base >> 6) & 15) == 0) atomicAdd(&w.cx_cnt[c], 16u * (uint32_t)__popcll(grp));
            }
            todo &= ~grp;
        }
        // earliest queue date among the barrier's waiting events.  A hot barrier's events would queue up on the one
        // word: its lanes reduce inside the wave first, and only an earlier date than the one already there is
        // worth an atomic (after the first few waves almost none is).
        if (live && !hot && mw) {
            const unsigned long long at = now + 1000ull * mw;
            if (at < a_load(&w.cx_atmin[c])) atomicMin(&w.cx_atmin[c], at);
        }
        uint64_t todo_mw = __ballot(hot && mw);
        while (todo_mw) {
            const int l0 = __ffsll((unsigned long long)todo_mw) - 1;
            const uint32_t c0 = __shfl(c, l0);
            const bool mine = hot && mw && c == c0;
            uint32_t m = mine ? mw : 0xFFFFFFFFu;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(m, off);
                if (o < m) m = o;
            }
            if ((int)lane == l0) {
                const unsigned long long at = now + 1000ull * m;
                if (at < a_load(&w.cx_atmin[c0])) atomicMin(&w.cx_atmin[c0], at);
            }
            todo_mw &= ~__ballot(mine);
        }
    }
}

// one wave per barrier; 64 consecutive positions (one bitmap word) per step, read coalesced
__global__ void __launch_bounds__(64) cx_finalize_kernel(Table t, Work w, uint64_t now) {
    const uint32_t lane = threadIdx.x;
    const uint32_t nc = min(w.counters[C_COMPLEX], CX_CAP);
    uint32_t longest = 0, n_mid = 0;
    for (uint32_t c = blockIdx.x; c < nc; c += gridDim.x) {
        const uint32_t off = w.cx_off[c];
        if (off == IDX_NONE) continue;
        const uint32_t s = w.complex_[c];
        Slot* sl = &t.slots[s];
        const uint32_t count = sl->count, slab = sl->slab;
        const uint64_t bits_pre = sl->bits;
        const uint32_t span = w.cx_hot[c];
        const bool hot = span != 0;
        const uint32_t pstride = hot ? span : 1u, replicas = hot ? cx_span_replicas(span) : 1u;
        uint32_t t_complete = 0, n_new = 0;
        uint64_t bits0 = bits_pre;
        const uint32_t words = (count + 63) >> 6;
        for (uint32_t wd = 0; wd < words; ++wd) {
            const uint32_t p = wd * 64 + lane;
            const uint64_t word = count > 64 ? slab_bitmap(t, slab)[wd] : bits_pre;
            uint32_t f = IDX_NONE;
            if (p < count) {
                uint2 mm = w.cx_mm[off + p * pstride];
                if (hot) {      // fold the position's lines into the first: the copy pass reads the last index there
                    for (uint32_t r = 1; r < replicas; ++r) {
                        const uint2 o = w.cx_mm[off + p * pstride + r * CX_HOT_STRIDE];
                        if (o.x < mm.x) mm.x = o.x;
                        if (o.y > mm.y) mm.y = o.y;
                    }
                    w.cx_mm[off + p * pstride] = mm;
                }
                f = mm.x;
            }
            const bool fresh = f != IDX_NONE && !((word >> lane) & 1);
            if (fresh && f > t_complete) t_complete = f;
            const uint64_t add = __ballot(fresh);
            n_new += (uint32_t)__popcll(add);          // the same in every lane
            if (add) {
                if (count > 64) { if (lane == 0) slab_bitmap(t, slab)[wd] = word | add; }
                else bits0 = word | add;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t v = __shfl_xor(t_complete, o);
            if (v > t_complete) t_complete = v;
        }
        if (lane == 0) {
            const uint32_t arrived_pre = count > 64 ? (uint32_t)bits_pre : (uint32_t)__popcll(bits_pre);
            const uint32_t arrived = arrived_pre + n_new;
            uint32_t T = IDX_NONE;
            if (arrived_pre == count) T = 0;
            else if (arrived == count) T = t_complete;
            const uint64_t pa_pre = q_primary_at(t.q[s]);
            const uint64_t at_min = w.cx_atmin[c];
            w.cx_T[c] = T;
            w.cx_nq[c] = pa_pre == NQ;
            sl->bits = count > 64 ? (uint64_t)arrived : bits0;
            sl->join_ts = now;
            sl->chain = 0;
            // events before T wait (store-nearest, redis.clj:269-271); from T on the barrier is queued at now
            if (T != IDX_NONE) { t.q[s] = now; ready_append_one(t, s, blockIdx.x); }
            else if (pa_pre != NQ && at_min < pa_pre) { t.q[s] = at_min; note_queue_date(t, s, at_min); }
            const uint32_t events = w.cx_cnt[c];
            if (events > longest) longest = events;
            if (events <= CHAIN_REG_WIDE) ++n_mid;
            if (events >= 256) sl->meta |= META_HOT; else sl->meta &= ~META_HOT;
        }
    }
    if (lane == 0 && longest) a_max(&w.counters[C_MAXCHAIN], longest);       // one atomic per block
    if (lane == 0 && n_mid) atomicAdd(&w.counters[C_MID], n_mid);
}

// pass 3: messages of surviving events into the barrier's slab.  G lanes cooperate on one
// message with 16-byte loads and stores; a wave handles 64/G events per step.
template <int G>
__global__ void __launch_bounds__(256) join_copy_kernel(Table t, Work w, const dj_join_rec* recs,
                                                        const uint8_t* payloads, uint64_t pitch, uint32_t n,
                                                        int32_t* status) {
    constexpr int EPW = 64 / G;                       // events per wave step
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t sub = lane / G, j0 = lane % G;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool hot = w.cx_on && w.counters[C_COMPLEX] != 0;
    for (uint64_t base = wave * EPW; base < n; base += waves * EPW) {
        const uint64_t i = base + sub;
        if (i >= n) continue;
        const uint64_t d = w.dst[i];
        if (d == NODST) continue;
        const uint32_t slab = (uint32_t)d, k = dst_class(d);
        // position and "full row" travel in d; the record is read only when more is needed
        uint4 b = make_uint4(dst_pos(d), t.stride, 0, 0);                    // position, len, max_wait
        const bool cx = hot && (d & DST_CX);
        if (cx || (d & DST_PM) || !(d & DST_FULL)) b = reinterpret_cast<const uint4*>(recs)[2 * i + 1];
        if (cx) {
            // an event of a hot barrier gets its answer here, from the tables of the cx_* kernels
            const uint32_t c = cx_lookup(w, slab);
            if (c != IDX_NONE) {
                const uint32_t T = w.cx_T[c];
                if (j0 == 0) {
                    if (T != IDX_NONE && i >= T) status[i] = DJ_JOIN_COMPLETE;
                    else if (b.z && w.cx_nq[c]) status[i] = DJ_JOIN_NEAREST_NOT_QUEUED;
                }
                if (w.cx_mm[w.cx_off[c] + b.x * (w.cx_hot[c] ? w.cx_hot[c] : 1u)].y != (uint32_t)i + 1) continue;   // a later event of the batch wrote this position
            }
        }
        if (j0 == 0 && (d & DST_PM)) slab_posmeta(t, slab, k)[b.x] = make_uint2(b.y, b.z);
        const uint4* src = reinterpret_cast<const uint4*>(payloads + i * pitch);
        uint4* dstp = reinterpret_cast<uint4*>(slab_payload(t, slab, k) + (uint64_t)b.x * t.stride);
        // the row is written in whole 64-byte lines (zeros behind the message): a partly written line costs a
        // read-modify-write in memory; the row itself is a multiple of 64 bytes
        const uint32_t chunks = (b.y + 15) >> 4, full = (chunks + 3) & ~3u;
        for (uint32_t c = j0; c < full; c += G) nt_store(&dstp[c], c < chunks ? nt_load(&src[c]) : make_uint4(0, 0, 0, 0));
    }
}

// The same pass for engines that accept messages longer than one inline row (max_message > stride).
// A long message gets a block of its own from the arena (wave-cooperative alloc / free, a block of
// the same size class is reused when a position is overwritten); the row keeps a dj_long_ref and
// posmeta the true length.  Long messages are copied by the whole wave, one after the other.
template <int G>
__global__ void __launch_bounds__(256) join_copy_long_kernel(Table t, Work w, const dj_join_rec* recs,
                                                             const uint8_t* payloads, uint64_t pitch, uint32_t n,
                                                             int32_t* status) {
    constexpr int EPW = 64 / G;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t sub = lane / G, j0 = lane % G;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool hot = w.cx_on && w.counters[C_COMPLEX] != 0;
    for (uint64_t base = wave * EPW; base < n; base += waves * EPW) {      // uniform across the wave
        const uint64_t i = base + sub;
        bool live = i < n;
        uint64_t d = NODST;
        uint4 b = make_uint4(0, 0, 0, 0);
        if (live) {
            d = w.dst[i];
            live = d != NODST;
        }
        const uint32_t slab = (uint32_t)d, k = live ? dst_class(d) : 0;
        if (live) {
            b = make_uint4(dst_pos(d), t.stride, 0, 0);            // position, len, max_wait
            const bool cx = hot && (d & DST_CX);
            if (cx || (d & DST_PM) || !(d & DST_FULL)) b = reinterpret_cast<const uint4*>(recs)[2 * i + 1];
            if (cx) {
                const uint32_t c = cx_lookup(w, slab);
                if (c != IDX_NONE) {
                    const uint32_t T = w.cx_T[c];
                    if (j0 == 0) {
                        if (T != IDX_NONE && i >= T) status[i] = DJ_JOIN_COMPLETE;
                        else if (b.z && w.cx_nq[c]) status[i] = DJ_JOIN_NEAREST_NOT_QUEUED;
                    }
                    if (w.cx_mm[w.cx_off[c] + b.x * (w.cx_hot[c] ? w.cx_hot[c] : 1u)].y != (uint32_t)i + 1) live = false;
                }
            }
        }
        // the first lane of each event decides what happens to the position's block
        const bool lead = live && j0 == 0;
        const bool new_long = live && b.y > t.stride;
        bool want = false, give_back = false;
        uint32_t k_new = 0, k_old = 0, unit_old = 0, unit_new = 0;
        uint8_t* row = nullptr;
        if (lead) {
            row = slab_payload(t, slab, k) + (uint64_t)b.x * t.stride;
            // only a barrier that held long messages before this batch can have one at this position
            const uint32_t old_len = (d & DST_OVF) ? slab_posmeta(t, slab, k)[b.x].x : 0u;
            const bool old_long = old_len > t.stride && old_len != DJ_NIL;
            if (old_long) {
                const dj_long_ref ref = *reinterpret_cast<const dj_long_ref*>(row);
                unit_old = ref.unit;
                k_old = ref.size_class;
            }
            if (new_long) {
                k_new = ovf_class(b.y);
                if (old_long && k_old == k_new) unit_new = unit_old;      // same size class: write in place
                else { want = true; give_back = old_long; }
            } else {
                give_back = old_long;
            }
        }
        wave_ovf_free(t, give_back, k_old, unit_old);
        const uint32_t fresh = wave_ovf_alloc(t, want, k_new);
        if (lead) {
            if (want) unit_new = fresh;
            uint32_t len = b.y;
            if (new_long && unit_new == 0) {                  // arena exhausted: the position keeps an empty message
                status[i] = DJ_JOIN_NO_SPACE;
                len = 0;
            }
            if (d & DST_PM) slab_posmeta(t, slab, k)[b.x] = make_uint2(len, b.z);    // always set for a long message
            if (new_long && unit_new != 0) {
                dj_long_ref ref;
                ref.unit = unit_new; ref.size_class = k_new; ref.len = b.y; ref.zero = 0;
                *reinterpret_cast<dj_long_ref*>(row) = ref;
                if (!(d & DST_OVF)) atomicOr(&t.slots[w.ev_slot[i]].meta, META_OVF);   // not flagged yet at link time
            }
        }
        if (live && !new_long) {
            const uint4* src = reinterpret_cast<const uint4*>(payloads + i * pitch);
            uint4* dstp = reinterpret_cast<uint4*>(slab_payload(t, slab, k) + (uint64_t)b.x * t.stride);
            const uint32_t chunks = (b.y + 15) >> 4, full = (chunks + 3) & ~3u;     // whole lines, as in join_copy
            for (uint32_t c = j0; c < full; c += G) nt_store(&dstp[c], c < chunks ? nt_load(&src[c]) : make_uint4(0, 0, 0, 0));
        }
        // long messages: the whole wave copies one after the other
        uint64_t todo = __ballot(lead && new_long && unit_new != 0);
        while (todo) {
            const int l = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t ev_lo = __shfl((uint32_t)i, l), ev_hi = __shfl((uint32_t)(i >> 32), l);
            const uint64_t ev = ((uint64_t)ev_hi << 32) | ev_lo;
            const uint32_t unit = __shfl(unit_new, l), len = __shfl(b.y, l);
            const uint4* src = reinterpret_cast<const uint4*>(payloads + ev * pitch);
            uint4* dstp = reinterpret_cast<uint4*>(t.arena + (uint64_t)unit * 64);
            const uint32_t chunks = (len + 15) >> 4;
            for (uint32_t c = lane; c < chunks; c += 64) nt_store(&dstp[c], nt_load(&src[c]));
        }
    }
}

// Copies long messages out of the arena (dj_fetch_long): one wave per reference.
__global__ void __launch_bounds__(256) fetch_long_kernel(Table t, uint32_t n, const dj_long_ref* refs,
                                                         const uint64_t* offsets, uint8_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n; i += waves) {
        const dj_long_ref r = refs[i];
        const uint64_t block_bytes = 64ull << (r.size_class < OVF_CLASSES ? r.size_class : 0);
        if (r.size_class >= OVF_CLASSES || r.unit == 0 || r.len > block_bytes ||
            (uint64_t)r.unit + (block_bytes >> 6) > t.arena_units || (offsets[i] & 15))
            continue;                                          // not a reference into this arena
        const uint4* src = reinterpret_cast<const uint4*>(t.arena + (uint64_t)r.unit * 64);
        uint4* dstp = reinterpret_cast<uint4*>(out + offsets[i]);
        const uint32_t chunks = (r.len + 15) >> 4;
        for (uint32_t c = lane; c < chunks; c += 64) dstp[c] = src[c];
    }
}

// =====================================================================================
// sweep: redis/all-publish-messages (redis.clj:283-289) then, per popped barrier,
// join-metadata, timeout classification (core.clj:86-95) and redis/all-joins (redis.clj:210-243)
// =====================================================================================

__device__ inline void fill_cont(dj_cont* c, const Slot& sl, uint64_t now, uint64_t publish_at, uint64_t pos_base,
                                 uint64_t slot_hint = 0) {
    c->fp_lo = sl.fp_lo;
    c->fp_hi = sl.fp_hi;
    c->publish_at = publish_at;
    c->pos_base = pos_base;
    c->reserved = slot_hint;     // slot index + 1: lets the ack find the barrier without probing
    if (sl.meta & META_HAS) {
        c->deadline_ms = sl.deadline;
        c->count = sl.count;
        c->flags = now > sl.deadline ? DJ_CONT_TIMEOUT : 0;     // is-timeout?, strict (core.clj:86-89)
        c->retry_publish = retry_to_api(meta_retry_pub(sl.meta));
        c->retry_timeout = retry_to_api(meta_retry_to(sl.meta));
    } else {
        c->deadline_ms = 0;
        c->count = 0;
        c->flags = DJ_CONT_NO_METADATA;
        c->retry_publish = DJ_NIL;
        c->retry_timeout = DJ_NIL;
    }
}

// The pop of a sweep: every slot with q <= now (in the primary queue, score <= now) is popped
// (ZRANGEBYSCORE + ZREMRANGEBYSCORE 0 now).  Two forms, picked on the device, the same for every block:
//   drain  now has not reached the scan bound *due_cur: whatever is due sits in the ready list (barriers
//          that join batches completed since the last sweep).  A block takes whole shards of the list; an
//          entry is popped when its queue word says so, decided by a compare-and-swap (the list may name a
//          slot twice, or a slot whose barrier was re-forked or deleted since).
//   scan   one pass over q[].  A block scans chunks of 256 * POP_ITEMS queue words with POP_ITEMS loads
//          in flight per thread and collects the due slots in a list in LDS.  The pass also recomputes the
//          bound: the smallest primary score that stays queued goes to *due_next.
// Either way the due slots are then worked off one entry per thread (one slot read each, descriptors
// written to consecutive indices).  Output indices come from a block-wide prefix sum and ONE atomic on the
// packed cursor word per 256 entries.  The cursors are raw (they may run past the caps); the gather pass
// (or finish_result_kernel) publishes the clamped dj_sweep_result.
constexpr int POP_ITEMS = 16;

struct PopShared {
    uint32_t wave_c[4], wave_p[4];
    unsigned long long base_s;
};

// One round of up to 256 due slots, one per thread (has).  popped: the queue word already carries Q_POPPED
// (drain form); a barrier that does not fit the caller's buffers is then put back.
__device__ inline void pop_emit_round(const Table& t, const Work& w, PopShared& sh, bool has, uint64_t s, uint64_t pa,
                                      bool popped, uint64_t now, uint32_t cap_conts, uint64_t cap_pos, int alloc_pos,
                                      dj_cont* conts) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t cnt = 0;
    Slot sl;
    if (has) {
        sl = t.slots[s];
        cnt = (alloc_pos && (sl.meta & META_HAS)) ? sl.count : 0;
    }
    const uint32_t incl_c = wave_incl_scan(has ? 1u : 0u), incl_p = wave_incl_scan(cnt);
    if (lane == 63) { sh.wave_c[wv] = incl_c; sh.wave_p[wv] = incl_p; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long tc = sh.wave_c[0] + sh.wave_c[1] + sh.wave_c[2] + sh.wave_c[3];
        const unsigned long long tp = sh.wave_p[0] + sh.wave_p[1] + sh.wave_p[2] + sh.wave_p[3];
        sh.base_s = tc ? a_add(&w.sweep[S_CURSOR], tc | (tp << CURSOR_SHIFT)) : 0ull;
    }
    __syncthreads();
    uint32_t c_idx = (uint32_t)(sh.base_s & CURSOR_CONT_MASK) + incl_c - 1;
    uint64_t p_idx = (sh.base_s >> CURSOR_SHIFT) + incl_p - cnt;
    for (uint32_t v = 0; v < wv; ++v) { c_idx += sh.wave_c[v]; p_idx += sh.wave_p[v]; }
    if (has) {
        bool fits = true;
        if (c_idx >= cap_conts) {
            fits = false;
        } else if (alloc_pos && p_idx + cnt > cap_pos) {
            fits = false;
            dj_cont v = {};                      // index taken, barrier stays queued
            v.flags = CONT_VOID;
            conts[c_idx] = v;
            w.cont_slot[c_idx] = 0xFFFFFFFFu;
        }
        if (fits) {
            if (!popped) t.q[s] = pa | Q_POPPED;     // popped; the backup entry stays (redis.clj:288)
            fill_cont(&conts[c_idx], sl, now, pa, p_idx, s + 1);
            w.cont_slot[c_idx] = (uint32_t)s;
        } else {
            w.sweep[S_MORE] = 1;
            if (popped) t.q[s] = pa;
            // still due: the next sweep has to look for it (its region's bound may just have been raised past it)
            if (pa < a_load(w.due_next)) __hip_atomic_fetch_min(w.due_next, (unsigned long long)pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t.ready_on) __hip_atomic_fetch_min(&t.qmin[s >> REGION_SHIFT], (unsigned long long)pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// drain form (launched only by engines that keep the ready list)
__global__ void __launch_bounds__(256) pop_ready_kernel(Table t, Work w, uint64_t now, uint32_t cap_conts,
                                                        uint64_t cap_pos, int alloc_pos, dj_cont* conts) {
    __shared__ PopShared sh;
    const uint32_t tid = threadIdx.x;
    if (*t.ready_ovf != 0) return;      // the list is incomplete: every queue word is read instead (nobody writes this word during a sweep's pop kernels)
    // block b: shard b mod READY_SHARDS, every (gridDim.x / READY_SHARDS)-th piece of 256 entries of it
    const uint32_t shard = blockIdx.x & (READY_SHARDS - 1), per_shard = gridDim.x / READY_SHARDS;    // grid: a multiple of READY_SHARDS
    const uint32_t filled = t.ready_n[shard * READY_PAD];
    const uint32_t ln = filled < t.ready_shard_cap ? filled : t.ready_shard_cap;
    const uint32_t* entries = t.ready + (size_t)shard * t.ready_shard_cap;
    for (uint32_t r = (blockIdx.x / READY_SHARDS) * 256; r < ln; r += per_shard * 256) {     // uniform across the block
        const uint32_t e = r + tid;
        bool has = e < ln;
        uint64_t s = 0, pa = NQ;
        if (has) {
            s = entries[e];
            pa = a_load(&t.q[s]);
            if (pa <= now) {
                has = a_cas(&t.q[s], pa, pa | Q_POPPED);       // false: another entry of the same slot got it
            } else {
                has = false;
                // queued for later (a re-fork, or a caller whose clock went back): the list is emptied after this sweep,
                // so both bounds must cover the date from here on (a completion writes its date to the list only, and
                // its region's bound may still be the fork's deadline)
                if (!(pa >> 63)) {
                    if (pa < a_load(w.due_next))
                        __hip_atomic_fetch_min(w.due_next, (unsigned long long)pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long* rb = &t.qmin[s >> REGION_SHIFT];
                    if (pa < a_load(rb)) __hip_atomic_fetch_min(rb, (unsigned long long)pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        pop_emit_round(t, w, sh, has, s, pa, true, now, cap_conts, cap_pos, alloc_pos, conts);
    }
}

// The due slots a block has found and not yet worked off.  A block collects them over ALL the chunks it scans and
// works them off in rounds of 256 (one slot per thread), so the cursor atomic is paid once per 256 due slots and
// once more per block at the end — not once per chunk: on a large table with few due slots per chunk (65 536 due
// among 32 M slots: 8 per chunk of 4096) the per-chunk form queued 8192 returning atomics on the one cursor word,
// about 100 us of the pass whatever the scan itself took.
constexpr uint32_t POP_CHUNK = 256 * POP_ITEMS;
struct PopList {
    uint32_t slot[POP_CHUNK + 256];      // a round's remainder (< 256) plus one chunk's worth
    uint32_t n;
};

// Appends the slots cb + j * 256 + tid of the bits set in due_mask (wave-cooperative: one LDS atomic per wave).
__device__ inline void pop_list_append(PopList& L, uint32_t due_mask, uint64_t slot0, uint32_t step) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nd = (uint32_t)__popc(due_mask);
    const uint32_t incl = wave_incl_scan(nd);
    const uint32_t wtot = __shfl(incl, 63);
    if (wtot == 0) return;                      // uniform across the wave
    uint32_t wbase = 0;
    if (lane == 63) wbase = atomicAdd(&L.n, wtot);
    wbase = __shfl(wbase, 63);
    uint32_t at = wbase + incl - nd;
    for (uint32_t mk = due_mask; mk; mk &= mk - 1) L.slot[at++] = (uint32_t)(slot0 + (uint64_t)(__ffs(mk) - 1) * step);
}

// Works off whole rounds of 256 listed slots (all of them with `rest`).  Called by every thread of the block after a
// barrier that made the appends visible; leaves fewer than 256 entries at the front of the list (none with `rest`).
__device__ inline void pop_list_drain(const Table& t, const Work& w, PopList& L, PopShared& sh, bool rest, uint64_t now,
                                      uint32_t cap_conts, uint64_t cap_pos, int alloc_pos, dj_cont* conts) {
    const uint32_t tid = threadIdx.x;
    const uint32_t ln = L.n;                    // the same in every thread
    uint32_t r = 0;
    for (; r + 256 <= ln || (rest && r < ln); r += 256) {       // uniform across the block
        const bool has = r + tid < ln;
        uint64_t s = 0, pa = 0;
        if (has) {
            s = L.slot[r + tid];
            pa = t.q[s];
        }
        pop_emit_round(t, w, sh, has, s, pa, false, now, cap_conts, cap_pos, alloc_pos, conts);
    }
    const uint32_t rem = r < ln ? ln - r : 0;   // < 256
    const uint32_t keep = tid < rem ? L.slot[r + tid] : 0;
    __syncthreads();
    if (tid < rem) L.slot[tid] = keep;
    if (tid == 0) L.n = rem;
    __syncthreads();
}

// scan form of engines without a ready list
__device__ inline void pop_due_body(const Table& t, const Work& w, uint64_t now, uint32_t cap_conts,
                                                      uint64_t cap_pos, int alloc_pos, dj_cont* conts) {
    __shared__ PopList L;
    __shared__ PopShared sh;
    const uint32_t tid = threadIdx.x;
    const uint64_t nslots = t.mask + 1;
    if (tid == 0) L.n = 0;
    __syncthreads();
    for (uint64_t cb = blockIdx.x * (uint64_t)POP_CHUNK; cb < nslots; cb += (uint64_t)gridDim.x * POP_CHUNK) {
        uint32_t due_mask = 0;
#pragma unroll
        for (int j = 0; j < POP_ITEMS; ++j) {
            const uint64_t s = cb + (uint64_t)j * 256 + tid;
            const uint64_t qv = s < nslots ? t.q[s] : NQ;
            if (qv <= now) due_mask |= 1u << j;
        }
        pop_list_append(L, due_mask, cb + tid, 256);
        __syncthreads();
        pop_list_drain(t, w, L, sh, false, now, cap_conts, cap_pos, alloc_pos, conts);
    }
    pop_list_drain(t, w, L, sh, true, now, cap_conts, cap_pos, alloc_pos, conts);
}

// (at most 80 scalar registers: from 82 on a compute unit admits seven blocks of 256 threads instead of eight)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8), amdgpu_num_sgpr(80)))
pop_due_kernel(Table t, Work w, uint64_t now, uint32_t cap_conts, uint64_t cap_pos, int alloc_pos, dj_cont* conts) {
    pop_due_body(t, w, now, cap_conts, cap_pos, alloc_pos, conts);
}

// ---- scan form for engines with a ready list: two passes over REGIONS of 64 slots -----------------------------
// Every writer of a primary score that is not on the ready list keeps Table::qmin[slot / 64] at or below it.
// Pass 1 reads the 8-byte bound of every region and lists the regions whose bound has been reached (all of them
// after a list overflow, a compaction or a restore); pass 2 reads only the listed regions' queue words, pops what
// is due and recomputes their bounds.  Cost: 8 bytes per region plus 512 bytes per region with something due,
// where a plain scan reads every queue word.  Both leave at once while now has not reached the global bound.
constexpr int SELECT_BLOCK = 1024;

__global__ void __launch_bounds__(SELECT_BLOCK) select_regions_kernel(Table t, Work w, uint64_t now) {
    __shared__ uint32_t n_blk, base_blk;
    __shared__ unsigned long long later_blk;
    const bool all = *t.ready_ovf != 0;
    if (now < *w.due_cur && !all) return;            // the ready list holds everything that is due
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x == 0 && tid == 0) w.sweep[S_SCANNED] = 1;
    const uint64_t nreg = (t.mask + 1) >> REGION_SHIFT;
    if (tid == 0) later_blk = NQ;
    unsigned long long later = NQ;                   // smallest bound of the regions this thread leaves alone
    for (uint64_t r0 = (uint64_t)blockIdx.x * SELECT_BLOCK; r0 < nreg; r0 += (uint64_t)gridDim.x * SELECT_BLOCK) {
        if (tid == 0) n_blk = 0;
        __syncthreads();
        const uint64_t r = r0 + tid;
        const unsigned long long qm = r < nreg ? t.qmin[r] : NQ;
        const bool due = r < nreg && (all || qm <= now);
        if (!due && qm < later) later = qm;
        // a place in the block's part of the list: one LDS atomic per wave, one global atomic per block
        const uint64_t m = __ballot(due);
        uint32_t wbase = 0;
        if (m && lane == 0) wbase = atomicAdd(&n_blk, (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0);
        __syncthreads();
        if (tid == 0 && n_blk) base_blk = (uint32_t)a_add(&w.sweep[S_REGIONS], (unsigned long long)n_blk);
        __syncthreads();
        if (due) w.regions[base_blk + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint32_t)r;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(later >> 32), off) << 32) | __shfl_xor((uint32_t)later, off);
        if (o < later) later = o;
    }
    if (lane == 0 && later != NQ) atomicMin(&later_blk, later);
    __syncthreads();
    if (tid == 0 && later_blk < a_load(w.due_next))
        __hip_atomic_fetch_min(w.due_next, later_blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pass 2: a block works on 64 listed regions at a time (what pop_due_body calls a chunk: 256 threads x POP_ITEMS
// queue words; wave wv reads region 4 j + wv of the 64 in step j), the regions' numbers in LDS
__global__ void __launch_bounds__(256) pop_regions_kernel(Table t, Work w, uint64_t now, uint32_t cap_conts,
                                                          uint64_t cap_pos, int alloc_pos, dj_cont* conts) {
    __shared__ PopList L;
    __shared__ uint32_t reg[64];
    __shared__ PopShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t nr = w.sweep[S_REGIONS];          // 0 unless pass 1 found it was its turn
    uint64_t later = NQ;
    if (tid == 0) L.n = 0;
    __syncthreads();
    for (uint64_t c0 = (uint64_t)blockIdx.x * 64; c0 < nr; c0 += (uint64_t)gridDim.x * 64) {
        if (tid < 64) reg[tid] = c0 + tid < nr ? w.regions[c0 + tid] : 0xFFFFFFFFu;
        __syncthreads();
        uint64_t qv[POP_ITEMS];
#pragma unroll
        for (int j = 0; j < POP_ITEMS; ++j) {
            const uint32_t r = reg[j * 4 + wv];
            qv[j] = r != 0xFFFFFFFFu ? t.q[((uint64_t)r << REGION_SHIFT) + lane] : NQ;
        }
#pragma unroll
        for (int j = 0; j < POP_ITEMS; ++j) {
            // the region's new bound: the smallest score beyond now that stays (popped and absent entries have bit 63
            // set; what is due goes, or lowers the bound again where it does not fit the caller's buffers)
            uint64_t m = (qv[j] > now && !(qv[j] >> 63)) ? qv[j] : NQ;
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(m >> 32), off) << 32) | __shfl_xor((uint32_t)m, off);
                if (o < m) m = o;
            }
            const uint32_t r = reg[j * 4 + wv];
            if (lane == 0 && r != 0xFFFFFFFFu) {
                a_store(&t.qmin[r], (unsigned long long)m);
                if (m < later) later = m;
            }
            // the region's due slots onto the block's list (every lane of the wave reads the same region: one append per step)
            const uint64_t dm = __ballot(qv[j] <= now);
            if (dm) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&L.n, (uint32_t)__popcll(dm));
                wbase = __shfl(wbase, 0);
                if (qv[j] <= now) L.slot[wbase + (uint32_t)__popcll(dm & ((1ull << lane) - 1))] = (r << REGION_SHIFT) + lane;
            }
        }
        __syncthreads();
        pop_list_drain(t, w, L, sh, false, now, cap_conts, cap_pos, alloc_pos, conts);     // (its barriers also guard reg[])
    }
    pop_list_drain(t, w, L, sh, true, now, cap_conts, cap_pos, alloc_pos, conts);
    if (later != NQ && later < a_load(w.due_next))   // lane 0 of each wave carries the regions' new bounds
        __hip_atomic_fetch_min(w.due_next, (unsigned long long)later, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// clamp the raw cursors into the caller-visible result; resets the other sweep bank, hands the scan bound to
// the next sweep and empties the ready list.  Called by all threads of ONE block after the pop kernels have
// finished.
__device__ inline void publish_result(const Table& t, const Work& w, dj_sweep_result* res, uint32_t cap_conts, uint64_t cap_pos) {
    if (t.ready_on)
        for (uint32_t k = threadIdx.x; k < READY_SHARDS; k += blockDim.x) t.ready_n[k * READY_PAD] = 0;
    if (threadIdx.x != 0) return;
    const unsigned long long cur = w.sweep[S_CURSOR];
    const unsigned long long nc = cur & CURSOR_CONT_MASK, np = cur >> CURSOR_SHIFT;
    res->n_conts = nc > cap_conts ? cap_conts : (uint32_t)nc;
    res->n_positions = np > cap_pos ? cap_pos : np;
    res->more = w.sweep[S_MORE] ? 1u : 0u;
    w.sweep_next[S_CURSOR] = 0;
    w.sweep_next[S_MORE] = 0;
    w.sweep_next[S_REGIONS] = 0;
    if (!t.ready_on) return;
    w.sweep_next[S_SCANNED] = 0;
    // a scan has recomputed the bound into *due_next; a drain leaves what is queued for later where it was
    if (!w.sweep[S_SCANNED] && *w.due_cur < *w.due_next) *w.due_next = *w.due_cur;
    *w.due_cur = NQ;                                 // the bank after next starts empty
    *t.ready_ovf = 0;
}

__global__ void finish_result_kernel(Table t, Work w, dj_sweep_result* res, uint32_t cap_conts, uint64_t cap_pos) {
    if (blockIdx.x == 0) publish_result(t, w, res, cap_conts, cap_pos);
}

// redis/all-joins for n continuations.  GS lanes work on one continuation, 64 / GS continuations per
// wave are in flight together.  update_ts: join-ts-read := join-ts (redis.clj:222-229).
template <int GS>
__device__ inline void gather_body(const Table& t, const Work& w, const dj_cont* conts, uint32_t n, int update_ts,
                                   uint32_t* lens, uint32_t* max_waits, uint8_t* payloads) {
    const uint64_t gtid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t group = gtid / GS, groups = ((uint64_t)gridDim.x * blockDim.x) / GS;
    const uint32_t j = (uint32_t)(gtid % GS);
    const uint32_t CH = t.stride >> 4;                      // 16-byte chunks per position row
    for (uint64_t kc = group; kc < n; kc += groups) {
        const uint32_t s = w.cont_slot[kc];
        if (s == 0xFFFFFFFFu) continue;
        const uint4 head = reinterpret_cast<const uint4*>(&conts[kc])[2];      // count, flags, retry counters
        if (head.y & (DJ_CONT_NO_METADATA | CONT_VOID)) continue;
        const uint64_t pos_base = conts[kc].pos_base;
        Slot* sl = &t.slots[s];
        const uint32_t count = head.x, slab = sl->slab, smeta = sl->meta, k = slab_class(count, smeta);
        const bool bitmap = class_has_bitmap(k);
        const bool stored_meta = (smeta & META_PMD) != 0;    // else every present position is a full row, no max-wait
        const uint64_t bits0 = sl->bits;
        if (update_ts && j == 0) sl->join_ts_read = sl->join_ts;
        const uint64_t* bm = slab_bitmap(t, slab);
        const uint2* pm = slab_posmeta(t, slab, k);
        const uint8_t* pay = slab_payload(t, slab, k);
        for (uint32_t p = j; p < count; p += GS) {
            const bool present = bitmap ? ((bm[p >> 6] >> (p & 63)) & 1) : ((bits0 >> p) & 1);
            const uint2 meta = !present ? make_uint2(DJ_NIL, 0) : stored_meta ? pm[p] : make_uint2(t.stride, 0);
            if (lens) lens[pos_base + p] = meta.x;
            if (max_waits) max_waits[pos_base + p] = meta.y;
        }
        if (!payloads) continue;
        // (position, chunk) pairs flattened over the group's lanes
        const uint32_t items = count * CH;
        for (uint32_t it = j; it < items; it += GS) {
            const uint32_t p = it / CH, ch = it - p * CH;
            const bool present = bitmap ? ((bm[p >> 6] >> (p & 63)) & 1) : ((bits0 >> p) & 1);
            if (!present) continue;
            const uint32_t len = stored_meta ? pm[p].x : t.stride;   // a long message leaves its 16-byte reference in the row
            // whole 64-byte lines of the caller's row are written (zeros behind the message), never a part of one
            const uint32_t chunks = len > t.stride ? 1u : ((len + 15) >> 4);
            if (ch >= ((chunks + 3) & ~3u)) continue;
            nt_store(&reinterpret_cast<uint4*>(payloads + (pos_base + p) * (uint64_t)t.stride)[ch],
                     ch < chunks ? nt_load(&reinterpret_cast<const uint4*>(pay + (uint64_t)p * t.stride)[ch])
                                 : make_uint4(0, 0, 0, 0));
        }
    }
}

// The gather pass for popped barriers (from_sweep: the raw cursor says how many; the pass also publishes
// the sweep's result block) or for n_fixed listed continuations.  Lanes per continuation: after a sweep
// from what was actually popped (16-byte chunks per continuation on average: whole waves from 96 on,
// measured on counts 4..64 x 64..1024 bytes), else the caller's choice.
// (at most 80 scalar registers: from 82 on a compute unit admits seven blocks of 256 threads instead of eight)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
gather_kernel(Table t, Work w, const dj_cont* conts, int from_sweep,
                                                     uint32_t n_fixed, uint32_t cap_conts, uint64_t cap_pos,
                                                     int update_ts, uint32_t* lens, uint32_t* max_waits,
                                                     uint8_t* payloads, dj_sweep_result* res, int gs_fixed) {
    uint32_t n = n_fixed;
    int gs = gs_fixed;
    if (from_sweep) {
        const unsigned long long cur = w.sweep[S_CURSOR];
        const unsigned long long nc = cur & CURSOR_CONT_MASK, np = cur >> CURSOR_SHIFT;
        n = nc > cap_conts ? cap_conts : (uint32_t)nc;
        const unsigned long long chunks = nc ? np * (t.stride >> 4) / nc : 0;
        gs = chunks >= 96 ? 64 : chunks > 16 ? 16 : 8;
    }
    if (gs == 64) gather_body<64>(t, w, conts, n, update_ts, lens, max_waits, payloads);
    else if (gs == 16) gather_body<16>(t, w, conts, n, update_ts, lens, max_waits, payloads);
    else gather_body<8>(t, w, conts, n, update_ts, lens, max_waits, payloads);
    if (from_sweep && res && blockIdx.x == 0) publish_result(t, w, res, cap_conts, cap_pos);
}

// =====================================================================================
// ack / recover: core/complete (core.clj:119-130), reset-publish (core.clj:113-117),
// attempt-retry-publish (core.clj:132-155)
// =====================================================================================

// core/complete on a slot this thread owns: redis/complete (Lua, redis.clj:177-188) and on
// "Join Updated" reset-publish.  Returns true when the barrier is to be deleted; the caller
// then frees the slab (wave-cooperatively) and clears the slot.
__device__ inline bool complete_decide(const Table& t, uint64_t s, uint64_t now, bool reset = true) {
    Slot* sl = &t.slots[s];
    if (sl->join_ts == sl->join_ts_read) return true;          // two nils compare equal
    if (!reset) return false;                                  // redis/complete alone: the Lua error reply, nothing written
    // reset-publish: both retry counters dropped, :publish-retry-cnt 0, re-queued at now
    sl->meta = meta_pack(sl->meta & META_HAS, 0, RETRY_NIL) | (sl->meta & META_KEEP);
    t.q[s] = now;
    note_queue_date(t, s, now);
    return false;
}

// attempt-retry-publish's counter selection (core.clj:137-141): which counter, its value, its cap
__device__ inline bool retries_left(const Table& t, uint64_t deadline, uint32_t rp, uint32_t rt, uint64_t now,
                                    bool* timeout, uint32_t* cnt) {
    *timeout = now > deadline;                                            // core.clj:137
    const uint32_t cur = *timeout ? rt : rp;
    *cnt = cur == RETRY_NIL ? 0 : cur;                                    // core.clj:139
    return *cnt < (*timeout ? t.max_retry_timeout_publish : t.max_retry_publish);   // core.clj:142
}

// update-to-retry-publish (core.clj:107-111): counter + 1, re-queued at now
__device__ inline void retry_requeue(const Table& t, uint64_t s, bool timeout, uint32_t cnt, uint32_t rp, uint32_t rt,
                                     uint64_t now) {
    if (timeout) rt = cnt + 1; else rp = cnt + 1;
    t.slots[s].meta = meta_pack(true, rp, rt) | (t.slots[s].meta & META_KEEP);
    t.q[s] = now;
    note_queue_date(t, s, now);
}

constexpr int ACK_BLOCK = 512;

// CMD = false: the publisher's outcome per continuation, ok[i] != 0 = the POST succeeded.
// CMD = true (the single-key store commands of the host, one row per launch): ok[i] is a command,
//   0 / 1 as above, ACK_OP_COMPLETE = redis/complete alone (redis.clj:191-205: delete, or answer DJ_ACK_DIRTY and
//   write nothing), ACK_OP_SET_METADATA = redis/update-metadata alone (redis.clj:106-111: the row is the metadata
//   map, written as it is: no counter is raised, no queue is touched).
constexpr uint8_t ACK_OP_COMPLETE = 2, ACK_OP_SET_METADATA = 3;

template <bool CMD>
__global__ void __launch_bounds__(ACK_BLOCK) ack_kernel(Table t, uint32_t n_max, const uint32_t* d_n, const dj_cont* conts,
                                                        const uint8_t* ok, uint64_t now, int32_t* status) {
    __shared__ uint32_t lds[BLOCK_SLAB_LDS];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = d_n ? (*d_n < n_max ? *d_n : n_max) : n_max;
    if (blockIdx.x * blockDim.x >= n) return;          // the whole block is beyond the end
    // ---- phase A (per lane): decide -----------------------------------------------------------
    int32_t st = 0;
    bool clear = false, free_old = false, alloc = false, carry = false, write_back = false, timeout = false;
    uint64_t s_u = 0;
    uint32_t k_old = 0, old_slab = 0, k_new = 0, kx_new = 0, cnt = 0, rp = RETRY_NIL, rt = RETRY_NIL;
    // long-message blocks to give back: positions [rel_from, rel_to) of slab rel_slab (class rel_k)
    const bool long_on = t.max_message > t.stride;
    bool rel = false;
    uint32_t rel_slab = 0, rel_k = 0, rel_from = 0, rel_to = 0;
    dj_cont c = {};
    const bool live = i < n;
    if (live) c = conts[i];
    // a row that no sweep of this engine can have written (a snapshot with metadata whose :count is outside
    // 1..max_count): answered 0 like a void row, nothing changes
    const uint8_t op = CMD && live && ok ? ok[i] : 0;
    const bool only_complete = CMD && op == ACK_OP_COMPLETE, only_set = CMD && op == ACK_OP_SET_METADATA;
    const bool sane = only_complete || (!only_set && (c.flags & DJ_CONT_NO_METADATA)) || (c.count >= 1 && c.count <= t.max_count);
    if (live && !(c.flags & CONT_VOID) && sane) {
        const bool published = ok == nullptr || ok[i] != 0;
        const bool has_md = only_complete || only_set || !(c.flags & DJ_CONT_NO_METADATA);
        rp = retry_from_api(c.retry_publish);
        rt = retry_from_api(c.retry_timeout);
        bool complete_path = !only_set && (!has_md || published);     // core.clj:159-161 / 134-136 (no metadata), core.clj:168 (published)
        int32_t deleted_code = has_md ? DJ_ACK_DONE : DJ_ACK_DONE_NO_METADATA;
        if (!complete_path && !only_set && !retries_left(t, c.deadline_ms, rp, rt, now, &timeout, &cnt)) {
            complete_path = true;                      // core.clj:142-147: retries exhausted
            deleted_code = DJ_ACK_DROPPED_EXHAUSTED;
        }
        if (complete_path) {
            int64_t s = -1;
            if (c.reserved && c.reserved - 1 <= t.mask) {      // slot hint written by the sweep
                const Slot* hs = &t.slots[c.reserved - 1];
                if (hs->fp_lo == c.fp_lo && hs->fp_hi == c.fp_hi) s = (int64_t)(c.reserved - 1);
            }
            if (s < 0) s = find_slot(t, c.fp_lo, c.fp_hi);
            if (s < 0) {
                st = deleted_code;                     // nothing left to delete
            } else if (complete_decide(t, (uint64_t)s, now, !only_complete)) {
                st = deleted_code;
                clear = true;
                s_u = (uint64_t)s;
                const uint32_t m = t.slots[s].meta;
                old_slab = t.slots[s].slab;
                k_old = slab_class(t.slots[s].count, m);
                free_old = old_slab != 0;
                if (long_on && (m & META_OVF)) {
                    rel = true; rel_slab = old_slab; rel_k = k_old; rel_from = 0;
                    rel_to = (m & META_KX_MASK) ? class_positions(k_old) : t.slots[s].count;
                }
            } else {
                st = only_complete ? DJ_ACK_DIRTY : DJ_ACK_REQUEUED_DIRTY;
            }
        } else {
            // publish failed, retries left: update-metadata writes the publisher's snapshot back
            // (redis.clj:106-111), creating the key if the barrier was deleted meanwhile
            bool created;
            const int64_t s = find_or_claim(t, c.fp_lo, c.fp_hi, &created);
            if (s >= 0) {
                s_u = (uint64_t)s;
                Slot* sl = &t.slots[s];
                if (created) {
                    sl->count = 0; sl->meta = 0; sl->slab = 0; sl->chain = 0;
                    sl->deadline = 0; sl->join_ts = 0; sl->join_ts_read = 0; sl->bits = 0;
                    t.q[s] = Q_NONE;
                }
                write_back = true;
                k_new = class_of(c.count);
                // The snapshot's :count may differ from the live one (the barrier was re-forked or deleted while its
                // publish was in flight).  SET writes the metadata only: the messages stay, HLEN keeps counting all
                // of them, reads see those below the count written here.  The slab therefore keeps its class when
                // that is large enough, and grows (everything stored moves over) when it is not.
                if (sl->slab == 0) {
                    alloc = true;
                } else {
                    const uint32_t k_live = slab_class(sl->count, sl->meta);
                    if (k_new > k_live) {
                        old_slab = sl->slab;
                        k_old = k_live;
                        alloc = true;
                        carry = true;
                    } else {
                        // messages at or beyond the count written back may now exist (the count went down), or already
                        // did: the barrier is marked, and replayed event by event from here on, also when its slab's
                        // class happens to be that of the new count
                        const bool marked = (sl->meta & META_KX_MASK) != 0;
                        kx_new = (marked || c.count < sl->count || k_live != k_new) ? k_live + 1 : 0;
                    }
                }
            }
        }
    }
    // ---- phase B (block- and wave-cooperative): slab traffic, one atomic per size class ---------
    if (long_on) wave_release_long_each(t, rel, rel_slab, rel_k, rel_from, rel_to);
    const uint32_t fresh = wave_slab_alloc(t, alloc, k_new);
    if (carry && fresh == 0) carry = false;            // arena exhausted (engine limit): the barrier keeps slab and count
    wave_carry_each(t, carry, old_slab, k_old, fresh, k_new);
    block_slab_free(t, free_old || carry, k_old, old_slab, lds);
    // ---- phase C (per lane): finish ------------------------------------------------------------------
    if (clear) clear_slot(t, s_u);
    if (write_back) {
        Slot* sl = &t.slots[s_u];
        bool set_count = true;
        if (carry) {
            // presence moves with the messages: a word becomes the first word of a bitmap, a bitmap a longer one
            if (class_has_bitmap(k_new)) {
                uint64_t* nb = slab_bitmap(t, fresh);
                const uint32_t nw = class_positions(k_new) >> 6;
                uint32_t have = 1;
                if (class_has_bitmap(k_old)) {
                    const uint64_t* ob = slab_bitmap(t, old_slab);
                    have = class_positions(k_old) >> 6;
                    for (uint32_t j = 0; j < have; ++j) nb[j] = ob[j];
                } else {
                    nb[0] = sl->bits;
                    sl->bits = (uint64_t)__popcll(sl->bits);
                }
                for (uint32_t j = have; j < nw; ++j) nb[j] = 0;
            }
            sl->slab = fresh;
        } else if (alloc && old_slab != 0) {
            set_count = false;                         // the slab could not grow
            kx_new = (sl->meta & META_KX_MASK) >> META_KX_SHIFT;
        } else if (alloc) {
            sl->slab = fresh;
            sl->bits = 0;
            if (fresh) fill_posmeta_default(t, fresh, k_new);
            if (c.count > 64 && fresh) {
                uint64_t* bm = slab_bitmap(t, fresh);
                for (uint32_t j = 0; j < (class_positions(k_new) >> 6); ++j) bm[j] = 0;
            }
        }
        if (set_count) sl->count = c.count;
        sl->deadline = c.deadline_ms;
        if (only_set) sl->meta = meta_pack(true, rp, rt) | (sl->meta & META_KEEP);      // SET <id>:metadata, redis.clj:110
        else retry_requeue(t, s_u, timeout, cnt, rp, rt, now);             // core.clj:151
        sl->meta = (sl->meta & ~META_KX_MASK) | (kx_new << META_KX_SHIFT);
        st = only_set ? DJ_ACK_METADATA_SET : DJ_ACK_RETRY;
    }
    if (live) status[i] = st;
}

// core/recover-unprocessed-publish-messages: ZRANGEBYSCORE backup 0 (now - delay), then the
// 2-arity attempt-retry-publish with freshly read metadata (core.clj:153-155,193-201)
constexpr int RECOVER_BLOCK = 1024;

__global__ void __launch_bounds__(RECOVER_BLOCK) recover_kernel(Table t, uint64_t now, uint32_t cap, dj_cont* conts,
                                                                int32_t* status, dj_sweep_result* res) {
    __shared__ uint32_t lds[BLOCK_SLAB_LDS];
    __shared__ uint32_t wave_n[RECOVER_BLOCK / 64];
    __shared__ uint32_t out_base;
    const uint64_t nslots = t.mask + 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t max_score = now - t.check_publish_delay_ms;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // all threads of a block iterate together (nslots is a multiple of the block size): output indices
    // and slab frees cost one atomic per block and round
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nslots; s += stride) {
        const uint64_t qv = t.q[s];
        const uint64_t ba = q_backup_at(qv);
        const bool due = qv != Q_NONE && ba <= max_score;
        if (__syncthreads_or(due) == 0) continue;
        const uint64_t dm = __ballot(due);
        if (lane == 0) wave_n[wv] = (uint32_t)__popcll(dm);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (uint32_t v = 0; v < RECOVER_BLOCK / 64; ++v) { const uint32_t c = wave_n[v]; wave_n[v] = tot; tot += c; }
            out_base = a_add(&res->n_conts, tot);
        }
        __syncthreads();
        const uint32_t at = out_base + wave_n[wv] + (uint32_t)__popcll(dm & ((1ull << lane) - 1));
        int32_t st = 0;
        bool clear = false;
        uint32_t k_old = 0, old_slab = 0;
        Slot sl = {};
        if (due) {
            sl = t.slots[s];
            bool timeout = false, drop;
            uint32_t cnt = 0;
            int32_t deleted_code;
            if (!(sl.meta & META_HAS)) {                                   // core.clj:134-136
                drop = true;
                deleted_code = DJ_ACK_DONE_NO_METADATA;
            } else {
                drop = !retries_left(t, sl.deadline, meta_retry_pub(sl.meta), meta_retry_to(sl.meta), now, &timeout, &cnt);
                deleted_code = DJ_ACK_DROPPED_EXHAUSTED;
            }
            if (!drop) {
                retry_requeue(t, s, timeout, cnt, meta_retry_pub(sl.meta), meta_retry_to(sl.meta), now);
                st = DJ_ACK_RETRY;
            } else if (complete_decide(t, s, now)) {
                st = deleted_code;
                clear = true;
                old_slab = sl.slab;
                k_old = slab_class(sl.count, sl.meta);
            } else {
                st = DJ_ACK_REQUEUED_DIRTY;
            }
        }
        if (t.max_message > t.stride)
            wave_release_long_each(t, clear && (sl.meta & META_OVF) != 0, old_slab, k_old, 0,
                                   (sl.meta & META_KX_MASK) ? class_positions(k_old) : sl.count);
        block_slab_free(t, clear && old_slab != 0, k_old, old_slab, lds);
        if (clear) clear_slot(t, s);
        if (due) {
            if (at < cap) {
                fill_cont(&conts[at], sl, now, ba, 0, s + 1);
                status[at] = st;
            } else {
                res->more = 1;
            }
        }
    }
}

// live slots and tombstones, counted on demand (dj_get_stats)
__global__ void __launch_bounds__(256) stats_kernel(Table t) {
    const uint64_t nslots = t.mask + 1;
    unsigned long long live = 0, tomb = 0;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = t.slots[s].fp_lo;
        live += k >= 2;
        tomb += k == FP_TOMB;
    }
    for (int off = 32; off > 0; off >>= 1) {
        live += __shfl_xor(live, off);
        tomb += __shfl_xor(tomb, off);
    }
    if (lane_id() == 0) {
        if (live) a_add(&t.stat[0], live);
        if (tomb) a_add(&t.stat[1], tomb);
    }
}

// =====================================================================================
// single-key helpers
// =====================================================================================

// probe one fingerprint and describe it as a continuation (no queue change)
__global__ void lookup_one_kernel(Table t, Work w, uint64_t lo, uint64_t hi, dj_cont* cont, uint64_t* scores) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t s = find_slot(t, lo, hi);
    scores[0] = NQ;
    scores[1] = NQ;
    if (s < 0) {
        Slot none = {};
        none.fp_lo = lo;
        none.fp_hi = hi;
        fill_cont(cont, none, 0, NQ, 0);
        w.cont_slot[0] = 0xFFFFFFFFu;
        return;
    }
    fill_cont(cont, t.slots[s], 0, q_primary_at(t.q[s]), 0, (uint64_t)s + 1);
    cont->flags &= ~DJ_CONT_TIMEOUT;
    w.cont_slot[0] = (uint32_t)s;
    scores[0] = q_primary_at(t.q[s]);
    scores[1] = q_backup_at(t.q[s]);
}

// redis/store-in-publish-queue for one id (redis.clj:255-259); creates a metadata-less slot
// when the id is unknown (a queue entry without metadata, core.clj:159-161)
__global__ void store_in_queue_one_kernel(Table t, uint64_t lo, uint64_t hi, uint64_t at, int32_t* err) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool created;
    int64_t s = find_or_claim(t, lo, hi, &created);
    if (s < 0) { *err = 1; return; }
    if (created) {
        Slot* sl = &t.slots[s];
        sl->count = 0; sl->meta = 0; sl->slab = 0; sl->chain = 0;
        sl->deadline = 0; sl->join_ts = 0; sl->join_ts_read = 0; sl->bits = 0;
    }
    t.q[s] = at;
    note_queue_date(t, (uint64_t)s, at);
    *err = 0;
}

// redis/store-nearest-in-publish-queue for one id (Lua, redis.clj:261-281): ZSCORE of the primary queue; a member
// that is not there makes the script's comparison raise (*out = -1); a later date than `at` moves both queues'
// entries to `at`.  The script's answer is ZADD's count of NEW members, 0 either way (*out = 0).
__global__ void store_nearest_one_kernel(Table t, uint64_t lo, uint64_t hi, uint64_t at, int32_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t s = find_slot(t, lo, hi);
    const uint64_t pa = s < 0 ? NQ : q_primary_at(t.q[s]);
    if (pa == NQ) { *out = -1; return; }
    if (pa > at) {
        t.q[s] = at;
        note_queue_date(t, (uint64_t)s, at);
    }
    *out = 0;
}

// redis/get-unprocessed-publish-messages (redis.clj:291-293): every member of the backup queue with a score of at
// most max_score, described as a continuation (fingerprint, metadata, publish_at = the backup score).  Read-only.
__global__ void __launch_bounds__(256) list_unprocessed_kernel(Table t, uint64_t max_score, uint32_t cap, dj_cont* conts,
                                                               dj_sweep_result* res) {
    const uint64_t nslots = t.mask + 1;
    for (uint64_t s0 = (uint64_t)blockIdx.x * blockDim.x; s0 < nslots; s0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = s0 + threadIdx.x;       // slots are a multiple of the block size: whole waves iterate together
        const uint64_t qv = t.q[s];
        const uint64_t ba = q_backup_at(qv);
        const bool due = qv != Q_NONE && ba <= max_score;
        const uint32_t at = wave_append(&res->n_conts, due);
        if (!due) continue;
        if (at < cap) fill_cont(&conts[at], t.slots[s], 0, ba, 0, s + 1);
        else res->more = 1;
    }
}

// Table hygiene: deleted barriers leave tombstones, and a lookup of an absent key only stops at a
// never-used slot.  rebuild re-inserts every live slot of `from` into the empty table `to`
// (same size); the host then swaps the two and clears the old one.  live[0] += live slots.
__global__ void __launch_bounds__(256) rebuild_kernel(Table from, Table to, unsigned long long* live, unsigned long long* live_next) {
    const uint64_t nslots = from.mask + 1;
    unsigned long long mine = 0;
    // the ready list names slots of the old table: the next sweep scans; the next compaction counts from zero
    if (blockIdx.x == 0 && threadIdx.x == 0) { a_store(from.ready_ovf, 1u); *live_next = 0; }
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
        // The old table is left all-empty on the way (it is the next compaction's target): a never-used slot is all
        // zeros already, a deleted one all zeros but the first half of its key (clear_slot), and both have no queue
        // entry; a live slot is cleared after it has moved.
        const ulonglong2 key = *reinterpret_cast<const ulonglong2*>(&from.slots[s]);
        if (key.x == FP_EMPTY) continue;
        if (key.x == FP_TOMB) { from.slots[s].fp_lo = FP_EMPTY; continue; }
        const Slot sl = from.slots[s];
        const uint64_t qv = from.q[s];
        ulonglong2* line = reinterpret_cast<ulonglong2*>(&from.slots[s]);
        line[0] = line[1] = line[2] = line[3] = make_ulonglong2(0, 0);
        from.q[s] = Q_NONE;
        bool created;
        const int64_t d = find_or_claim(to, sl.fp_lo, sl.fp_hi, &created);
        if (d < 0) continue;   // cannot happen: `to` has as many slots as `from` has live keys at most
        Slot* dl = &to.slots[d];
        dl->count = sl.count; dl->meta = sl.meta; dl->slab = sl.slab; dl->chain = 0;
        dl->deadline = sl.deadline; dl->join_ts = sl.join_ts; dl->join_ts_read = sl.join_ts_read; dl->bits = sl.bits;
        to.q[d] = qv;
        ++mine;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane_id() == 0 && mine) a_add(live, mine);
}

__global__ void clamp_count_kernel(dj_sweep_result* res, uint32_t cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && res->n_conts > cap) res->n_conts = cap;
}

// =====================================================================================
// multi-GPU routing: stable partition of a join batch by owner rank
// =====================================================================================
__host__ __device__ inline uint32_t owner_of(uint64_t lo, uint64_t hi, uint32_t n_ranks) {
    (void)lo;
    return (uint32_t)((hi >> 17) % n_ranks);
}

constexpr int PART_BLOCK = 256;
constexpr int PART_EPT = 4;                                  // events per thread
constexpr int PART_TILE = PART_BLOCK * PART_EPT;             // events per block
constexpr int PART_GROUPS = PART_EPT * (PART_BLOCK / 64);    // runs of 64 consecutive events per tile, one wave each
constexpr int PART_MAX_RANKS = 64;
constexpr uint32_t PART_NONE = 0xFFFFFFFFu;

// A tile is PART_TILE consecutive events; thread t of its block takes events tile0 + k * PART_BLOCK + t, k = 0..3, so
// that a wave's loads are contiguous and group g = k * (waves per block) + wave covers events tile0 + 64 g .. + 63.
// Block s of either pass owns SEGMENT s: tiles s * tiles_per_seg .. + tiles_per_seg - 1.  The host keeps
// segments at 1024 and segments x ranks at 32768 or less, so that every block of the scatter pass can add the segment
// counts up itself (32 KB of reads that hit the L2 with eight ranks) and no pass of a single block stands between the two.

// The lanes of a wave that hold the same owner, in lane order: -> this lane's rank among them; lane 0 leaves the
// number of lanes per owner in row[owner] (LDS, zeroed by the caller: owners no lane holds are not written).
__device__ inline uint32_t wave_rank_by_owner(uint32_t owner, uint32_t* row) {
    const uint32_t lane = lane_id();
    uint32_t rank_in = 0;
    uint64_t todo = __ballot(owner != PART_NONE);
    while (todo) {
        const uint32_t r = __shfl(owner, __ffsll((unsigned long long)todo) - 1);
        const uint64_t m = __ballot(owner == r);
        if (owner == r) rank_in = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (lane == 0) row[r] = (uint32_t)__popcll(m);
        todo &= ~m;
    }
    return rank_in;
}

// seg_counts[r * n_seg + s] = events of segment s owned by rank r
__global__ void __launch_bounds__(PART_BLOCK) partition_count_kernel(const dj_join_rec* recs, uint32_t n, uint32_t n_ranks,
                                                                     uint32_t tiles_per_seg, uint32_t* seg_counts) {
    __shared__ uint32_t cnt[PART_MAX_RANKS];
    if (threadIdx.x < PART_MAX_RANKS) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t tl = 0; tl < tiles_per_seg; ++tl) {
        const uint64_t tile0 = ((uint64_t)blockIdx.x * tiles_per_seg + tl) * PART_TILE;
        if (tile0 >= n) break;
        uint32_t owner[PART_EPT];
#pragma unroll
        for (int k = 0; k < PART_EPT; ++k) {
            const uint64_t i = tile0 + k * PART_BLOCK + threadIdx.x;
            owner[k] = i < n ? owner_of(0, reinterpret_cast<const uint64_t*>(recs)[4 * i + 1], n_ranks) : PART_NONE;
        }
#pragma unroll
        for (int k = 0; k < PART_EPT; ++k) {
            uint64_t todo = __ballot(owner[k] != PART_NONE);
            while (todo) {
                const uint32_t r = __shfl(owner[k], __ffsll((unsigned long long)todo) - 1);
                const uint64_t m = __ballot(owner[k] == r);
                if (lane_id() == 0) atomicAdd(&cnt[r], (uint32_t)__popcll(m));
                todo &= ~m;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < n_ranks) seg_counts[threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

// scatter records (and the origin index) to their rank group, keeping array order inside a group;
// the messages follow in the same pass (G lanes per row, 16 bytes per lane, destinations handed
// from lane to lane inside the wave, four rows per lane group in flight).  Block 0 also writes the
// totals per rank to counts_out.
template <int G>
__global__ void __launch_bounds__(PART_BLOCK) partition_scatter_kernel(const dj_join_rec* recs, uint32_t n, uint32_t n_ranks,
                                                                       uint32_t tiles_per_seg, const uint32_t* seg_counts,
                                                                       uint32_t* counts_out, dj_join_rec* recs_out,
                                                                       uint32_t* origin, const uint8_t* pay_in,
                                                                       uint8_t* pay_out, uint64_t pitch, uint32_t self_rank,
                                                                       dj_join_rec* recs_self, uint8_t* pay_self) {
    // self_rank < n_ranks: the rows of that rank's group go to recs_self / pay_self instead (row j of the group to row j
    // there: the router lets this rank's own bucket land where its engine reads it); origin[] is indexed as if they
    // had gone to their place in recs_out like everybody else's
    __shared__ uint32_t cnt[PART_GROUPS][PART_MAX_RANKS];
    __shared__ uint32_t tot[PART_MAX_RANKS], base[PART_MAX_RANKS];
    __shared__ uint32_t self_base;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n_seg = gridDim.x;
    // where this segment's events start in each rank's group: the rank's events in earlier segments, behind
    // everything of the ranks before it
    for (uint32_t r = wv; r < n_ranks; r += PART_BLOCK / 64) {
        uint32_t all = 0, before = 0;
        for (uint32_t j = lane; j < n_seg; j += 64) {
            const uint32_t v = seg_counts[r * n_seg + j];
            all += v;
            if (j < blockIdx.x) before += v;
        }
        for (int off = 32; off > 0; off >>= 1) { all += __shfl_xor(all, off); before += __shfl_xor(before, off); }
        if (lane == 0) { tot[r] = all; base[r] = before; }
    }
    __syncthreads();
    uint32_t mine = 0;
    if (threadIdx.x < n_ranks) {
        mine = base[threadIdx.x];
        uint32_t before = 0;
        for (uint32_t r = 0; r < threadIdx.x; ++r) before += tot[r];
        mine += before;
        if (threadIdx.x == self_rank) self_base = before;
        if (blockIdx.x == 0) counts_out[threadIdx.x] = tot[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x < n_ranks) base[threadIdx.x] = mine;

    constexpr int RPW = 64 / G;                                  // rows a wave copies per step
    constexpr int FLIGHT = 4;                                    // steps whose loads are issued before the first store
    const uint32_t sub = lane / G, j0 = lane % G;
    const uint32_t chunks = (uint32_t)(pitch >> 4);
    for (uint32_t tl = 0; tl < tiles_per_seg; ++tl) {
        const uint64_t tile0 = ((uint64_t)blockIdx.x * tiles_per_seg + tl) * PART_TILE;
        if (tile0 >= n) break;                                   // uniform across the block
        __syncthreads();                                         // the previous tile's offsets have been read
        for (uint32_t j = threadIdx.x; j < PART_GROUPS * PART_MAX_RANKS; j += PART_BLOCK) (&cnt[0][0])[j] = 0;
        __syncthreads();
        ulonglong2 a[PART_EPT], b[PART_EPT];
        uint32_t owner[PART_EPT], o[PART_EPT];
#pragma unroll
        for (int k = 0; k < PART_EPT; ++k) {
            const uint64_t i = tile0 + k * PART_BLOCK + threadIdx.x;
            owner[k] = PART_NONE;
            if (i < n) {
                a[k] = reinterpret_cast<const ulonglong2*>(recs)[2 * i];
                b[k] = reinterpret_cast<const ulonglong2*>(recs)[2 * i + 1];
                owner[k] = owner_of(a[k].x, a[k].y, n_ranks);
            }
        }
#pragma unroll
        for (int k = 0; k < PART_EPT; ++k) o[k] = wave_rank_by_owner(owner[k], cnt[k * (PART_BLOCK / 64) + wv]);
        __syncthreads();
        if (threadIdx.x < n_ranks) {            // per rank: where each group's events start in the output
            uint32_t run = base[threadIdx.x];
            for (int g = 0; g < PART_GROUPS; ++g) {
                const uint32_t v = cnt[g][threadIdx.x];
                cnt[g][threadIdx.x] = run;
                run += v;
            }
            base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PART_EPT; ++k) {
            if (owner[k] == PART_NONE) { o[k] = PART_NONE; continue; }
            o[k] += cnt[k * (PART_BLOCK / 64) + wv][owner[k]];
            origin[o[k]] = (uint32_t)(tile0 + k * PART_BLOCK + threadIdx.x);
            dj_join_rec* to = recs_out;
            if (owner[k] == self_rank) {             // from here on o[k] is the row in the self buffers, flagged
                to = recs_self;
                o[k] -= self_base;
            }
            reinterpret_cast<ulonglong2*>(to)[2 * (size_t)o[k]] = a[k];
            reinterpret_cast<ulonglong2*>(to)[2 * (size_t)o[k] + 1] = b[k];
            if (owner[k] == self_rank) o[k] |= 0x80000000u;
        }
        if (pay_in == nullptr) continue;
#pragma unroll 1
        for (int k = 0; k < PART_EPT; ++k) {
            const uint64_t wave_i0 = tile0 + k * PART_BLOCK + wv * 64;
            if (wave_i0 >= n) break;
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < chunks; c0 += G) {
                const uint32_t c = c0 + j0;
#pragma unroll 1
                for (int step0 = 0; step0 < G; step0 += FLIGHT) {
                    uint4 v[FLIGHT];
                    uint32_t oo[FLIGHT];
#pragma unroll
                    for (int f = 0; f < FLIGHT; ++f) {
                        const uint32_t src_lane = (step0 + f) * RPW + sub;
                        oo[f] = __shfl(o[k], src_lane);
                        if (c >= chunks) oo[f] = PART_NONE;
                        if (oo[f] != PART_NONE)
                            v[f] = nt_load(&reinterpret_cast<const uint4*>(pay_in + (wave_i0 + src_lane) * pitch)[c]);
                    }
#pragma unroll
                    for (int f = 0; f < FLIGHT; ++f)
                        if (oo[f] != PART_NONE)
                            nt_store(&reinterpret_cast<uint4*>(((oo[f] & 0x80000000u) ? pay_self : pay_out) + (uint64_t)(oo[f] & 0x7FFFFFFFu) * pitch)[c], v[f]);
                }
            }
        }
    }
}

// The way back of a routed batch: sorted[j] answers the record that dj_partition_joins moved from place origin[j] of
// the caller's batch to place j: out[origin[j]] = sorted[j].  Reads stream, the 4-byte stores scatter.
__global__ void __launch_bounds__(256) scatter_statuses_kernel(uint32_t n, const uint32_t* origin, const int32_t* sorted, int32_t* out) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        out[origin[j]] = sorted[j];
}

// Measurement aid: a plain streaming copy, one 16-byte load and store per lane, no loop: the grid covers the buffer.
// What this GPU's memory delivers to the simplest possible kernel is the ceiling the byte-moving passes are held against.
// (Of the shapes tools/copy_sweep.hip tries on 1 GiB this one is the fastest, 6.2 TB/s read + written; a grid-stride loop
// with four loads in flight per lane reaches 4.1 to 4.7, chunks of 4 x 256 lanes per block 5.4 to 5.9.)
__global__ void __launch_bounds__(256) stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

}  // namespace dj
