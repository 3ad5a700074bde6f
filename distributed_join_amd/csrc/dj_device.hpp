// dj_device.hpp — device-side data layout and helpers of the barrier engine (gfx950).
//
// HBM layout (one engine = one GPU):
//   slots[]      64-byte records, open addressing with linear probing, power-of-two size.
//                One record = one Redis key family of the reference:
//                "<id>:metadata", "<id>:join-ts", "<id>:join-ts-read" and the presence set of
//                the hash "<id>" (reference redis.clj:42-73).
//   q[]          one u64 per slot for BOTH sorted sets "publish-queue" / "publish-queue-backup"
//                (redis.clj:36-37).  Every writer of the reference keeps a member's two scores
//                equal (store-in-publish-queue redis.clj:257-258, store-nearest redis.clj:270-271)
//                and the only asymmetric operation removes it from the primary set
//                (all-publish-messages, redis.clj:288), so a member is in one of three states:
//                  in both, same score : q = score
//                  backup only         : q = score | Q_POPPED
//                  in neither          : q = Q_NONE
//                `q <= now` is then exactly "in the primary set with score <= now": the sweep
//                streams 8 bytes per slot and every queue update touches one line.
//   arena        payload slabs in 64-byte units, one slab per barrier, size class = count
//                rounded up to a power of two:
//                  [bitmap words, only when class > 64 positions]
//                  [posmeta: class * {u32 len, u32 max_wait}]           (64-byte aligned)
//                  [payload: class * payload_stride bytes]              (64-byte aligned)
//   free rings   per size class, slabs returned by deleted barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dj_engine.h"

namespace dj {

constexpr uint64_t FP_EMPTY = 0;       // never-used slot (fp_lo)
constexpr uint64_t FP_TOMB = 1;        // deleted slot (fp_lo), fp_hi = 0
constexpr uint64_t NQ = DJ_NOT_QUEUED;
constexpr uint64_t Q_NONE = ~0ull;          // in neither queue
constexpr uint64_t Q_POPPED = 1ull << 63;   // removed from the primary queue, still in the backup queue
constexpr uint32_t META_HAS = 1u;      // metadata key exists
constexpr uint32_t META_HOT = 2u;      // hint: the barrier received hundreds of events in its last batch
constexpr uint32_t META_OVF = 4u;      // some position of the barrier holds (or held) a long message in a block of its own
constexpr uint32_t META_PMD = 8u;      // some position holds (or held) a message that does not fill its row exactly, or a
                                       // max-wait: the per-position {len, max_wait} words are authoritative.  Without the
                                       // flag every present position is {payload_stride, 0}; the words (pre-filled with
                                       // that value at fork) are then neither written by joins nor read by gathers
// Bits 4..7 of meta: 0, or size class of the barrier's slab + 1 on a barrier that may hold messages at or beyond its
// count.  Only a failed publish can bring that about: it writes its metadata snapshot back (redis.clj:106-111) and the
// snapshot's :count may be older, and smaller, than the barrier's (re-forked with another :count meanwhile).  The
// messages stored under the other count stay where they are, are counted by HLEN as before and are invisible to reads
// beyond the new count, so the slab keeps its class (or grows, when a count is written back that it cannot hold) and
// the presence word or bitmap keeps covering all its positions.  While the bits are set, HLEN can pass the count and
// come back: the join kernels replay such a barrier event by event.
constexpr uint32_t META_KX_SHIFT = 4;
constexpr uint32_t META_KX_MASK = 0xFu << META_KX_SHIFT;
constexpr uint32_t META_KEEP = META_OVF | META_PMD | META_KX_MASK;     // what a rewrite of the retry counters leaves alone
constexpr int OVF_CLASSES = 15;        // blocks for long messages: 1 << k units, 64 B .. 1 MiB
constexpr uint32_t RETRY_NIL = 0xFFu;  // retry counter absent
constexpr int NUM_CLASSES = 11;        // 1, 2, 4 ... 1024 positions
constexpr uint64_t NODST = ~0ull;
constexpr uint32_t CONT_VOID = DJ_CONT_VOID;  // descriptor index allocated but not popped
constexpr uint32_t READY_SHARDS = 256;        // sub-lists of the ready list, each with a counter on a line of its own
constexpr uint32_t READY_PAD = 32;            // counters are READY_PAD words apart (128 bytes)

struct __attribute__((aligned(64))) Slot {
    uint64_t fp_lo, fp_hi;
    uint32_t count;         // :count of the metadata, 0 without metadata
    uint32_t meta;          // bit0 META_HAS | publish-retry-cnt << 8 | timeout-publish-retry-cnt << 16
    uint32_t slab;          // arena offset in 64-byte units, 0 = none
    uint32_t chain;         // transient per-batch link head (event index + 1); 0 between batches
    uint64_t deadline;      // :timeout-secs
    uint64_t join_ts_read;  // "<id>:join-ts-read", 0 = nil
    uint64_t join_ts;       // "<id>:join-ts", 0 = nil            } the last 16 bytes: what a join batch
    uint64_t bits;          // count <= 64: presence bitmap;       } rewrites, stored together
                            // count > 64: number of present positions
};
static_assert(sizeof(Slot) == 64, "slot must be one 64-byte line");

// join_ts and bits with one 16-byte store
__device__ inline void store_ts_bits(Slot* sl, uint64_t join_ts, uint64_t bits) {
    reinterpret_cast<ulonglong2*>(sl)[3] = make_ulonglong2(join_ts, bits);
}

__host__ __device__ inline uint64_t q_primary_at(uint64_t v) { return (v >> 63) ? NQ : v; }
__host__ __device__ inline uint64_t q_backup_at(uint64_t v) { return v == Q_NONE ? NQ : (v & ~Q_POPPED); }
__host__ __device__ inline uint32_t meta_pack(bool has, uint32_t rp, uint32_t rt) {
    return (has ? META_HAS : 0u) | ((rp & 0xFFu) << 8) | ((rt & 0xFFu) << 16);
}
__host__ __device__ inline uint32_t meta_retry_pub(uint32_t m) { return (m >> 8) & 0xFFu; }
__host__ __device__ inline uint32_t meta_retry_to(uint32_t m) { return (m >> 16) & 0xFFu; }
__host__ __device__ inline uint32_t retry_to_api(uint32_t r) { return r == RETRY_NIL ? DJ_NIL : r; }
__host__ __device__ inline uint32_t retry_from_api(uint32_t r) { return r == DJ_NIL ? RETRY_NIL : (r > 0xFEu ? 0xFEu : r); }

// ---- slab geometry ---------------------------------------------------------------
__host__ __device__ inline uint32_t class_of(uint32_t count) {
    uint32_t k = 0;
    while ((1u << k) < count) ++k;
    return k;
}
__host__ __device__ inline uint32_t class_positions(uint32_t k) { return 1u << k; }
// size class of the slab of a barrier with this :count and meta word (see META_KX_MASK)
__host__ __device__ inline uint32_t slab_class(uint32_t count, uint32_t meta) {
    const uint32_t kx = (meta >> 4) & 0xFu;
    return kx ? kx - 1 : class_of(count);
}
// presence is a bitmap in the slab (and Slot::bits a counter) for classes above 64 positions, else the word Slot::bits
__host__ __device__ inline bool class_has_bitmap(uint32_t k) { return k > 6; }
__host__ __device__ inline uint32_t slab_bitmap_bytes(uint32_t k) {
    uint32_t p = 1u << k;
    return p > 64 ? ((p / 64 * 8 + 63) & ~63u) : 0;
}
__host__ __device__ inline uint32_t slab_posmeta_off(uint32_t k) { return slab_bitmap_bytes(k); }
__host__ __device__ inline uint32_t slab_payload_off(uint32_t k) {
    return slab_posmeta_off(k) + (((1u << k) * 8 + 63) & ~63u);
}
__host__ __device__ inline uint32_t slab_units(uint32_t k, uint32_t stride) {
    return (slab_payload_off(k) + (1u << k) * stride) / 64;
}

struct Table {
    Slot* slots;
    uint64_t* q;                 // queue word per slot, see the layout note above
    uint64_t mask;               // slots - 1
    uint8_t* arena;
    uint64_t arena_units;
    unsigned long long* bump;    // next never-used arena unit
    uint32_t* ring[NUM_CLASSES]; // freed slabs per class
    uint32_t ring_cap[NUM_CLASSES];
    unsigned int* ring_head;     // [NUM_CLASSES] pop side
    unsigned int* ring_tail;     // [NUM_CLASSES] push side
    unsigned int* ring_tail_snap;// [NUM_CLASSES] tail at the start of the current call: pops stop here
    unsigned long long* stat;    // scratch of stats_kernel: [0] live slots, [1] tombstones
    // blocks for messages longer than one inline row (only with max_message > stride): the slot row of
    // such a position starts with a dj_long_ref, its posmeta len is > stride
    uint32_t* oring[OVF_CLASSES];     // freed blocks per class
    uint32_t oring_cap[OVF_CLASSES];
    unsigned int* oring_head;         // [OVF_CLASSES]
    unsigned int* oring_tail;
    unsigned int* oring_tail_snap;    // tails before the current join batch's copy pass: pops stop here
    // Ready list (the device-side form of "the primary queue, ordered by score"): a join batch appends the slot
    // of every barrier it completes (queued at now); the next sweep drains the list instead of scanning q[].
    // Every OTHER writer of a primary score (fork deadlines, max-wait dates, re-queues) lowers *due_min
    // instead; a sweep scans q[] only when its now has reached that bound (or the list overflowed / was
    // invalidated), and the scan recomputes the bound.  The list may hold stale or repeated slots: a pop is
    // decided by q[slot] alone.
    uint32_t* ready;                  // [READY_SHARDS][ready_shard_cap]
    unsigned int* ready_n;            // [READY_SHARDS * READY_PAD] entries appended per shard (may run past the cap)
    unsigned int* ready_ovf;          // != 0: the list is incomplete, the next sweep scans
    unsigned long long* due_min;      // lower bound of the primary scores that are in no list (the bank the next sweep reads)
    unsigned long long* qmin;         // the same bound per REGION of 64 consecutive slots (512 bytes of queue words): a scan
                                      // reads 8 bytes per region and only the regions that can hold something due
    uint32_t ready_shard_cap;
    uint32_t ready_on;                // 0: no list and no bound are kept, every sweep scans (small tables: the scan is cheap
                                      // and hands the gather pass its barriers in slot order)
    uint32_t max_message;             // longest message accepted, >= stride
    uint32_t stride;
    uint32_t fork_pmd;                // 1: barriers are forked with META_PMD set (recent joins were mostly not full rows
                                      // without max-wait: the flag would be set by their first join anyway, with an atomic)
    uint32_t max_count;
    uint32_t max_retry_publish, max_retry_timeout_publish;
    uint64_t check_publish_delay_ms;
};

// per-batch scratch, sized by max_batch
struct Work {
    uint64_t* node;      // join link: next(25) | position(10) << 25 | max_wait(29) << 35
    uint64_t* dst;       // join: where the event's message goes, NODST = superseded / rejected
    uint32_t* ev_slot;   // join: slot of event i (0xFFFFFFFF when the event was refused)
    uint32_t* complex_;  // slots whose chain is too long for the register path
    uint32_t* slot_of;   // fork: slot of record i; SLOT_CREATED set when the record's own claim made the slot
    unsigned int* fork_contested;        // fork: set by a record whose barrier existed already (from before the batch or made by
                                         // another record of it); while zero every record owns a fresh slot and wins it
    unsigned int* fork_contested_next;   // the next fork batch's word, zeroed by this one
    uint32_t* cont_slot; // sweep: slot of continuation k
    uint32_t* regions;   // sweep (engines with a ready list): the 64-slot regions of the table that may hold something due
    unsigned int* counters;       // C_* below, bank of the current join batch
    unsigned int* counters_next;  // the other bank: zeroed by the current batch for the next one
    unsigned int* hint;           // pinned HOST memory: [0] barriers with long chains, [1] longest chain of the latest
                                  // join batch, written by its last kernel (the host reads them without waiting);
                                  // [4..7] how many of the batch's first 4 x 64 events store a {len, max-wait} word
    unsigned long long* sweep;    // S_* below, bank of the current sweep
    unsigned long long* sweep_next;
    unsigned long long* due_cur;  // scan bound the current sweep reads (what Table::due_min pointed to since the last one)
    unsigned long long* due_next; // scan bound the current sweep leaves for the next one
    // hot-barrier path (barriers with long chains), see the cx_* kernels
    uint32_t* cx_keys;            // [CX_HASH] slab of a hot barrier, 0 = empty
    uint32_t* cx_val;             // [CX_HASH] its index in complex_[]
    uint32_t* cx_off;             // [CX_CAP] where the barrier's per-position tables start in the pool, IDX_NONE = not registered
    uint2* cx_mm;                 // [CX_POOL] per position: {first event index, last event index + 1}
    unsigned long long* cx_atmin; // [CX_CAP] smallest now + 1000 * max-wait among the barrier's events
    uint32_t* cx_cnt;             // [CX_CAP] events of the barrier in this batch
    uint32_t* cx_T;               // [CX_CAP] first event index that answers COMPLETE, IDX_NONE = none
    uint32_t* cx_hot;             // [CX_CAP] 0, or for a barrier flagged hot before the batch the pool entries per position
                                  // (cx_hot_span: a power of two, 1 .. 128)
    uint32_t* cx_nq;              // [CX_CAP] 1 when the barrier was not in the primary queue before the batch
    uint32_t cx_on;               // 1: this batch runs the parallel hot-barrier kernels
    uint32_t publish_hints;       // 1: the link pass hands the previous batch's counters (the other bank) to the host
    uint32_t hlen;                // 1: every accepted event's status word also carries HLEN of its barrier's hash right
                                  // after the event, above DJ_JOIN_HLEN_SHIFT (dj_set_join_hlen)
    uint32_t fused;               // 1: the link pass has copied every accepted event's message into its row already; the
                                  // resolve passes copy again where a later event of the batch wrote the same position
    // bucketed join path (join_link_bucket_kernel / join_resolve_bucket_kernel)
    uint64_t* bkeys;              // [BK_MAX_BUCKETS][RB_CAP] one key per linked event, in the bucket of its slot's range
    unsigned int* bcursor;        // [BK_MAX_BUCKETS] keys per bucket; zero between batches (the second pass clears its own)
    uint32_t bbuckets;            // buckets of this batch (a power of two)
    uint32_t bshift;              // bucket of a slot = slot >> bshift
    uint32_t bcap;                // keys a bucket holds (bfat x the second pass's LDS capacity)
    uint32_t bfat;                // sub-ranges the second pass cuts a bucket's slot range into (1: the bucket fits its LDS)
    uint32_t bsubshift;           // sub-range of a key = its slot inside the bucket's range >> bsubshift
    uint32_t bmwbits;             // bits of a key above its slot field that carry a short max-wait (0: none)
};
constexpr uint32_t SLOT_CREATED = 0x80000000u;      // tables have at most 2^30 slots
enum { C_COMPLEX = 1, C_MAXCHAIN = 2, C_CXPOOL = 3, C_MID = 4, C_RECOPY = 5, C_BOVF = 6, C_COUNT = 8 };    // C_BOVF: a bucket overflowed
constexpr uint32_t CX_CAP = 65536;     // barriers with long chains per batch handled by the parallel path
constexpr uint32_t CX_HASH = 262144;   // power of two, >= 2 * CX_CAP
constexpr uint32_t CX_HOT_STRIDE = 8;  // pool entries per table line (64 bytes)
// A barrier flagged hot spreads every position over several 64-byte lines (its events pick one by wave, the
// finalize pass folds them into the first): thousands of atomics on ONE line would queue up.  The fewer
// positions a barrier has, the more events each of them takes, the more lines it gets: about 2048 pool
// entries (16 KB) per hot barrier whatever its count.  Below 8 entries per position there is one copy and
// positions share lines.
__host__ __device__ inline uint32_t cx_hot_span(uint32_t count) {
    uint32_t p2 = 1;
    while (p2 < count) p2 <<= 1;
    const uint32_t span = 2048u / p2;
    return span > 128u ? 128u : span < 1u ? 1u : span;
}
__host__ __device__ inline uint32_t cx_span_replicas(uint32_t span) { return span >= 2 * CX_HOT_STRIDE ? span / CX_HOT_STRIDE : 1u; }
constexpr uint32_t CX_POOL = 1u << 24; // positions of all of them together (the rest takes the serial kernel)
constexpr uint32_t IDX_NONE = 0xFFFFFFFFu;
// sweep scratch: S_CURSOR packs both output cursors so a block takes its output range with one
// atomic: continuations in the low 31 bits, positions above (positions * 64 B <= arena <= 256 GiB)
enum { S_CURSOR = 0, S_MORE = 1, S_SCANNED = 2, S_REGIONS = 3, S_COUNT = 4 };    // S_REGIONS: entries of Work::regions
constexpr int CURSOR_SHIFT = 31;
constexpr unsigned long long CURSOR_CONT_MASK = (1ull << CURSOR_SHIFT) - 1;

// ---- atomics (agent scope, relaxed) ------------------------------------------------
template <class T>
__device__ inline T a_load(T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ inline void a_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ inline T a_add(T* p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ inline T a_max(T* p, T v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ inline T a_xchg(T* p, T v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ inline bool a_cas(T* p, T expect, T v) {
    return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline uint32_t lane_id() { return __lane_id(); }

// Streamed bytes (messages on their way into or out of a slab) are touched once: non-temporal
// loads and stores keep them from pushing the barrier table out of the caches.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint4 nt_load(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline void nt_store(uint4* p, uint4 v) {
    u32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(p));
}

// Append one element per lane with pred set; one atomic per wave.  Must be reached by the
// wave in converged control flow.  Returns the index, or 0xFFFFFFFF for lanes without pred.
__device__ inline uint32_t wave_append(unsigned int* counter, bool pred) {
    uint64_t m = __ballot(pred);
    if (m == 0) return 0xFFFFFFFFu;
    uint32_t lane = lane_id();
    int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = a_add(counter, (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    return pred ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1)) : 0xFFFFFFFFu;
}

// A primary score written without a ready-list entry: keep the sweep's scan bound below it.  Most callers see
// a bound that is already low enough and issue no atomic.
// (A plain read of a bound can only be too high inside one launch, never too low: bounds go down between sweeps, and
// only a sweep's scan raises them.  Too high costs an atomic that changes nothing.)
constexpr uint32_t REGION_SHIFT = 6;
__device__ inline void note_region_date(const Table& t, uint64_t s, uint64_t at) {
    if (!t.ready_on) return;
    unsigned long long* r = &t.qmin[s >> REGION_SHIFT];
    if (at < *r) __hip_atomic_fetch_min(r, (unsigned long long)at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void note_global_date(const Table& t, uint64_t at) {
    if (t.ready_on && at < a_load(t.due_min)) __hip_atomic_fetch_min(t.due_min, (unsigned long long)at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void note_queue_date(const Table& t, uint64_t s, uint64_t at) {
    note_region_date(t, s, at);
    note_global_date(t, at);
}

// Append the slots of lanes with pred to the ready list: one atomic per wave, on the counter of the shard
// `shard`.  Must be reached by the wave in converged control flow.
__device__ inline void ready_append_wave(const Table& t, bool pred, uint32_t slot, uint32_t shard) {
    if (!t.ready_on) return;
    const uint64_t m = __ballot(pred);
    if (m == 0) return;
    shard &= READY_SHARDS - 1;
    const uint32_t lane = lane_id();
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = a_add(&t.ready_n[shard * READY_PAD], (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    if (pred) {
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (at < t.ready_shard_cap) t.ready[(size_t)shard * t.ready_shard_cap + at] = slot;
        else a_store(t.ready_ovf, 1u);
    }
}

// one entry from one thread (rare paths)
__device__ inline void ready_append_one(const Table& t, uint32_t slot, uint32_t shard) {
    if (!t.ready_on) return;
    shard &= READY_SHARDS - 1;
    const uint32_t at = a_add(&t.ready_n[shard * READY_PAD], 1u);
    if (at < t.ready_shard_cap) t.ready[(size_t)shard * t.ready_shard_cap + at] = slot;
    else a_store(t.ready_ovf, 1u);
}

// ---- probing -----------------------------------------------------------------------
// The memory system moves aligned 128-byte blocks (two slots): a second slot of the same block
// costs nothing, a slot of another block costs a full random access.  The probe sequence
// therefore visits the home slot h, its block partner h ^ 1, then the following blocks in order:
//   p = 0: h,  p = 1: h ^ 1,  p >= 2: ((h & ~1) + p) & mask          (every slot exactly once)
__device__ inline uint64_t probe_at(uint64_t h, uint64_t p, uint64_t mask) {
    return p == 0 ? h : (p == 1 ? (h ^ 1) : (((h & ~1ull) + p) & mask));
}

// Read-only lookup.  Valid in every kernel that does not insert (joins, sweeps, acks of
// existing barriers): slot keys only change between launches there.  Both slots of a block are
// requested together.
// *info (optional) receives the slot's second 16 bytes {count, meta, slab, chain} as they were at the lookup.
__device__ inline int64_t find_slot(const Table& t, uint64_t lo, uint64_t hi, uint4* info = nullptr) {
    const uint64_t h = lo & t.mask;
    uint64_t s0 = h, s1 = h ^ 1;
    for (uint64_t p = 0; p <= t.mask; p += 2) {
        const ulonglong2 k0 = *reinterpret_cast<const ulonglong2*>(&t.slots[s0]);
        const ulonglong2 k1 = *reinterpret_cast<const ulonglong2*>(&t.slots[s1]);
        uint4 i0 = make_uint4(0, 0, 0, 0), i1 = i0;
        if (info) {
            i0 = reinterpret_cast<const uint4*>(&t.slots[s0])[1];
            i1 = reinterpret_cast<const uint4*>(&t.slots[s1])[1];
        }
        if (k0.x == lo && k0.y == hi) { if (info) *info = i0; return (int64_t)s0; }
        if (k0.x == FP_EMPTY) return -1;
        if (k1.x == lo && k1.y == hi) { if (info) *info = i1; return (int64_t)s1; }
        if (k1.x == FP_EMPTY) return -1;
        s0 = ((h & ~1ull) + p + 2) & t.mask;
        s1 = s0 + 1;
    }
    return -1;
}

// Lookup-or-insert for kernels that create slots (fork, retry write-back, orphan queue
// entries).  Slots only go EMPTY/TOMB -> key inside such a kernel, so a key once seen is final.
// Returns the slot or -1 when the table is full (or a half-written slot never completed); *created says whether
// this call claimed it.
__device__ inline int64_t find_or_claim(const Table& t, uint64_t lo, uint64_t hi, bool* created) {
    *created = false;
    for (int attempt = 0; attempt < 64; ++attempt) {
        const uint64_t h = lo & t.mask;
        int64_t tomb = -1;
        bool restart = false;
        for (uint64_t p = 0; p <= t.mask && !restart; ++p) {
            const uint64_t s = probe_at(h, p, t.mask);
            Slot* sl = &t.slots[s];
            uint64_t klo = a_load(&sl->fp_lo);       // never from a stale cache line: slots are being claimed
            if (klo == lo) {
                // same low half: wait (without spinning inside the iteration) for the claimer's
                // high half, then compare
                uint64_t khi = a_load(&sl->fp_hi);
                for (int spin = 0; khi == 0 && spin < (1 << 20); ++spin) {
                    __builtin_amdgcn_s_sleep(1);
                    khi = a_load(&sl->fp_hi);
                }
                if (khi == hi) return (int64_t)s;
                // Still half-written after ~30 ms: its claimer is not making progress.  The slot may yet turn out to be
                // this key, so going on (and inserting the key a second time) would be worse than giving up.
                if (khi == 0) return -1;
                continue;  // another key with the same low half
            }
            if (klo == FP_TOMB) {
                if (tomb < 0) tomb = (int64_t)s;
                continue;
            }
            if (klo != FP_EMPTY) continue;  // another key
            // end of the probe chain: the key is absent, claim the first tombstone or this slot
            uint64_t target = tomb >= 0 ? (uint64_t)tomb : s;
            uint64_t expect = tomb >= 0 ? FP_TOMB : FP_EMPTY;
            Slot* tl = &t.slots[target];
            if (a_cas(&tl->fp_lo, expect, lo)) {
                a_store(&tl->fp_hi, hi);
                *created = true;
                return (int64_t)target;
            }
            restart = true;  // somebody else claimed it: probe again from the home slot
        }
        if (!restart) return -1;
    }
    return -1;
}

// ---- slab allocation ---------------------------------------------------------------
// All three helpers are WAVE-COOPERATIVE: every lane of the wave must call them in converged
// control flow (lanes with nothing to do pass want/has = false).  A single hot word sustains
// only ~90 atomics per microsecond on this chip, so the wave issues one atomic per distinct
// size class (rings) or one per wave (bump pointer), never one per lane.

// in-wave inclusive prefix sum
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(v, off);
        if ((int)lane >= off) v += o;
    }
    return v;
}

// Returns a slab of class k for lanes with want, or 0 when the arena is exhausted.
// Freed slabs first (ring pops are bounded by the tail snapshot taken at the start of the
// call, so a pop never reads an entry that a push of the same launch has not written yet;
// head and tail are free-running 32-bit counters, ring_cap is a power of two).
__device__ inline uint32_t wave_slab_alloc(const Table& t, bool want, uint32_t k) {
    const uint32_t lane = lane_id();
    uint32_t slab = 0;
    uint64_t todo = __ballot(want);
    while (todo) {
        const int l0 = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(k, l0);
        const uint64_t grp = __ballot(want && k == k0);
        const uint32_t m = (uint32_t)__popcll(grp);
        const uint32_t rank = (uint32_t)__popcll(grp & ((1ull << lane) - 1));
        uint32_t h = 0;
        int got = 0;
        if ((int)lane == l0) {
            const unsigned int snap = t.ring_tail_snap[k0];
            if ((int)(snap - a_load(&t.ring_head[k0])) > 0) {
                h = a_add(&t.ring_head[k0], m);
                const int avail = (int)(snap - h);
                got = avail < 0 ? 0 : (avail > (int)m ? (int)m : avail);
                if (got < (int)m) a_add(&t.ring_head[k0], (unsigned int)(got - (int)m));   // give back the overshoot
            }
        }
        h = __shfl(h, l0);
        got = __shfl(got, l0);
        if (want && k == k0 && (int)rank < got) slab = t.ring[k0][(h + rank) & (t.ring_cap[k0] - 1)];
        todo &= ~grp;
    }
    const bool need = want && slab == 0;
    const uint32_t units = need ? slab_units(k, t.stride) : 0;
    const uint32_t incl = wave_incl_scan(units);
    const uint32_t total = __shfl(incl, 63);
    if (total == 0) return slab;
    unsigned long long base = 0;
    if (lane == 0) base = a_add(t.bump, (unsigned long long)total);
    base = __shfl(base, 0);
    if (need) {
        const unsigned long long u = base + incl - units;
        if (u + units <= t.arena_units) slab = (uint32_t)u;
    }
    return slab;
}

// Rank of every lane with pred inside its class, counted over the whole block: one LDS atomic per
// wave and class.  cnt: NUM_CLASSES words of shared memory, zeroed, barrier before and after by the caller.
__device__ inline uint32_t block_class_rank(bool pred, uint32_t k, uint32_t* cnt) {
    const uint32_t lane = lane_id();
    uint32_t rank = 0;
    uint64_t todo = __ballot(pred);
    while (todo) {
        const int l0 = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(k, l0);
        const uint64_t grp = __ballot(pred && k == k0);
        uint32_t base = 0;
        if ((int)lane == l0) base = atomicAdd(&cnt[k0], (uint32_t)__popcll(grp));
        base = __shfl(base, l0);
        if (pred && k == k0) rank = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1));
        todo &= ~grp;
    }
    return rank;
}

constexpr int BLOCK_SLAB_LDS = 4 * NUM_CLASSES + 4;     // words of shared memory the two block helpers need

// Block-cooperative form: ONE ring pop per size class and ONE bump add per block, whatever mix of
// classes the block's threads ask for (barriers of many sizes in one batch must not fall back to an
// atomic per wave and class).  Every thread of the block must call it.  lds: BLOCK_SLAB_LDS words.
__device__ inline uint32_t block_slab_alloc(const Table& t, bool want, uint32_t k, uint32_t* lds) {
    uint32_t* cnt = lds;                         // requests per class
    uint32_t* head = lds + NUM_CLASSES;          // ring position of the class's pop
    uint32_t* got = lds + 2 * NUM_CLASSES;       // how many of them the ring could serve
    uint32_t* off = lds + 3 * NUM_CLASSES;       // unit offset of the class inside the block's bump reservation
    uint32_t* base = lds + 4 * NUM_CLASSES;      // [0..1] the reservation's first unit, [2] anything requested
    if (threadIdx.x < NUM_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t rank = block_class_rank(want, k, cnt);
    __syncthreads();
    if (threadIdx.x < NUM_CLASSES) {
        const uint32_t c = threadIdx.x, m = cnt[c];
        uint32_t h = 0;
        int g = 0;
        if (m) {
            const unsigned int snap = t.ring_tail_snap[c];
            h = a_add(&t.ring_head[c], m);
            const int avail = (int)(snap - h);
            g = avail < 0 ? 0 : (avail > (int)m ? (int)m : avail);
            if (g < (int)m) a_add(&t.ring_head[c], (unsigned int)(g - (int)m));    // give back the overshoot
        }
        head[c] = h;
        got[c] = (uint32_t)g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
        uint32_t any = 0;
        for (uint32_t c = 0; c < NUM_CLASSES; ++c) {
            off[c] = (uint32_t)total;
            total += (unsigned long long)(cnt[c] - got[c]) * slab_units(c, t.stride);
            any |= cnt[c];
        }
        const unsigned long long b = total ? a_add(t.bump, total) : 0ull;
        base[0] = (uint32_t)b;
        base[1] = (uint32_t)(b >> 32);
        base[2] = any;
    }
    __syncthreads();
    uint32_t slab = 0;
    if (want) {
        if (rank < got[k]) {
            slab = t.ring[k][(head[k] + rank) & (t.ring_cap[k] - 1)];
        } else {
            const uint32_t units = slab_units(k, t.stride);
            const unsigned long long u = (((unsigned long long)base[1] << 32) | base[0]) + off[k] +
                                         (unsigned long long)(rank - got[k]) * units;
            if (u + units <= t.arena_units) slab = (uint32_t)u;
        }
    }
    __syncthreads();
    return slab;
}

// Returns the slabs of lanes with has (class k) to the free rings.
__device__ inline void wave_slab_free(const Table& t, bool has, uint32_t k, uint32_t slab) {
    const uint32_t lane = lane_id();
    has = has && slab != 0;
    uint64_t todo = __ballot(has);
    while (todo) {
        const int l0 = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(k, l0);
        const uint64_t grp = __ballot(has && k == k0);
        const uint32_t rank = (uint32_t)__popcll(grp & ((1ull << lane) - 1));
        unsigned int tl = 0;
        if ((int)lane == l0) tl = a_add(&t.ring_tail[k0], (unsigned int)__popcll(grp));
        tl = __shfl(tl, l0);
        if (has && k == k0) t.ring[k0][(tl + rank) & (t.ring_cap[k0] - 1)] = slab;
        todo &= ~grp;
    }
}

// Block-cooperative form: ONE ring push per size class and block (acks of a sweep free thousands of
// slabs; a push per wave would queue up on the rings' tail words).  Every thread of the block must
// call it.  lds: BLOCK_SLAB_LDS words of shared memory.
__device__ inline void block_slab_free(const Table& t, bool has, uint32_t k, uint32_t slab, uint32_t* lds) {
    uint32_t* cnt = lds;
    uint32_t* tail = lds + NUM_CLASSES;
    has = has && slab != 0;
    if (threadIdx.x < NUM_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t rank = block_class_rank(has, k, cnt);
    __syncthreads();
    if (threadIdx.x < NUM_CLASSES) {
        const uint32_t c = threadIdx.x;
        tail[c] = cnt[c] ? a_add(&t.ring_tail[c], cnt[c]) : 0u;
    }
    __syncthreads();
    if (has) t.ring[k][(tail[k] + rank) & (t.ring_cap[k] - 1)] = slab;
    __syncthreads();
}

// ---- blocks for long messages ----------------------------------------------------------------
__host__ __device__ inline uint32_t ovf_class(uint32_t len) {
    const uint32_t units = (len + 63) >> 6;
    uint32_t k = 0;
    while ((1u << k) < units) ++k;
    return k;
}

// Wave-cooperative like the slab helpers.  Returns a block of 1 << k units or 0 (arena exhausted).
__device__ inline uint32_t wave_ovf_alloc(const Table& t, bool want, uint32_t k) {
    const uint32_t lane = lane_id();
    uint32_t unit = 0;
    uint64_t todo = __ballot(want);
    while (todo) {
        const int l0 = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(k, l0);
        const uint64_t grp = __ballot(want && k == k0);
        const uint32_t m = (uint32_t)__popcll(grp);
        const uint32_t rank = (uint32_t)__popcll(grp & ((1ull << lane) - 1));
        uint32_t h = 0;
        int got = 0;
        if ((int)lane == l0) {
            const unsigned int snap = t.oring_tail_snap[k0];
            if ((int)(snap - a_load(&t.oring_head[k0])) > 0) {
                h = a_add(&t.oring_head[k0], m);
                const int avail = (int)(snap - h);
                got = avail < 0 ? 0 : (avail > (int)m ? (int)m : avail);
                if (got < (int)m) a_add(&t.oring_head[k0], (unsigned int)(got - (int)m));
            }
        }
        h = __shfl(h, l0);
        got = __shfl(got, l0);
        if (want && k == k0 && (int)rank < got) unit = t.oring[k0][(h + rank) & (t.oring_cap[k0] - 1)];
        todo &= ~grp;
    }
    const bool need = want && unit == 0;
    const uint32_t units = need ? (1u << k) : 0;
    const uint32_t incl = wave_incl_scan(units);
    const uint32_t total = __shfl(incl, 63);
    if (total == 0) return unit;
    unsigned long long base = 0;
    if (lane == 0) base = a_add(t.bump, (unsigned long long)total);
    base = __shfl(base, 0);
    uint32_t missed = 0;
    if (need) {
        const unsigned long long u = base + incl - units;
        if (u + units <= t.arena_units) unit = (uint32_t)u;
        else missed = units;
    }
    // what did not fit goes back to the pointer, so that smaller requests can still use the tail
    const uint32_t back = __shfl(wave_incl_scan(missed), 63);
    if (back && lane == 0) a_add(t.bump, 0ull - (unsigned long long)back);
    return unit;
}

__device__ inline void wave_ovf_free(const Table& t, bool has, uint32_t k, uint32_t unit) {
    const uint32_t lane = lane_id();
    has = has && unit != 0 && k < OVF_CLASSES;
    uint64_t todo = __ballot(has);
    while (todo) {
        const int l0 = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k0 = __shfl(k, l0);
        const uint64_t grp = __ballot(has && k == k0);
        const uint32_t rank = (uint32_t)__popcll(grp & ((1ull << lane) - 1));
        unsigned int tl = 0;
        if ((int)lane == l0) tl = a_add(&t.oring_tail[k0], (unsigned int)__popcll(grp));
        tl = __shfl(tl, l0);
        if (has && k == k0) t.oring[k0][(tl + rank) & (t.oring_cap[k0] - 1)] = unit;
        todo &= ~grp;
    }
}

// redis/complete's deletions (redis.clj:180-185) for a slot this thread owns.  The slab goes
// back through wave_slab_free by the caller.
__device__ inline void clear_slot(const Table& t, uint64_t s) {
    Slot* sl = &t.slots[s];
    sl->count = 0;
    sl->meta = 0;
    sl->slab = 0;
    sl->chain = 0;
    sl->deadline = 0;
    sl->join_ts = 0;
    sl->join_ts_read = 0;
    sl->bits = 0;
    sl->fp_hi = 0;
    sl->fp_lo = FP_TOMB;
    t.q[s] = Q_NONE;
}

__device__ inline uint64_t* slab_bitmap(const Table& t, uint32_t slab) {
    return reinterpret_cast<uint64_t*>(t.arena + (uint64_t)slab * 64);
}
__device__ inline uint2* slab_posmeta(const Table& t, uint32_t slab, uint32_t k) {
    return reinterpret_cast<uint2*>(t.arena + (uint64_t)slab * 64 + slab_posmeta_off(k));
}
__device__ inline uint8_t* slab_payload(const Table& t, uint32_t slab, uint32_t k) {
    return t.arena + (uint64_t)slab * 64 + slab_payload_off(k);
}

// Whole wave, uniform arguments: gives back the long-message blocks of positions [from, to) of one
// slab (class k) and zeroes their lengths.
__device__ inline void wave_release_long(const Table& t, uint32_t slab, uint32_t k, uint32_t from, uint32_t to) {
    uint2* pm = slab_posmeta(t, slab, k);
    const uint8_t* pay = slab_payload(t, slab, k);
    for (uint32_t p0 = from; p0 < to; p0 += 64) {
        const uint32_t p = p0 + lane_id();
        bool has = false;
        uint32_t kk = 0, unit = 0;
        if (p < to) {
            const uint32_t len = pm[p].x;
            if (len > t.stride && len != DJ_NIL) {
                const dj_long_ref ref = *reinterpret_cast<const dj_long_ref*>(pay + (uint64_t)p * t.stride);
                has = true;
                kk = ref.size_class;
                unit = ref.unit;
                pm[p].x = 0;
            }
        }
        wave_ovf_free(t, has, kk, unit);
    }
}

// Every lane of the wave calls this; a lane with `mine` owns a barrier whose blocks must go.
__device__ inline void wave_release_long_each(const Table& t, bool mine, uint32_t slab, uint32_t k, uint32_t from,
                                              uint32_t to) {
    uint64_t todo = __ballot(mine && slab != 0);
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        wave_release_long(t, __shfl(slab, l), __shfl(k, l), __shfl(from, l), __shfl(to, l));
    }
}

// Every lane of the wave calls this; a lane with `mine` owns a barrier that moves from slab `from` (class kf) to the
// larger slab `to` (class kt): the {len, max-wait} words and the rows of all positions of the old class are copied by
// the whole wave (a long message's reference moves with its row), the new positions get the default word.
__device__ inline void wave_carry_each(const Table& t, bool mine, uint32_t from, uint32_t kf, uint32_t to, uint32_t kt) {
    uint64_t todo = __ballot(mine);
    const uint32_t lane = lane_id();
    while (todo) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t f = __shfl(from, l), a = __shfl(kf, l), d = __shfl(to, l), b = __shfl(kt, l);
        const uint32_t np = class_positions(a), nq = class_positions(b);
        const uint2* pf = slab_posmeta(t, f, a);
        uint2* pt = slab_posmeta(t, d, b);
        for (uint32_t p = lane; p < nq; p += 64) pt[p] = p < np ? pf[p] : make_uint2(t.stride, 0);
        const uint4* rf = reinterpret_cast<const uint4*>(slab_payload(t, f, a));
        uint4* rt = reinterpret_cast<uint4*>(slab_payload(t, d, b));
        const uint64_t chunks = (uint64_t)np * (t.stride >> 4);
        for (uint64_t c = lane; c < chunks; c += 64) rt[c] = rf[c];
    }
}

// A slab that starts a new life (fork): every position's {len, max_wait} word holds the value that
// needs no write later, a full row without max-wait (so no position holds a long message either).
// The area is a multiple of 64 bytes.
__device__ inline void fill_posmeta_default(const Table& t, uint32_t slab, uint32_t k) {
    uint4* pm = reinterpret_cast<uint4*>(slab_posmeta(t, slab, k));
    const uint32_t n16 = (slab_payload_off(k) - slab_posmeta_off(k)) >> 4;
    const uint4 v = make_uint4(t.stride, 0, t.stride, 0);
    for (uint32_t j = 0; j < n16; ++j) pm[j] = v;
}

// The same for every lane of a wave that wants it (call with the whole wave).  Slabs of up to 16 positions are
// filled by their own lane (a line or two); larger ones one after the other by all 64 lanes, 16 bytes each per
// step, instead of one lane writing up to 8 KB in a loop while its neighbours wait.
__device__ inline void wave_fill_posmeta_each(const Table& t, bool want, uint32_t slab, uint32_t k) {
    const uint32_t n16 = want ? (slab_payload_off(k) - slab_posmeta_off(k)) >> 4 : 0;
    if (want && n16 <= 8) fill_posmeta_default(t, slab, k);
    const uint4 v = make_uint4(t.stride, 0, t.stride, 0);
    for (uint64_t m = __ballot(want && n16 > 8); m; m &= m - 1) {
        const int l = __ffsll((unsigned long long)m) - 1;
        const uint32_t sb = __shfl(slab, l), kk = __shfl(k, l), cnt = __shfl(n16, l);
        uint4* pm = reinterpret_cast<uint4*>(slab_posmeta(t, sb, kk));
        for (uint32_t j = lane_id(); j < cnt; j += 64) pm[j] = v;
    }
}

// node packing of the join link pass: next (25) | position (10) | max-wait (29)
constexpr uint32_t NODE_MW_LIMIT = 1u << 29;
__device__ inline uint64_t node_pack(uint32_t next, uint32_t pos, uint32_t mw) {
    return (uint64_t)next | ((uint64_t)pos << 25) | ((uint64_t)mw << 35);
}
__device__ inline uint32_t node_next(uint64_t x) { return (uint32_t)(x & 0x1FFFFFFu); }
__device__ inline uint32_t node_pos(uint64_t x) { return (uint32_t)((x >> 25) & 0x3FFu); }
__device__ inline uint32_t node_mw(uint64_t x) { return (uint32_t)(x >> 35); }

// where an event's message goes: slab (32) | size class (4) | position (10) | flags.  With DST_FULL and without
// DST_PM the copy pass needs nothing else from the event's record.
constexpr uint64_t DST_FULL = 1ull << 46;   // the message fills its row exactly
constexpr uint64_t DST_OVF = 1ull << 47;    // the barrier holds (or held) long messages: the position's old length matters
constexpr uint64_t DST_CX = 1ull << 48;     // set by cx_scan: the event's barrier is registered with the parallel long-chain kernels
constexpr uint64_t DST_PM = 1ull << 62;     // also write the position's {len, max_wait}
__device__ inline uint64_t dst_pack(uint32_t slab, uint32_t k, uint32_t pos) {
    return (uint64_t)slab | ((uint64_t)k << 32) | ((uint64_t)pos << 36);
}
__device__ inline uint32_t dst_class(uint64_t d) { return (uint32_t)(d >> 32) & 15u; }
__device__ inline uint32_t dst_pos(uint64_t d) { return (uint32_t)(d >> 36) & 0x3FFu; }

}  // namespace dj
