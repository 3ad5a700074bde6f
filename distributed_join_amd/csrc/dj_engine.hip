// dj_engine.hip — host side of the C ABI declared in include/dj_engine.h.
//
// One engine owns one GPU, one HIP stream and all device memory of its barrier table.
// Every entry point validates its arguments on the host (sizes, alignment, clock range)
// before a kernel is launched, so that a kernel never sees a shape it was not written for.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "dj_kernels.hpp"

using namespace dj;

// ---- scratch of dj_join_batch_static -----------------------------------------------------------
// The per-batch work arrays of a join batch (link words, message destinations, event-to-slot map, list of
// long chains, counters) for batches of at most DJ_STATIC_JOIN_MAX events, as a device array of the code
// object instead of hipMalloc memory.  Every GPU the library runs on holds its own copy at its own
// address; `scratch` itself is only the host-side handle of the symbol: host code uses the per-device
// address from hipGetSymbolAddress (g_scratch), resolved by dj_create with that device current.
namespace {
constexpr size_t SCR_NODE = 0;
constexpr size_t SCR_DST = SCR_NODE + (size_t)DJ_STATIC_JOIN_MAX * 8;
constexpr size_t SCR_EV_SLOT = SCR_DST + (size_t)DJ_STATIC_JOIN_MAX * 8;
constexpr size_t SCR_COMPLEX = SCR_EV_SLOT + (size_t)DJ_STATIC_JOIN_MAX * 4;
constexpr size_t SCR_COUNTERS = SCR_COMPLEX + (size_t)DJ_STATIC_JOIN_MAX * 4;
constexpr size_t SCR_BYTES = SCR_COUNTERS + 256;       // two banks of C_COUNT counters
static_assert(2 * C_COUNT * 4 <= 256, "counter banks must fit their 256 bytes");
constexpr int SCR_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[SCR_BYTES];
static unsigned char* g_scratch[SCR_MAX_DEVICES];      // device address of `scratch` on each device, nullptr = not resolved
static std::mutex g_scratch_mu;

struct dj_engine {
    dj_config cfg;
    Table t;
    Work w;
    hipStream_t stream;
    // single-key staging (device) for dj_snapshot_one / dj_queue_scores
    dj_cont* d_one_cont;
    uint64_t* d_one_scores;
    uint32_t* d_one_lens;
    uint32_t* d_one_mws;
    uint8_t* d_one_payload;
    int32_t* d_one_err;
    dj_sweep_result* d_result;   // internal result block when the caller passes none
    uint32_t* d_part_counts;     // partition scratch: segments * ranks, at most 32768 words (engine stream)
    uint32_t* d_part_counts2;    // same, for partitions launched on a caller stream
    void* allocs[128];
    int n_allocs;
    std::string err;
    std::recursive_mutex mu;     // entry points serialise themselves: callers may share an engine between threads
    Slot* slots_alt;             // second table for dj_compact (always all-empty between calls)
    uint64_t* q_alt;
    uint32_t n_cus;              // compute units of the device
    uint32_t bfat_max;           // largest number of sub-ranges per bucket the key buffer was sized for
    int bucket_mode;             // -1: the bucketed join passes where they pay; 0: never; 1: wherever they apply (tests)
    int fuse_off;                // 1: the link pass never copies messages itself (dj_set_hot_path mode + 4: tests of the separate copy pass)
    int hlen_mode;               // 1: join statuses carry HLEN above DJ_JOIN_HLEN_SHIFT (dj_set_join_hlen)
    int cx_force;                // -1: automatic; else bit 0: parallel hot-barrier kernels, bit 1: wide register replay (tests)
    unsigned int* h_maxchain;    // pinned: longest chain of the previous join batch (picks the hot-barrier kernels)
    uint64_t used_upper;         // upper bound of non-empty slots (live + tombstones) since the last compaction
    // an automatic compaction does not wait for its live count: it arrives in pinned memory
    // (h_maxchain + 2, 8 bytes) and is picked up by the next fork
    hipEvent_t compact_ev;
    bool compact_pending;
    uint64_t claimed_since_compact;
    int ctr_bank, sweep_bank;    // which half of the double-banked scratch the next call uses
    unsigned int* d_counters;    // 2 x C_COUNT
    unsigned int* d_fork_contested;      // 2 words: this fork batch's and the next one's (see Work::fork_contested)
    int fork_bank;
    unsigned long long* d_live;  // 2 words: live keys counted by this compaction and by the next one (each zeroes the other's)
    int live_bank;
    unsigned long long* d_sweep; // 2 x S_COUNT
    unsigned long long* d_due;   // the two banks of the sweep's scan bound (Table::due_min points to one of them)
    // optional per-kernel timing (dj_profile_enable): one event pair per kernel id
    int profiling;
    hipEvent_t ev_start[DJ_PROF_KERNELS], ev_stop[DJ_PROF_KERNELS];
    int ev_used[DJ_PROF_KERNELS];
    unsigned char* scratch_ptr;  // this device's copy of the static scratch array (dj_join_batch_static)
};

// Launch `call` on the engine's stream; with profiling on, bracket it with the kernel id's events.
#define LAUNCH(e, kid, call)                                                   \
    do {                                                                       \
        if ((e)->profiling) (void)hipEventRecord((e)->ev_start[kid], (e)->stream); \
        call;                                                                  \
        if ((e)->profiling) {                                                  \
            (void)hipEventRecord((e)->ev_stop[kid], (e)->stream);              \
            (e)->ev_used[kid] = 1;                                             \
        }                                                                      \
    } while (0)

static thread_local std::string g_create_err;   // dj_create's failure text, read through dj_last_error(NULL) by the same thread

static int fail(dj_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(e, call)                                                                            \
    do {                                                                                            \
        hipError_t _r = (call);                                                                     \
        if (_r != hipSuccess) return fail(e, DJ_E_HIP, "%s failed: %s", #call, hipGetErrorString(_r)); \
    } while (0)

static int dev_alloc(dj_engine* e, void** p, size_t bytes) {
    if (e->n_allocs >= 128) return fail(e, DJ_E_INVALID, "allocation table full");
    hipError_t r = hipMalloc(p, bytes ? bytes : 16);
    if (r != hipSuccess) return fail(e, DJ_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(r));
    e->allocs[e->n_allocs++] = *p;
    return DJ_OK;
}
#define ALLOC(e, ptr, bytes)                                         \
    do {                                                             \
        int _rc = dev_alloc(e, (void**)&(ptr), (bytes));             \
        if (_rc != DJ_OK) return _rc;                                \
    } while (0)

// Table size from which a sweep drains the ready list instead of scanning the queue words.  Below it the scan
// costs less than the list does (measured at 2 M slots: the list's appends add 4 us to a join batch of 2^20
// events, the pop is no faster, and the gather and ack passes lose 8 us because the continuations no longer
// come in slot order); the scan grows with the table (8 bytes per slot), the list with what is due.  Measured pop
// step with 65 536 barriers due per sweep, scan against list: 4 M slots 22 / 19 us when they completed, 22 / 32 us
// when they timed out; 8 M slots 36 / 16 and 36 / 40; 16 M slots 62 / 16 and 62 / 39.
constexpr uint64_t READY_AUTO_SLOTS = 1ull << 23;

static bool clock_ok(uint64_t now_ms) { return now_ms >= 1000000000000ull && now_ms < 9000000000000ull; }
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static uint32_t grid_for(uint64_t n, uint32_t block) { return (uint32_t)((n + block - 1) / block); }
static uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

// ---- fingerprint ------------------------------------------------------------------
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

extern "C" void dj_fingerprint(const void* id, size_t len, uint64_t* fp_lo, uint64_t* fp_hi) {
    const uint8_t* p = static_cast<const uint8_t*>(id);
    uint64_t h1 = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
    uint64_t h2 = 0xC2B2AE3D27D4EB4Full + ((uint64_t)len << 32);
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h1 = rotl64(h1 ^ w, 27) * 0x9FB21C651E98DF25ull + 0x52DCE729ull;
        h2 = rotl64(h2 + w, 31) * 0xD6E8FEB86659FD93ull ^ (h1 >> 29);
    }
    uint64_t w = 0;
    memcpy(&w, p + i, len - i);
    w ^= (uint64_t)(len - i) << 56;
    h1 = rotl64(h1 ^ w, 27) * 0x9FB21C651E98DF25ull + 0x52DCE729ull;
    h2 = rotl64(h2 + w, 31) * 0xD6E8FEB86659FD93ull ^ (h1 >> 29);
    uint64_t lo = fmix64(h1 ^ rotl64(h2, 23));
    uint64_t hi = fmix64(h2 + 0x9E3779B97F4A7C15ull * (h1 | 1));
    if (lo < 2) lo += 2;   // 0 and 1 mark empty and deleted slots
    if (hi == 0) hi = 1;   // 0 marks a slot whose key is still being written
    *fp_lo = lo;
    *fp_hi = hi;
}

extern "C" void dj_fingerprint_batch(const uint8_t* bytes, const uint64_t* offsets, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; ++i) dj_fingerprint(bytes + offsets[i], offsets[i + 1] - offsets[i], &out[2 * i], &out[2 * i + 1]);
}

extern "C" uint32_t dj_owner_of(uint64_t fp_lo, uint64_t fp_hi, uint32_t n_ranks) {
    return n_ranks ? owner_of(fp_lo, fp_hi, n_ranks) : 0;
}

// ---- lifecycle -----------------------------------------------------------------------
extern "C" const char* dj_last_error(const dj_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

extern "C" void dj_destroy(dj_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    for (int k = 0; k < DJ_PROF_KERNELS; ++k) {
        if (e->ev_start[k]) (void)hipEventDestroy(e->ev_start[k]);
        if (e->ev_stop[k]) (void)hipEventDestroy(e->ev_stop[k]);
    }
    for (int i = 0; i < e->n_allocs; ++i) (void)hipFree(e->allocs[i]);
    if (e->h_maxchain) (void)hipHostFree(e->h_maxchain);
    if (e->compact_ev) (void)hipEventDestroy(e->compact_ev);
    delete e;
}

static int create_impl(dj_engine* e) {
    const dj_config& c = e->cfg;
    HIP_TRY(e, hipSetDevice(c.device));
    HIP_TRY(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (c.device < 0 || c.device >= SCR_MAX_DEVICES) return fail(e, DJ_E_INVALID, "device ordinal must be 0..%d", SCR_MAX_DEVICES - 1);
    {   // once per device, with that device current
        std::lock_guard<std::mutex> g(g_scratch_mu);
        if (!g_scratch[c.device]) {
            unsigned char* p = nullptr;
            HIP_TRY(e, hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
            g_scratch[c.device] = p;
        }
        e->scratch_ptr = g_scratch[c.device];
    }
    Table& t = e->t;
    memset(&t, 0, sizeof t);
    uint64_t slots = next_pow2(c.capacity_barriers * 2);
    if (slots < 1024) slots = 1024;
    if (slots > (1ull << 30)) return fail(e, DJ_E_INVALID, "capacity_barriers too large (max 2^29)");
    t.mask = slots - 1;
    t.stride = c.payload_stride;
    t.max_message = c.max_message_bytes > c.payload_stride ? c.max_message_bytes : c.payload_stride;
    t.max_count = c.max_count;
    t.max_retry_publish = c.max_retry_publish;
    t.max_retry_timeout_publish = c.max_retry_timeout_publish;
    t.check_publish_delay_ms = c.check_publish_delay_ms;
    t.arena_units = c.arena_bytes / 64;
    if (t.arena_units >= (1ull << 32)) return fail(e, DJ_E_INVALID, "arena_bytes too large (max 256 GiB)");
    ALLOC(e, t.slots, slots * sizeof(Slot));
    ALLOC(e, t.q, slots * 8);
    ALLOC(e, e->slots_alt, slots * sizeof(Slot));
    ALLOC(e, e->q_alt, slots * 8);
    HIP_TRY(e, hipMemsetAsync(e->slots_alt, 0, slots * sizeof(Slot), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->q_alt, 0xFF, slots * 8, e->stream));
    e->used_upper = 0;
    e->compact_pending = false;
    e->claimed_since_compact = 0;
    HIP_TRY(e, hipEventCreateWithFlags(&e->compact_ev, hipEventDisableTiming));
    ALLOC(e, t.arena, t.arena_units * 64);
    ALLOC(e, t.bump, 8);
    ALLOC(e, t.ring_head, NUM_CLASSES * 4);
    ALLOC(e, t.ring_tail, NUM_CLASSES * 4);
    ALLOC(e, t.ring_tail_snap, NUM_CLASSES * 4);
    ALLOC(e, t.stat, 4 * 8);
    const uint32_t max_class = class_of(c.max_count);
    for (uint32_t k = 0; k < NUM_CLASSES; ++k) {
        // slabs of a class that can exist at once: bounded by arena / slab size, and (one slab per barrier, the
        // ring holding only slabs that no slot refers to) by the number of table slots, which is what bounds
        // the live barriers (more than capacity_barriers of them are accepted while slots are left)
        uint64_t cap = k <= max_class ? t.arena_units / slab_units(k, t.stride) + 1 : 1;
        if (cap > slots + 1) cap = slots + 1;
        cap = next_pow2(cap);
        t.ring_cap[k] = (uint32_t)cap;
        ALLOC(e, t.ring[k], cap * 4);
    }
    // free lists of the long-message blocks: only the classes a message longer than one row can need
    ALLOC(e, t.oring_head, OVF_CLASSES * 4);
    ALLOC(e, t.oring_tail, OVF_CLASSES * 4);
    ALLOC(e, t.oring_tail_snap, OVF_CLASSES * 4);
    for (uint32_t k = 0; k < OVF_CLASSES; ++k) {
        const bool used = t.max_message > t.stride && (64ull << k) > t.stride && k <= ovf_class(t.max_message);
        const uint64_t cap = used ? next_pow2(t.arena_units / (1ull << k) + 1) : 1;
        t.oring_cap[k] = (uint32_t)cap;
        ALLOC(e, t.oring[k], cap * 4);
    }
    HIP_TRY(e, hipMemsetAsync(t.oring_head, 0, OVF_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.oring_tail, 0, OVF_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.oring_tail_snap, 0, OVF_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.slots, 0, slots * sizeof(Slot), e->stream));
    HIP_TRY(e, hipMemsetAsync(t.q, 0xFF, slots * 8, e->stream));
    const unsigned long long one = 1;   // arena unit 0 is reserved: slab 0 means "no slab"
    HIP_TRY(e, hipMemcpyAsync(t.bump, &one, 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.ring_head, 0, NUM_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.ring_tail, 0, NUM_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.ring_tail_snap, 0, NUM_CLASSES * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.stat, 0, 4 * 8, e->stream));
    // ready list: room for the completions of four batches of max_batch events between two sweeps; beyond
    // that the list overflows and the next sweep scans the queue words instead
    t.ready_shard_cap = (uint32_t)next_pow2(((uint64_t)c.max_batch * 4 + READY_SHARDS - 1) / READY_SHARDS);
    if (t.ready_shard_cap < 64) t.ready_shard_cap = 64;
    ALLOC(e, t.ready, (size_t)READY_SHARDS * t.ready_shard_cap * 4);
    ALLOC(e, t.ready_n, (size_t)READY_SHARDS * READY_PAD * 4);
    ALLOC(e, t.ready_ovf, 4);
    ALLOC(e, e->d_due, 2 * 8);
    HIP_TRY(e, hipMemsetAsync(t.ready_n, 0, (size_t)READY_SHARDS * READY_PAD * 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(t.ready_ovf, 0, 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_due, 0xFF, 2 * 8, e->stream));      // nothing queued: no bound
    t.due_min = e->d_due;
    ALLOC(e, t.qmin, (slots >> REGION_SHIFT) * 8);
    HIP_TRY(e, hipMemsetAsync(t.qmin, 0xFF, (slots >> REGION_SHIFT) * 8, e->stream));
    t.ready_on = slots >= READY_AUTO_SLOTS ? 1u : 0u;

    Work& w = e->w;
    const uint64_t nb = c.max_batch;
    ALLOC(e, w.node, nb * 8);
    ALLOC(e, w.dst, nb * 8);
    ALLOC(e, w.ev_slot, nb * 4);
    ALLOC(e, w.complex_, nb * 4);
    ALLOC(e, w.slot_of, nb * 4);
    ALLOC(e, w.cont_slot, (nb > slots ? nb : slots) * 4);
    ALLOC(e, w.regions, (slots >> REGION_SHIFT) * 4);
    ALLOC(e, e->d_fork_contested, 2 * 4);
    HIP_TRY(e, hipMemsetAsync(e->d_fork_contested, 0, 2 * 4, e->stream));
    ALLOC(e, e->d_live, 2 * 8);
    HIP_TRY(e, hipMemsetAsync(e->d_live, 0, 2 * 8, e->stream));
    e->live_bank = 0;
    ALLOC(e, e->d_counters, 2 * C_COUNT * 4);
    HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 2 * C_COUNT * 4, e->stream));
    // room for the keys of one batch of max_batch events: each of the 1024 buckets holds bfat_max x RB_CAP keys
    e->bfat_max = 1;
    while ((uint64_t)BK_MAX_BUCKETS * BK_MEAN_MAX * e->bfat_max < nb && e->bfat_max < 16) e->bfat_max <<= 1;
    ALLOC(e, w.bkeys, (size_t)BK_MAX_BUCKETS * RB_CAP * e->bfat_max * 8);
    ALLOC(e, w.bcursor, BK_MAX_BUCKETS * 4);
    HIP_TRY(e, hipMemsetAsync(w.bcursor, 0, BK_MAX_BUCKETS * 4, e->stream));
    w.bbuckets = w.bshift = w.bsubshift = w.bmwbits = 0;
    w.bcap = RB_CAP;
    w.bfat = 1;
    ALLOC(e, w.cx_keys, CX_HASH * 4);
    ALLOC(e, w.cx_val, CX_HASH * 4);
    ALLOC(e, w.cx_off, CX_CAP * 4);
    ALLOC(e, w.cx_mm, (size_t)CX_POOL * 8);
    ALLOC(e, w.cx_atmin, CX_CAP * 8);
    ALLOC(e, w.cx_cnt, CX_CAP * 4);
    ALLOC(e, w.cx_T, CX_CAP * 4);
    ALLOC(e, w.cx_nq, CX_CAP * 4);
    ALLOC(e, w.cx_hot, CX_CAP * 4);
    w.cx_on = 0;
    HIP_TRY(e, hipHostMalloc((void**)&e->h_maxchain, 64, hipHostMallocDefault));
    memset(e->h_maxchain, 0, 64);
    w.hint = e->h_maxchain;       // hipHostMalloc memory is addressable from the device
    ALLOC(e, e->d_sweep, 2 * S_COUNT * 8);
    HIP_TRY(e, hipMemsetAsync(e->d_sweep, 0, 2 * S_COUNT * 8, e->stream));
    e->ctr_bank = e->sweep_bank = e->fork_bank = 0;
    w.counters = e->d_counters;
    w.counters_next = e->d_counters + C_COUNT;
    w.sweep = e->d_sweep;
    w.sweep_next = e->d_sweep + S_COUNT;

    ALLOC(e, e->d_one_cont, sizeof(dj_cont));
    ALLOC(e, e->d_one_scores, 16);
    ALLOC(e, e->d_one_lens, (size_t)c.max_count * 4);
    ALLOC(e, e->d_one_mws, (size_t)c.max_count * 4);
    ALLOC(e, e->d_one_payload, (size_t)c.max_count * c.payload_stride);
    ALLOC(e, e->d_one_err, 4);
    ALLOC(e, e->d_result, sizeof(dj_sweep_result));
    ALLOC(e, e->d_part_counts, 32768 * 4);
    ALLOC(e, e->d_part_counts2, 32768 * 4);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return DJ_OK;
}

extern "C" int dj_create(const dj_config* cfg, dj_engine** out) {
    if (!cfg || !out) return fail(nullptr, DJ_E_INVALID, "null argument");
    if (cfg->struct_size != sizeof(dj_config)) return fail(nullptr, DJ_E_INVALID, "dj_config.struct_size mismatch");
    if (cfg->max_count < 1 || cfg->max_count > 999) return fail(nullptr, DJ_E_INVALID, "max_count must be 1..999");
    if (cfg->payload_stride < 64 || cfg->payload_stride % 64 || cfg->payload_stride > (1u << 20))
        return fail(nullptr, DJ_E_INVALID, "payload_stride must be a multiple of 64 in [64, 1 MiB]");
    if (cfg->max_message_bytes > (1u << 20)) return fail(nullptr, DJ_E_INVALID, "max_message_bytes must be <= 1 MiB");
    if (cfg->max_batch < 1 || cfg->max_batch > (1u << 24)) return fail(nullptr, DJ_E_INVALID, "max_batch must be 1..2^24");
    if (cfg->capacity_barriers < 1 || cfg->capacity_barriers > (1ull << 29))
        return fail(nullptr, DJ_E_INVALID, "capacity_barriers must be 1..2^29");
    if (cfg->max_retry_publish > 200 || cfg->max_retry_timeout_publish > 200)
        return fail(nullptr, DJ_E_INVALID, "retry caps must be <= 200");
    if (cfg->arena_bytes / 64 < 1 + (uint64_t)slab_units(class_of(cfg->max_count), cfg->payload_stride))
        return fail(nullptr, DJ_E_INVALID, "arena_bytes smaller than one slab of max_count positions");
    dj_engine* e = new (std::nothrow) dj_engine();
    if (!e) return fail(nullptr, DJ_E_NOMEM, "out of host memory");
    e->cfg = *cfg;
    e->n_allocs = 0;
    e->stream = nullptr;
    e->profiling = 0;
    e->h_maxchain = nullptr;
    e->compact_ev = nullptr;
    e->cx_force = -1;
    e->hlen_mode = 0;
    e->fuse_off = 0;
    e->bucket_mode = -1;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || cus < 1) cus = 256;
        e->n_cus = (uint32_t)cus;
    }
    e->scratch_ptr = nullptr;
    for (int k = 0; k < DJ_PROF_KERNELS; ++k) { e->ev_start[k] = nullptr; e->ev_stop[k] = nullptr; e->ev_used[k] = 0; }
    int rc = create_impl(e);
    if (rc != DJ_OK) {
        g_create_err = e->err;
        dj_destroy(e);
        return rc;
    }
    *out = e;
    return DJ_OK;
}

extern "C" int dj_sync(dj_engine* e) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return DJ_OK;
}

extern "C" void* dj_stream(dj_engine* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int dj_get_config(const dj_engine* e, dj_config* out) {
    if (!e || !out) return DJ_E_INVALID;
    *out = e->cfg;
    return DJ_OK;
}

extern "C" int dj_set_hot_path(dj_engine* e, int mode) {
    if (!e || mode < -1 || mode > 23) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (mode < 0) {
        e->bucket_mode = -1; e->fuse_off = 0; e->cx_force = -1;
        return DJ_OK;
    }
    // + 8: the bucketed join passes never run; + 16: they run wherever they apply; both leave the other choices to the
    // engine unless lower bits pin them as well (8 and 16 alone: everything else automatic)
    e->bucket_mode = mode >= 16 ? 1 : mode >= 8 ? 0 : -1;
    const bool only_bucket = mode == 8 || mode == 16;
    mode &= 7;
    e->fuse_off = mode >= 4 ? 1 : 0;
    if (mode >= 4) mode -= 4;
    e->cx_force = only_bucket ? -1 : mode;
    return DJ_OK;
}

extern "C" int dj_set_join_hlen(dj_engine* e, int on) {
    if (!e || on < 0 || on > 1) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    e->hlen_mode = on;
    return DJ_OK;
}

extern "C" int dj_set_sweep_path(dj_engine* e, int mode) {
    if (!e || mode < -1 || mode > 1) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    e->t.ready_on = mode < 0 ? (e->t.mask + 1 >= READY_AUTO_SLOTS ? 1u : 0u) : (uint32_t)mode;
    // whatever was queued while no list was kept is in no list: the next sweep scans and recomputes the bound
    HIP_TRY(e, hipMemsetAsync(e->t.ready_ovf, 1, 4, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_due, 0, 2 * 8, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->t.ready_n, 0, (size_t)READY_SHARDS * READY_PAD * 4, e->stream));
    return DJ_OK;
}

extern "C" int dj_profile_enable(dj_engine* e, int on) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (on)
        for (int k = 0; k < DJ_PROF_KERNELS; ++k) {
            if (!e->ev_start[k]) HIP_TRY(e, hipEventCreate(&e->ev_start[k]));
            if (!e->ev_stop[k]) HIP_TRY(e, hipEventCreate(&e->ev_stop[k]));
        }
    for (int k = 0; k < DJ_PROF_KERNELS; ++k) e->ev_used[k] = 0;
    e->profiling = on ? 1 : 0;
    return DJ_OK;
}

extern "C" int dj_profile_read(dj_engine* e, float* h_ms) {
    if (!e || !h_ms) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (int k = 0; k < DJ_PROF_KERNELS; ++k) {
        h_ms[k] = -1.0f;
        if (e->ev_used[k]) HIP_TRY(e, hipEventElapsedTime(&h_ms[k], e->ev_start[k], e->ev_stop[k]));
        e->ev_used[k] = 0;
    }
    return DJ_OK;
}

static uint32_t scan_blocks(const dj_engine* e) {
    uint64_t b = (e->t.mask + 1) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (uint32_t)b;
}

// ---- table hygiene ------------------------------------------------------------------------
// Re-inserts the live barriers into the spare table and swaps: tombstones disappear, probe chains
// shrink.  Outstanding continuations stay valid: their slot hints are verified against the
// fingerprint before use.  compact_start queues the work and the read-back of the live count;
// compact_finish waits for the count (or, with wait = false, takes it only if it has arrived).
static int compact_start(dj_engine* e) {
    Table to = e->t;
    to.slots = e->slots_alt;
    to.q = e->q_alt;
    if (e->profiling) (void)hipEventRecord(e->ev_start[DJ_PROF_COMPACT], e->stream);
    unsigned long long* live = e->d_live + e->live_bank;
    e->live_bank ^= 1;
    rebuild_kernel<<<scan_blocks(e), 256, 0, e->stream>>>(e->t, to, live, e->d_live + e->live_bank);
    HIP_TRY(e, hipGetLastError());      // (the kernel leaves the old table all-empty and marks the ready list as incomplete)
    if (e->profiling) { (void)hipEventRecord(e->ev_stop[DJ_PROF_COMPACT], e->stream); e->ev_used[DJ_PROF_COMPACT] = 1; }
    HIP_TRY(e, hipMemcpyAsync(e->h_maxchain + 2, live, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipEventRecord(e->compact_ev, e->stream));
    e->slots_alt = e->t.slots;
    e->q_alt = e->t.q;
    e->t.slots = to.slots;
    e->t.q = to.q;
    e->compact_pending = true;
    e->claimed_since_compact = 0;
    return DJ_OK;
}

static int compact_finish(dj_engine* e, bool wait) {
    if (!e->compact_pending) return DJ_OK;
    if (wait) HIP_TRY(e, hipEventSynchronize(e->compact_ev));
    else if (hipEventQuery(e->compact_ev) != hipSuccess) return DJ_OK;
    unsigned long long live;
    memcpy(&live, (const void*)(e->h_maxchain + 2), 8);
    e->used_upper = live + e->claimed_since_compact;
    e->compact_pending = false;
    return DJ_OK;
}

extern "C" int dj_compact(dj_engine* e, uint64_t* h_live) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    int rc = compact_finish(e, true);
    if (rc != DJ_OK) return rc;
    rc = compact_start(e);
    if (rc != DJ_OK) return rc;
    rc = compact_finish(e, true);
    if (rc != DJ_OK) return rc;
    if (h_live) *h_live = e->used_upper;
    return DJ_OK;
}

// ---- hot path ----------------------------------------------------------------------------
extern "C" int dj_fork_batch(dj_engine* e, uint32_t n, const dj_fork_rec* d_recs, int32_t* d_status) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (n == 0) return DJ_OK;
    if (!d_recs || !d_status || !aligned16(d_recs)) return fail(e, DJ_E_INVALID, "fork: null or misaligned pointer");
    if (n > e->cfg.max_batch) return fail(e, DJ_E_TOO_LARGE, "fork: n=%u exceeds max_batch=%u", n, e->cfg.max_batch);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // every record may claim a never-used slot; past 80 % non-empty slots lookups of absent keys get
    // long, so sweep the tombstones out first
    if (e->used_upper + n > (e->t.mask + 1) / 5 * 4) {
        // the bound may be stale: a compaction queued by an earlier call reports its live count late
        int rc = compact_finish(e, true);
        if (rc != DJ_OK) return rc;
        if (e->used_upper + n > (e->t.mask + 1) / 5 * 4) {
            // queued on the stream, nobody waits for it; used_upper stays the (now pessimistic) old
            // bound until the live count is in
            rc = compact_start(e);
            if (rc != DJ_OK) return rc;
        }
    } else {
        (void)compact_finish(e, false);
    }
    {   // of the latest join batch's first 256 events, did most store a {len, max-wait} word?
        const volatile unsigned int* hint = e->h_maxchain;
        e->t.fork_pmd = hint[4] + hint[5] + hint[6] + hint[7] >= 128 ? 1u : 0u;
    }
    e->used_upper += n;
    e->claimed_since_compact += n;
    e->w.fork_contested = e->d_fork_contested + e->fork_bank;
    e->w.fork_contested_next = e->d_fork_contested + (e->fork_bank ^ 1);
    e->fork_bank ^= 1;
    if (e->profiling) (void)hipEventRecord(e->ev_start[DJ_PROF_FORK_CLAIM], e->stream);
    fork_claim_kernel<<<grid_for(n, 256), 256, 0, e->stream>>>(e->t, e->w, d_recs, n, d_status);
    fork_contest_kernel<<<grid_for(n, 256) < 512 ? grid_for(n, 256) : 512, 256, 0, e->stream>>>(e->t, e->w, n);
    if (e->profiling) { (void)hipEventRecord(e->ev_stop[DJ_PROF_FORK_CLAIM], e->stream); e->ev_used[DJ_PROF_FORK_CLAIM] = 1; }
    LAUNCH(e, DJ_PROF_FORK_APPLY, (fork_apply_kernel<<<grid_for(n, FORK_APPLY_BLOCK), FORK_APPLY_BLOCK, 0, e->stream>>>(e->t, e->w, d_recs, n, d_status)));
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

template <int G>
static void launch_join_copy(dj_engine* e, const Work& w, uint32_t n, const dj_join_rec* recs, const uint8_t* payloads, uint64_t pitch,
                             int32_t* status) {
    const uint64_t steps = ((uint64_t)n + (64 / G) - 1) / (64 / G);          // wave steps
    uint64_t blocks = (steps + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (e->t.max_message > e->t.stride)
        join_copy_long_kernel<G><<<(uint32_t)blocks, 256, 0, e->stream>>>(e->t, w, recs, payloads, pitch, n, status);
    else
        join_copy_kernel<G><<<(uint32_t)blocks, 256, 0, e->stream>>>(e->t, w, recs, payloads, pitch, n, status);
}

constexpr unsigned int WIDE_ON = 2048;   // barriers with 9..16 events in a batch from which the wide replay pays

static int join_check(dj_engine* e, const char* what, uint32_t n, const dj_join_rec* d_recs, const uint8_t* d_payloads,
                      uint64_t pitch, uint64_t now_ms, const int32_t* d_status, uint32_t max_n, const char* limit) {
    if (!clock_ok(now_ms)) return fail(e, DJ_E_CLOCK_RANGE, "%s: now_ms=%llu outside [1e12, 9e12)", what, (unsigned long long)now_ms);
    if (n == 0) return DJ_OK;
    if (!d_recs || !d_status || !d_payloads) return fail(e, DJ_E_INVALID, "%s: null pointer", what);
    if (!aligned16(d_recs) || !aligned16(d_payloads) || pitch % 16 || pitch == 0)
        return fail(e, DJ_E_INVALID, "%s: records and payloads must be 16-byte aligned, pitch a positive multiple of 16", what);
    if (n > max_n) return fail(e, DJ_E_TOO_LARGE, "%s: n=%u exceeds %s=%u", what, n, limit, max_n);
    return DJ_OK;
}

// The join passes on the engine's stream with the work arrays of `w`.  serial_only: long chains take the
// one-wave-per-barrier kernel whatever the previous batch looked like (the parallel kernels' tables are not
// part of every work set); both forms give the same results.
static int join_launch(dj_engine* e, Work& w, bool serial_only, uint32_t n, const dj_join_rec* d_recs,
                       const uint8_t* d_payloads, uint64_t pitch, uint64_t now_ms, int32_t* d_status) {
    // h_maxchain[0] = barriers with long chains in the previous batch, [1] = its longest chain
    const volatile unsigned int* hint = e->h_maxchain;
    // The serial kernel walks one chain per wave on 2048 blocks, about 0.6 us per link: its time grows with
    // (rounds of 2048 barriers) x (longest chain).  The parallel kernels cost about 80 us whatever the chains
    // look like (measured: 100 barriers x 250 events and 500 x 200 break even, 8192 x 128 and anything longer
    // than the serial kernel's 256-entry chain copy go parallel, 2000 x 50 and 8000 x 12 stay serial).
    const unsigned int rounds = hint[0] / 2048 + 1;
    w.hlen = e->hlen_mode ? 1u : 0u;
    if (w.hlen) serial_only = true;      // HLEN after every single event: only the replays that go event by event have it
    w.cx_on = serial_only ? 0u : e->cx_force >= 0 ? (uint32_t)(e->cx_force & 1) : ((uint64_t)rounds * hint[1] > 256 ? 1u : 0u);
    // Fused form: the link pass copies every accepted event's message into its row itself (one pass less over the
    // events, and the copy's streaming overlaps the link's atomics); the resolve passes copy a position again where
    // several events of the batch wrote it.  Not with the parallel long-chain kernels (a barrier with thousands of
    // events per batch would have every one of them write its few rows) and not with messages in blocks of their own.
    w.fused = (!w.cx_on && e->t.max_message <= e->t.stride && e->fuse_off == 0) ? 1u : 0u;
    const uint32_t lanes = e->cfg.payload_stride / 16;
    // h_maxchain[8] = barriers of the previous batch with 9..16 events: while there are thousands of them the
    // register replay takes chains up to 16 (slower for everyone else, far cheaper than a wave per barrier)
    const bool wide = e->cx_force >= 0 ? (e->cx_force & 2) != 0 : hint[8] >= WIDE_ON;
    // Bucketed form of the link and resolve passes (no atomic per event): for batches that are large enough to give
    // every compute unit a tile of the first pass (measured, link + resolve pass in us, chain form -> bucketed: 36.8 ->
    // 39.8 at 131072 events, 45.1 -> 41.9 at 196608, 51.2 -> 48.1 at 262144, 64.6 -> 62.1 at 393216, 135 -> 119 at 2^20;
    // buckets of 512 to 1024 events on average up to 2^20, up to 1280 so that a rank's share of a routed batch of 2^20
    // events per rank, which comes a little above or below 2^20, keeps plain buckets), on tables of any size (2^25 slots,
    // batches of 2^20, link + resolve: 132 us chained, 121 us bucketed), and whose previous batch showed no long chains
    // (a hot barrier overflows its bucket, and the batch then falls back to chains).
    const uint64_t slots = e->t.mask + 1;
    // Batches beyond 1024 buckets x 1280 events: the buckets get fat (bfat x 2048 keys) and the second pass works through
    // a bucket's slot range in bfat sub-ranges; one launch pair whatever the batch size, up to what the key buffer holds.
    bool bucket = !serial_only && w.fused && !wide && n <= (uint64_t)BK_MAX_BUCKETS * BK_MEAN_MAX * e->bfat_max && e->bucket_mode != 0 &&
                  (e->bucket_mode == 1 || (n >= 196608 && hint[0] < 2048 && hint[1] < 512));
    if (bucket) {
        uint32_t B = 16;
        while (B < BK_MAX_BUCKETS && (uint64_t)B * 512 < n) B <<= 1;
        while (B > 1 && (uint64_t)B * 4 > slots) B >>= 1;
        uint32_t shift = 0;
        while ((slots >> shift) > B) ++shift;
        uint32_t fat = 1;
        while ((uint64_t)B * BK_MEAN_MAX * fat < n && fat < e->bfat_max && (shift >> 1) && (1u << shift) / (fat * 2) >= 4) fat <<= 1;
        uint32_t fat_log = 0;
        while ((1u << fat_log) < fat) ++fat_log;
        w.bbuckets = B;
        w.bshift = shift;
        w.bfat = fat;
        w.bcap = fat * RB_CAP;
        w.bsubshift = shift - fat_log;
        w.bmwbits = shift >= 29 ? 0u : (29 - shift > 14 ? 14u : 29 - shift);
        bucket = (uint64_t)B * BK_MEAN_MAX * fat >= n;        // (a tiny table cannot spread a large batch)
    }
    // equal tiles of the first pass: one per compute unit (two above 4096 events per unit; a unit holds two blocks),
    // none below 1024 events; 2^20 events on 256 units: 256 tiles of 4096.  A batch of 257 tiles of 4096 took
    // 108 us where 256 took 82.
    uint32_t tiles = 0, tile_events = 0, ept = 0;
    if (bucket) {
        const uint32_t most = e->n_cus * (n > (uint64_t)e->n_cus * BK_TILE ? 2u : 1u);
        tiles = grid_for(n, BK_BLOCK);
        if (tiles > most) tiles = most;
        if ((uint64_t)tiles * BK_TILE < n) {
            // more than two tiles of 4096 events per unit: whole rounds of (two tiles per unit), the tiles a little under 4096
            const uint32_t round = 2 * e->n_cus;
            tiles = (grid_for(n, BK_TILE) + round - 1) / round * round;
        }
        tile_events = (uint32_t)(((n + tiles - 1) / tiles + 63) & ~63ull);
        tiles = grid_for(n, tile_events);
        ept = (tile_events + BK_BLOCK - 1) / BK_BLOCK;
        bucket = ept <= 4;                              // (a GPU with few compute units: the chain form)
    }
    if (bucket) {
#define DJ_LINKB1(G, E) join_link_bucket_kernel<G, E><<<tiles, BK_BLOCK, 0, e->stream>>>(e->t, w, d_recs, n, tile_events, pitch, d_status, d_payloads)
#define DJ_LINKB(G) do { if (ept == 4) DJ_LINKB1(G, 4); else if (ept == 3) DJ_LINKB1(G, 3); else if (ept == 2) DJ_LINKB1(G, 2); else DJ_LINKB1(G, 1); } while (0)
        if (lanes <= 4) LAUNCH(e, DJ_PROF_JOIN_LINK, DJ_LINKB(4));
        else if (lanes <= 8) LAUNCH(e, DJ_PROF_JOIN_LINK, DJ_LINKB(8));
        else if (lanes <= 16) LAUNCH(e, DJ_PROF_JOIN_LINK, DJ_LINKB(16));
        else if (lanes <= 32) LAUNCH(e, DJ_PROF_JOIN_LINK, DJ_LINKB(32));
        else LAUNCH(e, DJ_PROF_JOIN_LINK, DJ_LINKB(64));
#undef DJ_LINKB1
#undef DJ_LINKB
        if (w.bfat > 1) {
            LAUNCH(e, DJ_PROF_JOIN_RESOLVE, (join_resolve_bucket_kernel<true><<<w.bbuckets * w.bfat, RB_T, 0, e->stream>>>(e->t, w, d_recs, n, now_ms, d_status)));
            HIP_TRY(e, hipMemsetAsync(w.bcursor, 0, (size_t)w.bbuckets * 4, e->stream));
        }
        else LAUNCH(e, DJ_PROF_JOIN_RESOLVE, (join_resolve_bucket_kernel<false><<<w.bbuckets, RB_T, 0, e->stream>>>(e->t, w, d_recs, n, now_ms, d_status)));
    } else {
#define DJ_LINK(HOT, G, BLOCK) join_link_kernel<HOT, G><<<grid_for(n, BLOCK), BLOCK, 0, e->stream>>>(e->t, w, d_recs, n, pitch, d_status, d_payloads)
    if (w.cx_on) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(true, 0, LINK_BLOCK_HOT)));
    else if (!w.fused) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 0, LINK_BLOCK)));
    else if (lanes <= 4) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 4, LINK_BLOCK)));
    else if (lanes <= 8) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 8, LINK_BLOCK)));
    else if (lanes <= 16) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 16, LINK_BLOCK)));
    else if (lanes <= 32) LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 32, LINK_BLOCK)));
    else LAUNCH(e, DJ_PROF_JOIN_LINK, (DJ_LINK(false, 64, LINK_BLOCK)));
#undef DJ_LINK
    if (wide)
        LAUNCH(e, DJ_PROF_JOIN_RESOLVE, (join_resolve_kernel<CHAIN_REG_WIDE><<<grid_for(n, RESOLVE_BLOCK), RESOLVE_BLOCK, 0, e->stream>>>(e->t, w, (uint32_t)n, now_ms, d_status)));
    else
        LAUNCH(e, DJ_PROF_JOIN_RESOLVE, (join_resolve_kernel<CHAIN_REG><<<grid_for(n, RESOLVE_BLOCK), RESOLVE_BLOCK, 0, e->stream>>>(e->t, w, (uint32_t)n, now_ms, d_status)));
    }
    // Long chains (hot barriers): the previous batch's longest chain, read back without waiting, picks
    // between the serial kernel and the parallel cx_* kernels.  Both give the same results.
    // One wave per long chain; the grid follows the previous batch's number of them (an empty pass of 4096
    // blocks costs 6 us, one of 256 blocks 2 us; a batch that suddenly has many more loops over them once).
    uint32_t cx_grid = 2 * hint[0];
    cx_grid = cx_grid < 256 ? 256 : cx_grid > 4096 ? 4096 : cx_grid;
    if (w.cx_on) {
        HIP_TRY(e, hipMemsetAsync(w.cx_keys, 0, CX_HASH * 4, e->stream));
        if (e->profiling) (void)hipEventRecord(e->ev_start[DJ_PROF_JOIN_COMPLEX], e->stream);
        cx_init_kernel<<<CX_CAP / CX_BPB, 256, 0, e->stream>>>(e->t, w);
        uint32_t sb = grid_for(n, 256);
        if (sb > 4096) sb = 4096;
        cx_scan_kernel<<<sb, 256, 0, e->stream>>>(w, n, now_ms);
        cx_finalize_kernel<<<1024, 64, 0, e->stream>>>(e->t, w, now_ms);
        join_resolve_complex_kernel<<<cx_grid, 64, 0, e->stream>>>(e->t, w, (uint32_t)n, now_ms, d_status, d_recs, d_payloads, pitch);
        if (e->profiling) { (void)hipEventRecord(e->ev_stop[DJ_PROF_JOIN_COMPLEX], e->stream); e->ev_used[DJ_PROF_JOIN_COMPLEX] = 1; }
    } else {
        LAUNCH(e, DJ_PROF_JOIN_COMPLEX, (join_resolve_complex_kernel<<<cx_grid, 64, 0, e->stream>>>(e->t, w, (uint32_t)n, now_ms, d_status, d_recs, d_payloads, pitch)));
    }
    if (w.fused) {
        HIP_TRY(e, hipGetLastError());
        return DJ_OK;
    }
    if (lanes <= 4) LAUNCH(e, DJ_PROF_JOIN_COPY, launch_join_copy<4>(e, w, n, d_recs, d_payloads, pitch, d_status));
    else if (lanes <= 8) LAUNCH(e, DJ_PROF_JOIN_COPY, launch_join_copy<8>(e, w, n, d_recs, d_payloads, pitch, d_status));
    else if (lanes <= 16) LAUNCH(e, DJ_PROF_JOIN_COPY, launch_join_copy<16>(e, w, n, d_recs, d_payloads, pitch, d_status));
    else if (lanes <= 32) LAUNCH(e, DJ_PROF_JOIN_COPY, launch_join_copy<32>(e, w, n, d_recs, d_payloads, pitch, d_status));
    else LAUNCH(e, DJ_PROF_JOIN_COPY, launch_join_copy<64>(e, w, n, d_recs, d_payloads, pitch, d_status));
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

extern "C" int dj_join_batch(dj_engine* e, uint32_t n, const dj_join_rec* d_recs, const uint8_t* d_payloads,
                             uint64_t pitch, uint64_t now_ms, int32_t* d_status) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    const int rc = join_check(e, "join", n, d_recs, d_payloads, pitch, now_ms, d_status, e->cfg.max_batch, "max_batch");
    if (rc != DJ_OK || n == 0) return rc;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // double-banked counters: this batch counts in one bank and zeroes the other for the next batch
    e->w.counters = e->d_counters + e->ctr_bank * C_COUNT;
    e->w.counters_next = e->d_counters + (e->ctr_bank ^ 1) * C_COUNT;
    e->ctr_bank ^= 1;
    e->w.publish_hints = 1;
    return join_launch(e, e->w, false, n, d_recs, d_payloads, pitch, now_ms, d_status);
}

extern "C" int dj_join_batch_static(dj_engine* e, uint32_t n, const dj_join_rec* d_recs, const uint8_t* d_payloads,
                                    uint64_t pitch, uint64_t now_ms, int32_t* d_status) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    const int rc = join_check(e, "join_static", n, d_recs, d_payloads, pitch, now_ms, d_status, DJ_STATIC_JOIN_MAX, "DJ_STATIC_JOIN_MAX");
    if (rc != DJ_OK || n == 0) return rc;
    if (!e->scratch_ptr) return fail(e, DJ_E_INVALID, "join_static: the scratch array was not resolved on device %d", e->cfg.device);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // the array keeps the previous call's contents: the counters must start at zero (the other arrays are
    // written before they are read, element by element, by the link pass)
    unsigned char* p = e->scratch_ptr;
    HIP_TRY(e, hipMemsetAsync(p + SCR_COUNTERS, 0, 256, e->stream));
    Work w = e->w;            // hint, sweep banks and fork scratch stay the engine's; the join arrays do not
    w.node = reinterpret_cast<uint64_t*>(p + SCR_NODE);
    w.dst = reinterpret_cast<uint64_t*>(p + SCR_DST);
    w.ev_slot = reinterpret_cast<uint32_t*>(p + SCR_EV_SLOT);
    w.complex_ = reinterpret_cast<uint32_t*>(p + SCR_COMPLEX);
    w.counters = reinterpret_cast<unsigned int*>(p + SCR_COUNTERS);
    w.counters_next = w.counters + C_COUNT;
    w.publish_hints = 0;        // the second bank is a dummy here: the engine's own batches keep steering the kernel choice
    return join_launch(e, w, true, n, d_recs, d_payloads, pitch, now_ms, d_status);
}

static uint32_t pop_blocks(const dj_engine* e) {
    uint64_t b = (e->t.mask + 1 + 256 * POP_ITEMS - 1) / (256 * POP_ITEMS);
    if (b > 2048) b = 2048;
    return (uint32_t)b;
}

static void dispatch_gather(dj_engine* e, const dj_cont* conts, int from_sweep, uint32_t n_fixed, uint32_t cap_conts,
                            uint64_t cap_pos, int update_ts, uint32_t* lens, uint32_t* mws, uint8_t* payloads,
                            dj_sweep_result* res) {
    // a listed continuation (snapshot) gets lanes by the largest continuation the engine can hold; after a
    // sweep the kernel decides from what was popped
    const uint64_t items = (uint64_t)e->cfg.max_count * (e->cfg.payload_stride / 16);
    const int gs_fixed = items <= 16 ? 8 : items < 96 ? 16 : 64;
    const uint64_t n = from_sweep ? cap_conts : n_fixed;
    uint64_t blocks = (n * 64 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    gather_kernel<<<(uint32_t)blocks, 256, 0, e->stream>>>(e->t, e->w, conts, from_sweep, n_fixed, cap_conts, cap_pos,
                                                            update_ts, lens, mws, payloads, res, gs_fixed);
}

static int sweep_impl(dj_engine* e, const char* what, uint64_t now_ms, uint32_t cap_conts, uint64_t cap_positions,
                      int gather, dj_cont* d_conts, uint32_t* d_lens, uint32_t* d_max_waits, uint8_t* d_payloads,
                      dj_sweep_result* d_result, dj_sweep_result* h_result) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(now_ms)) return fail(e, DJ_E_CLOCK_RANGE, "%s: now_ms=%llu outside [1e12, 9e12)", what, (unsigned long long)now_ms);
    if (!d_conts || cap_conts == 0) return fail(e, DJ_E_INVALID, "%s: no continuation buffer", what);
    if (cap_conts > e->t.mask + 1) cap_conts = (uint32_t)(e->t.mask + 1);
    if (gather) {
        if (cap_positions < e->cfg.max_count) return fail(e, DJ_E_INVALID, "%s: cap_positions must be >= max_count", what);
        if (cap_positions > (1ull << 32)) cap_positions = 1ull << 32;
        if (d_payloads && !aligned16(d_payloads)) return fail(e, DJ_E_INVALID, "%s: payload buffer must be 16-byte aligned", what);
    }
    dj_sweep_result* res = d_result ? d_result : e->d_result;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // double-banked cursors: this sweep counts in one bank, its last kernel zeroes the other
    e->w.sweep = e->d_sweep + e->sweep_bank * S_COUNT;
    e->w.sweep_next = e->d_sweep + (e->sweep_bank ^ 1) * S_COUNT;
    e->sweep_bank ^= 1;
    // the scan bound: this sweep reads the bank every queue write since the last sweep went to and leaves the
    // other one for the writers that follow
    e->w.due_cur = e->t.due_min;
    e->w.due_next = e->t.due_min == e->d_due ? e->d_due + 1 : e->d_due;
    e->t.due_min = e->w.due_next;
    if (e->profiling) (void)hipEventRecord(e->ev_start[DJ_PROF_POP], e->stream);
    // with a ready list: the drain, then the two passes over the table's regions, which find on the device whether
    // they have anything to do (a date that is not on the list has come) and then read only the regions with such a date
    if (e->t.ready_on) {
        // (modest grids: a kernel that has nothing to do must get out of the way quickly)
        pop_ready_kernel<<<READY_SHARDS * 4, 256, 0, e->stream>>>(e->t, e->w, now_ms, cap_conts, cap_positions, gather, d_conts);
        const uint64_t nreg = (e->t.mask + 1) >> REGION_SHIFT;
        const uint64_t sel = (nreg + SELECT_BLOCK - 1) / SELECT_BLOCK;
        select_regions_kernel<<<(uint32_t)(sel > 512 ? 512 : sel), SELECT_BLOCK, 0, e->stream>>>(e->t, e->w, now_ms);
        pop_regions_kernel<<<pop_blocks(e) > 2048 ? 2048 : pop_blocks(e), 256, 0, e->stream>>>(e->t, e->w, now_ms, cap_conts, cap_positions, gather, d_conts);
    } else {
        pop_due_kernel<<<pop_blocks(e), 256, 0, e->stream>>>(e->t, e->w, now_ms, cap_conts, cap_positions, gather, d_conts);
    }
    if (e->profiling) { (void)hipEventRecord(e->ev_stop[DJ_PROF_POP], e->stream); e->ev_used[DJ_PROF_POP] = 1; }
    if (gather)
        LAUNCH(e, DJ_PROF_GATHER, dispatch_gather(e, d_conts, 1, 0, cap_conts, cap_positions, 1, d_lens, d_max_waits, d_payloads, res));
    else
        finish_result_kernel<<<1, 64, 0, e->stream>>>(e->t, e->w, res, cap_conts, 0);
    HIP_TRY(e, hipGetLastError());
    if (h_result) {
        HIP_TRY(e, hipMemcpyAsync(h_result, res, sizeof(dj_sweep_result), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    return DJ_OK;
}

extern "C" int dj_pop_due(dj_engine* e, uint64_t now_ms, uint32_t cap_conts, dj_cont* d_conts,
                          dj_sweep_result* d_result, dj_sweep_result* h_result) {
    return sweep_impl(e, "pop_due", now_ms, cap_conts, 0, 0, d_conts, nullptr, nullptr, nullptr, d_result, h_result);
}

extern "C" int dj_sweep(dj_engine* e, uint64_t now_ms, uint32_t cap_conts, uint64_t cap_positions, dj_cont* d_conts,
                        uint32_t* d_lens, uint32_t* d_max_waits, uint8_t* d_payloads, dj_sweep_result* d_result,
                        dj_sweep_result* h_result) {
    return sweep_impl(e, "sweep", now_ms, cap_conts, cap_positions, 1, d_conts, d_lens, d_max_waits, d_payloads,
                      d_result, h_result);
}

extern "C" int dj_fetch_long(dj_engine* e, uint32_t n, const dj_long_ref* d_refs, const uint64_t* d_offsets, uint8_t* d_out) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (n == 0) return DJ_OK;
    if (!d_refs || !d_offsets || !d_out || !aligned16(d_refs) || !aligned16(d_out))
        return fail(e, DJ_E_INVALID, "fetch_long: null or misaligned pointer");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    uint64_t blocks = ((uint64_t)n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    fetch_long_kernel<<<(uint32_t)blocks, 256, 0, e->stream>>>(e->t, n, d_refs, d_offsets, d_out);
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

extern "C" int dj_ack_batch(dj_engine* e, uint32_t n, const uint32_t* d_n, const dj_cont* d_conts, const uint8_t* d_ok,
                            uint64_t now_ms, int32_t* d_status) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(now_ms)) return fail(e, DJ_E_CLOCK_RANGE, "ack: now_ms=%llu outside [1e12, 9e12)", (unsigned long long)now_ms);
    if (n == 0) return DJ_OK;
    if (!d_conts || !d_status) return fail(e, DJ_E_INVALID, "ack: null pointer");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    LAUNCH(e, DJ_PROF_ACK, (ack_kernel<false><<<grid_for(n, ACK_BLOCK), ACK_BLOCK, 0, e->stream>>>(e->t, n, d_n, d_conts, d_ok, now_ms, d_status)));
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

static uint32_t recover_blocks(const dj_engine* e) {
    uint64_t b = (e->t.mask + 1) / RECOVER_BLOCK;      // slots are a power of two >= 1024
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (uint32_t)b;
}

extern "C" int dj_recover(dj_engine* e, uint64_t now_ms, uint32_t cap, dj_cont* d_conts, int32_t* d_status,
                          dj_sweep_result* d_result, dj_sweep_result* h_result) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(now_ms)) return fail(e, DJ_E_CLOCK_RANGE, "recover: now_ms=%llu outside [1e12, 9e12)", (unsigned long long)now_ms);
    if (!d_conts || !d_status || cap == 0) return fail(e, DJ_E_INVALID, "recover: no output buffer");
    dj_sweep_result* res = d_result ? d_result : e->d_result;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemsetAsync(res, 0, sizeof(dj_sweep_result), e->stream));
    LAUNCH(e, DJ_PROF_RECOVER, (recover_kernel<<<recover_blocks(e), RECOVER_BLOCK, 0, e->stream>>>(e->t, now_ms, cap, d_conts, d_status, res)));
    clamp_count_kernel<<<1, 64, 0, e->stream>>>(res, cap);
    HIP_TRY(e, hipGetLastError());
    if (h_result) {
        HIP_TRY(e, hipMemcpyAsync(h_result, res, sizeof(dj_sweep_result), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    return DJ_OK;
}

// ---- single-key functions ---------------------------------------------------------------
extern "C" int dj_snapshot_one(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, int update_ts, dj_cont* h_meta,
                               uint32_t* h_lens, uint32_t* h_max_waits, uint8_t* h_payloads) {
    if (!e || !h_meta) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    lookup_one_kernel<<<1, 64, 0, e->stream>>>(e->t, e->w, fp_lo, fp_hi, e->d_one_cont, e->d_one_scores);
    gather_kernel<<<1, 64, 0, e->stream>>>(e->t, e->w, e->d_one_cont, 0, 1, 1, e->cfg.max_count, update_ts ? 1 : 0,
                                           e->d_one_lens, e->d_one_mws, h_payloads ? e->d_one_payload : nullptr, nullptr, 64);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(h_meta, e->d_one_cont, sizeof(dj_cont), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint32_t count = h_meta->count;
    if (count) {
        if (h_lens) HIP_TRY(e, hipMemcpyAsync(h_lens, e->d_one_lens, (size_t)count * 4, hipMemcpyDeviceToHost, e->stream));
        if (h_max_waits) HIP_TRY(e, hipMemcpyAsync(h_max_waits, e->d_one_mws, (size_t)count * 4, hipMemcpyDeviceToHost, e->stream));
        if (h_payloads)
            HIP_TRY(e, hipMemcpyAsync(h_payloads, e->d_one_payload, (size_t)count * e->cfg.payload_stride, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    return DJ_OK;
}

extern "C" int dj_store_in_publish_queue(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, uint64_t at_ms) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(at_ms)) return fail(e, DJ_E_CLOCK_RANGE, "store_in_publish_queue: at_ms outside [1e12, 9e12)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    store_in_queue_one_kernel<<<1, 64, 0, e->stream>>>(e->t, fp_lo, fp_hi, at_ms, e->d_one_err);
    HIP_TRY(e, hipGetLastError());
    int32_t err = 0;
    HIP_TRY(e, hipMemcpyAsync(&err, e->d_one_err, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (err) return fail(e, DJ_E_NOMEM, "store_in_publish_queue: table full");
    return DJ_OK;
}

extern "C" int dj_queue_scores(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, uint64_t* h_primary_at, uint64_t* h_backup_at) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    lookup_one_kernel<<<1, 64, 0, e->stream>>>(e->t, e->w, fp_lo, fp_hi, e->d_one_cont, e->d_one_scores);
    HIP_TRY(e, hipGetLastError());
    uint64_t sc[2];
    HIP_TRY(e, hipMemcpyAsync(sc, e->d_one_scores, 16, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (h_primary_at) *h_primary_at = sc[0];
    if (h_backup_at) *h_backup_at = sc[1];
    return DJ_OK;
}

// One row through the command form of the ack pass (redis/complete or redis/update-metadata alone).
static int command_one(dj_engine* e, uint8_t op, const dj_cont& row, int32_t* h_status) {
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // staging: the row in d_one_cont, the command byte in the first of the two score words, the answer in the second
    uint64_t words[2] = {op, 0};
    HIP_TRY(e, hipMemcpyAsync(e->d_one_cont, &row, sizeof row, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_one_scores, words, sizeof words, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));       // row and words live on this thread's stack
    ack_kernel<true><<<1, ACK_BLOCK, 0, e->stream>>>(e->t, 1, nullptr, e->d_one_cont, reinterpret_cast<const uint8_t*>(e->d_one_scores),
                                                     0, reinterpret_cast<int32_t*>(e->d_one_scores + 1));
    HIP_TRY(e, hipGetLastError());
    int32_t st = 0;
    HIP_TRY(e, hipMemcpyAsync(&st, e->d_one_scores + 1, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *h_status = st;
    return DJ_OK;
}

extern "C" int dj_complete_one(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, int32_t* h_status) {
    if (!e || !h_status) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    dj_cont row = {};
    row.fp_lo = fp_lo;
    row.fp_hi = fp_hi;
    return command_one(e, ACK_OP_COMPLETE, row, h_status);
}

extern "C" int dj_update_metadata_one(dj_engine* e, const dj_cont* h_meta, int32_t* h_status) {
    if (!e || !h_meta || !h_status) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    dj_cont row = {};
    row.fp_lo = h_meta->fp_lo;
    row.fp_hi = h_meta->fp_hi;
    row.count = h_meta->count;
    row.deadline_ms = h_meta->deadline_ms;
    row.retry_publish = h_meta->retry_publish;
    row.retry_timeout = h_meta->retry_timeout;
    return command_one(e, ACK_OP_SET_METADATA, row, h_status);
}

extern "C" int dj_store_nearest_in_publish_queue(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, uint64_t at_ms, int32_t* h_result) {
    if (!e || !h_result) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(at_ms)) return fail(e, DJ_E_CLOCK_RANGE, "store_nearest_in_publish_queue: at_ms outside [1e12, 9e12)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    store_nearest_one_kernel<<<1, 64, 0, e->stream>>>(e->t, fp_lo, fp_hi, at_ms, e->d_one_err);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(h_result, e->d_one_err, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return DJ_OK;
}

extern "C" int dj_unprocessed(dj_engine* e, uint64_t now_ms, uint32_t cap, dj_cont* d_conts, dj_sweep_result* d_result,
                              dj_sweep_result* h_result) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (!clock_ok(now_ms)) return fail(e, DJ_E_CLOCK_RANGE, "unprocessed: now_ms=%llu outside [1e12, 9e12)", (unsigned long long)now_ms);
    if (!d_conts || cap == 0) return fail(e, DJ_E_INVALID, "unprocessed: no output buffer");
    dj_sweep_result* res = d_result ? d_result : e->d_result;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemsetAsync(res, 0, sizeof(dj_sweep_result), e->stream));
    list_unprocessed_kernel<<<scan_blocks(e), 256, 0, e->stream>>>(e->t, now_ms - e->cfg.check_publish_delay_ms, cap, d_conts, res);
    clamp_count_kernel<<<1, 64, 0, e->stream>>>(res, cap);
    HIP_TRY(e, hipGetLastError());
    if (h_result) {
        HIP_TRY(e, hipMemcpyAsync(h_result, res, sizeof(dj_sweep_result), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    return DJ_OK;
}

extern "C" int dj_get_stats(dj_engine* e, dj_stats* h_stats) {
    if (!e || !h_stats) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    unsigned long long st[4], bump;
    unsigned int head[NUM_CLASSES], tail[NUM_CLASSES];
    HIP_TRY(e, hipMemsetAsync(e->t.stat, 0, 4 * 8, e->stream));
    stats_kernel<<<scan_blocks(e), 256, 0, e->stream>>>(e->t);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(st, e->t.stat, sizeof st, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(&bump, e->t.bump, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(head, e->t.ring_head, sizeof head, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(tail, e->t.ring_tail, sizeof tail, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    h_stats->slots = e->t.mask + 1;
    h_stats->live = st[0];
    h_stats->tombstones = st[1];
    h_stats->arena_units_used = bump < e->t.arena_units ? bump : e->t.arena_units;   // refused slab requests leave it past the end
    h_stats->arena_units_free = 0;
    for (uint32_t k = 0; k < NUM_CLASSES; ++k)
        h_stats->arena_units_free += (uint64_t)(tail[k] - head[k]) * slab_units(k, e->t.stride);
    unsigned int ohead[OVF_CLASSES], otail[OVF_CLASSES];
    HIP_TRY(e, hipMemcpy(ohead, e->t.oring_head, sizeof ohead, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(otail, e->t.oring_tail, sizeof otail, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < OVF_CLASSES; ++k) h_stats->arena_units_free += (uint64_t)(otail[k] - ohead[k]) << k;
    return DJ_OK;
}

// ---- durability -----------------------------------------------------------------------------
namespace {
struct CkptHeader {
    char magic[8];               // "DJCKPT5\0"
    uint64_t slots, arena_units_used, arena_units_total;
    uint32_t stride, max_count;
    uint32_t ring_n[NUM_CLASSES];    // free slabs per class, stored head-first after the arena
    uint32_t oring_n[OVF_CLASSES];   // free long-message blocks per class, after the slab lists
    uint32_t max_message;
    uint64_t used_upper;
    uint64_t payload_bytes;          // everything behind the header
    uint64_t checksum;               // of those bytes (fnv-1a over 64-bit words, tail bytes zero-extended)
};
const size_t CKPT_CHUNK = 64u << 20;

struct Fnv {       // over the byte stream, whatever pieces it arrives in
    uint64_t h = 0xCBF29CE484222325ull, bytes = 0;
    uint8_t pend[8];
    size_t npend = 0;
    void word(const uint8_t* b) { uint64_t w; memcpy(&w, b, 8); h = (h ^ w) * 0x100000001B3ull; }
    void add(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        bytes += n;
        while (npend && n) { pend[npend++] = *b++; --n; if (npend == 8) { word(pend); npend = 0; } }
        size_t i = 0;
        for (; i + 8 <= n; i += 8) word(b + i);
        for (; i < n; ++i) pend[npend++] = b[i];
    }
    uint64_t value() {
        uint8_t last[8] = {0};
        memcpy(last, pend, npend);
        uint64_t v = h;
        if (npend) { uint64_t w; memcpy(&w, last, 8); v = (v ^ w) * 0x100000001B3ull; }
        return v ^ bytes;
    }
};

int dev_to_file(dj_engine* e, FILE* f, const void* dev, size_t bytes, void* stage, Fnv* sum) {
    const uint8_t* p = static_cast<const uint8_t*>(dev);
    for (size_t off = 0; off < bytes; off += CKPT_CHUNK) {
        const size_t m = bytes - off < CKPT_CHUNK ? bytes - off : CKPT_CHUNK;      // chunks are multiples of 8 bytes but the last
        HIP_TRY(e, hipMemcpy(stage, p + off, m, hipMemcpyDeviceToHost));
        sum->add(stage, m);
        if (fwrite(stage, 1, m, f) != m) return fail(e, DJ_E_INVALID, "checkpoint: short write");
    }
    return DJ_OK;
}

int file_to_dev(dj_engine* e, FILE* f, void* dev, size_t bytes, void* stage) {
    uint8_t* p = static_cast<uint8_t*>(dev);
    for (size_t off = 0; off < bytes; off += CKPT_CHUNK) {
        const size_t m = bytes - off < CKPT_CHUNK ? bytes - off : CKPT_CHUNK;
        if (fread(stage, 1, m, f) != m) return fail(e, DJ_E_INVALID, "checkpoint: file is truncated");
        HIP_TRY(e, hipMemcpy(p + off, stage, m, hipMemcpyHostToDevice));
    }
    return DJ_OK;
}
}  // namespace

extern "C" int dj_checkpoint_save(dj_engine* e, const char* path) {
    if (!e || !path) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    int rc = compact_finish(e, true);
    if (rc != DJ_OK) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    CkptHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "DJCKPT5", 8);
    h.slots = e->t.mask + 1;
    h.arena_units_total = e->t.arena_units;
    h.stride = e->t.stride;
    h.max_count = e->t.max_count;
    h.max_message = e->t.max_message;
    h.used_upper = e->used_upper;
    unsigned long long bump;
    unsigned int head[NUM_CLASSES], tail[NUM_CLASSES];
    HIP_TRY(e, hipMemcpy(&bump, e->t.bump, 8, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(head, e->t.ring_head, sizeof head, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(tail, e->t.ring_tail, sizeof tail, hipMemcpyDeviceToHost));
    h.arena_units_used = bump < e->t.arena_units ? bump : e->t.arena_units;
    for (uint32_t k = 0; k < NUM_CLASSES; ++k) h.ring_n[k] = tail[k] - head[k];
    unsigned int ohead[OVF_CLASSES], otail[OVF_CLASSES];
    HIP_TRY(e, hipMemcpy(ohead, e->t.oring_head, sizeof ohead, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(otail, e->t.oring_tail, sizeof otail, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < OVF_CLASSES; ++k) h.oring_n[k] = otail[k] - ohead[k];
    // written beside the target and renamed over it once complete and on disk: a crash in the middle of a
    // periodic snapshot leaves the previous checkpoint as it was
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(e, DJ_E_INVALID, "checkpoint: cannot open %s for writing", tmp.c_str());
    void* stage = malloc(CKPT_CHUNK);
    if (!stage) { fclose(f); remove(tmp.c_str()); return fail(e, DJ_E_NOMEM, "checkpoint: no staging memory"); }
    Fnv sum;
    rc = fwrite(&h, sizeof h, 1, f) == 1 ? DJ_OK : fail(e, DJ_E_INVALID, "checkpoint: short write");
    if (rc == DJ_OK) rc = dev_to_file(e, f, e->t.slots, h.slots * sizeof(Slot), stage, &sum);
    if (rc == DJ_OK) rc = dev_to_file(e, f, e->t.q, h.slots * 8, stage, &sum);
    if (rc == DJ_OK) rc = dev_to_file(e, f, e->t.arena, h.arena_units_used * 64, stage, &sum);
    for (uint32_t k = 0; k < NUM_CLASSES && rc == DJ_OK; ++k) {
        // the ring's live window [head, tail) may wrap: two pieces, head-first
        const uint32_t cap = e->t.ring_cap[k], n = h.ring_n[k], at = head[k] & (cap - 1);
        const uint32_t first = n < cap - at ? n : cap - at;
        rc = dev_to_file(e, f, e->t.ring[k] + at, (size_t)first * 4, stage, &sum);
        if (rc == DJ_OK) rc = dev_to_file(e, f, e->t.ring[k], (size_t)(n - first) * 4, stage, &sum);
    }
    for (uint32_t k = 0; k < OVF_CLASSES && rc == DJ_OK; ++k) {
        const uint32_t cap = e->t.oring_cap[k], n = h.oring_n[k], at = ohead[k] & (cap - 1);
        const uint32_t first = n < cap - at ? n : cap - at;
        rc = dev_to_file(e, f, e->t.oring[k] + at, (size_t)first * 4, stage, &sum);
        if (rc == DJ_OK) rc = dev_to_file(e, f, e->t.oring[k], (size_t)(n - first) * 4, stage, &sum);
    }
    free(stage);
    if (rc == DJ_OK) {          // the header again, now with the size and checksum of what follows it
        h.payload_bytes = sum.bytes;
        h.checksum = sum.value();
        if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&h, sizeof h, 1, f) != 1) rc = fail(e, DJ_E_INVALID, "checkpoint: short write");
    }
    if (rc == DJ_OK && (fflush(f) != 0 || fsync(fileno(f)) != 0)) rc = fail(e, DJ_E_INVALID, "checkpoint: flush to disk failed");
    if (fclose(f) != 0 && rc == DJ_OK) rc = fail(e, DJ_E_INVALID, "checkpoint: close failed");
    if (rc == DJ_OK && rename(tmp.c_str(), path) != 0) rc = fail(e, DJ_E_INVALID, "checkpoint: cannot rename %s to %s", tmp.c_str(), path);
    if (rc != DJ_OK) remove(tmp.c_str());
    return rc;
}

extern "C" int dj_checkpoint_load(dj_engine* e, const char* path) {
    if (!e || !path) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    int rc = compact_finish(e, true);
    if (rc != DJ_OK) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(e, DJ_E_INVALID, "checkpoint: cannot open %s", path);
    CkptHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "DJCKPT5", 8) != 0) {
        fclose(f);
        return fail(e, DJ_E_INVALID, "checkpoint: %s is not a checkpoint file", path);
    }
    if (h.slots != e->t.mask + 1 || h.stride != e->t.stride || h.max_count != e->t.max_count ||
        h.max_message != e->t.max_message || h.arena_units_used > e->t.arena_units) {
        fclose(f);
        return fail(e, DJ_E_INVALID, "checkpoint: saved with slots=%llu stride=%u max_count=%u arena_used=%llu units, "
                    "engine has slots=%llu stride=%u max_count=%u arena=%llu units",
                    (unsigned long long)h.slots, h.stride, h.max_count, (unsigned long long)h.arena_units_used,
                    (unsigned long long)(e->t.mask + 1), e->t.stride, e->t.max_count, (unsigned long long)e->t.arena_units);
    }
    for (uint32_t k = 0; k < NUM_CLASSES; ++k)
        if (h.ring_n[k] > e->t.ring_cap[k]) {
            fclose(f);
            return fail(e, DJ_E_INVALID, "checkpoint: free list of class %u does not fit", k);
        }
    for (uint32_t k = 0; k < OVF_CLASSES; ++k)
        if (h.oring_n[k] > e->t.oring_cap[k]) {
            fclose(f);
            return fail(e, DJ_E_INVALID, "checkpoint: block free list of class %u does not fit", k);
        }
    void* stage = malloc(CKPT_CHUNK);
    if (!stage) { fclose(f); return fail(e, DJ_E_NOMEM, "checkpoint: no staging memory"); }
    {   // first pass, nothing on the device is touched: the file must be whole (size, checksum) and its free
        // lists must name units inside the saved arena
        uint64_t want = h.slots * (sizeof(Slot) + 8) + h.arena_units_used * 64;
        for (uint32_t k = 0; k < NUM_CLASSES; ++k) want += (uint64_t)h.ring_n[k] * 4;
        for (uint32_t k = 0; k < OVF_CLASSES; ++k) want += (uint64_t)h.oring_n[k] * 4;
        Fnv sum;
        const uint64_t lists_at = h.slots * (sizeof(Slot) + 8) + h.arena_units_used * 64;
        bool lists_ok = true;
        for (uint64_t off = 0; h.payload_bytes == want && off < want;) {
            const size_t m = want - off < CKPT_CHUNK ? (size_t)(want - off) : CKPT_CHUNK;
            if (fread(stage, 1, m, f) != m) break;
            sum.add(stage, m);
            // entries of the free lists (4 bytes each, 4-byte aligned in the stream: everything before them is a multiple of 8)
            const uint64_t from = off > lists_at ? off : lists_at;
            for (uint64_t p = from; p + 4 <= off + m; p += 4) {
                uint32_t unit;
                memcpy(&unit, static_cast<uint8_t*>(stage) + (p - off), 4);
                if (unit == 0 || unit >= h.arena_units_used) lists_ok = false;
            }
            off += m;
        }
        const bool whole = h.payload_bytes == want && sum.bytes == want && sum.value() == h.checksum && fgetc(f) == EOF;
        if (!whole || !lists_ok) {
            free(stage);
            fclose(f);
            return fail(e, DJ_E_INVALID, "checkpoint: %s is %s; the engine's state was left as it was", path,
                        whole ? "inconsistent (a free list names a unit outside the saved arena)" : "truncated or damaged (size or checksum mismatch)");
        }
        if (fseek(f, (long)sizeof h, SEEK_SET) != 0) { free(stage); fclose(f); return fail(e, DJ_E_INVALID, "checkpoint: seek failed"); }
    }
    rc = file_to_dev(e, f, e->t.slots, h.slots * sizeof(Slot), stage);
    if (rc == DJ_OK) rc = file_to_dev(e, f, e->t.q, h.slots * 8, stage);
    if (rc == DJ_OK) rc = file_to_dev(e, f, e->t.arena, h.arena_units_used * 64, stage);
    unsigned int head[NUM_CLASSES], tail[NUM_CLASSES];
    for (uint32_t k = 0; k < NUM_CLASSES && rc == DJ_OK; ++k) {
        rc = file_to_dev(e, f, e->t.ring[k], (size_t)h.ring_n[k] * 4, stage);
        head[k] = 0;
        tail[k] = h.ring_n[k];
    }
    unsigned int ohead[OVF_CLASSES], otail[OVF_CLASSES];
    for (uint32_t k = 0; k < OVF_CLASSES && rc == DJ_OK; ++k) {
        rc = file_to_dev(e, f, e->t.oring[k], (size_t)h.oring_n[k] * 4, stage);
        ohead[k] = 0;
        otail[k] = h.oring_n[k];
    }
    free(stage);
    fclose(f);
    if (rc != DJ_OK) return rc;      // the engine's state is undefined after a failed load of a valid header
    const unsigned long long bump = h.arena_units_used;
    HIP_TRY(e, hipMemcpy(e->t.bump, &bump, 8, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.ring_head, head, sizeof head, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.ring_tail, tail, sizeof tail, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.ring_tail_snap, tail, sizeof tail, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.oring_head, ohead, sizeof ohead, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.oring_tail, otail, sizeof otail, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->t.oring_tail_snap, otail, sizeof otail, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemset(e->d_counters, 0, 2 * C_COUNT * 4));
    HIP_TRY(e, hipMemset(e->d_sweep, 0, 2 * S_COUNT * 8));
    // no ready list and no scan bound come with the file: the first sweep scans and recomputes the bound
    HIP_TRY(e, hipMemset(e->t.ready_ovf, 1, 4));
    HIP_TRY(e, hipMemset(e->d_due, 0, 2 * 8));
    e->used_upper = h.used_upper;
    e->claimed_since_compact = 0;
    e->h_maxchain[0] = e->h_maxchain[1] = e->h_maxchain[8] = 0;
    return DJ_OK;
}

// ---- multi-GPU routing ------------------------------------------------------------------
extern "C" int dj_partition_joins(dj_engine* e, uint32_t n, uint32_t n_ranks, const dj_join_rec* d_recs,
                                  const uint8_t* d_payloads, uint64_t pitch, dj_join_rec* d_recs_out,
                                  uint8_t* d_payloads_out, uint32_t* d_origin, uint32_t* d_counts, void* stream) {
    return dj_partition_joins_self(e, n, n_ranks, 0xFFFFFFFFu, d_recs, d_payloads, pitch, d_recs_out, d_payloads_out, nullptr, nullptr,
                                   d_origin, d_counts, stream);
}

extern "C" int dj_partition_joins_self(dj_engine* e, uint32_t n, uint32_t n_ranks, uint32_t self_rank, const dj_join_rec* d_recs,
                                       const uint8_t* d_payloads, uint64_t pitch, dj_join_rec* d_recs_out,
                                       uint8_t* d_payloads_out, dj_join_rec* d_recs_self, uint8_t* d_payloads_self,
                                       uint32_t* d_origin, uint32_t* d_counts, void* stream) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (self_rank != 0xFFFFFFFFu) {
        if (self_rank >= n_ranks) return fail(e, DJ_E_INVALID, "partition: self_rank must be below n_ranks");
        if (!d_recs_self || !aligned16(d_recs_self) || (d_payloads && (!d_payloads_self || !aligned16(d_payloads_self))))
            return fail(e, DJ_E_INVALID, "partition: the buffers for this rank's own group are missing or misaligned");
    }
    if (n_ranks < 1 || n_ranks > PART_MAX_RANKS) return fail(e, DJ_E_INVALID, "partition: n_ranks must be 1..%d", PART_MAX_RANKS);
    if (!d_recs || !d_recs_out || !d_origin || !d_counts) return fail(e, DJ_E_INVALID, "partition: null pointer");
    if (!aligned16(d_recs) || !aligned16(d_recs_out)) return fail(e, DJ_E_INVALID, "partition: records must be 16-byte aligned");
    if (d_payloads && (!d_payloads_out || !aligned16(d_payloads) || !aligned16(d_payloads_out) || pitch % 16 || pitch == 0))
        return fail(e, DJ_E_INVALID, "partition: payload buffers must be 16-byte aligned, pitch a positive multiple of 16");
    if (n > e->cfg.max_batch) return fail(e, DJ_E_TOO_LARGE, "partition: n=%u exceeds max_batch=%u", n, e->cfg.max_batch);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    uint32_t* scratch = stream ? e->d_part_counts2 : e->d_part_counts;
    if (n == 0) {
        HIP_TRY(e, hipMemsetAsync(d_counts, 0, n_ranks * 4, st));
        return DJ_OK;
    }
    // at most 1024 segments and segments x ranks <= 32768: every scatter block adds the segment counts up itself
    uint32_t seg_max = 32768 / n_ranks;
    if (seg_max > 1024) seg_max = 1024;
    const uint32_t tiles = grid_for(n, PART_TILE);
    const uint32_t tiles_per_seg = (tiles + seg_max - 1) / seg_max;
    const uint32_t segs = (tiles + tiles_per_seg - 1) / tiles_per_seg;
    partition_count_kernel<<<segs, PART_BLOCK, 0, st>>>(d_recs, n, n_ranks, tiles_per_seg, scratch);
    const uint64_t lanes = d_payloads ? pitch / 16 : 1;
#define DJ_SCATTER(G) partition_scatter_kernel<G><<<segs, PART_BLOCK, 0, st>>>(d_recs, n, n_ranks, tiles_per_seg, scratch, d_counts, d_recs_out, d_origin, d_payloads, d_payloads_out, pitch, self_rank, d_recs_self, d_payloads_self)
    if (lanes <= 4) DJ_SCATTER(4);
    else if (lanes <= 8) DJ_SCATTER(8);
    else if (lanes <= 16) DJ_SCATTER(16);
    else if (lanes <= 32) DJ_SCATTER(32);
    else DJ_SCATTER(64);
#undef DJ_SCATTER
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

extern "C" int dj_scatter_statuses(dj_engine* e, uint32_t n, const uint32_t* d_origin, const int32_t* d_sorted, int32_t* d_out, void* stream) {
    if (!e) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    if (n == 0) return DJ_OK;
    if (!d_origin || !d_sorted || !d_out) return fail(e, DJ_E_INVALID, "scatter_statuses: null pointer");
    if (n > e->cfg.max_batch) return fail(e, DJ_E_TOO_LARGE, "scatter_statuses: n=%u exceeds max_batch=%u", n, e->cfg.max_batch);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    uint32_t blocks = grid_for(n, 256);
    if (blocks > 2048) blocks = 2048;
    scatter_statuses_kernel<<<blocks, 256, 0, stream ? (hipStream_t)stream : e->stream>>>(n, d_origin, d_sorted, d_out);
    HIP_TRY(e, hipGetLastError());
    return DJ_OK;
}

extern "C" int dj_measure_copy(dj_engine* e, uint64_t bytes, uint32_t iterations, double* h_gbps) {
    if (!e || !h_gbps || bytes < (1u << 20) || iterations < 1) return DJ_E_INVALID;
    std::lock_guard<std::recursive_mutex> guard(e->mu);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    bytes &= ~(uint64_t)15;
    void *src = nullptr, *dst = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    auto body = [&]() -> int {
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) return fail(e, DJ_E_NOMEM, "measure_copy: no room for two buffers of %llu bytes", (unsigned long long)bytes);
        HIP_TRY(e, hipMemsetAsync(src, 1, bytes, e->stream));
        HIP_TRY(e, hipEventCreate(&a));
        HIP_TRY(e, hipEventCreate(&b));
        if (bytes / 16 / 256 >= 0x7FFFFFFFull) return fail(e, DJ_E_INVALID, "measure_copy: %llu bytes are more than one launch covers", (unsigned long long)bytes);
        const uint32_t blocks = (uint32_t)((bytes / 16 + 255) / 256);
        stream_copy_kernel<<<blocks, 256, 0, e->stream>>>((const uint4*)src, (uint4*)dst, bytes / 16);      // warm-up
        HIP_TRY(e, hipEventRecord(a, e->stream));
        for (uint32_t k = 0; k < iterations; ++k) stream_copy_kernel<<<blocks, 256, 0, e->stream>>>((const uint4*)src, (uint4*)dst, bytes / 16);
        HIP_TRY(e, hipEventRecord(b, e->stream));
        HIP_TRY(e, hipGetLastError());
        HIP_TRY(e, hipEventSynchronize(b));
        float ms = 0;
        HIP_TRY(e, hipEventElapsedTime(&ms, a, b));
        *h_gbps = ms > 0 ? 2.0 * (double)bytes * iterations / (ms * 1e-3) / 1e9 : 0.0;      // bytes read + bytes written
        return DJ_OK;
    };
    const int rc = body();
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(src);
    (void)hipFree(dst);
    return rc;
}
