// fork_proto.hip — what one insert into the barrier table costs, by memory operation.
//
// A stand-alone measurement (not part of the engine): 2^20 distinct 128-bit keys are inserted into an empty
// open-addressing table of 2^21 64-byte slots, the way the fork pass claims slots, in several variants that
// differ in the atomic operations they issue per record.  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/fork_proto tools/fork_proto.hip && tools/_build/fork_proto
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Slot { uint64_t lo, hi; uint32_t count, meta, slab, chain; uint64_t deadline, ts_read, ts, bits; };
static_assert(sizeof(Slot) == 64, "slot is one line");

template <class T> __device__ inline T a_load(T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ inline void a_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ inline bool a_cas(T* p, T expect, T v) {
    return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline uint64_t probe_at(uint64_t h, uint64_t p, uint64_t mask) { return p == 0 ? h : (p == 1 ? (h ^ 1) : (((h & ~1ull) + p) & mask)); }

// V bit 0: no atomic max on the slot after the claim; bit 1: the first pass over the probe sequence reads with
// plain loads; bit 2: the claimer's index goes into the slot with an atomic store instead of the atomic max;
// bit 3: high half written by a plain store
template <int V>
__global__ void __launch_bounds__(256) claim(Slot* slots, uint64_t mask, const ulonglong2* keys, uint32_t n, uint32_t* slot_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ulonglong2 k = keys[i];
    int64_t found = -1;
    for (int attempt = 0; attempt < 64 && found < 0; ++attempt) {
        const uint64_t h = k.x & mask;
        bool restart = false;
        for (uint64_t p = 0; p <= mask && !restart; ++p) {
            Slot* sl = &slots[probe_at(h, p, mask)];
            const uint64_t klo = ((V & 2) && attempt == 0) ? sl->lo : a_load(&sl->lo);
            if (klo == k.x) {
                uint64_t khi = a_load(&sl->hi);
                for (int spin = 0; khi == 0 && spin < (1 << 20); ++spin) { __builtin_amdgcn_s_sleep(1); khi = a_load(&sl->hi); }
                if (khi == k.y) { found = sl - slots; break; }
                continue;
            }
            if (klo != 0) continue;
            if (a_cas(&sl->lo, (uint64_t)0, (uint64_t)k.x)) {
                if (V & 8) sl->hi = k.y; else a_store(&sl->hi, (uint64_t)k.y);
                found = sl - slots;
                break;
            }
            restart = true;
        }
        if (!restart && found < 0) break;
    }
    slot_of[i] = (uint32_t)found;
    if (found < 0) return;
    if (V & 4) a_store(&slots[found].chain, i + 1);
    else if (!(V & 1)) __hip_atomic_fetch_max(&slots[found].chain, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the apply half in its barest form: read the slot's winner word, write the rest of the line and a queue word
template <int V>
__global__ void __launch_bounds__(256) apply(Slot* slots, uint64_t* q, const ulonglong2* keys, uint32_t n, const uint32_t* slot_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    if (s == 0xFFFFFFFFu) return;
    Slot* sl = &slots[s];
    if (!(V & 1) && sl->chain != i + 1) return;
    uint4* w = reinterpret_cast<uint4*>(sl);
    w[1] = make_uint4(8, 1, i + 1, 0);
    w[2] = make_uint4(i, 0, 0, 0);
    w[3] = make_uint4(0, 0, 0, 0);
    q[s] = i;
}

static void* g_evict = nullptr;            // 1 GiB written between the table's memset and the claim pass (cold table)
static bool g_cold = false;

template <int V>
static void run(const char* what, Slot* slots, uint64_t nslots, uint64_t* q, const ulonglong2* keys, uint32_t n, uint32_t* slot_of) {
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    float best_c = 1e9f, best_a = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(slots, 0, nslots * sizeof(Slot)));
        if (g_cold) CHECK(hipMemset(g_evict, 1, 1ull << 30));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        claim<V><<<(n + 255) / 256, 256>>>(slots, nslots - 1, keys, n, slot_of);
        CHECK(hipEventRecord(e1));
        apply<V><<<(n + 255) / 256, 256>>>(slots, q, keys, n, slot_of);
        CHECK(hipEventRecord(e2));
        CHECK(hipDeviceSynchronize());
        float c, a;
        CHECK(hipEventElapsedTime(&c, e0, e1)); CHECK(hipEventElapsedTime(&a, e1, e2));
        if (c < best_c) best_c = c;
        if (a < best_a) best_a = a;
    }
    std::vector<uint32_t> so(n);
    CHECK(hipMemcpy(so.data(), slot_of, n * 4ull, hipMemcpyDeviceToHost));
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += so[i] == 0xFFFFFFFFu;
    printf("%-58s claim %7.1f us   apply %6.1f us   (unplaced %u)\n", what, best_c * 1e3, best_a * 1e3, bad);
}

int main() {
    const uint32_t n = 1u << 20;
    const uint64_t nslots = 1ull << 21;
    std::vector<ulonglong2> h(n);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (uint32_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i].x = x | 1;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i].y = x | 1;
    }
    Slot* slots; uint64_t* q; ulonglong2* keys; uint32_t* slot_of;
    CHECK(hipMalloc(&slots, nslots * sizeof(Slot)));
    CHECK(hipMalloc(&q, nslots * 8));
    CHECK(hipMalloc(&keys, n * sizeof(ulonglong2)));
    CHECK(hipMalloc(&slot_of, n * 4ull));
    CHECK(hipMemcpy(keys, h.data(), n * sizeof(ulonglong2), hipMemcpyHostToDevice));
    printf("2^20 inserts into 2^21 empty slots of 64 bytes\n");
    run<0>("atomic loads, CAS, atomic store of the high half, atomic max", slots, nslots, q, keys, n, slot_of);
    run<1>("... without the atomic max", slots, nslots, q, keys, n, slot_of);
    run<2>("... first pass with plain loads", slots, nslots, q, keys, n, slot_of);
    run<3>("... plain loads and no atomic max", slots, nslots, q, keys, n, slot_of);
    run<4>("... atomic store in place of the atomic max", slots, nslots, q, keys, n, slot_of);
    run<6>("... plain loads, atomic store in place of the max", slots, nslots, q, keys, n, slot_of);
    run<11>("... plain loads, no max, high half by a plain store", slots, nslots, q, keys, n, slot_of);
    run<0>("(first variant again)", slots, nslots, q, keys, n, slot_of);
    // the table was zeroed just before every run above, so its lines sat in the 256 MB Infinity Cache; the engine's
    // table was zeroed a job earlier
    CHECK(hipMalloc(&g_evict, 1ull << 30));
    g_cold = true;
    run<0>("first variant, 1 GiB written between memset and claim", slots, nslots, q, keys, n, slot_of);
    run<1>("without the atomic max, 1 GiB written between", slots, nslots, q, keys, n, slot_of);
    return 0;
}
