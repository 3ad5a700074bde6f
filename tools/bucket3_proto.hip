// Prototype of a second slot-bucketed join path: PROBE FIRST, then bucket 8-byte keys.
//
// tools/bucket_proto.hip partitioned the 32-byte records by home-slot range before anything else and lost to the
// chain path.  This form keeps the link pass's order of work (records and messages are read in array order, the table
// is probed, the message goes to its row) and replaces only the per-event atomic exchange on the slot by an 8-byte key
// {slot, event index, position} written to the bucket of the slot's range: tiles of 4096 events are counted per bucket
// in LDS, one cursor atomic per bucket and tile reserves the place, the keys leave LDS in runs.  The second pass has
// one block per bucket: keys into LDS, chains built with LDS exchanges on a directly indexed head array, the first
// event of every barrier replays its chain from LDS and reads / writes the slot once.  No barrier logic beyond the
// memory pattern of it (bitmap OR, one status store per completing event, one queue word per completed barrier).
// Build: hipcc --offload-arch=gfx950 -O3 -o bucket3_proto tools/bucket3_proto.hip ; run: ./bucket3_proto
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}

struct Rec { uint64_t lo, hi; uint32_t pos, len, mw, idx; };
constexpr int TILE = 4096, PB = 1024;

__global__ void gen_kernel(Rec* recs, uint32_t n, uint32_t nb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = mix(i * 77ull + 5) % nb;
    recs[i] = Rec{mix(b), mix(b + 99), (uint32_t)(mix(i) & 7), 64, 0, 0};
}

// key: slot << 32 | event index << 8 | position (positions below 256 here)
// mode bits: 1 probe, 2 bucket the keys, 4 message copy
template <int B>
__global__ void __launch_bounds__(PB) link_bucket_kernel(const Rec* recs, uint32_t n, const uint4* slots, uint64_t mask, int shift,
                                                         const uint4* payloads, uint4* arena, uint64_t arena_lines,
                                                         int32_t* status, uint32_t* cursor, uint64_t* keys, uint32_t cap, int mode) {
    __shared__ uint32_t hist[B], off[B], base[B];
    __shared__ uint64_t stage[TILE];
    __shared__ uint32_t wsum[PB / 64];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint32_t b = t; b < B; b += PB) hist[b] = 0;
    __syncthreads();
    uint64_t key[4];
    uint32_t bk[4], rank[4];
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = blockIdx.x * TILE + wave * 256 + k * 64 + lane;
        valid[k] = i < n;
        key[k] = 0; bk[k] = 0; rank[k] = 0;
        uint64_t line = 0;
        if (valid[k]) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(&recs[i]);
            const ulonglong2 a = p[0], c = p[1];
            uint64_t s = a.x & mask;
            if (mode & 1) {
                const uint4 k0 = slots[s * 4], k1 = slots[(s ^ 1) * 4], i0 = slots[s * 4 + 1];
                if (k0.x == 0x12345 && k1.x == 77 && i0.x == 3) s ^= 1;
            }
            key[k] = (s << 32) | ((uint64_t)i << 8) | ((uint32_t)c.x & 255);
            bk[k] = (uint32_t)(s >> shift);
            status[i] = 1;
            line = mix(a.x + (uint32_t)c.x) % arena_lines;
        }
        if (mode & 4) {       // the wave's 64 messages, four lanes each, into their rows
            const uint32_t lo32 = (uint32_t)line, hi32 = (uint32_t)(line >> 32);
            const uint32_t i0 = blockIdx.x * TILE + wave * 256 + k * 64;
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int src = step * 16 + (lane >> 2);
                const uint64_t ln = ((uint64_t)__shfl(hi32, src) << 32) | __shfl(lo32, src);
                if (i0 + src < n) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&payloads[(size_t)(i0 + src) * 4 + (lane & 3)]));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(&arena[ln * 4 + (lane & 3)]));
                }
            }
        }
        if ((mode & 2) && valid[k]) rank[k] = atomicAdd(&hist[bk[k]], 1u);
    }
    if (!(mode & 2)) return;
    __syncthreads();
    // exclusive scan of the B counts (B <= 1024: one per thread), and one cursor atomic per bucket that has events
    {
        const uint32_t c = t < B ? hist[t] : 0;
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t bw = 0;
        for (uint32_t v = 0; v < wave; ++v) bw += wsum[v];
        if (t < B) {
            off[t] = bw + incl - c;
            base[t] = c ? atomicAdd(&cursor[t], c) : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (valid[k]) stage[off[bk[k]] + rank[k]] = key[k];
    __syncthreads();
    // the staged keys leave in runs: entry e belongs to the bucket whose range holds it (binary search over off[])
    const uint32_t total = off[B - 1] + hist[B - 1];
    for (uint32_t e = t; e < total; e += PB) {
        const uint64_t kv = stage[e];
        const uint32_t b = (uint32_t)((kv >> 32) >> shift);
        const uint32_t at = base[b] + (e - off[b]);
        if (at < cap) keys[(size_t)b * cap + at] = kv;
    }
}

// one block per bucket; up to EPT events per thread.  mode bits: 1 chains in LDS, 2 slot read + write, 4 status / queue stores
template <int T, int EPT, int RANGE>
__global__ void __launch_bounds__(T) resolve_bucket_kernel(const uint32_t* cursor, const uint64_t* keys, uint32_t cap, uint4* slots,
                                                           int32_t* status, uint64_t* q, int mode, uint64_t* sink) {
    __shared__ uint32_t head[RANGE];
    __shared__ uint32_t nxt[T * EPT];
    __shared__ uint64_t kbuf[T * EPT];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    uint32_t nb = cursor[b];
    if (nb > cap) nb = cap;
    if (nb > T * EPT) nb = T * EPT;
    for (uint32_t s = t; s < RANGE; s += T) head[s] = 0;
    uint64_t key[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t e = k * T + t;
        key[k] = e < nb ? keys[(size_t)b * cap + e] : 0;
        kbuf[e] = key[k];
    }
    __syncthreads();
    uint32_t prev[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t e = k * T + t;
        prev[k] = 1;
        if (e < nb && (mode & 1)) {
            prev[k] = atomicExch(&head[(uint32_t)(key[k] >> 32) & (RANGE - 1)], e + 1);
            nxt[e] = prev[k];
        }
    }
    __syncthreads();
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t e = k * T + t;
        if (e >= nb || prev[k] != 0) continue;          // the barrier's first event resolves it
        const uint64_t s = key[k] >> 32;
        uint4 info = make_uint4(0, 0, 0, 0), tail = info;
        if (mode & 2) { info = slots[s * 4 + 1]; tail = slots[s * 4 + 3]; }
        uint32_t bits = tail.z, m = 0, last_idx = 0;
        uint32_t cur = (mode & 1) ? head[(uint32_t)s & (RANGE - 1)] : e + 1;
        while (cur != 0 && m < 64) {
            const uint64_t kv = kbuf[cur - 1];
            bits |= 1u << ((uint32_t)kv & 31);
            const uint32_t idx = (uint32_t)(kv >> 8) & 0xFFFFFF;
            if (idx > last_idx) last_idx = idx;
            cur = (mode & 1) ? nxt[cur - 1] : 0;
            ++m;
        }
        if (mode & 2) {
            slots[s * 4 + 3] = make_uint4(tail.x, tail.y, bits, m + info.x);
            reinterpret_cast<uint64_t*>(&slots[s * 4 + 2])[1] = 1700000000000ull;      // join-ts
        }
        if ((mode & 4) && (s & 7) == 0) { status[last_idx] = 2; q[s] = 1700000000000ull; }   // one barrier in eight completes
        acc += bits;
    }
    if (sink && acc == 0x123456789ull) sink[0] = acc;
}

template <class F> static float time_us(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

template <int B, int T, int EPT>
static void run(const Rec* recs, uint32_t n, uint4* tab, uint64_t mask, const uint4* pay, uint4* arena, uint64_t arena_lines,
                int32_t* status, uint64_t* q, uint64_t* sink) {
    constexpr int RANGE = (1 << 21) / B;
    int bbits = 0; while ((1u << bbits) < (unsigned)B) ++bbits;
    const int shift = 21 - bbits;
    const uint32_t cap = 2 * (n / B) + 512, tiles = (n + TILE - 1) / TILE;
    uint32_t* cursor; uint64_t* keys;
    CK(hipMalloc(&cursor, B * 4)); CK(hipMalloc(&keys, (size_t)B * cap * 8));
    printf("B = %d buckets of %d slots, %u keys each at most, second pass: %d threads x %d events\n", B, RANGE, cap, T, EPT);
    for (int mode : {0, 1, 5, 3, 7}) {
        float us = time_us([&] {
            (void)hipMemsetAsync(cursor, 0, B * 4, 0);
            link_bucket_kernel<B><<<tiles, PB>>>(recs, n, tab, mask, shift, pay, arena, arena_lines, status, cursor, keys, cap, mode); }, 20);
        printf("  first pass  mode %d (1 probe, 2 keys to buckets, 4 message copy): %6.1f us\n", mode, us);
    }
    uint32_t* hc = (uint32_t*)malloc(B * 4);
    CK(hipMemcpy(hc, cursor, B * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0; uint32_t mx = 0;
    for (int b = 0; b < B; ++b) { tot += hc[b]; if (hc[b] > mx) mx = hc[b]; }
    printf("  keys in buckets: %llu of %u, largest bucket %u\n", (unsigned long long)tot, n, mx);
    free(hc);
    for (int mode : {0, 1, 3, 7}) {
        float us = time_us([&] { resolve_bucket_kernel<T, EPT, RANGE><<<B, T>>>(cursor, keys, cap, tab, status, q, mode, sink); }, 20);
        printf("  second pass mode %d (1 chains in LDS, 2 slot read + write, 4 status and queue stores): %6.1f us\n", mode, us);
    }
    float us = time_us([&] {
        (void)hipMemsetAsync(cursor, 0, B * 4, 0);
        link_bucket_kernel<B><<<tiles, PB>>>(recs, n, tab, mask, shift, pay, arena, arena_lines, status, cursor, keys, cap, 7);
        resolve_bucket_kernel<T, EPT, RANGE><<<B, T>>>(cursor, keys, cap, tab, status, q, 7, sink); }, 20);
    printf("  both passes, everything on: %6.1f us\n", us);
    CK(hipFree(cursor)); CK(hipFree(keys));
}

int main() {
    const uint32_t n = 1u << 20, nb = 1u << 20;
    const uint64_t slots = 1ull << 21, mask = slots - 1;
    Rec* recs; uint4 *tab, *pay, *arena; int32_t* status; uint64_t *sink, *q;
    const uint64_t arena_lines = (600ull << 20) / 64;
    CK(hipMalloc(&recs, (size_t)n * 32));
    CK(hipMalloc(&tab, slots * 64)); CK(hipMemset(tab, 0, slots * 64));
    CK(hipMalloc(&pay, (size_t)n * 64)); CK(hipMemset(pay, 1, (size_t)n * 64));
    CK(hipMalloc(&arena, arena_lines * 64)); CK(hipMalloc(&status, n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&q, slots * 8));
    gen_kernel<<<n / 256, 256>>>(recs, n, nb);
    CK(hipDeviceSynchronize());
    run<256, 1024, 8>(recs, n, tab, mask, pay, arena, arena_lines, status, q, sink);
    run<512, 1024, 4>(recs, n, tab, mask, pay, arena, arena_lines, status, q, sink);
    run<1024, 512, 4>(recs, n, tab, mask, pay, arena, arena_lines, status, q, sink);
    run<1024, 1024, 2>(recs, n, tab, mask, pay, arena, arena_lines, status, q, sink);
    return 0;
}
