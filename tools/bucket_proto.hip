// Prototype of the slot-bucketed join path: what would a batch cost if its events were first partitioned (stable)
// by home-slot range and each range were then worked off by one block that groups its events in LDS?
// Measures the partition passes (count, scan, scatter) and the memory pattern of the bucket pass (probe read,
// random message read, random row write, slot write-back) with a real in-LDS sort in between.
// Build: hipcc --offload-arch=gfx950 -O3 -o bucket_proto tools/bucket_proto.hip ; run: ./bucket_proto
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
constexpr int WIN = 4096, PB = 1024;       // events per window, threads per block

__global__ void gen_kernel(Rec* recs, uint32_t n, uint32_t nb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = mix(i * 77ull + 5) % nb;
    recs[i] = Rec{mix(b), mix(b + 99), (uint32_t)(mix(i) & 7), 64, 0, 0};
}

__global__ void __launch_bounds__(PB) count_kernel(const Rec* recs, uint32_t n, uint64_t mask, int shift, uint32_t B,
                                                   uint32_t W, uint32_t* counts, uint32_t* totals) {
    extern __shared__ uint32_t hist[];
    for (uint32_t b = threadIdx.x; b < B; b += PB) hist[b] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = blockIdx.x * WIN + wave * 256 + r * 64 + lane;
        if (i < n) atomicAdd(&hist[(uint32_t)((recs[i].lo & mask) >> shift)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += PB) {
        const uint32_t c = hist[b];
        counts[(size_t)b * W + blockIdx.x] = c;
        if (c) atomicAdd(&totals[b], c);
    }
}

// one wave per bucket: start of the bucket = sum of the totals before it; then an exclusive scan of its row
__global__ void __launch_bounds__(256) scan_kernel(uint32_t* counts, const uint32_t* totals, uint32_t B, uint32_t W,
                                                   uint32_t* bucket_begin) {
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= B) return;
    uint32_t s = 0;
    for (uint32_t j = lane; j < b; j += 64) s += totals[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { bucket_begin[b] = s; if (b == B - 1) bucket_begin[B] = s + totals[b]; }
    const uint32_t per = (W + 63) / 64;
    uint32_t* row = counts + (size_t)b * W;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t w = lane * per + k; if (w < W) mine += row[w]; }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
    uint32_t run = s + incl - mine;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t w = lane * per + k;
        if (w < W) { const uint32_t c = row[w]; row[w] = run; run += c; }
    }
}

__device__ inline uint64_t match_bucket(uint32_t b, int bits, bool valid) {
    uint64_t m = __ballot(valid);
    for (int j = 0; j < bits; ++j) {
        const uint64_t bal = __ballot((b >> j) & 1);
        m &= ((b >> j) & 1) ? bal : ~bal;
    }
    return valid ? m : 0;
}

__global__ void __launch_bounds__(PB) scatter_kernel(const Rec* recs, uint32_t n, uint64_t mask, int shift, uint32_t B,
                                                     int bbits, uint32_t W, const uint32_t* offsets, Rec* out, int32_t* status) {
    extern __shared__ uint32_t lds[];
    uint16_t* hist = reinterpret_cast<uint16_t*>(lds);            // [16][B]
    uint32_t* base = lds + 16 * B / 2;                            // [B]
    for (uint32_t k = threadIdx.x; k < 16 * B / 2; k += PB) lds[k] = 0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Rec r[4];
    uint32_t bk[4], pre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = blockIdx.x * WIN + wave * 256 + k * 64 + lane;
        if (i < n) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(&recs[i]);
            const ulonglong2 a = p[0], c = p[1];
            r[k].lo = a.x; r[k].hi = a.y; r[k].pos = (uint32_t)c.x; r[k].len = (uint32_t)(c.x >> 32); r[k].mw = (uint32_t)c.y;
            r[k].idx = i;
            bk[k] = (uint32_t)((a.x & mask) >> shift);
        } else bk[k] = 0;
    }
    for (uint32_t b = threadIdx.x; b < B; b += PB) base[b] = offsets[(size_t)b * W + blockIdx.x];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = blockIdx.x * WIN + wave * 256 + k * 64 + lane;
        const bool valid = i < n;
        const uint64_t m = match_bucket(bk[k], bbits, valid);
        uint16_t* h = &hist[wave * B + bk[k]];
        uint32_t old = valid ? *h : 0;
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (valid && (m >> lane) >> 1 == 0) *h = (uint16_t)(old + (uint32_t)__popcll(m));     // highest lane of the group
        pre[k] = old + rank;
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += PB) {
        uint32_t s = 0;
        for (int v = 0; v < 16; ++v) { const uint32_t c = hist[v * B + b]; hist[v * B + b] = (uint16_t)s; s += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = blockIdx.x * WIN + wave * 256 + k * 64 + lane;
        if (i >= n) continue;
        const uint32_t pos = base[bk[k]] + hist[wave * B + bk[k]] + pre[k];
        ulonglong2* p = reinterpret_cast<ulonglong2*>(&out[pos]);
        p[0] = make_ulonglong2(r[k].lo, r[k].hi);
        p[1] = make_ulonglong2((uint64_t)r[k].pos | ((uint64_t)r[k].len << 32), (uint64_t)r[k].mw | ((uint64_t)i << 32));
        status[i] = 1;
    }
}

// in-LDS bitonic sort of 1024 keys, one per thread; strides below 64 stay inside the wave (shuffles)
__device__ inline uint64_t sort1024(uint64_t key, uint64_t* buf) {
    const uint32_t t = threadIdx.x;
    for (uint32_t k = 2; k <= 1024; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            uint64_t other;
            if (j >= 64) {
                buf[t] = key;
                __syncthreads();
                other = buf[t ^ j];
                __syncthreads();
            } else {
                other = ((uint64_t)__shfl_xor((uint32_t)(key >> 32), j) << 32) | __shfl_xor((uint32_t)key, j);
            }
            const bool up = (t & k) == 0, lower = (t & j) == 0;
            const bool take_min = up == lower;
            key = take_min ? (key < other ? key : other) : (key > other ? key : other);
        }
    }
    return key;
}

// mode bits: 1 probe, 2 sort, 4 slot write-back, 8 message copy, 16 status scatter
__global__ void __launch_bounds__(1024) bucket_kernel(const Rec* part, const uint32_t* bucket_begin, uint4* slots, uint64_t mask,
                                                      const uint4* payloads, uint4* arena, uint64_t arena_lines,
                                                      int32_t* status, int mode, uint64_t* sink) {
    __shared__ uint64_t buf[1024];
    __shared__ uint32_t info[1024];
    const uint32_t begin = bucket_begin[blockIdx.x], end = bucket_begin[blockIdx.x + 1];
    const uint32_t t = threadIdx.x;
    for (uint32_t c0 = begin; c0 < end; c0 += 1024) {
        const uint32_t e = c0 + t;
        const bool valid = e < end;
        Rec r = {};
        uint64_t key = ~0ull;
        uint64_t s = 0;
        uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0, m2 = m0, m3 = m0;
        if (valid) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(&part[e]);
            const ulonglong2 a = p[0], c = p[1];
            r.lo = a.x; r.hi = a.y; r.pos = (uint32_t)c.x; r.idx = (uint32_t)(c.y >> 32);
            if (mode & 8) {      // the message is requested before anything is known about the event
                const uint4* src = payloads + (size_t)r.idx * 4;
                m0 = src[0]; m1 = src[1]; m2 = src[2]; m3 = src[3];
            }
            s = r.lo & mask;
            if (mode & 1) {
                const uint4 k0 = slots[s * 4], k1 = slots[(s ^ 1) * 4], i0 = slots[s * 4 + 1];
                if (k0.x == 0x12345 && k1.x == 77) s ^= 1;
                info[t] = i0.x;
            }
            key = (s << 32) | t;
        }
        if (mode & 2) key = sort1024(key, buf);
        __syncthreads();
        buf[t] = key;
        __syncthreads();
        const bool alive = key != ~0ull;
        const bool head = alive && (t == 0 || (buf[t - 1] >> 32) != (key >> 32));
        if (head && (mode & 4)) {
            uint32_t m = 1;
            while (t + m < 1024 && (buf[t + m] >> 32) == (key >> 32)) ++m;
            const uint64_t sl = key >> 32;
            slots[sl * 4 + 3] = make_uint4(m, info[(uint32_t)key & 1023], t, 1);
        }
        if (valid && (mode & 8)) {
            const uint64_t line = mix(r.lo + r.pos) % arena_lines;
            uint4* d = arena + line * 4;
            d[0] = m0; d[1] = m1; d[2] = m2; d[3] = m3;
        }
        if (valid && (mode & 16) && (r.idx & 7) == 0) status[r.idx] = 2;
        __syncthreads();
    }
    if (sink && t == 5000) sink[0] = 1;
}

// second form: grouping by an LDS hash (no sort), messages copied by 4 lanes each at the end
// mode bits: 1 probe, 2 group, 4 slot write-back, 8 message copy, 16 status scatter
template <int T>
__global__ void __launch_bounds__(T) bucket2_kernel(const Rec* part, const uint32_t* bucket_begin, uint4* slots, uint64_t mask,
                                                    const uint4* payloads, uint4* arena, uint64_t arena_lines,
                                                    int32_t* status, int mode, uint64_t* sink) {
    constexpr int H = 2 * T;
    __shared__ uint32_t hkey[H];
    __shared__ uint32_t hcnt[H];
    __shared__ uint32_t hoff[H];
    __shared__ uint32_t run[T];
    __shared__ uint32_t wsum[T / 64];
    const uint32_t begin = bucket_begin[blockIdx.x], end = bucket_begin[blockIdx.x + 1];
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t c0 = begin; c0 < end; c0 += T) {
        const uint32_t e = c0 + t;
        const bool valid = e < end;
        Rec r = {};
        uint64_t s = 0;
        uint32_t info = 0;
        hkey[t] = 0; hkey[t + T] = 0; hcnt[t] = 0; hcnt[t + T] = 0;
        if (valid) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(&part[e]);
            const ulonglong2 a = p[0], c = p[1];
            r.lo = a.x; r.hi = a.y; r.pos = (uint32_t)c.x; r.idx = (uint32_t)(c.y >> 32);
            s = r.lo & mask;
            if (mode & 1) {
                const uint4 k0 = slots[s * 4], k1 = slots[(s ^ 1) * 4], i0 = slots[s * 4 + 1];
                if (k0.x == 0x12345 && k1.x == 77) s ^= 1;
                info = i0.x;
            }
        }
        __syncthreads();
        uint32_t cell = 0, rank = 0;
        if (mode & 2) {
            if (valid) {
                cell = (uint32_t)(s * 2654435761u) >> 21 & (H - 1);
                for (;; cell = (cell + 1) & (H - 1)) {
                    const uint32_t k = atomicCAS(&hkey[cell], 0u, (uint32_t)s + 1);
                    if (k == 0 || k == (uint32_t)s + 1) break;
                }
                rank = atomicAdd(&hcnt[cell], 1u);
            }
            __syncthreads();
            // exclusive scan over the 2T cells: two per thread
            const uint32_t c0_ = hcnt[2 * t], c1_ = hcnt[2 * t + 1];
            uint32_t incl = c0_ + c1_;
            for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t basew = 0;
            for (uint32_t v = 0; v < wv; ++v) basew += wsum[v];
            hoff[2 * t] = basew + incl - c0_ - c1_;
            hoff[2 * t + 1] = basew + incl - c1_;
            __syncthreads();
            if (valid) run[hoff[cell] + rank] = t;
            __syncthreads();
        }
        if (valid && rank == 0 && (mode & 4)) {
            const uint32_t m = (mode & 2) ? hcnt[cell] : 1;
            uint32_t acc = 0;
            if (mode & 2) for (uint32_t k = 0; k < m && k < 8; ++k) acc += run[hoff[cell] + k];
            slots[s * 4 + 3] = make_uint4(m, info, acc, 1);
        }
        if (mode & 8) {
            const uint64_t line = valid ? mix(r.lo + r.pos) % arena_lines : 0;
            const uint32_t lo32 = (uint32_t)line, hi32 = (uint32_t)(line >> 32);
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int src_lane = step * 16 + (lane >> 2);
                const uint32_t idx = __shfl(r.idx, src_lane);
                const uint64_t ln = ((uint64_t)__shfl(hi32, src_lane) << 32) | __shfl(lo32, src_lane);
                const bool ok = __shfl((int)valid, src_lane);
                if (ok) arena[ln * 4 + (lane & 3)] = payloads[(size_t)idx * 4 + (lane & 3)];
            }
        }
        if (valid && (mode & 16) && (r.idx & 7) == 0) status[r.idx] = 2;
        __syncthreads();
    }
    if (sink && t == 5000) sink[0] = 1;
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

int main() {
    const uint32_t n = 1u << 20, nb = 1u << 20;
    const uint64_t slots = 1ull << 21, mask = slots - 1;
    Rec *recs, *part; uint4 *tab, *pay, *arena; int32_t* status; uint32_t *counts, *totals, *bb; uint64_t* sink;
    const uint64_t arena_lines = (600ull << 20) / 64;
    CK(hipMalloc(&recs, (size_t)n * 32)); CK(hipMalloc(&part, (size_t)n * 32));
    CK(hipMalloc(&tab, slots * 64)); CK(hipMemset(tab, 0, slots * 64));
    CK(hipMalloc(&pay, (size_t)n * 64)); CK(hipMemset(pay, 1, (size_t)n * 64));
    CK(hipMalloc(&arena, arena_lines * 64)); CK(hipMalloc(&status, n * 4)); CK(hipMalloc(&sink, 64));
    gen_kernel<<<n / 256, 256>>>(recs, n, nb);
    for (uint32_t B : {512u, 1024u, 2048u}) {
        int bbits = 0; while ((1u << bbits) < B) ++bbits;
        const int shift = 21 - bbits;
        const uint32_t W = (n + WIN - 1) / WIN;
        CK(hipMalloc(&counts, (size_t)B * W * 4)); CK(hipMalloc(&totals, B * 4)); CK(hipMalloc(&bb, (B + 1) * 4));
        printf("B = %u buckets, %u windows\n", B, W);
        float us = time_us([&] { hipMemsetAsync(totals, 0, B * 4, 0); count_kernel<<<W, PB, B * 4>>>(recs, n, mask, shift, B, W, counts, totals); }, 20);
        printf("  memset + count : %6.1f us\n", us);
        us = time_us([&] { scan_kernel<<<(B * 64 + 255) / 256, 256>>>(counts, totals, B, W, bb); }, 1);     // (in place: once)
        // redo count for a clean scan
        hipMemsetAsync(totals, 0, B * 4, 0); count_kernel<<<W, PB, B * 4>>>(recs, n, mask, shift, B, W, counts, totals);
        scan_kernel<<<(B * 64 + 255) / 256, 256>>>(counts, totals, B, W, bb);
        CK(hipDeviceSynchronize());
        printf("  scan (twice)   : %6.1f us\n", us);
        if (B <= 1024) {
            us = time_us([&] { scatter_kernel<<<W, PB, 16 * B * 2 + B * 4>>>(recs, n, mask, shift, B, bbits, W, counts, part, status); }, 20);
            printf("  scatter        : %6.1f us\n", us);
        } else { CK(hipFree(counts)); CK(hipFree(totals)); CK(hipFree(bb)); continue; }
        us = time_us([&] {
            hipMemsetAsync(totals, 0, B * 4, 0); count_kernel<<<W, PB, B * 4>>>(recs, n, mask, shift, B, W, counts, totals);
            scan_kernel<<<(B * 64 + 255) / 256, 256>>>(counts, totals, B, W, bb);
            scatter_kernel<<<W, PB, 16 * B * 2 + B * 4>>>(recs, n, mask, shift, B, bbits, W, counts, part, status); }, 20);
        printf("  all three      : %6.1f us\n", us);
        // check: partition is a stable permutation
        {
            Rec* h = (Rec*)malloc((size_t)n * 32); uint32_t* hb = (uint32_t*)malloc((B + 1) * 4);
            CK(hipMemcpy(h, part, (size_t)n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb, bb, (B + 1) * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; uint32_t maxb = 0;
            for (uint32_t b = 0; b < B; ++b) {
                if (hb[b + 1] - hb[b] > maxb) maxb = hb[b + 1] - hb[b];
                for (uint32_t e = hb[b]; e < hb[b + 1]; ++e) {
                    if (((h[e].lo & mask) >> shift) != b) ++bad;
                    if (e > hb[b] && h[e].idx <= h[e - 1].idx) ++bad;
                }
            }
            printf("  partition check: end %u of %u, %zu errors, largest bucket %u\n", hb[B], n, bad, maxb);
            free(h); free(hb);
        }
        for (int mode : {0, 1, 3, 7, 8, 9, 15, 31}) {
            us = time_us([&] { bucket_kernel<<<B, 1024>>>(part, bb, tab, mask, pay, arena, arena_lines, status, mode, sink); }, 20);
            printf("  bucket pass mode %2d : %6.1f us\n", mode, us);
        }
        for (int mode : {0, 1, 3, 7, 8, 9, 15, 31}) {
            us = time_us([&] { bucket2_kernel<1024><<<B, 1024>>>(part, bb, tab, mask, pay, arena, arena_lines, status, mode, sink); }, 20);
            printf("  bucket2<1024> mode %2d : %6.1f us\n", mode, us);
        }
        for (int mode : {0, 1, 3, 7, 8, 9, 15, 31}) {
            us = time_us([&] { bucket2_kernel<512><<<B, 512>>>(part, bb, tab, mask, pay, arena, arena_lines, status, mode, sink); }, 20);
            printf("  bucket2<512> mode %2d : %6.1f us\n", mode, us);
        }
        CK(hipFree(counts)); CK(hipFree(totals)); CK(hipFree(bb));
    }
    return 0;
}
