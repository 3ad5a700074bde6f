// Memory-system ceilings of one GPU for the access shapes the barrier engine uses:
//   copy      : streaming 16-byte loads/stores over 1 GiB (the classic HBM number)
//   rand_rd   : one random 64-byte table line per thread, 8 bytes read from it
//   rand_rmw  : the same, plus an 8-byte store into the line (what join_link does to a slot)
//   rand_xchg : the same with an atomic exchange instead of load+store
//   rand_wr64 : one full random 64-byte line written per 4 lanes (what join_copy does at P=64)
// Build: hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip ; run: ./membench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) dst[i] = src[i];
}

__global__ void rand_rd_kernel(const uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                               uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    if (v == 0x1234567) out[0] = v;          // never true; keeps the load
}

// two lines per thread: pair = 1 -> both halves of one aligned 128-byte block, pair = 0 -> line and its
// successor (crosses a 128-byte boundary for odd lines)
__global__ void rand_rd2_kernel(const uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                int pair, uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t other = pair ? (line ^ 1) : ((line + 1) & line_mask);
    uint64_t v = tab[line * 8], u = tab[other * 8];
    if (v + u == 0x1234567) out[0] = v;
}

__global__ void rand_rmw_kernel(uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    tab[line * 8 + 3] = v + i;
}

__global__ void rand_xchg_kernel(uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                 uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    uint32_t old = atomicExch(reinterpret_cast<uint32_t*>(&tab[line * 8 + 3]), i + (uint32_t)v);
    if (old == 0xfffffff3u) out[0] = old;
}

// read + atomic max whose result is not used (the compiler emits the non-returning form)
__global__ void rand_amax_kernel(uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                 uint64_t* __restrict__ out, int with_read) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = with_read ? tab[line * 8] : 0;
    atomicMax(reinterpret_cast<uint32_t*>(&tab[line * 8 + 3]), i);
    if (v == 0x1234567) out[0] = v;
}

// read + 64-bit compare-and-swap on the word read + a non-returning max on a neighbour word (fork_claim)
__global__ void rand_cas_kernel(uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                uint64_t* __restrict__ out, int with_max) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[line * 8]), (unsigned long long)v,
                                       (unsigned long long)v);
    if (with_max) atomicMax(reinterpret_cast<uint32_t*>(&tab[line * 8 + 3]), i);
    if (old == 0x1234567) out[0] = old;
}

// the slot line is read, the exchange goes to a dense 4-byte column indexed by the line number
__global__ void rand_colxchg_kernel(const uint64_t* __restrict__ tab, uint32_t* __restrict__ col, uint64_t line_mask,
                                    uint32_t n, uint64_t seed, uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    uint32_t old = atomicExch(&col[line], i);
    if (old + v == 0xfffffff3u) out[0] = old;
}

// K independent random lines per thread (more requests in flight per wave)
template <int K>
__global__ void rand_rdk_kernel(const uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * K >= n) return;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += tab[(mix(seed + i * K + k) & line_mask) * 8];
    if (acc == 0x1234567) out[0] = acc;
}

// exchange with workgroup scope: the hardware may serve it in the XCD's own L2 (NOT coherent across XCDs;
// only a yardstick for what a device-wide exchange costs beyond that)
__global__ void rand_xchg_wg_kernel(uint64_t* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed,
                                    uint64_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    uint64_t v = tab[line * 8];
    uint32_t old = __hip_atomic_exchange(reinterpret_cast<uint32_t*>(&tab[line * 8 + 3]), i + (uint32_t)v, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == 0xfffffff3u) out[0] = old;
}

__global__ void rand_wr64_kernel(uint4* __restrict__ tab, uint64_t line_mask, uint32_t n, uint64_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = t >> 2, c = t & 3;
    if (i >= n) return;
    uint64_t line = mix(seed + i) & line_mask;
    tab[line * 4 + c] = make_uint4(i, c, i, c);
}

// what join_copy does at P = 64: row i of a contiguous source goes to a random table line, 4 lanes x 16 bytes;
// idx != nullptr: the destination line comes from an 8-byte word per row (as from the dst array)
template <bool NT>
__global__ void scatter_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ tab, const uint64_t* __restrict__ idx,
                                    uint64_t line_mask, uint32_t n, uint64_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = t >> 2, c = t & 3;
    if (i >= n) return;
    uint64_t line = idx ? (idx[i] & line_mask) : (mix(seed + i) & line_mask);
    if (NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&src[(size_t)i * 4 + c]));
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(&tab[line * 4 + c]));
    } else {
        tab[line * 4 + c] = src[(size_t)i * 4 + c];
    }
}

__global__ void fill_idx_kernel(uint64_t* idx, uint32_t n, uint64_t seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = mix(seed + i);
}

template <class F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f(r + 1);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t GiB = 1ull << 30;
    uint4 *src, *dst; uint64_t* out;
    CK(hipMalloc(&src, GiB)); CK(hipMalloc(&dst, GiB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, GiB)); CK(hipMemset(dst, 0, GiB));
    const size_t n16 = GiB / 16;
    float ms = time_ms([&](int) { copy_kernel<<<256 * 16, 256>>>(src, dst, n16); }, 10);
    printf("copy 1 GiB            : %.3f ms  %.0f GB/s (read+write)\n", ms, 2.0 * GiB / ms / 1e6);

    uint32_t* col; CK(hipMalloc(&col, GiB / 16)); CK(hipMemset(col, 0, GiB / 16));
    for (uint32_t n : {1u << 20, 1u << 22})
    for (size_t tab_bytes : {size_t(8) << 20, size_t(128) << 20, size_t(1) << 30}) {
        uint64_t line_mask = tab_bytes / 64 - 1;
        uint64_t* tab = reinterpret_cast<uint64_t*>(dst);
        printf("table %4zu MiB, %u random 64-byte lines per launch\n", tab_bytes >> 20, n);
        ms = time_ms([&](int r) { rand_rd_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_rd   : %7.1f us  %.1f G lines/s\n", ms * 1e3, n / ms / 1e6);
        ms = time_ms([&](int r) { rand_rd2_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, 1, out); }, 20);
        printf("  rand_rd2 same 128 B : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_rd2_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, 0, out); }, 20);
        printf("  rand_rd2 line, line+1: %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_rmw_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r); }, 20);
        printf("  rand_rmw  : %7.1f us  %.1f G lines/s\n", ms * 1e3, n / ms / 1e6);
        ms = time_ms([&](int r) { rand_xchg_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_xchg : %7.1f us  %.1f G lines/s\n", ms * 1e3, n / ms / 1e6);
        ms = time_ms([&](int r) { rand_xchg_wg_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_xchg, workgroup scope : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_amax_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out, 1); }, 20);
        printf("  rand_rd + max (no return) : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_amax_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out, 0); }, 20);
        printf("  max (no return) alone : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_cas_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out, 0); }, 20);
        printf("  rand_rd + cas : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_cas_kernel<<<n / 256, 256>>>(tab, line_mask, n, 7777ull * r, out, 1); }, 20);
        printf("  rand_rd + cas + max : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_colxchg_kernel<<<n / 256, 256>>>(tab, col, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_rd + xchg on a 4-byte column : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_rdk_kernel<2><<<n / 2 / 256, 256>>>(tab, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_rd 2 lines per thread : %7.1f us\n", ms * 1e3);
        ms = time_ms([&](int r) { rand_rdk_kernel<4><<<n / 4 / 256, 256>>>(tab, line_mask, n, 7777ull * r, out); }, 20);
        printf("  rand_rd 4 lines per thread : %7.1f us\n", ms * 1e3);
        {
            uint64_t* idx = reinterpret_cast<uint64_t*>(col);
            fill_idx_kernel<<<n / 256, 256>>>(idx, n, 99);
            const uint4* rows = src;
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            ms = time_ms([&](int r) { scatter_rows_kernel<false><<<n * 4 / 256, 256>>>(rows, t4, nullptr, line_mask, n, 7777ull * r); }, 20);
            printf("  rows -> random lines              : %7.1f us\n", ms * 1e3);
            ms = time_ms([&](int r) { scatter_rows_kernel<true><<<n * 4 / 256, 256>>>(rows, t4, nullptr, line_mask, n, 7777ull * r); }, 20);
            printf("  rows -> random lines, non-temporal: %7.1f us\n", ms * 1e3);
            ms = time_ms([&](int r) { scatter_rows_kernel<false><<<n * 4 / 256, 256>>>(rows, t4, idx, line_mask, n, 0); }, 20);
            printf("  rows -> lines from an index array : %7.1f us\n", ms * 1e3);
            ms = time_ms([&](int r) { scatter_rows_kernel<true><<<n * 4 / 256, 256>>>(rows, t4, idx, line_mask, n, 0); }, 20);
            printf("  same, non-temporal                : %7.1f us\n", ms * 1e3);
        }
        ms = time_ms([&](int r) { rand_wr64_kernel<<<n * 4 / 256, 256>>>(reinterpret_cast<uint4*>(tab), line_mask, n, 7777ull * r); }, 20);
        printf("  rand_wr64 : %7.1f us  %.1f G lines/s\n", ms * 1e3, n / ms / 1e6);
    }
    CK(hipFree(col)); CK(hipFree(src)); CK(hipFree(dst)); CK(hipFree(out));
    return 0;
}
