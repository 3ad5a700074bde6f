// What a streaming copy reaches on this GPU, by kernel shape: blocks per compute unit, loads in flight per lane,
// the stride between a lane's loads (whole grid / one block's chunk), plain or non-temporal accesses.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/copy_sweep tools/copy_sweep.hip && tools/_build/copy_sweep
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid-stride: lane's K loads are a whole grid apart
template <int K, bool NT>
__global__ void __launch_bounds__(256) copy_grid(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (; i + (K - 1) * stride < n16; i += K * stride) {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < K; ++k) { if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// block-contiguous: a block owns chunks of K x 256 x 16 bytes
template <int K, bool NT>
__global__ void __launch_bounds__(256) copy_chunk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
    const uint64_t chunk = (uint64_t)K * 256;
    for (uint64_t c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        const uint64_t base = c * chunk + threadIdx.x;
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) if (base + k * 256 < n16) v[k] = NT ? __builtin_nontemporal_load(&src[base + k * 256]) : src[base + k * 256];
#pragma unroll
        for (int k = 0; k < K; ++k) if (base + k * 256 < n16) { if (NT) __builtin_nontemporal_store(v[k], &dst[base + k * 256]); else dst[base + k * 256] = v[k]; }
    }
}

// one shot: every lane moves K x 16 bytes, no loop (grid covers the buffer)
template <int K, bool NT>
__global__ void __launch_bounds__(256) copy_once(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * (K * 256) + threadIdx.x;
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) if (base + k * 256 < n16) v[k] = NT ? __builtin_nontemporal_load(&src[base + k * 256]) : src[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; ++k) if (base + k * 256 < n16) { if (NT) __builtin_nontemporal_store(v[k], &dst[base + k * 256]); else dst[base + k * 256] = v[k]; }
}

template <class F>
static double run(F launch, uint64_t bytes, hipStream_t s) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipEventRecord(a, s));
    for (int k = 0; k < 8; ++k) launch();
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return 2.0 * bytes * 8 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024ull) << 20;
    const uint64_t n16 = bytes / 16;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    void *src, *dst;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipMemsetAsync(src, 1, bytes, s));
    CHECK(hipMemsetAsync(dst, 2, bytes, s));
    printf("%d CUs, %llu MiB per buffer; GB/s of bytes read + written\n", cus, (unsigned long long)(bytes >> 20));
    const u32x4* S = (const u32x4*)src;
    u32x4* D = (u32x4*)dst;
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        CHECK(hipEventRecord(a, s));
        for (int k = 0; k < 8; ++k) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("hipMemcpyAsync D2D            %8.1f\n", 2.0 * bytes * 8 / (ms * 1e-3) / 1e9);
    }
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipMemsetAsync(dst, 3, bytes, s));
        CHECK(hipEventRecord(a, s));
        for (int k = 0; k < 8; ++k) CHECK(hipMemsetAsync(dst, 3, bytes, s));
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("hipMemsetAsync (written only) %8.1f\n", 1.0 * bytes * 8 / (ms * 1e-3) / 1e9);
    }
#define GRID(K, NT) for (int bpc : {2, 4, 8, 16, 32}) printf("grid-stride K=%d nt=%d blocks/CU=%-2d %8.1f\n", K, NT, bpc, run([&] { copy_grid<K, NT><<<cus * bpc, 256, 0, s>>>(S, D, n16); }, bytes, s));
#define CHUNK(K, NT) for (int bpc : {2, 4, 8, 16, 32}) printf("chunked     K=%d nt=%d blocks/CU=%-2d %8.1f\n", K, NT, bpc, run([&] { copy_chunk<K, NT><<<cus * bpc, 256, 0, s>>>(S, D, n16); }, bytes, s));
#define ONCE(K, NT) printf("one-shot    K=%d nt=%d              %8.1f\n", K, NT, run([&] { copy_once<K, NT><<<(unsigned)((n16 + K * 256 - 1) / (K * 256)), 256, 0, s>>>(S, D, n16); }, bytes, s));
    GRID(1, false) GRID(2, false) GRID(4, false) GRID(8, false) GRID(4, true) GRID(8, true)
    CHUNK(1, false) CHUNK(4, false) CHUNK(8, false) CHUNK(4, true)
    ONCE(1, false) ONCE(2, false) ONCE(4, false) ONCE(8, false) ONCE(4, true)
    CHECK(hipFree(src));
    CHECK(hipFree(dst));
    return 0;
}
