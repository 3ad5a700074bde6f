// Resident blocks per compute unit of the engine's kernels, as the runtime counts them (registers, LDS, waves).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idistributed_join_amd/csrc -o tools/_build/occupancy tools/occupancy.hip
#include <cstdio>
#include <hip/hip_runtime.h>
#include "dj_kernels.hpp"

#define SHOW(kernel, threads)                                                                             \
    do {                                                                                                  \
        int nb = -1;                                                                                      \
        hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0);            \
        printf("%-44s %4d threads: %d blocks per compute unit (%s)\n", #kernel, threads, nb, hipGetErrorString(rc)); \
    } while (0)

int main() {
    using namespace dj;
    SHOW((join_link_bucket_kernel<4, 4>), BK_BLOCK);
    SHOW((join_link_bucket_kernel<4, 3>), BK_BLOCK);
    SHOW((join_link_bucket_kernel<4, 2>), BK_BLOCK);
    SHOW((join_link_bucket_kernel<4, 1>), BK_BLOCK);
    SHOW(join_resolve_bucket_kernel, RB_T);
    SHOW((join_link_kernel<false, 4>), LINK_BLOCK);
    SHOW((join_resolve_kernel<CHAIN_REG>), RESOLVE_BLOCK);
    SHOW(pop_due_kernel, 256);
    SHOW(gather_kernel, 256);
    SHOW((ack_kernel<false>), ACK_BLOCK);
    SHOW(fork_claim_kernel, 256);
    SHOW(fork_apply_kernel, FORK_APPLY_BLOCK);
    return 0;
}
