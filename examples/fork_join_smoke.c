/* fork_join_smoke.c — the reference's fork-join-test.sh against the C ABI, from plain C (no Python, no torch): one
 * barrier "smoke~1" of count 3, three joins, the publisher's poll (dj_sweep), the ack; then a barrier that times out
 * with one of its three parts.  What a host in any language does through its FFI is what this file does through the
 * header: device buffers for the batch calls, host pointers for the single-key ones.
 *
 * Build (from the repository root, after __graft_entry__.build() made the library):
 *   gcc -std=c11 -O1 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o examples/fork_join_smoke \
 *       examples/fork_join_smoke.c -Ldistributed_join_amd -ldj_engine -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,'$ORIGIN/../distributed_join_amd' -Wl,-rpath,/opt/rocm/lib
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dj_engine.h"

#define OK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dj_last_error(e)); return 1; } } while (0)
#define HIP(call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(rc_)); return 1; } } while (0)

enum { STRIDE = 64, CAP = 16 };

int main(void) {
    dj_engine* e = NULL;
    dj_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.capacity_barriers = 1024;
    cfg.arena_bytes = 1u << 20;
    cfg.max_count = 8;
    cfg.payload_stride = STRIDE;
    cfg.max_batch = 1024;
    cfg.max_retry_publish = 3;
    cfg.max_retry_timeout_publish = 3;
    cfg.check_publish_delay_ms = 10000;
    if (dj_create(&cfg, &e) != 0) { fprintf(stderr, "dj_create: %s\n", dj_last_error(NULL)); return 1; }

    const uint64_t t0 = 1700000000000ull;
    const char* ids[2] = {"smoke~1", "late~1"};
    dj_fork_rec forks[2];
    memset(forks, 0, sizeof forks);
    for (int k = 0; k < 2; ++k) {
        dj_fingerprint(ids[k], strlen(ids[k]), &forks[k].fp_lo, &forks[k].fp_hi);
        forks[k].count = 3;
        forks[k].deadline_ms = t0 + 10000;            /* max-wait 10 s */
    }
    /* device buffers of the batch calls */
    dj_fork_rec* d_forks; dj_join_rec* d_joins; uint8_t* d_msgs; int32_t* d_status;
    dj_cont* d_conts; uint32_t* d_lens; uint8_t* d_rows;
    HIP(hipMalloc((void**)&d_forks, sizeof forks));
    HIP(hipMalloc((void**)&d_joins, 4 * sizeof(dj_join_rec)));
    HIP(hipMalloc((void**)&d_msgs, 4 * STRIDE));
    HIP(hipMalloc((void**)&d_status, CAP * sizeof(int32_t)));
    HIP(hipMalloc((void**)&d_conts, CAP * sizeof(dj_cont)));
    HIP(hipMalloc((void**)&d_lens, CAP * 8 * sizeof(uint32_t)));
    HIP(hipMalloc((void**)&d_rows, CAP * 8 * STRIDE));
    int32_t status[CAP];

    /* POST /v1/fork x 2 */
    HIP(hipMemcpy(d_forks, forks, sizeof forks, hipMemcpyHostToDevice));
    OK(dj_fork_batch(e, 2, d_forks, d_status));
    OK(dj_sync(e));
    HIP(hipMemcpy(status, d_status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (status[0] != DJ_FORK_CREATED || status[1] != DJ_FORK_CREATED) { fprintf(stderr, "fork statuses %d %d\n", status[0], status[1]); return 1; }

    /* POST /v1/join: the three parts of smoke~1 and part 1 of late~1, one batch */
    dj_join_rec joins[4];
    uint8_t msgs[4 * STRIDE];
    memset(joins, 0, sizeof joins);
    memset(msgs, 0, sizeof msgs);
    const char* texts[4] = {"part zero", "part one", "part two", "only part"};
    for (int k = 0; k < 4; ++k) {
        const dj_fork_rec* f = &forks[k < 3 ? 0 : 1];
        joins[k].fp_lo = f->fp_lo;
        joins[k].fp_hi = f->fp_hi;
        joins[k].position = k < 3 ? (uint32_t)k : 1u;
        joins[k].len = (uint32_t)strlen(texts[k]);
        memcpy(msgs + k * STRIDE, texts[k], joins[k].len);
    }
    HIP(hipMemcpy(d_joins, joins, sizeof joins, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_msgs, msgs, sizeof msgs, hipMemcpyHostToDevice));
    OK(dj_join_batch(e, 4, d_joins, d_msgs, STRIDE, t0 + 5, d_status));
    OK(dj_sync(e));
    HIP(hipMemcpy(status, d_status, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int32_t want[4] = {DJ_JOIN_WAITING, DJ_JOIN_WAITING, DJ_JOIN_COMPLETE, DJ_JOIN_WAITING};
    for (int k = 0; k < 4; ++k)
        if (status[k] != want[k]) { fprintf(stderr, "join %d answered %d, expected %d\n", k, status[k], want[k]); return 1; }

    /* the publisher's poll, twice: right away (smoke~1 is complete), and past the deadline (late~1 times out) */
    for (int poll = 0; poll < 2; ++poll) {
        const uint64_t now = poll == 0 ? t0 + 6 : t0 + 10001;
        dj_sweep_result res;
        OK(dj_sweep(e, now, CAP, CAP * 8, d_conts, d_lens, NULL, d_rows, NULL, &res));
        if (res.n_conts != 1 || res.more) { fprintf(stderr, "poll %d popped %u continuations\n", poll, res.n_conts); return 1; }
        dj_cont c;
        uint32_t lens[8];
        uint8_t rows[8 * STRIDE];
        HIP(hipMemcpy(&c, d_conts, sizeof c, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(lens, d_lens, c.count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(rows, d_rows, (size_t)c.count * STRIDE, hipMemcpyDeviceToHost));
        const int timeout = (c.flags & DJ_CONT_TIMEOUT) != 0;
        printf("%s -> %s destination:", ids[poll], timeout ? "timeout" : "normal");
        for (uint32_t p = 0; p < c.count; ++p) {
            if (lens[p] == DJ_NIL) printf(" nil");
            else printf(" \"%.*s\"", (int)lens[p], (const char*)rows + p * STRIDE);
        }
        printf("\n");
        if (timeout != poll || c.fp_lo != forks[poll].fp_lo || c.count != 3) { fprintf(stderr, "unexpected continuation\n"); return 1; }
        if (poll == 0 && (lens[0] != 9 || lens[2] != 8 || memcmp(rows + STRIDE, "part one", 8) != 0)) { fprintf(stderr, "wrong messages\n"); return 1; }
        if (poll == 1 && (lens[0] != DJ_NIL || lens[1] != 9 || lens[2] != DJ_NIL)) { fprintf(stderr, "wrong presence\n"); return 1; }
        /* the POST succeeded: complete */
        OK(dj_ack_batch(e, 1, NULL, d_conts, NULL, now, d_status));
        OK(dj_sync(e));
        HIP(hipMemcpy(status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (status[0] != DJ_ACK_DONE) { fprintf(stderr, "ack answered %d\n", status[0]); return 1; }
    }
    /* GET /v1/status of a barrier that is gone: no metadata */
    dj_cont meta;
    OK(dj_snapshot_one(e, forks[0].fp_lo, forks[0].fp_hi, 0, &meta, NULL, NULL, NULL));
    if (!(meta.flags & DJ_CONT_NO_METADATA)) { fprintf(stderr, "smoke~1 still has metadata\n"); return 1; }
    dj_destroy(e);
    printf("ok\n");
    return 0;
}
