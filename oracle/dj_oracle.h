/*
 * dj_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  The oracle is a sequential, string-keyed CPU
 * restatement of the reference's barrier logic (src/distributed_join/core.clj) and of the
 * Redis commands and Lua scripts it drives (src/distributed_join/redis.clj).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker.  The engine in distributed_join_amd/ never links or calls it.
 *
 * Parity pinning: the reference (Clojure + an external Redis server) cannot be built or
 * run in this environment, so the oracle is pinned by the known-answer assertions the
 * reference's own tests make (test/distributed_join/publish_or_timeout_test.clj,
 * test/distributed_join/ring_test.clj, fork-join-test.sh), transcribed into
 * tests/golden/scenarios.json with a virtual clock.
 *
 * Ids are passed by index into a table registered with djo_set_ids, so that large event
 * streams cross the ctypes boundary as flat arrays.
 */
#ifndef DJ_ORACLE_H
#define DJ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct djo djo;

/* same numeric values as the engine's DJ_* codes, restated here so the oracle has no
 * dependency on product headers */
#define DJO_NIL 0xFFFFFFFFu
#define DJO_NOT_QUEUED 0xFFFFFFFFFFFFFFFFull
#define DJO_JOIN_WAITING 1
#define DJO_JOIN_COMPLETE 2
#define DJO_JOIN_NOT_FOUND 3
#define DJO_JOIN_BAD_POSITION 4
#define DJO_JOIN_NEAREST_NOT_QUEUED 5
#define DJO_JOIN_TOO_LONG 6
#define DJO_FORK_CREATED 1
#define DJO_FORK_BAD_COUNT 4
#define DJO_ACK_DONE 1
#define DJO_ACK_REQUEUED_DIRTY 2
#define DJO_ACK_RETRY 3
#define DJO_ACK_DROPPED_EXHAUSTED 4
#define DJO_ACK_DONE_NO_METADATA 5
#define DJO_CONT_TIMEOUT 1u
#define DJO_CONT_NO_METADATA 2u

typedef struct djo_cont {
    uint32_t id_idx;
    uint32_t count;
    uint32_t flags;
    uint32_t retry_publish;
    uint32_t retry_timeout;
    uint32_t pad;
    uint64_t deadline_ms;
    uint64_t publish_at;
    uint64_t pos_base;
} djo_cont;

djo* djo_create(uint64_t check_publish_delay_ms, uint32_t max_retry_publish,
                uint32_t max_retry_timeout_publish, uint32_t max_count, uint32_t max_message_len);
void djo_destroy(djo* o);
void djo_set_ids(djo* o, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_ids);

void djo_fork_batch(djo* o, uint32_t n, const uint32_t* id_idx, const uint32_t* count,
                    const uint64_t* deadline_ms, int32_t* status);
void djo_join_batch(djo* o, uint32_t n, const uint32_t* id_idx, const uint32_t* position,
                    const uint32_t* len, const uint32_t* max_wait_s, const uint8_t* payloads,
                    uint64_t pitch, uint64_t now_ms, int32_t* status);
/* returns number popped, or -1 if cap is too small (nothing is modified then) */
int64_t djo_pop_due(djo* o, uint64_t now_ms, uint32_t cap, djo_cont* out);
int64_t djo_sweep(djo* o, uint64_t now_ms, uint32_t cap_conts, uint64_t cap_positions,
                  djo_cont* conts, uint32_t* lens, uint32_t* max_waits, uint8_t* payloads,
                  uint64_t stride, uint64_t* n_positions);
void djo_ack_batch(djo* o, uint32_t n, const djo_cont* conts, const uint8_t* ok,
                   uint64_t now_ms, int32_t* status);
int64_t djo_recover(djo* o, uint64_t now_ms, uint32_t cap, djo_cont* out, int32_t* status);
int djo_snapshot_one(djo* o, uint32_t id_idx, int update_ts, djo_cont* meta, uint32_t* lens,
                     uint32_t* max_waits, uint8_t* payloads, uint64_t stride);
void djo_store_in_publish_queue(djo* o, uint32_t id_idx, uint64_t at_ms);
void djo_queue_scores(djo* o, uint32_t id_idx, uint64_t* primary_at, uint64_t* backup_at);
uint64_t djo_live_barriers(djo* o);

/* The public functions of redis.clj one by one (no core.clj logic around them): what a host that keeps
 * core.clj and replaces only the namespace calls.  redis/join-metadata, all-joins, all-joins-status,
 * store-in-publish-queue and get-publish-date are djo_snapshot_one, djo_store_in_publish_queue and
 * djo_queue_scores above. */
void djo_redis_fork(djo* o, uint32_t id_idx, uint32_t count, uint64_t timeout_secs);                  /* redis.clj:90-104 */
void djo_redis_update_metadata(djo* o, uint32_t id_idx, uint32_t count, uint64_t timeout_secs,
                               uint32_t retry_publish, uint32_t retry_timeout);                      /* redis.clj:106-111 */
/* 0 with *hlen = HLEN, or DJO_JOIN_NOT_FOUND / DJO_JOIN_BAD_POSITION                                   redis.clj:143-159 */
int djo_redis_join(djo* o, uint32_t id_idx, uint32_t position, const uint8_t* message, uint32_t len,
                   uint32_t max_wait_s, uint64_t now_ms, uint32_t* hlen);
int djo_redis_complete(djo* o, uint32_t id_idx);            /* 1 deleted, 0 "Join Updated" error        redis.clj:191-205 */
int djo_redis_store_nearest_in_publish_queue(djo* o, uint32_t id_idx, uint64_t publish_at);   /* 1, or 0: the script raised, redis.clj:261-281 */
/* ids in (score, member) order; -1 when cap is too small (nothing popped then)                         redis.clj:283-293 */
int64_t djo_redis_all_publish_messages(djo* o, uint64_t now_ms, uint32_t cap, uint32_t* id_idx);
int64_t djo_redis_get_unprocessed_publish_messages(djo* o, uint64_t now_ms, uint32_t cap, uint32_t* id_idx);

#ifdef __cplusplus
}
#endif
#endif
