/*
 * dj_batcher.h — per-request front of the barrier engine (C ABI).
 *
 * The reference's handlers work one request at a time: POST /v1/fork and POST /v1/join each call
 * core/fork / core/join once (reference src/distributed_join/ring.clj:54-72, core.clj:24-58), from
 * as many server threads as there are connections.  The engine works on batches.  The batcher sits
 * between the two: request threads hand over ONE record (and its message) and get the record's
 * status back through a callback (or by waiting); one drain thread collects what has arrived,
 * copies it to the device from pinned staging buffers and applies it with dj_fork_batch /
 * dj_join_batch.  Requests of one batch are applied in arrival order, which is one valid
 * serialisation of concurrent HTTP requests; a request is answered only after its batch has been
 * applied, so "fork answered 201, then the client joins" keeps its order.
 *
 * The drain thread is also the engine's single caller: everything else that touches the engine
 * (the publisher's sweep / ack, recover, status lookups, checkpoints) goes through
 * dj_batcher_run(), which runs a function on that thread between two batches.
 *
 * No counterpart in the reference (Redis serialises commands itself, redis.clj:13-21).
 */
#ifndef DJ_BATCHER_H
#define DJ_BATCHER_H

#include "dj_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dj_batcher dj_batcher;

/* Called on the drain thread with the DJ_JOIN_* / DJ_FORK_* status of one request.  Keep it short
 * (complete a future, push to a queue); it must not call back into the batcher's blocking calls. */
typedef void (*dj_status_cb)(void* user, int32_t status);

typedef struct dj_batcher_config {
    uint32_t struct_size;       /* sizeof(dj_batcher_config) */
    uint32_t max_batch;         /* requests per kind and batch, 1..dj_config.max_batch of the engine */
    uint32_t max_wait_us;       /* a batch is closed at the latest this long after its first request */
    uint32_t max_message;       /* longest message in bytes the staging buffers hold (<= the engine's limit) */
} dj_batcher_config;

typedef struct dj_batcher_stats {
    uint64_t batches;           /* drain cycles that applied something */
    uint64_t forks, joins;      /* requests applied */
    uint64_t largest_batch;     /* most requests of one kind in one cycle */
    uint64_t tasks;             /* dj_batcher_run calls executed */
} dj_batcher_stats;

/* The engine must outlive the batcher.  Other threads may still call the engine directly (its entry
 * points serialise themselves), but such a call lands between any two calls of the drain thread,
 * e.g. between the forks and the joins of one batch; dj_batcher_run() places it between two batches. */
int dj_batcher_create(dj_engine* e, const dj_batcher_config* cfg, dj_batcher** out);
/* Applies what is still queued, answers it, stops the thread. */
void dj_batcher_destroy(dj_batcher* b);

/* Time source of the batches (the reference reads System/currentTimeMillis per request,
 * core.clj:15); the default is the wall clock in epoch milliseconds.  Tests install a virtual one. */
int dj_batcher_set_clock(dj_batcher* b, uint64_t (*now_ms)(void* user), void* user);

/* Thread-safe, return as soon as the request is staged (they wait only while a batch is full).
 * message: rec->len bytes (may be NULL when len is 0).  A message longer than max_message is
 * answered DJ_JOIN_TOO_LONG without reaching the engine.  Returns DJ_OK or DJ_E_INVALID. */
int dj_batcher_join(dj_batcher* b, const dj_join_rec* rec, const void* message, dj_status_cb cb, void* user);
int dj_batcher_fork(dj_batcher* b, const dj_fork_rec* rec, dj_status_cb cb, void* user);

/* Same, but wait for the answer. */
int dj_batcher_join_wait(dj_batcher* b, const dj_join_rec* rec, const void* message, int32_t* status);
int dj_batcher_fork_wait(dj_batcher* b, const dj_fork_rec* rec, int32_t* status);

/* Runs fn(user, now_ms) on the drain thread between two batches (the engine is idle and
 * synchronised at that moment; now_ms is the batcher's clock) and waits for it.  Returns fn's
 * return value. */
int dj_batcher_run(dj_batcher* b, int (*fn)(void* user, dj_engine* e, uint64_t now_ms), void* user);

int dj_batcher_get_stats(dj_batcher* b, dj_batcher_stats* out);

#ifdef __cplusplus
}
#endif
#endif
