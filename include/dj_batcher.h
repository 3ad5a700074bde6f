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
/* The batcher of ONE RANK of a node whose barriers are sharded over several engines (one engine = one GPU, owner =
 * dj_owner_of): requests enter at any rank, the rank's windows are exchanged through its router (dj_router_create,
 * include/dj_engine.h), the owner's engine applies every record, and the record's status returns to the rank that
 * holds the request (dj_route_return / dj_route_statuses) and to the caller's callback — what the reference's handlers
 * answer per request (ring.clj:54-69; 404 / 400 redis.clj:152-153,159) whichever replica took it.  Every rank of the
 * router's communicator creates one; their drain threads then run the same cycles: an agreement (dj_route_agree: how many
 * batches of forks and of joins each rank has staged, whether all want to stop), then as many routed fork batches and
 * routed join batches as the busiest rank has, empty ones where a rank has run out.  A rank with nothing staged joins
 * the next agreement after max_wait_us (100 us when that is 0): a request waits at most that long for idle peers.
 * Requirements: router pitch = max_message rounded up to 16 (1 -> 16), router max_send >= max_batch, the engine's
 * max_batch >= the router's max_recv.  dj_batcher_destroy is a collective here: a rank's drain thread leaves only when
 * every rank has been asked to stop and has nothing staged.  Everything else (dj_batcher_join / _fork / _run, the
 * clock) works as on a plain batcher; dj_batcher_run runs on this rank's engine (sweeps, acks and recovers are
 * GPU-local). */
int dj_batcher_create_routed(dj_engine* e, dj_router* router, const dj_batcher_config* cfg, dj_batcher** out);
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
