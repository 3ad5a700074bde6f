/*
 * dj_engine.h — C ABI of the MI355X barrier engine.
 *
 * The engine replaces the state half of the reference service: the namespace
 * distributed-join.redis (reference src/distributed_join/redis.clj), which is the
 * only way src/distributed_join/core.clj touches barrier state.  One engine owns one
 * GPU: a device-resident open-addressing barrier table in HBM, per-barrier position
 * bitmaps and payload slabs, and the primary / backup publish queues as one queue
 * word per slot.  Every call is a batch; per-item results come back in
 * caller-provided arrays.
 *
 * Conventions
 *  - plain C, no exceptions across the boundary; every function returns 0 on
 *    success or a negative DJ_E_* code, and dj_last_error() gives the text.
 *  - pointers named d_* are DEVICE pointers (hipMalloc memory on the engine's
 *    device); pointers named h_* are host pointers.  The engine never frees or keeps
 *    caller buffers past the call's completion on its stream.
 *  - calls are asynchronous on the engine's HIP stream unless documented
 *    otherwise; dj_sync() waits for them.  Calls on one engine serialise themselves
 *    (one engine = one stream = one internal mutex), so threads may share an engine;
 *    the ORDER of their batches is then theirs to arrange (include/dj_batcher.h does).
 *    The stream is the engine's own (non-blocking: it does not wait for the null stream):
 *    a d_* input written by work on another stream must be complete before the call
 *    (wait for that stream, or make dj_stream() wait for an event recorded on it), and
 *    outputs are to be read after dj_sync() or from work ordered behind dj_stream().
 *  - the engine never reads a clock: now_ms is always an argument.  It must be a
 *    13-digit epoch-milliseconds value (1e12 <= now_ms < 9e12): inside that range
 *    the reference's Lua string comparison of scores (redis.clj:268-269) equals the
 *    numeric comparison the engine performs.
 *  - barriers are keyed by a 128-bit fingerprint of the barrier-id bytes
 *    "join-id~transaction-id" (redis.clj:42-47); dj_fingerprint() computes it.  The id
 *    strings and destination URLs stay on the host, keyed by fingerprint.
 */
#ifndef DJ_ENGINE_H
#define DJ_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dj_engine dj_engine;

/* ---- error codes (function return values) ------------------------------------ */
#define DJ_OK              0
#define DJ_E_INVALID      -1   /* bad argument / config */
#define DJ_E_HIP          -2   /* a HIP runtime call failed */
#define DJ_E_NOMEM        -3   /* device allocation failed */
#define DJ_E_CLOCK_RANGE  -4   /* now_ms outside [1e12, 9e12) */
#define DJ_E_TOO_LARGE    -5   /* batch larger than dj_config.max_batch */

/* "absent" marker for optional 32-bit fields (nil in the reference's maps) */
#define DJ_NIL 0xFFFFFFFFu
/* score of a barrier that is not in a queue */
#define DJ_NOT_QUEUED 0xFFFFFFFFFFFFFFFFull

/* ---- per-item status codes ----------------------------------------------------- */
/* dj_join_batch: what core/join (core.clj:44-58) answers for the event            */
#define DJ_JOIN_WAITING             1  /* 202 "...waiting on more chunks."          core.clj:58  */
#define DJ_JOIN_COMPLETE            2  /* 202 "...request will be completed."       core.clj:55  */
#define DJ_JOIN_NOT_FOUND           3  /* 404 "barrier-id does not exist"           redis.clj:152,159 */
#define DJ_JOIN_BAD_POSITION        4  /* 400 "position is greater than total"      redis.clj:153 */
#define DJ_JOIN_NEAREST_NOT_QUEUED  5  /* 500: join carried max-wait while the barrier was not in the
                                          primary queue; the Lua compare at redis.clj:269 raises, after
                                          the position was already stored                           */
#define DJ_JOIN_TOO_LONG            6  /* 400: message longer than dj_config.payload_stride
                                          (store.clj:24-26 with the engine's own limit)             */
#define DJ_JOIN_NO_SPACE            8  /* engine limit: the arena has no room for this long message; the position
                                          is stored with an empty message                           */
#define DJ_JOIN_BAD_MAX_WAIT        7  /* 400: max_wait_s >= 2^29 (store.clj:27-28 with the engine's own
                                          limit; the reference's configured limits are 500..6000 s)  */
/* With dj_set_join_hlen(e, 1) the status word of every event that was stored (WAITING, COMPLETE, NEAREST_NOT_QUEUED)
 * also carries what redis/join returns as :count (redis.clj:137,157): HLEN of the barrier's hash right after this
 * event's HSET, i.e. the number of positions that hold a message, whatever the metadata's :count says. */
#define DJ_JOIN_HLEN_SHIFT 8
#define DJ_JOIN_STATUS(word) ((int32_t)((word) & 0xFF))
#define DJ_JOIN_HLEN(word)   ((uint32_t)(word) >> DJ_JOIN_HLEN_SHIFT)
/* dj_fork_batch: core/fork (core.clj:24-33) */
#define DJ_FORK_CREATED             1  /* 201, barrier created or fully reset (redis.clj:90-104)    */
#define DJ_FORK_TABLE_FULL          2  /* engine limit: no free table slot                          */
#define DJ_FORK_NO_SPACE            3  /* engine limit: payload arena exhausted                     */
#define DJ_FORK_BAD_COUNT           4  /* count outside 1..dj_config.max_count (store.clj:32-34)    */
/* dj_ack_batch / dj_recover: what core/complete, reset-publish and attempt-retry-publish did      */
#define DJ_ACK_DONE                 1  /* redis/complete deleted the barrier        redis.clj:179-185 */
#define DJ_ACK_REQUEUED_DIRTY       2  /* a join arrived after the snapshot: reset-publish core.clj:113-117,127-129 */
#define DJ_ACK_RETRY                3  /* retry counter +1, re-queued at now        core.clj:148-152 */
#define DJ_ACK_DROPPED_EXHAUSTED    4  /* retries exhausted, barrier deleted        core.clj:142-147 */
#define DJ_ACK_DONE_NO_METADATA     5  /* metadata was missing: completed as failed core.clj:134-136,159-161 */
/* the single-key store commands below (dj_complete_one, dj_update_metadata_one) */
#define DJ_ACK_DIRTY                6  /* redis/complete alone: a join arrived after the snapshot; the Lua script
                                          answers its error and changes nothing             redis.clj:186-187 */
#define DJ_ACK_METADATA_SET         7  /* redis/update-metadata wrote the map               redis.clj:106-111 */

/* ---- records ------------------------------------------------------------------- */

typedef struct dj_config {
    uint32_t struct_size;               /* = sizeof(dj_config) */
    int32_t  device;                    /* HIP device ordinal */
    uint64_t capacity_barriers;         /* max live barriers, 1..2^29; table slots = next pow2 >= 2x.  The table keeps
                                           working beyond it (DJ_FORK_TABLE_FULL only when no slot is left), but
                                           lookups lengthen steeply past 1.8x: size it for the peak */
    uint64_t arena_bytes;               /* payload arena size in bytes */
    uint32_t max_count;                 /* :fork-count-max, <= 999 (config.edn:33) */
    uint32_t payload_stride;            /* bytes reserved per position (inline row); multiple of 64 */
    uint32_t max_batch;                 /* largest n accepted by one batch call, <= 2^24 */
    uint32_t max_retry_publish;         /* :max-retry-post-publish          (config.edn:24) */
    uint32_t max_retry_timeout_publish; /* :max-retry-post-timeout-publish  (config.edn:25) */
    uint32_t max_message_bytes;         /* longest message accepted (:join-message-len-max, config.edn:38), up to
                                           1 MiB; 0 or <= payload_stride: messages are limited to payload_stride.
                                           Messages longer than payload_stride live in blocks of their own
                                           (dj_long_ref below) */
    uint64_t check_publish_delay_ms;    /* :check-publish-delay-ms          (config.edn:16) */
} dj_config;

/* one /v1/fork after core/fork computed the deadline (core.clj:28): 32 bytes */
typedef struct dj_fork_rec {
    uint64_t fp_lo, fp_hi;      /* fingerprint of "join-id~transaction-id" */
    uint32_t count;             /* :count */
    uint32_t reserved;
    uint64_t deadline_ms;       /* :timeout-secs = now + 1000 * max-wait (absolute epoch ms) */
} dj_fork_rec;

/* one /v1/join (store.clj:19-28): 32-byte event header; the message bytes of event i
 * are at d_payloads + i * pitch */
typedef struct dj_join_rec {
    uint64_t fp_lo, fp_hi;      /* fingerprint of :barrier-id */
    uint32_t position;          /* :position */
    uint32_t len;               /* message length in bytes */
    uint32_t max_wait_s;        /* :max-wait in seconds, 0 = absent */
    uint32_t reserved;
} dj_join_rec;

#define DJ_CONT_TIMEOUT      1u /* now > :timeout-secs, strict (core.clj:86-89): goes to :timeout-destination */
#define DJ_CONT_NO_METADATA  2u /* id was queued but has no metadata (core.clj:159-161) */
#define DJ_CONT_VOID 0x80000000u /* unused output row: the barrier's positions did not fit under cap_positions, so it
                                    stays queued (dj_sweep_result.more is set).  Every other field of the row is 0.
                                    Readers skip it; dj_ack_batch answers it with status 0 */

/* one popped barrier = one continuation to publish: 64 bytes.  It doubles as the
 * metadata snapshot that core/publish-message holds while it publishes
 * (core.clj:175-177) and hands to attempt-retry-publish on failure (core.clj:167,170). */
typedef struct dj_cont {
    uint64_t fp_lo, fp_hi;
    uint64_t deadline_ms;       /* :timeout-secs of the snapshot */
    uint64_t publish_at;        /* the primary-queue score it was popped with */
    uint32_t count;             /* :count of the snapshot, 0 with DJ_CONT_NO_METADATA */
    uint32_t flags;             /* DJ_CONT_* */
    uint32_t retry_publish;     /* :publish-retry-cnt, DJ_NIL when absent */
    uint32_t retry_timeout;     /* :timeout-publish-retry-cnt, DJ_NIL when absent */
    uint64_t pos_base;          /* index of position 0 of this barrier in d_lens / d_max_waits;
                                   its message bytes start at d_payloads + pos_base * payload_stride */
    uint64_t reserved;
} dj_cont;

/* A message longer than payload_stride is not copied into the sweep's / snapshot's payload rows:
 * its row starts with this 16-byte reference instead (d_lens[...] still holds the true length, which
 * is how a reader tells: len > payload_stride).  dj_fetch_long copies the bytes out.  A reference
 * stays valid until the next call that can change the barrier (join of that position, ack, fork of
 * the same id, recover); fetch right after the sweep. */
typedef struct dj_long_ref {
    uint32_t unit;              /* arena offset in 64-byte units */
    uint32_t size_class;        /* the block holds 64 << size_class bytes */
    uint32_t len;               /* message length in bytes */
    uint32_t zero;
} dj_long_ref;

/* counters a sweep leaves behind (device memory, also copied to the host on request) */
typedef struct dj_sweep_result {
    uint32_t n_conts;           /* continuations written */
    uint32_t more;              /* 1: due barriers were left queued because an output cap was hit */
    uint64_t n_positions;       /* positions written (sum of counts) */
} dj_sweep_result;

/* ---- lifecycle ------------------------------------------------------------------ */
int  dj_create(const dj_config* cfg, dj_engine** out);
void dj_destroy(dj_engine* e);
const char* dj_last_error(const dj_engine* e);   /* e may be NULL: last create error */
int  dj_sync(dj_engine* e);                        /* wait for all queued work */
void* dj_stream(dj_engine* e);                     /* the engine's hipStream_t */
int  dj_get_config(const dj_engine* e, dj_config* out);   /* the configuration the engine was created with */

/* Fingerprint of a barrier-id (host side, pure function).  Never yields a reserved
 * value.  Replaces the Redis key strings built by make-key (redis.clj:42-47). */
void dj_fingerprint(const void* id, size_t len, uint64_t* fp_lo, uint64_t* fp_hi);
/* Same for n ids stored back to back; offsets has n+1 entries. */
void dj_fingerprint_batch(const uint8_t* bytes, const uint64_t* offsets, size_t n, uint64_t* fp_lo_hi);

/* ---- hot path (device pointers) ------------------------------------------------- */

/* core/fork for n barriers: redis/fork (redis.clj:90-104: set metadata, drop joins and both
 * join timestamps, remove from both queues) followed by redis/store-in-publish-queue at the
 * deadline (core.clj:30, redis.clj:255-259).  Records are applied in array order. */
int dj_fork_batch(dj_engine* e, uint32_t n, const dj_fork_rec* d_recs, int32_t* d_status);

/* core/join for n events, applied in array order with one timestamp: redis/join
 * (redis.clj:143-159, Lua add-join-and-count-them redis.clj:122-141), all-there?
 * (core.clj:36-38), then store-in-publish-queue at now (core.clj:54) or, for a join with
 * max-wait, store-nearest-in-publish-queue at now + 1000*max-wait (core.clj:40-42,
 * redis.clj:261-281).  pitch: bytes between consecutive events' messages, multiple of 16. */
int dj_join_batch(dj_engine* e, uint32_t n, const dj_join_rec* d_recs,
                  const uint8_t* d_payloads, uint64_t pitch, uint64_t now_ms, int32_t* d_status);

/* dj_join_batch for callers that must not allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results, the chain
 * form of dj_join_batch's kernels (never the bucketed one, whose key buffers belong to the engine); the per-batch work arrays (link words, destinations, event-to-slot map, list of
 * long chains, counters: 24 bytes per event) live in a device array of the library's code object instead of
 * memory the engine allocated, 1.5 MiB per GPU that stays until the process ends.  Limit: n <=
 * DJ_STATIC_JOIN_MAX events per call (DJ_E_TOO_LARGE beyond).  The call itself allocates and frees nothing
 * (no hipMalloc, hipFree or allocating library routine).  Every engine on one GPU shares that GPU's array:
 * two such calls that overlap on one GPU (two engines, i.e. two streams) race; calls on one engine are
 * ordered by its stream.  Barriers with very many events in one batch always take the serial kernel here. */
#define DJ_STATIC_JOIN_MAX 65536u
int dj_join_batch_static(dj_engine* e, uint32_t n, const dj_join_rec* d_recs,
                         const uint8_t* d_payloads, uint64_t pitch, uint64_t now_ms, int32_t* d_status);

/* redis/all-publish-messages (redis.clj:283-289): pop every barrier whose primary score is
 * <= now_ms; the backup entry stays.  Writes fp, publish_at (and the metadata snapshot) of at
 * most cap_conts barriers; d_result->more says whether some were left.  Order unspecified. */
int dj_pop_due(dj_engine* e, uint64_t now_ms, uint32_t cap_conts,
               dj_cont* d_conts, dj_sweep_result* d_result, dj_sweep_result* h_result);

/* The publisher's read side for one poll, core/process-publish-messages up to the POST
 * (core.clj:180-188): pop (as dj_pop_due), then per popped barrier join-metadata
 * (redis.clj:248-253), timeout classification (core.clj:86-95), and redis/all-joins
 * (redis.clj:210-243): join-ts-read := join-ts, messages gathered in position order.
 * Position p of continuation k: d_lens[pos_base+p] = message length or DJ_NIL when the
 * position is missing; d_max_waits[...] = its :max-wait (0 absent; may be NULL);
 * bytes at d_payloads + (pos_base+p) * payload_stride (the row may be zero-filled behind the message up
 * to the next 64-byte boundary).  A barrier is popped only if its
 * count still fits under cap_positions; one that does not may leave a DJ_CONT_VOID row among the
 * n_conts rows written.  h_result (optional) makes the call synchronous. */
int dj_sweep(dj_engine* e, uint64_t now_ms, uint32_t cap_conts, uint64_t cap_positions,
             dj_cont* d_conts, uint32_t* d_lens, uint32_t* d_max_waits, uint8_t* d_payloads,
             dj_sweep_result* d_result, dj_sweep_result* h_result);

/* Copies n long messages out of the arena: message i (d_refs[i], as found in a payload row) goes to
 * d_out + d_offsets[i] (16-byte aligned offsets; the caller sizes d_out from the lengths).  A
 * reference that does not point into the arena is skipped. */
int dj_fetch_long(dj_engine* e, uint32_t n, const dj_long_ref* d_refs, const uint64_t* d_offsets, uint8_t* d_out);

/* Outcome of publishing, for n (or *d_n when d_n != NULL, then n is the upper bound)
 * continuations previously returned by dj_sweep / dj_pop_due; d_ok[i] != 0 (or d_ok == NULL)
 * means the POST succeeded.
 *   ok   -> core/complete (core.clj:119-130): redis/complete (Lua complete-if-no-update,
 *           redis.clj:162-205) deletes the barrier unless a join arrived after the snapshot,
 *           in which case reset-publish (core.clj:113-117) re-queues it at now_ms.
 *   fail -> attempt-retry-publish with the snapshot (core.clj:132-152).
 * The same barrier must not appear twice in one call. */
int dj_ack_batch(dj_engine* e, uint32_t n, const uint32_t* d_n, const dj_cont* d_conts,
                 const uint8_t* d_ok, uint64_t now_ms, int32_t* d_status);

/* core/recover-unprocessed-publish-messages (core.clj:193-201): every backup entry with score
 * <= now_ms - check_publish_delay_ms (redis.clj:291-293) goes through the 2-arity
 * attempt-retry-publish (core.clj:153-155) with freshly read metadata.  Writes fp and outcome
 * of at most cap entries (d_conts[i].fp_*, d_status[i]); all due entries are processed. */
int dj_recover(dj_engine* e, uint64_t now_ms, uint32_t cap, dj_cont* d_conts, int32_t* d_status,
               dj_sweep_result* d_result, dj_sweep_result* h_result);

/* ---- single-key store functions (host pointers, synchronous, low rate) ----------- */

/* core/status read side (core.clj:60-68): join-metadata + all-joins-status (redis.clj:245-246,
 * no timestamp update), or with update_ts != 0 redis/all-joins (redis.clj:242-243).
 * Returns 0 and fills *h_meta (flags has DJ_CONT_NO_METADATA when the id has no metadata;
 * pos_base = 0); h_lens / h_max_waits need max_count entries, h_payloads max_count *
 * payload_stride bytes (any of the three may be NULL). */
int dj_snapshot_one(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, int update_ts,
                    dj_cont* h_meta, uint32_t* h_lens, uint32_t* h_max_waits, uint8_t* h_payloads);

/* redis/store-in-publish-queue (redis.clj:255-259) for one id, with or without metadata. */
int dj_store_in_publish_queue(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, uint64_t at_ms);
/* redis/get-publish-date (redis.clj:295-297) and the backup score: DJ_NOT_QUEUED when absent. */
int dj_queue_scores(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi,
                    uint64_t* h_primary_at, uint64_t* h_backup_at);

/* ---- the remaining commands of the store namespace, one key at a time ----------------------------
 * With these, every public function of distributed-join.redis (redis.clj:90-300) has an entry point of its own,
 * so a host can keep core.clj as it is and replace only the namespace behind it (the thirteen: fork,
 * update-metadata, join, complete, join-metadata, all-joins, all-joins-status, store-in-publish-queue,
 * store-nearest-in-publish-queue, all-publish-messages, get-unprocessed-publish-messages, get-publish-date, shutdown).
 * The batch calls above fuse what core.clj does between two of these commands; these do not. */

/* redis/complete (redis.clj:191-205, Lua complete-if-no-update redis.clj:162-189): deletes the barrier (metadata,
 * messages, both timestamps, both queue entries) when no join arrived since the last redis/all-joins, else
 * changes nothing.  *h_status = DJ_ACK_DONE or DJ_ACK_DIRTY (the error deferred that carries
 * :updated-join-before-complete, redis.clj:203-205). */
int dj_complete_one(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, int32_t* h_status);

/* redis/update-metadata (redis.clj:106-111): SET of the metadata map held in *h_meta (fp_lo, fp_hi, count,
 * deadline_ms = :timeout-secs, retry_publish, retry_timeout with DJ_NIL = key absent; the other fields are
 * ignored).  Creates the metadata when the barrier has none; stored messages and queue entries stay as they are
 * (a count that differs from the live one is handled as dj_ack_batch's failed publish handles it).
 * *h_status = DJ_ACK_METADATA_SET, or 0 when nothing was written (count outside 1..max_count, table full). */
int dj_update_metadata_one(dj_engine* e, const dj_cont* h_meta, int32_t* h_status);

/* redis/store-nearest-in-publish-queue (redis.clj:277-281, Lua redis.clj:261-275): when the id's primary-queue
 * score is later than at_ms, both queues' entries move to at_ms.  *h_result = 0 (the script's answer: ZADD adds no
 * new member either way), or -1 when the id is not in the primary queue (the script's comparison of a nil score
 * raises; the reference's caller gets an error deferred). */
int dj_store_nearest_in_publish_queue(dj_engine* e, uint64_t fp_lo, uint64_t fp_hi, uint64_t at_ms, int32_t* h_result);

/* redis/get-unprocessed-publish-messages (redis.clj:291-293): the members of the backup queue with a score of at
 * most now_ms - check_publish_delay_ms, at most cap of them (d_result->more tells whether some were left out),
 * each described by a dj_cont (fingerprint, metadata snapshot, publish_at = the backup score).  Read-only; order
 * unspecified (Redis' is by score, then member bytes).  h_result (optional) makes the call synchronous. */
int dj_unprocessed(dj_engine* e, uint64_t now_ms, uint32_t cap, dj_cont* d_conts,
                   dj_sweep_result* d_result, dj_sweep_result* h_result);

/* table occupancy, for monitoring and tests */
typedef struct dj_stats {
    uint64_t slots;             /* table slots */
    uint64_t live;              /* slots holding a barrier or an orphan queue entry */
    uint64_t tombstones;
    uint64_t arena_units_used;  /* 64-byte units handed out by the bump pointer */
    uint64_t arena_units_free;  /* 64-byte units sitting in the free lists */
} dj_stats;
int dj_get_stats(dj_engine* e, dj_stats* h_stats);

/* Table hygiene, no counterpart in the reference (Redis reclaims deleted keys itself): re-inserts
 * the live barriers into a spare table so that the tombstones of deleted barriers stop lengthening
 * lookups.  dj_fork_batch queues the same work on the stream by itself (without waiting) when the
 * table may be more than 80 % non-empty; this call waits.  *h_live (optional) = live slots. */
int dj_compact(dj_engine* e, uint64_t* h_live);

/* Durability.  The reference keeps all of this state in Redis and inherits Redis' persistence
 * (RDB / AOF; the service itself only holds a connection, redis.clj:13-15).  dj_checkpoint_save
 * waits for the stream and writes the whole barrier state (table, queue words, the used part of the
 * message arena, free lists) to a file; dj_checkpoint_load replaces the state of an engine created
 * with the same capacity_barriers, payload_stride and max_count (and an arena at least as large as
 * the saved one's used part) with the file's.  Outstanding continuations of the saved engine stay
 * valid on the restored one.  Both are synchronous and meant for quiet moments (shutdown, start,
 * periodic snapshots between batches). */
int dj_checkpoint_save(dj_engine* e, const char* path);
int dj_checkpoint_load(dj_engine* e, const char* path);

/* Barriers that receive very many events in one batch (hot keys, replays) are resolved by a
 * parallel set of kernels instead of a serial walk; the engine switches by itself from the longest
 * chain of the previous batch.  Both forms give identical results; this call pins the choice
 * (-1 automatic, 0 serial, 1 parallel) so tests can exercise each.  Likewise the per-thread replay of
 * short chains has a form for up to 8 events and one for up to 16, picked from the previous batch:
 * adding 2 to the mode (2 serial, 3 parallel) pins the second, 0 and 1 pin the first.  Messages that
 * fit a payload row are copied into the table by the first pass over the events itself when the serial
 * form runs; adding 4 (modes 4..7) keeps the separate copy pass that the other cases use.  The first two
 * passes over a join batch also have a bucketed form without an atomic per event, which the engine picks for
 * batches of 196608 to 1310720 events on tables of up to 2^22 slots whose previous batch showed no long chains: adding
 * 8 never runs it, adding 16 runs it wherever it applies (8 and 16 alone leave the other choices automatic).
 * Valid modes: -1 .. 23.  No reference counterpart. */
int dj_set_hot_path(dj_engine* e, int mode);

/* redis/join answers {:metadata m :count hlen} (redis.clj:137,157); core/join compares that HLEN with the metadata's
 * :count (all-there?, core.clj:36-38).  The batch calls make that comparison on the device and answer WAITING or
 * COMPLETE; a host that keeps core.clj as it is needs the number itself.  on = 1: every status word written by
 * dj_join_batch / dj_join_batch_static for a stored event is `status | HLEN << DJ_JOIN_HLEN_SHIFT` (read with
 * DJ_JOIN_STATUS / DJ_JOIN_HLEN; refused events keep their plain code).  HLEN can exceed :count after
 * redis/update-metadata wrote a smaller :count back (redis.clj:106-111).  Barriers with many events in one batch are
 * then always replayed event by event (the serial form), so leave it off (the default) where nobody reads the number. */
int dj_set_join_hlen(dj_engine* e, int on);

/* How a sweep finds what is due.  The reference pops a sorted set by score (redis.clj:283-289), which costs
 * what is due, not what is stored.  The engine has two forms with identical results: a scan of the queue
 * words (8 bytes per table slot), and a ready list that join batches append completed barriers to, with lower
 * bounds of every other queue date (deadlines, max-waits, re-queues): one for the table, which tells a sweep
 * whether anything but the list can be due, and one per region of 64 slots, which tells it where (it then
 * reads 8 bytes per region and the queue words of the regions that hold something due).  mode -1: by table
 * size (the list from 2^23 slots), 0: always scan, 1: keep the list.  No reference counterpart. */
int dj_set_sweep_path(dj_engine* e, int mode);

/* ---- per-kernel timing (measurement aid; no counterpart in the reference) ---------- */
/* Kernel ids for dj_profile_read.  With profiling enabled every launch of these kernels is
 * bracketed by a pair of HIP events on the engine's stream. */
enum {
    DJ_PROF_FORK_CLAIM = 0, DJ_PROF_FORK_APPLY, DJ_PROF_JOIN_LINK, DJ_PROF_JOIN_RESOLVE,
    DJ_PROF_JOIN_COMPLEX, DJ_PROF_JOIN_COPY, DJ_PROF_POP, DJ_PROF_GATHER, DJ_PROF_ACK,
    DJ_PROF_RECOVER, DJ_PROF_COMPACT, DJ_PROF_KERNELS
};
int dj_profile_enable(dj_engine* e, int on);
/* Waits for the stream; h_ms[DJ_PROF_KERNELS] = duration in ms of the most recent launch of
 * each kernel since the previous read, or -1 when it was not launched. */
int dj_profile_read(dj_engine* e, float* h_ms);

/* The streaming rate of this GPU's memory as the library's own plainest kernel sees it (16 bytes per lane copied from one
 * buffer of `bytes` bytes to another, `iterations` times after a warm-up; both buffers are allocated and freed by the
 * call, which waits): *h_gbps = (bytes read + bytes written) / time in GB/s.  Measurement aid, no reference counterpart. */
int dj_measure_copy(dj_engine* e, uint64_t bytes, uint32_t iterations, double* h_gbps);

/* ---- multi-GPU routing helper (device pointers) ---------------------------------- */

/* owner rank of a barrier among n_ranks engines */
uint32_t dj_owner_of(uint64_t fp_lo, uint64_t fp_hi, uint32_t n_ranks);

/* Stable partition of n join events by owner rank: d_counts[r] = events owned by r;
 * d_recs_out / d_payloads_out hold the events grouped by rank in array order inside each
 * group; d_origin[j] = input index of output event j (to route statuses back).
 * The exchange itself (all-to-all-v) is the caller's collective.  Any 32-byte record whose
 * first 16 bytes are the fingerprint can be routed (dj_fork_rec too); d_payloads may be NULL.
 * stream: hipStream_t to launch on, NULL = the engine's stream.  The partition reads no barrier
 * state, so a router may run it on its own stream while the engine applies the previous batch
 * (one partition in flight per stream). */
int dj_partition_joins(dj_engine* e, uint32_t n, uint32_t n_ranks,
                       const dj_join_rec* d_recs, const uint8_t* d_payloads, uint64_t pitch,
                       dj_join_rec* d_recs_out, uint8_t* d_payloads_out,
                       uint32_t* d_origin, uint32_t* d_counts, void* stream);

/* The same partition with ONE rank's group written elsewhere: row j of self_rank's group goes to d_recs_self[j] (and
 * d_payloads_self + j * pitch) instead of its place in d_recs_out; all other groups, d_counts and d_origin are as above
 * (d_origin is indexed as if the group had stayed in d_recs_out, whose rows for it are left untouched).  The router
 * uses it to let a rank's own bucket land directly where its engine reads it: no copy through the exchange. */
int dj_partition_joins_self(dj_engine* e, uint32_t n, uint32_t n_ranks, uint32_t self_rank,
                            const dj_join_rec* d_recs, const uint8_t* d_payloads, uint64_t pitch,
                            dj_join_rec* d_recs_out, uint8_t* d_payloads_out,
                            dj_join_rec* d_recs_self, uint8_t* d_payloads_self,
                            uint32_t* d_origin, uint32_t* d_counts, void* stream);

/* The inverse of the partition's move, for per-record answers: d_out[d_origin[j]] = d_sorted[j] for j < n, where
 * d_sorted[j] answers the record that dj_partition_joins put at place j.  Every origin index must be below the n of
 * that partition call (they are, when d_origin is what it wrote).  stream as in dj_partition_joins. */
int dj_scatter_statuses(dj_engine* e, uint32_t n, const uint32_t* d_origin, const int32_t* d_sorted, int32_t* d_out, void* stream);

/* ---- the exchange itself (no counterpart in the reference, whose replicas share one Redis:
 * redis.clj:13-15) ------------------------------------------------------------------------------
 *
 * The plan of one batch's exchange, a pure function every rank evaluates on the same matrix
 * (matrix[s * n_ranks + r] = records rank s holds for rank r, gathered from all ranks): a receiver's
 * incoming records, taken in source-rank order, are cut into pieces of at most max_recv rows, one piece per
 * round.  Returns the number of rounds (the same on every rank, at least 1) and, for round `round`, what rank
 * `rank` sends (send_cnt[r] records of its bucket for r, starting send_off[r] records into that bucket) and
 * receives (recv_cnt[s] records from s); any of the three arrays may be NULL.  A bucket that is larger than a
 * receiver's buffer (hot keys) thus costs more rounds, never an error on one rank while the others wait in a
 * collective.  Applying the pieces one after the other is the same as applying the whole stream: a batch is
 * defined as its records taken one by one. */
uint32_t dj_route_plan(uint32_t n_ranks, uint32_t rank, const uint32_t* matrix, uint64_t max_recv, uint32_t round,
                       uint32_t* send_off, uint32_t* send_cnt, uint32_t* recv_cnt);

/* A router owns two streams of its own (one for the partition kernels, one for everything that goes through
 * RCCL), `depth` sets of send/receive buffers and an RCCL communicator over the
 * n_ranks engines of one node (librccl is bound at run time; only these calls need it).  Rank 0 makes an id
 * with dj_route_unique_id and hands the 128 bytes to every rank by whatever channel the host has; every rank
 * then calls dj_router_create (a collective).  One batch:
 *   dj_route_start   partition + all-gather of the bucket sizes, queued on the router's streams; the host
 *                    does not wait.  Start the next batches while the engine applies this one.
 *   dj_route_advance optional: queues the exchange of the batch's FIRST piece now, without making the engine's
 *                    stream wait for it, so that batch k+1's rows cross the links while the engine applies batch
 *                    k and batch k+2 is partitioned (call it for k+1 once batch k's engine calls are queued, and
 *                    before the dj_route_start of k+2, whose all-gather would otherwise stand in front of it on
 *                    the communication stream).  The host waits for the bucket sizes as in dj_route_next.
 *   dj_route_next    exchanges the next piece (grouped ncclSend/ncclRecv of records, then messages) unless
 *                    dj_route_advance already has, and makes
 *                    the ENGINE's stream wait for its arrival; the host waits only for the bucket sizes, which
 *                    are in when the start was issued a batch ahead.  Returns device pointers to this rank's
 *                    records / messages, their number and whether more pieces follow (skew beyond max_recv).
 *                    Queue the engine call that consumes a piece before asking for the next one.
 *   dj_route_release after the last piece's engine call has been queued: the set may be reused.
 * Every rank makes the same calls in the same order, and all ranks route a batch either with messages or without
 * (d_payloads NULL on all or on none: a disagreement is reported by every rank's first dj_route_next / dj_route_advance).  A piece's record order: source rank 0's records in
 * array order, then rank 1's, ...: a fixed global order, so results do not depend on timing. */
typedef struct dj_router dj_router;
int  dj_route_unique_id(uint8_t* id128);
int  dj_router_create(dj_engine* e, uint32_t n_ranks, uint32_t rank, const uint8_t* id128, uint32_t max_send,
                      uint32_t max_recv, uint64_t pitch, uint32_t depth, dj_router** out);
void dj_router_destroy(dj_router* r);
const char* dj_router_last_error(const dj_router* r);     /* r may be NULL: last create / id error of this thread */
int  dj_route_start(dj_router* r, uint32_t n, const dj_join_rec* d_recs, const uint8_t* d_payloads, int* handle);
int  dj_route_advance(dj_router* r, int handle);
int  dj_route_next(dj_router* r, int handle, const dj_join_rec** d_recs_out, const uint8_t** d_payloads_out,
                   uint32_t* n_out, int* more);
int  dj_route_release(dj_router* r, int handle);

/* The way back.  The reference answers every /v1/join and /v1/fork to the client that sent it (ring.clj:54-69: 202 with
 * one of two texts core.clj:55,58; 404 / 400 redis.clj:152-153,159; 201 core.clj:32); with barriers sharded, the answer
 * is computed on the owner's GPU and must return to the rank that holds the request.  Per piece:
 *   dj_route_status_buffer  where the engine call that applies the piece handed out last writes its statuses (max_recv
 *                           words of the handle's buffer set; pass it as d_status of dj_join_batch / dj_fork_batch).
 *   dj_route_return         after that engine call has been queued: sends every status of the piece to the rank its record
 *                           came from (4 bytes per record, one grouped send / receive over the same plan read backwards, on
 *                           the router's communication stream behind an event on the engine's stream; the host does not
 *                           wait).  A collective like dj_route_next: every rank calls it once per piece, in order, before
 *                           asking for the next piece.
 * and once per batch, after the last piece's dj_route_return and before dj_route_release:
 *   dj_route_statuses       d_status_out[i] = the answer to record i of the batch this rank gave dj_route_start (n words,
 *                           this rank's own submission order; device memory).  `stream`: a hipStream_t that is made to wait
 *                           for the answers (read them from work ordered behind it), or NULL: the call itself waits. */
int  dj_route_status_buffer(dj_router* r, int handle, int32_t** d_status_piece);
int  dj_route_return(dj_router* r, int handle);
int  dj_route_statuses(dj_router* r, int handle, int32_t* d_status_out, void* stream);


/* One word from every rank to every rank's host: h_words[s] = the word rank s passed (n_ranks entries).  A collective
 * that waits: the hosts of a node use it to agree on what the next exchanges are (how many batches each has staged,
 * whether all want to stop) before any of them enters one — see dj_batcher_create_routed in dj_batcher.h. */
int  dj_route_agree(dj_router* r, uint32_t word, uint32_t* h_words);
/* What the router was created with (any pointer may be NULL). */
int  dj_router_get_limits(const dj_router* r, uint32_t* n_ranks, uint32_t* rank, uint32_t* max_send, uint32_t* max_recv, uint64_t* pitch);

#ifdef __cplusplus
}
#endif
#endif /* DJ_ENGINE_H */
