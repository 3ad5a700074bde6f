"""The namespace distributed-join.redis on the HIP engine, function for function.

core.clj reaches barrier state only through the thirteen public functions of
reference src/distributed_join/redis.clj:90-300.  EngineRedis offers the same thirteen, with the same
arguments, return shapes and failures, each on ONE entry point of include/dj_engine.h — so the request
logic above it (core_unchanged.py, a function-by-function restatement of core.clj) runs as it stands.
It is the tested twin of clojure/src/distributed_join/redis.clj (the JNA binding of the same calls).

store.EngineStore is the other seam: one batch call per request, with the decisions core.clj takes
between two commands made on the device.  This one is the reference's own granularity, one key per call.

    redis.clj function                  engine entry point
    fork                 :90-104        dj_fork_batch (one record; it also queues the barrier at :timeout-secs,
                                        which core.clj:30 asks for in its next call)
    update-metadata      :106-111       dj_update_metadata_one
    join                 :143-159       dj_join_batch (one event, with dj_set_join_hlen: the status word carries HLEN;
                                        the queue update core.clj:52-58 asks for next is already applied)
    complete             :191-205       dj_complete_one
    all-joins            :242-243       dj_snapshot_one, update_ts = 1
    all-joins-status     :245-246       dj_snapshot_one, update_ts = 0
    join-metadata        :248-253       dj_snapshot_one (metadata only)
    store-in-publish-queue        :255-259   dj_store_in_publish_queue
    store-nearest-in-publish-queue :277-281  dj_store_nearest_in_publish_queue
    all-publish-messages :283-289       dj_pop_due
    get-unprocessed-publish-messages :291-293  dj_unprocessed
    get-publish-date     :295-297       dj_queue_scores
    shutdown             :299-300       dj_destroy

A deferred becomes a plain return value, an error deferred an exception.
"""
import numpy as np

from . import capi


class StoreError(Exception):
    """ex-info of a failed deferred; `data` is its ex-data map."""

    def __init__(self, message, data=None):
        super().__init__(message)
        self.data = dict(data or {})


def err(reason, message):
    """http-handling/err (http_handling.clj:14-19): what redis.clj:152-153,159 answer with."""
    return StoreError(message, {"reason": reason, "message": message})


def database_failure(cause):
    """redis.clj:17-21: an internal failure, HTTP 500 at the handler."""
    return StoreError("Database failure(s)", {"causes": [cause]})


def make_key(join_id, transaction_id):
    return f"{join_id}~{transaction_id}"                  # redis.clj:42-45


class EngineRedis:
    def __init__(self, engine, clock, env):
        self.e = engine
        self.e.set_join_hlen(True)      # redis/join answers HLEN (redis.clj:157): the status words carry it
        self.clock = clock              # time-util/now-milliseconds, time_util.clj:3-4
        self.env = env
        # the host half of "<id>:metadata": the strings of the Fork map (ids, URLs).  Count, :timeout-secs and
        # the retry counters live in the engine and are read from there.
        self.records = {}
        self.names = {}                 # fingerprint -> barrier-id of every id this host has queued or forked

    def _fp(self, barrier_id, register=False):
        fp = capi.fingerprint(barrier_id)
        if register:
            self.names[fp] = barrier_id
        return fp

    def _name(self, c):
        fp = (int(c["fp_lo"]), int(c["fp_hi"]))
        return self.names.get(fp, f"<unknown {fp[0]:016x}{fp[1]:016x}>")

    def _ordered_ids(self, conts):
        # ZRANGEBYSCORE order: score, then member bytes (redis.clj:286,293)
        rows = [(int(c["publish_at"]), self._name(c).encode()) for c in conts if not int(c["flags"]) & capi.CONT_VOID]
        return [m.decode() for _, m in sorted(rows)]

    # redis.clj:90-104
    def fork(self, record):
        key = make_key(record["join-id"], record["transaction-id"])
        lo, hi = self._fp(key, register=True)
        rec = np.zeros(1, dtype=capi.FORK_REC)
        rec[0]["fp_lo"], rec[0]["fp_hi"], rec[0]["count"], rec[0]["deadline_ms"] = lo, hi, record["count"], record["timeout-secs"]
        st = int(self.e.fork_batch(rec)[0])
        if st != capi.DJ_FORK_CREATED:
            raise database_failure(f"engine limit reached ({st})")
        self.records[key] = {k: v for k, v in record.items() if k not in ("publish-retry-cnt", "timeout-publish-retry-cnt")}
        return key

    # redis.clj:106-111
    def update_metadata(self, metadata):
        key = make_key(metadata.get("join-id"), metadata.get("transaction-id"))
        lo, hi = self._fp(key, register=True)
        st = self.e.update_metadata_one(lo, hi, metadata["count"], metadata["timeout-secs"],
                                        metadata.get("publish-retry-cnt"), metadata.get("timeout-publish-retry-cnt"))
        if st != capi.DJ_ACK_METADATA_SET:
            raise database_failure(f"metadata refused ({st})")
        self.records[key] = {k: v for k, v in metadata.items() if k not in ("publish-retry-cnt", "timeout-publish-retry-cnt")}
        return key + ":metadata"

    # redis.clj:143-159
    def join(self, record):
        bid = record["barrier-id"]
        lo, hi = self._fp(bid)
        msg = record["message"].encode()
        rec = np.zeros(1, dtype=capi.JOIN_REC)
        rec[0]["fp_lo"], rec[0]["fp_hi"], rec[0]["position"], rec[0]["len"] = lo, hi, min(record["position"], 0xFFFFFFFF), len(msg)
        rec[0]["max_wait_s"] = record.get("max-wait") or 0
        pitch = (min(max(len(msg), 1), self.e.max_message) + 15) // 16 * 16
        pay = np.zeros((1, pitch), dtype=np.uint8)
        if len(msg) <= pitch:
            pay[0, :len(msg)] = np.frombuffer(msg, dtype=np.uint8)
        word = int(self.e.join_batch(rec, pay, self.clock())[0])
        st, hlen = capi.join_status(word), capi.join_hlen(word)
        if st == capi.DJ_JOIN_NOT_FOUND:
            raise err("not-found", "barrier-id does not exist")               # redis.clj:152,159
        if st == capi.DJ_JOIN_BAD_POSITION:
            raise err("bad-data", "position is greater than total")           # redis.clj:153
        if st == capi.DJ_JOIN_TOO_LONG:
            raise err("bad-data", f"Message length cannot exceed {self.env['join-message-len-max']}")
        if st not in (capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE, capi.DJ_JOIN_NEAREST_NOT_QUEUED):
            raise database_failure(f"join status {st}")
        # :count is HLEN of the hash after this HSET (redis.clj:137,157), whatever the metadata's :count says
        return {"metadata": self.join_metadata(bid), "count": hlen}

    # redis.clj:191-205
    def complete(self, barrier_id):
        lo, hi = self._fp(barrier_id)
        st = self.e.complete_one(lo, hi)
        if st == capi.DJ_ACK_DIRTY:
            raise StoreError("Join was updated before fork could be completed", {"updated-join-before-complete": True})
        self.records.pop(barrier_id, None)
        self.names.pop((lo, hi), None)
        return 1

    def _all_joins(self, barrier_id, count, update_ts):
        lo, hi = self._fp(barrier_id)
        meta, lens, mws, pay = self.e.snapshot_one(lo, hi, update_ts)
        if count is None:                                                     # as-array, redis.clj:76-80
            return None
        out = []
        for p in range(count):
            if p >= len(lens) or int(lens[p]) == capi.DJ_NIL:
                out.append(None)
                continue
            j = {"barrier-id": barrier_id, "position": p, "message": bytes(pay[p, :int(lens[p])]).decode()}
            if int(mws[p]):
                j["max-wait"] = int(mws[p])
            out.append(j)
        return out

    # redis.clj:242-243
    def all_joins(self, barrier_id, count):
        return self._all_joins(barrier_id, count, True)

    # redis.clj:245-246
    def all_joins_status(self, barrier_id, count):
        return self._all_joins(barrier_id, count, False)

    # redis.clj:248-253
    def join_metadata(self, barrier_id):
        lo, hi = self._fp(barrier_id)
        meta = self.e.snapshot_one(lo, hi, False)[0]
        if int(meta["flags"]) & capi.DJ_CONT_NO_METADATA:
            return None
        md = dict(self.records.get(barrier_id, {}))
        md["count"] = int(meta["count"])
        md["timeout-secs"] = int(meta["deadline_ms"])
        for key, field in (("publish-retry-cnt", "retry_publish"), ("timeout-publish-retry-cnt", "retry_timeout")):
            if int(meta[field]) != capi.DJ_NIL:
                md[key] = int(meta[field])
        return md

    # redis.clj:255-259
    def store_in_publish_queue(self, barrier_id, publish_at):
        lo, hi = self._fp(barrier_id, register=True)
        self.e.store_in_publish_queue(lo, hi, publish_at)
        return [publish_at]

    # redis.clj:277-281
    def store_nearest_in_publish_queue(self, barrier_id, publish_at):
        lo, hi = self._fp(barrier_id)
        r = self.e.store_nearest_in_publish_queue(lo, hi, publish_at)
        if r < 0:
            # ZSCORE of an absent member is a nil bulk, Lua false; `false > string` raises (redis.clj:268-269)
            raise database_failure("attempt to compare boolean with string")
        return [publish_at, r]

    # redis.clj:283-289
    def all_publish_messages(self):
        now, conts = self.clock(), []
        while True:
            part, more = self.e.pop_due(now)
            conts += list(part)
            if not more:
                return self._ordered_ids(conts)

    # redis.clj:291-293
    def get_unprocessed_publish_messages(self):
        conts, more = self.e.unprocessed(self.clock())
        if more:
            raise database_failure("more unprocessed entries than the host buffer holds")
        return self._ordered_ids(conts)

    # redis.clj:295-297
    def get_publish_date(self, barrier_id):
        lo, hi = self._fp(barrier_id)
        at = self.e.queue_scores(lo, hi)[0]
        return None if at == capi.DJ_NOT_QUEUED else at

    # redis.clj:299-300
    def shutdown(self):
        self.e.close()
