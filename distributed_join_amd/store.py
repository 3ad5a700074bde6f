"""Host-side store layer.

Two things live here:
  * the request limits of the reference's schemas (reference src/distributed_join/store.clj:9-77),
    restated as plain validation functions with the limits of config/prod/config.edn:33-39;
  * EngineStore, the string-keyed adapter over Engine that plays the role of the namespace
    distributed-join.redis for core.py: it maps barrier-id strings to 128-bit fingerprints,
    packs requests into the C ABI's records and unpacks continuations.
"""
from dataclasses import dataclass, field
from typing import List, Optional
from urllib.parse import urlsplit

import numpy as np

from . import capi

# config/prod/config.edn:13-14,33-39
DEFAULT_ENV = {
    "fork-max-wait": 6000,
    "join-max-wait": 6000,
    "check-publish-delay-ms": 10000,
    "publish-all-complete-timeout-ms": 30000,
    "max-retry-post-publish": 3,
    "max-retry-post-timeout-publish": 3,
    "fork-count-max": 999,
    "join-id-len-max": 255,
    "transaction-id-len-max": 255,
    "destination-url-len-max": 2049,
    "timeout-destination-url-len-max": 2049,
    "join-message-len-max": 262144,
    "barrier-id-len-max": 511,
}


def clojure_count(s):
    """(count s) of a java.lang.String: UTF-16 code units (a character beyond the BMP counts as two)."""
    return len(s.encode("utf-16-le")) // 2


def engine_message_bytes(env=DEFAULT_ENV):
    """dj_config.max_message_bytes for an engine behind these limits: the reference limits a message's
    CHARACTERS (store.clj:24-26), the engine stores UTF-8 bytes, at most three per UTF-16 code unit."""
    return min(3 * env["join-message-len-max"], 1 << 20)


def _is_int(v):
    return isinstance(v, int) and not isinstance(v, bool)


def _validate_id(v, invalid_chars, max_len):
    # store.clj:9-12: not blank, none of the invalid characters, length limit
    return isinstance(v, str) and v.strip() != "" and not any(c in v for c in invalid_chars) and len(v) <= max_len


def _validate_url(v, max_len):
    # store.clj:36-46: length limit, parses as a URL, protocol http/https, non-empty host
    if not isinstance(v, str) or len(v) > max_len:
        return False
    try:
        u = urlsplit(v)
    except ValueError:
        return False
    return u.scheme.lower() in ("http", "https") and bool(u.hostname)


def validate_fork(req, env=DEFAULT_ENV):
    """None when req satisfies the Fork schema (store.clj:53-69), else the error text."""
    if not isinstance(req, dict):
        return "body must be a map"
    allowed = {"join-id", "transaction-id", "count", "destination", "timeout-destination", "max-wait"}
    if set(req) - allowed:
        return "disallowed key"
    if not _validate_id(req.get("join-id"), "~:", env["join-id-len-max"]):
        return f"Join id must have a value, not exceed {env['join-id-len-max']} characters in length and not contain [~:]"
    if not _validate_id(req.get("transaction-id"), "~:", env["transaction-id-len-max"]):
        return (f"Transaction id must have a value, not exceed {env['transaction-id-len-max']} characters in length "
                "and not contain [~:]")
    c = req.get("count")
    if not _is_int(c) or c < 1 or c > env["fork-count-max"]:
        return f"Count must be positive and less than or equal to {env['fork-count-max']}"
    if not _validate_url(req.get("destination"), env["destination-url-len-max"]):
        return f"Destination does not contain a valid HTTP/HTTPS URL or is longer than {env['destination-url-len-max']}"
    if "timeout-destination" in req and not _validate_url(req["timeout-destination"], env["timeout-destination-url-len-max"]):
        return ("Timeout destination does not contain a valid HTTP/HTTPS URL or is longer than "
                f"{env['timeout-destination-url-len-max']}")
    if "max-wait" in req:
        mw = req["max-wait"]
        if not _is_int(mw) or mw <= 0 or mw > env["fork-max-wait"]:
            return f"max-wait must be positive and less than or equal to {env['fork-max-wait']}"
    return None


def validate_join(req, env=DEFAULT_ENV):
    """None when req satisfies the Join schema (store.clj:19-28), else the error text."""
    if not isinstance(req, dict):
        return "body must be a map"
    if set(req) - {"barrier-id", "position", "message", "max-wait"}:
        return "disallowed key"
    if not _validate_id(req.get("barrier-id"), ":", env["barrier-id-len-max"]):
        return f"Barrier id must have a value, not exceed {env['barrier-id-len-max']} characters in length and not contain :"
    p = req.get("position")
    if not _is_int(p) or p < 0:
        return "position must be a non-negative integer"
    m = req.get("message")
    if not isinstance(m, str) or clojure_count(m) > env["join-message-len-max"]:           # (count %), store.clj:25
        return f"Message length cannot exceed {env['join-message-len-max']}"
    if "max-wait" in req:
        mw = req["max-wait"]
        if not _is_int(mw) or mw <= 0 or mw >= env["join-max-wait"]:   # store.clj:27: strictly below
            return f"max-wait must be positive and less than or equal to {env['join-max-wait']}"
    return None


@dataclass
class Continuation:
    """One popped barrier: what core/publish-message holds while it publishes (core.clj:173-178)."""
    barrier_id: str
    count: int
    timeout: bool
    no_metadata: bool
    deadline_ms: int
    publish_at: int
    retry_publish: Optional[int]
    retry_timeout: Optional[int]
    messages: List[Optional[bytes]] = field(default_factory=list)
    max_waits: List[Optional[int]] = field(default_factory=list)
    raw: object = None      # the record to hand back to ack()


def _opt(v):
    return None if int(v) == capi.DJ_NIL else int(v)


class EngineStore:
    """String-keyed view of an Engine (the role of distributed-join.redis for core.py)."""

    def __init__(self, engine):
        self.e = engine
        self.names = {}          # (fp_lo, fp_hi) -> barrier-id

    def _fp(self, barrier_id, register=False):
        """Fingerprint of an id.  Only forks (and explicit queue entries) register the name: joins and status
        lookups of ids nobody forked must not grow the table, and a barrier's name goes when the barrier does."""
        fp = capi.fingerprint(barrier_id)
        if register:
            self.names[fp] = barrier_id
        return fp

    def _forget(self, conts, statuses):
        gone = (capi.DJ_ACK_DONE, capi.DJ_ACK_DROPPED_EXHAUSTED, capi.DJ_ACK_DONE_NO_METADATA)
        for c, st in zip(conts, statuses):
            if int(st) in gone:
                self.names.pop(capi.fingerprint(c.barrier_id), None)

    def _name(self, lo, hi):
        return self.names.get((int(lo), int(hi)), f"<unknown {int(lo):016x}{int(hi):016x}>")

    # redis/fork + store-in-publish-queue (core.clj:29-30)
    def fork_many(self, ids, counts, deadlines):
        recs = np.zeros(len(ids), dtype=capi.FORK_REC)
        for k, i in enumerate(ids):
            recs[k]["fp_lo"], recs[k]["fp_hi"] = self._fp(i, register=True)
        recs["count"] = np.asarray(counts, dtype=np.int64).clip(0, 0xFFFFFFFF)
        recs["deadline_ms"] = deadlines
        return self.e.fork_batch(recs)

    def fork(self, barrier_id, count, deadline_ms):
        return int(self.fork_many([barrier_id], [count], [deadline_ms])[0])

    # redis/join + queue update (core.clj:49-58)
    def join_many(self, ids, positions, messages, max_waits, now_ms):
        n = len(ids)
        recs = np.zeros(n, dtype=capi.JOIN_REC)
        longest = max([len(m) for m in messages] + [1])
        pitch = (min(longest, self.e.max_message) + 15) // 16 * 16
        pay = np.zeros((n, pitch), dtype=np.uint8)
        for k in range(n):
            recs[k]["fp_lo"], recs[k]["fp_hi"] = self._fp(ids[k])
            m = bytes(messages[k])
            recs[k]["len"] = len(m)
            if len(m) <= pitch:
                pay[k, :len(m)] = np.frombuffer(m, dtype=np.uint8)
        recs["position"] = np.asarray(positions, dtype=np.int64).clip(0, 0xFFFFFFFF)
        recs["max_wait_s"] = [0 if w is None else w for w in max_waits]
        return self.e.join_batch(recs, pay, now_ms)

    def join(self, barrier_id, position, message, max_wait, now_ms):
        return int(self.join_many([barrier_id], [position], [message], [max_wait], now_ms)[0])

    def _conts(self, conts, lens=None, mws=None, pay=None):
        out = []
        for c in conts:
            if int(c["flags"]) & capi.CONT_VOID:
                continue
            k = Continuation(self._name(c["fp_lo"], c["fp_hi"]), int(c["count"]),
                             bool(int(c["flags"]) & capi.DJ_CONT_TIMEOUT),
                             bool(int(c["flags"]) & capi.DJ_CONT_NO_METADATA), int(c["deadline_ms"]),
                             int(c["publish_at"]), _opt(c["retry_publish"]), _opt(c["retry_timeout"]), raw=c.copy())
            if lens is not None and not k.no_metadata:
                b = int(c["pos_base"])
                for p in range(k.count):
                    ln = int(lens[b + p])
                    if ln == capi.DJ_NIL:
                        k.messages.append(None)
                        k.max_waits.append(None)
                    else:
                        k.messages.append(bytes(pay[b + p, :ln]))
                        k.max_waits.append(int(mws[b + p]) or None)
            out.append(k)
        return out

    @staticmethod
    def _redis_order(conts):
        """ZRANGEBYSCORE order (redis.clj:286): by score, then by member bytes.  The engine hands a sweep's
        barriers over in table order; the host, which has the id strings, restores Redis' order."""
        return sorted(conts, key=lambda k: (k.publish_at, k.barrier_id.encode()))

    # redis/all-publish-messages (redis.clj:283-289)
    def pop_due(self, now_ms):
        out = []
        while True:
            conts, more = self.e.pop_due(now_ms)
            out += self._conts(conts)
            if not more:
                return self._redis_order(out)

    # pop + join-metadata + all-joins for every due barrier (core.clj:184-188,175,163)
    def sweep(self, now_ms):
        out = []
        while True:
            conts, lens, mws, pay, more = self.e.sweep(now_ms)
            out += self._conts(conts, lens, mws, pay)
            if not more:
                return self._redis_order(out)

    # core/complete / attempt-retry-publish for published (ok) or failed continuations
    def ack(self, conts, oks, now_ms):
        # one barrier may appear only once per engine call: later duplicates go to follow-up calls
        statuses = [0] * len(conts)
        pending = list(range(len(conts)))
        while pending:
            seen, batch, rest = set(), [], []
            for i in pending:
                (rest if conts[i].barrier_id in seen else batch).append(i)
                seen.add(conts[i].barrier_id)
            raw = np.array([conts[i].raw for i in batch], dtype=capi.CONT)
            st = self.e.ack_batch(raw, np.array([1 if oks[i] else 0 for i in batch], dtype=np.uint8), now_ms)
            for i, s in zip(batch, st):
                statuses[i] = int(s)
            pending = rest
        self._forget(conts, statuses)
        return statuses

    # core/recover-unprocessed-publish-messages (core.clj:193-201)
    def recover(self, now_ms):
        out = []
        while True:
            conts, st, more = self.e.recover(now_ms)
            out += list(zip(self._conts(conts), [int(s) for s in st]))
            if not more:
                self._forget([c for c, _ in out], [s for _, s in out])
                return out

    # join-metadata + all-joins-status / all-joins (core.clj:63-66, redis.clj:242-253)
    def snapshot(self, barrier_id, update_ts=False):
        lo, hi = self._fp(barrier_id)
        meta, lens, mws, pay = self.e.snapshot_one(lo, hi, update_ts)
        meta = meta.copy()
        meta["pos_base"] = 0
        return self._conts([meta], lens, mws, pay)[0]

    def store_in_publish_queue(self, barrier_id, at_ms):
        lo, hi = self._fp(barrier_id, register=True)
        self.e.store_in_publish_queue(lo, hi, at_ms)

    def queue_scores(self, barrier_id):
        lo, hi = self._fp(barrier_id)
        a, b = self.e.queue_scores(lo, hi)
        return (None if a == capi.DJ_NOT_QUEUED else a, None if b == capi.DJ_NOT_QUEUED else b)
