"""ctypes binding of the CPU oracle (oracle/dj_oracle.cpp).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module, and only as the checker.

The class exposes the same store interface as distributed_join_amd.engine.Engine
(fork_batch / join_batch / pop_due / sweep / ack_batch / recover / snapshot_one /
store_in_publish_queue / queue_scores), keyed by barrier-id strings, so that
distributed_join_amd.core.ForkJoinService and the parity tests can drive either one.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdj_oracle.so")

NIL = 0xFFFFFFFF
NOT_QUEUED = 0xFFFFFFFFFFFFFFFF

OCONT = np.dtype([("id_idx", "<u4"), ("count", "<u4"), ("flags", "<u4"), ("retry_publish", "<u4"),
                  ("retry_timeout", "<u4"), ("pad", "<u4"), ("deadline_ms", "<u8"),
                  ("publish_at", "<u8"), ("pos_base", "<u8")])
assert OCONT.itemsize == 48


def build():
    """Compile the oracle with g++ (no-op when it is up to date)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    vp, u32, u64, i64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64
    lib.djo_create.restype = vp
    lib.djo_create.argtypes = [u64, u32, u32, u32, u32]
    lib.djo_destroy.argtypes = [vp]
    lib.djo_set_ids.argtypes = [vp, vp, vp, u64]
    lib.djo_fork_batch.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.djo_join_batch.argtypes = [vp, u32, vp, vp, vp, vp, vp, u64, u64, vp]
    lib.djo_pop_due.restype = i64
    lib.djo_pop_due.argtypes = [vp, u64, u32, vp]
    lib.djo_sweep.restype = i64
    lib.djo_sweep.argtypes = [vp, u64, u32, u64, vp, vp, vp, vp, u64, vp]
    lib.djo_ack_batch.argtypes = [vp, u32, vp, vp, u64, vp]
    lib.djo_recover.restype = i64
    lib.djo_recover.argtypes = [vp, u64, u32, vp, vp]
    lib.djo_snapshot_one.restype = ctypes.c_int
    lib.djo_snapshot_one.argtypes = [vp, u32, ctypes.c_int, vp, vp, vp, vp, u64]
    lib.djo_store_in_publish_queue.argtypes = [vp, u32, u64]
    lib.djo_queue_scores.argtypes = [vp, u32, vp, vp]
    lib.djo_live_barriers.restype = u64
    lib.djo_live_barriers.argtypes = [vp]
    lib.djo_redis_fork.argtypes = [vp, u32, u32, u64]
    lib.djo_redis_update_metadata.argtypes = [vp, u32, u32, u64, u32, u32]
    lib.djo_redis_join.restype = ctypes.c_int
    lib.djo_redis_join.argtypes = [vp, u32, u32, ctypes.c_char_p, u32, u32, u64, ctypes.POINTER(u32)]
    lib.djo_redis_complete.restype = ctypes.c_int
    lib.djo_redis_complete.argtypes = [vp, u32]
    lib.djo_redis_store_nearest_in_publish_queue.restype = ctypes.c_int
    lib.djo_redis_store_nearest_in_publish_queue.argtypes = [vp, u32, u64]
    lib.djo_redis_all_publish_messages.restype = i64
    lib.djo_redis_all_publish_messages.argtypes = [vp, u64, u32, vp]
    lib.djo_redis_get_unprocessed_publish_messages.restype = i64
    lib.djo_redis_get_unprocessed_publish_messages.argtypes = [vp, u64, u32, vp]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    """Sequential string-keyed restatement of core.clj + redis.clj (see dj_oracle.cpp)."""

    def __init__(self, ids, max_count=999, payload_stride=64, check_publish_delay_ms=10000,
                 max_retry_publish=3, max_retry_timeout_publish=3):
        self.L = lib()
        self.stride = int(payload_stride)
        self.max_count = int(max_count)
        self.h = self.L.djo_create(check_publish_delay_ms, max_retry_publish, max_retry_timeout_publish,
                                   max_count, payload_stride)
        self.set_ids(ids)

    def set_ids(self, ids):
        self.ids = [i.encode() if isinstance(i, str) else bytes(i) for i in ids]
        self.index = {b: k for k, b in enumerate(self.ids)}
        blob = np.frombuffer(b"".join(self.ids) or b"\0", dtype=np.uint8).copy()
        offs = np.zeros(len(self.ids) + 1, dtype=np.uint64)
        np.cumsum([len(b) for b in self.ids], out=offs[1:])
        self.L.djo_set_ids(self.h, _p(blob), _p(offs), len(self.ids))

    def close(self):
        if self.h:
            self.L.djo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def idx(self, barrier_id):
        b = barrier_id.encode() if isinstance(barrier_id, str) else bytes(barrier_id)
        return self.index[b]

    # ---- batch interface: ids by index -------------------------------------------
    def fork_batch(self, id_idx, count, deadline_ms):
        id_idx = np.ascontiguousarray(id_idx, dtype=np.uint32)
        count = np.ascontiguousarray(count, dtype=np.uint32)
        deadline_ms = np.ascontiguousarray(deadline_ms, dtype=np.uint64)
        st = np.zeros(len(id_idx), dtype=np.int32)
        self.L.djo_fork_batch(self.h, len(id_idx), _p(id_idx), _p(count), _p(deadline_ms), _p(st))
        return st

    def join_batch(self, id_idx, position, length, max_wait_s, payloads, now_ms):
        id_idx = np.ascontiguousarray(id_idx, dtype=np.uint32)
        n = len(id_idx)
        position = np.ascontiguousarray(position, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        max_wait_s = np.ascontiguousarray(max_wait_s, dtype=np.uint32)
        payloads = np.ascontiguousarray(payloads, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 16), np.uint8)
        st = np.zeros(n, dtype=np.int32)
        self.L.djo_join_batch(self.h, n, _p(id_idx), _p(position), _p(length), _p(max_wait_s), _p(payloads),
                              payloads.shape[1], int(now_ms), _p(st))
        return st

    def pop_due(self, now_ms, cap=1 << 20):
        out = np.zeros(cap, dtype=OCONT)
        n = self.L.djo_pop_due(self.h, int(now_ms), cap, _p(out))
        if n < 0:
            raise RuntimeError("oracle pop_due: cap too small")
        return out[:n].copy()

    def sweep(self, now_ms, cap_conts=1 << 16, cap_positions=1 << 20):
        conts = np.zeros(cap_conts, dtype=OCONT)
        lens = np.full(cap_positions, NIL, dtype=np.uint32)
        mws = np.zeros(cap_positions, dtype=np.uint32)
        pay = np.zeros((cap_positions, self.stride), dtype=np.uint8)
        npos = ctypes.c_uint64(0)
        n = self.L.djo_sweep(self.h, int(now_ms), cap_conts, cap_positions, _p(conts), _p(lens), _p(mws), _p(pay),
                             self.stride, ctypes.byref(npos))
        if n < 0:
            raise RuntimeError("oracle sweep: caps too small")
        k = npos.value
        return conts[:n].copy(), lens[:k].copy(), mws[:k].copy(), pay[:k].copy()

    def ack_batch(self, conts, ok, now_ms):
        conts = np.ascontiguousarray(conts, dtype=OCONT)
        okb = None if ok is None else np.ascontiguousarray(ok, dtype=np.uint8)
        st = np.zeros(len(conts), dtype=np.int32)
        self.L.djo_ack_batch(self.h, len(conts), _p(conts), _p(okb), int(now_ms), _p(st))
        return st

    def recover(self, now_ms, cap=1 << 20):
        out = np.zeros(cap, dtype=OCONT)
        st = np.zeros(cap, dtype=np.int32)
        n = self.L.djo_recover(self.h, int(now_ms), cap, _p(out), _p(st))
        if n < 0:
            raise RuntimeError("oracle recover: cap too small")
        return out[:n].copy(), st[:n].copy()

    def snapshot_one(self, id_idx, update_ts=False):
        meta = np.zeros(1, dtype=OCONT)
        lens = np.full(self.max_count, NIL, dtype=np.uint32)
        mws = np.zeros(self.max_count, dtype=np.uint32)
        pay = np.zeros((self.max_count, self.stride), dtype=np.uint8)
        self.L.djo_snapshot_one(self.h, int(id_idx), 1 if update_ts else 0, _p(meta), _p(lens), _p(mws), _p(pay),
                                self.stride)
        c = int(meta[0]["count"])
        return meta[0], lens[:c], mws[:c], pay[:c]

    def store_in_publish_queue(self, id_idx, at_ms):
        self.L.djo_store_in_publish_queue(self.h, int(id_idx), int(at_ms))

    def queue_scores(self, id_idx):
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.L.djo_queue_scores(self.h, int(id_idx), ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def live_barriers(self):
        return int(self.L.djo_live_barriers(self.h))


class OracleStore:
    """String-keyed view of an Oracle with the interface of distributed_join_amd.store.EngineStore,
    so the host mirror (core.ForkJoinService) and the scenario tests can run on the checker."""

    def __init__(self, oracle):
        self.o = oracle

    def _idx(self, barrier_id):
        b = barrier_id.encode() if isinstance(barrier_id, str) else bytes(barrier_id)
        if b not in self.o.index:
            self.o.set_ids([i.decode() for i in self.o.ids] + [b.decode()])
        return self.o.index[b]

    def fork_many(self, ids, counts, deadlines):
        return self.o.fork_batch([self._idx(i) for i in ids], np.asarray(counts, dtype=np.int64).clip(0, 0xFFFFFFFF),
                                 deadlines)

    def fork(self, barrier_id, count, deadline_ms):
        return int(self.fork_many([barrier_id], [count], [deadline_ms])[0])

    def join_many(self, ids, positions, messages, max_waits, now_ms):
        n = len(ids)
        pitch = (max([len(m) for m in messages] + [1]) + 15) // 16 * 16
        pay = np.zeros((n, pitch), dtype=np.uint8)
        for k, m in enumerate(messages):
            pay[k, :len(m)] = np.frombuffer(bytes(m), dtype=np.uint8)
        return self.o.join_batch([self._idx(i) for i in ids], np.asarray(positions, dtype=np.int64).clip(0, 0xFFFFFFFF),
                                 [len(m) for m in messages], [0 if w is None else w for w in max_waits], pay, now_ms)

    def join(self, barrier_id, position, message, max_wait, now_ms):
        return int(self.join_many([barrier_id], [position], [message], [max_wait], now_ms)[0])

    def _conts(self, conts, lens=None, mws=None, pay=None):
        from distributed_join_amd.store import Continuation
        out = []
        for c in conts:
            opt = lambda v: None if int(v) == NIL else int(v)
            k = Continuation(self.o.ids[int(c["id_idx"])].decode(), int(c["count"]), bool(int(c["flags"]) & 1),
                             bool(int(c["flags"]) & 2), int(c["deadline_ms"]), int(c["publish_at"]),
                             opt(c["retry_publish"]), opt(c["retry_timeout"]), raw=c.copy())
            if lens is not None and not k.no_metadata:
                b = int(c["pos_base"])
                for p in range(k.count):
                    ln = int(lens[b + p])
                    if ln == NIL:
                        k.messages.append(None)
                        k.max_waits.append(None)
                    else:
                        k.messages.append(bytes(pay[b + p, :ln]))
                        k.max_waits.append(int(mws[b + p]) or None)
            out.append(k)
        return out

    def pop_due(self, now_ms):
        return self._conts(self.o.pop_due(now_ms))

    def sweep(self, now_ms):
        return self._conts(*self.o.sweep(now_ms))

    def ack(self, conts, oks, now_ms):
        raw = np.array([c.raw for c in conts], dtype=OCONT)
        return [int(s) for s in self.o.ack_batch(raw, np.array([1 if k else 0 for k in oks], dtype=np.uint8), now_ms)]

    def recover(self, now_ms):
        conts, st = self.o.recover(now_ms)
        return list(zip(self._conts(conts), [int(s) for s in st]))

    def snapshot(self, barrier_id, update_ts=False):
        meta, lens, mws, pay = self.o.snapshot_one(self._idx(barrier_id), update_ts)
        meta = meta.copy()
        meta["pos_base"] = 0
        return self._conts([meta], lens, mws, pay)[0]

    def store_in_publish_queue(self, barrier_id, at_ms):
        self.o.store_in_publish_queue(self._idx(barrier_id), at_ms)

    def queue_scores(self, barrier_id):
        a, b = self.o.queue_scores(self._idx(barrier_id))
        return (None if a == NOT_QUEUED else a, None if b == NOT_QUEUED else b)


class OracleRedis:
    """The namespace distributed-join.redis on the checker, function for function (redis.clj:90-300): the same
    thirteen functions, arguments, return shapes and failures as distributed_join_amd.redis_namespace.EngineRedis,
    each on the oracle's restatement of that one Redis command sequence.  core_unchanged.Core runs on either."""

    def __init__(self, oracle, clock, env):
        from distributed_join_amd import redis_namespace as ns
        self.ns = ns
        self.o = oracle
        self.clock = clock
        self.env = env
        self.records = {}       # the strings of "<id>:metadata" (ids, URLs); numbers live in the oracle

    def _idx(self, barrier_id):
        b = barrier_id.encode()
        if b not in self.o.index:
            self.o.set_ids([i.decode() for i in self.o.ids] + [barrier_id])
        return self.o.index[b]

    def _ids(self, fn, now):
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = fn(self.o.h, int(now), len(out), _p(out))
        if n < 0:
            raise RuntimeError("oracle: id buffer too small")
        return [self.o.ids[int(k)].decode() for k in out[:n]]

    @staticmethod
    def _strings(m):
        return {k: v for k, v in m.items() if k not in ("publish-retry-cnt", "timeout-publish-retry-cnt")}

    def fork(self, record):
        key = self.ns.make_key(record["join-id"], record["transaction-id"])
        self.o.L.djo_redis_fork(self.o.h, self._idx(key), record["count"], record["timeout-secs"])
        self.records[key] = self._strings(record)
        return key

    def update_metadata(self, metadata):
        key = self.ns.make_key(metadata.get("join-id"), metadata.get("transaction-id"))
        opt = lambda v: NIL if v is None else v
        self.o.L.djo_redis_update_metadata(self.o.h, self._idx(key), metadata["count"], metadata["timeout-secs"],
                                           opt(metadata.get("publish-retry-cnt")), opt(metadata.get("timeout-publish-retry-cnt")))
        self.records[key] = self._strings(metadata)
        return key + ":metadata"

    def join(self, record):
        bid = record["barrier-id"]
        md = self.join_metadata(bid)                                       # GET, redis.clj:149
        msg = record["message"].encode()
        hlen = ctypes.c_uint32(0)
        st = self.o.L.djo_redis_join(self.o.h, self._idx(bid), min(record["position"], 0xFFFFFFFF), msg, len(msg),
                                     record.get("max-wait") or 0, int(self.clock()), ctypes.byref(hlen))
        if st == 3:
            raise self.ns.err("not-found", "barrier-id does not exist")
        if st == 4:
            raise self.ns.err("bad-data", "position is greater than total")
        return {"metadata": md, "count": hlen.value}

    def complete(self, barrier_id):
        if not self.o.L.djo_redis_complete(self.o.h, self._idx(barrier_id)):
            raise self.ns.StoreError("Join was updated before fork could be completed", {"updated-join-before-complete": True})
        self.records.pop(barrier_id, None)
        return 1

    def _all_joins(self, barrier_id, count, update_ts):
        meta, lens, mws, pay = self.o.snapshot_one(self._idx(barrier_id), update_ts)
        if count is None:
            return None
        out = []
        for p in range(count):
            if p >= len(lens) or int(lens[p]) == NIL:
                out.append(None)
                continue
            j = {"barrier-id": barrier_id, "position": p, "message": bytes(pay[p, :int(lens[p])]).decode()}
            if int(mws[p]):
                j["max-wait"] = int(mws[p])
            out.append(j)
        return out

    def all_joins(self, barrier_id, count):
        return self._all_joins(barrier_id, count, True)

    def all_joins_status(self, barrier_id, count):
        return self._all_joins(barrier_id, count, False)

    def join_metadata(self, barrier_id):
        meta = np.zeros(1, dtype=OCONT)
        self.o.L.djo_snapshot_one(self.o.h, self._idx(barrier_id), 0, _p(meta), None, None, None, self.o.stride)
        if int(meta[0]["flags"]) & 2:
            return None
        md = dict(self.records.get(barrier_id, {}))
        md["count"], md["timeout-secs"] = int(meta[0]["count"]), int(meta[0]["deadline_ms"])
        for key, field in (("publish-retry-cnt", "retry_publish"), ("timeout-publish-retry-cnt", "retry_timeout")):
            if int(meta[0][field]) != NIL:
                md[key] = int(meta[0][field])
        return md

    def store_in_publish_queue(self, barrier_id, publish_at):
        self.o.store_in_publish_queue(self._idx(barrier_id), publish_at)
        return [publish_at]

    def store_nearest_in_publish_queue(self, barrier_id, publish_at):
        if not self.o.L.djo_redis_store_nearest_in_publish_queue(self.o.h, self._idx(barrier_id), int(publish_at)):
            raise self.ns.database_failure("attempt to compare boolean with string")
        return [publish_at, 0]

    def all_publish_messages(self):
        return self._ids(self.o.L.djo_redis_all_publish_messages, self.clock())

    def get_unprocessed_publish_messages(self):
        return self._ids(self.o.L.djo_redis_get_unprocessed_publish_messages, self.clock())

    def get_publish_date(self, barrier_id):
        a, _ = self.o.queue_scores(self._idx(barrier_id))
        return None if a == NOT_QUEUED else a

    def shutdown(self):
        self.o.close()
