"""ctypes front of include/dj_batcher.h: single requests from many threads, batched onto the engine.

The reference's handlers call core/fork and core/join once per HTTP request
(reference src/distributed_join/ring.clj:54-72); this is the piece that lets them keep doing so."""
import ctypes

import numpy as np

from . import capi


class Batcher:
    def __init__(self, engine, max_batch=1 << 14, max_wait_us=100, max_message=None):
        self.e = engine
        self.L = engine.L
        cfg = capi.BatcherConfig(ctypes.sizeof(capi.BatcherConfig), int(max_batch), int(max_wait_us),
                                 int(max_message if max_message is not None else engine.max_message))
        h = ctypes.c_void_p()
        rc = self.L.dj_batcher_create(engine.h, ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"dj_batcher_create failed ({rc})")
        self.h = h
        self._keep = []          # callbacks that C may still call

    def close(self):
        if self.h:
            self.L.dj_batcher_destroy(self.h)
            self.h = None
            self._keep.clear()

    def set_clock(self, fn):
        """fn() -> epoch milliseconds, called on the drain thread once per batch."""
        cb = capi.CLOCK_CB(lambda _u: int(fn()))
        self._keep.append(cb)
        self._check(self.L.dj_batcher_set_clock(self.h, cb, None))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"batcher call failed ({rc})")

    @staticmethod
    def _join_rec(fp, position, length, max_wait):
        r = np.zeros(1, dtype=capi.JOIN_REC)
        r[0] = (fp[0], fp[1], position, length, max_wait, 0)
        return r

    @staticmethod
    def _fork_rec(fp, count, deadline_ms):
        r = np.zeros(1, dtype=capi.FORK_REC)
        r[0] = (fp[0], fp[1], count, 0, deadline_ms)
        return r

    def fork_wait(self, fp, count, deadline_ms):
        r = self._fork_rec(fp, count, deadline_ms)
        st = ctypes.c_int32(0)
        self._check(self.L.dj_batcher_fork_wait(self.h, r.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st)))
        return st.value

    def join_wait(self, fp, position, message=b"", max_wait=0):
        r = self._join_rec(fp, position, len(message), max_wait)
        buf = ctypes.create_string_buffer(bytes(message), len(message)) if message else None
        st = ctypes.c_int32(0)
        self._check(self.L.dj_batcher_join_wait(self.h, r.ctypes.data_as(ctypes.c_void_p),
                                                ctypes.cast(buf, ctypes.c_void_p) if buf is not None else None,
                                                ctypes.byref(st)))
        return st.value

    def join_async(self, fp, position, message, on_status, max_wait=0):
        """on_status(status) runs on the drain thread."""
        r = self._join_rec(fp, position, len(message), max_wait)
        buf = ctypes.create_string_buffer(bytes(message), len(message)) if message else None
        cb = capi.STATUS_CB(lambda _u, st: on_status(int(st)))
        self._keep.append(cb)
        self._check(self.L.dj_batcher_join(self.h, r.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.cast(buf, ctypes.c_void_p) if buf is not None else None, cb, None))

    def fork_async(self, fp, count, deadline_ms, on_status):
        """on_status(status) runs on the drain thread."""
        r = self._fork_rec(fp, count, deadline_ms)
        cb = capi.STATUS_CB(lambda _u, st: on_status(int(st)))
        self._keep.append(cb)
        self._check(self.L.dj_batcher_fork(self.h, r.ctypes.data_as(ctypes.c_void_p), cb, None))

    def run(self, fn):
        """fn(now_ms) -> int runs on the drain thread between two batches, with the engine idle."""
        out = {}

        def tramp(_u, _e, now):
            try:
                out["value"] = fn(int(now))
                return 0
            except BaseException as ex:      # noqa: BLE001 - must not unwind into C
                out["error"] = ex
                return -1

        cb = capi.TASK_CB(tramp)
        rc = self.L.dj_batcher_run(self.h, cb, None)
        if "error" in out:
            raise out["error"]
        self._check(rc)
        return out.get("value")

    def stats(self):
        s = capi.BatcherStats()
        self._check(self.L.dj_batcher_get_stats(self.h, ctypes.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in capi.BatcherStats._fields_}
