"""Engine: one GPU's barrier table behind the C ABI (include/dj_engine.h).

PyTorch is used only as plumbing: device buffers (uint8 tensors) and stream handles.
Two layers:
  *_dev methods   take device tensors and launch on the engine's stream without
                  synchronising — what bench.py and the multi-GPU router use;
  plain methods   take / return numpy arrays, copy through the GPU and synchronise —
                  what the host mirror (core.py) and the tests use.
"""
import ctypes

import numpy as np

from . import capi
from .capi import CONT, FORK_REC, JOIN_REC, STATS, SWEEP_RESULT


class EngineError(RuntimeError):
    pass


def _torch():
    import torch
    return torch


def to_device(arr, device):
    """numpy array (any dtype, structured included) -> uint8 device tensor with the same bytes."""
    torch = _torch()
    a = np.ascontiguousarray(arr)
    flat = a.view(np.uint8).reshape(-1) if a.size else np.zeros(0, np.uint8)
    t = torch.from_numpy(flat.copy())
    return t.to(device, non_blocking=False)


def from_device(t, dtype, count=None):
    """device tensor -> numpy array of dtype (count elements)."""
    h = t.cpu().numpy().view(np.uint8).reshape(-1)
    out = h.view(dtype)
    return out[:count].copy() if count is not None else out.copy()


class Engine:
    def __init__(self, capacity_barriers, arena_bytes, max_count=999, payload_stride=64, max_batch=1 << 20,
                 check_publish_delay_ms=10000, max_retry_publish=3, max_retry_timeout_publish=3, device=0,
                 host_cap_conts=1 << 16, host_cap_positions=1 << 20, max_message_bytes=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the barrier engine has no CPU implementation")
        self.L = capi.load()
        self.device = torch.device("cuda", device)
        self.stride = int(payload_stride)
        self.max_message = max(int(max_message_bytes), self.stride)     # messages above stride live in blocks of their own
        self.max_count = int(max_count)
        self.max_batch = int(max_batch)
        cfg = capi.Config(ctypes.sizeof(capi.Config), device, int(capacity_barriers), int(arena_bytes),
                          int(max_count), int(payload_stride), int(max_batch), int(max_retry_publish),
                          int(max_retry_timeout_publish), int(max_message_bytes), int(check_publish_delay_ms))
        h = ctypes.c_void_p()
        torch.cuda.init()
        with torch.cuda.device(self.device):
            rc = self.L.dj_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise EngineError(f"dj_create failed ({rc}): {self.L.dj_last_error(None).decode()}")
        self.h = h
        self.stream_ptr = self.L.dj_stream(self.h)
        self.host_cap_conts = int(host_cap_conts)
        self.host_cap_positions = max(int(host_cap_positions), self.max_count)
        self._host_bufs = None

    # ---- plumbing ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.dj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"engine call failed ({rc}): {self.L.dj_last_error(self.h).decode()}")

    def sync(self):
        self._check(self.L.dj_sync(self.h))

    def torch_stream(self):
        """The engine's HIP stream as a torch stream (for events / ordering with torch ops).
        close() destroys the stream: tensors in pinned memory that were copied on it must be freed before
        (torch's pinned allocator touches the streams a block was used on when the block is freed)."""
        torch = _torch()
        return torch.cuda.ExternalStream(self.stream_ptr, device=self.device)

    def empty(self, nbytes):
        torch = _torch()
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)

    # ---- device layer -------------------------------------------------------------------
    def fork_batch_dev(self, n, d_recs, d_status):
        self._check(self.L.dj_fork_batch(self.h, n, d_recs.data_ptr(), d_status.data_ptr()))

    def join_batch_dev(self, n, d_recs, d_payloads, pitch, now_ms, d_status):
        self._check(self.L.dj_join_batch(self.h, n, d_recs.data_ptr(), d_payloads.data_ptr(), int(pitch), int(now_ms),
                                         d_status.data_ptr()))

    def join_batch_static_dev(self, n, d_recs, d_payloads, pitch, now_ms, d_status):
        """dj_join_batch_static: the same pass with its work arrays in the library's static device array
        (n <= capi.DJ_STATIC_JOIN_MAX; allocates nothing)."""
        self._check(self.L.dj_join_batch_static(self.h, n, d_recs.data_ptr(), d_payloads.data_ptr(), int(pitch),
                                                int(now_ms), d_status.data_ptr()))

    def sweep_dev(self, now_ms, cap_conts, cap_positions, d_conts, d_lens, d_mws, d_payloads, d_result, h_result=None):
        self._check(self.L.dj_sweep(self.h, int(now_ms), int(cap_conts), int(cap_positions), d_conts.data_ptr(),
                                    d_lens.data_ptr() if d_lens is not None else None,
                                    d_mws.data_ptr() if d_mws is not None else None,
                                    d_payloads.data_ptr() if d_payloads is not None else None,
                                    d_result.data_ptr() if d_result is not None else None,
                                    h_result.ctypes.data_as(ctypes.c_void_p) if h_result is not None else None))

    def pop_due_dev(self, now_ms, cap_conts, d_conts, d_result, h_result=None):
        self._check(self.L.dj_pop_due(self.h, int(now_ms), int(cap_conts), d_conts.data_ptr(),
                                      d_result.data_ptr() if d_result is not None else None,
                                      h_result.ctypes.data_as(ctypes.c_void_p) if h_result is not None else None))

    def ack_batch_dev(self, n, d_n, d_conts, d_ok, now_ms, d_status):
        self._check(self.L.dj_ack_batch(self.h, int(n), d_n.data_ptr() if d_n is not None else None, d_conts.data_ptr(),
                                        d_ok.data_ptr() if d_ok is not None else None, int(now_ms),
                                        d_status.data_ptr()))

    def recover_dev(self, now_ms, cap, d_conts, d_status, d_result, h_result=None):
        self._check(self.L.dj_recover(self.h, int(now_ms), int(cap), d_conts.data_ptr(), d_status.data_ptr(),
                                      d_result.data_ptr() if d_result is not None else None,
                                      h_result.ctypes.data_as(ctypes.c_void_p) if h_result is not None else None))

    def partition_joins_dev(self, n, n_ranks, d_recs, d_payloads, pitch, d_recs_out, d_payloads_out, d_origin, d_counts,
                            stream_ptr=None, self_rank=None, d_recs_self=None, d_payloads_self=None):
        """dj_partition_joins; with self_rank, dj_partition_joins_self: that rank's group goes to d_recs_self / d_payloads_self."""
        ptr = lambda t: t.data_ptr() if t is not None else None
        if self_rank is None:
            self._check(self.L.dj_partition_joins(self.h, int(n), int(n_ranks), d_recs.data_ptr(), ptr(d_payloads), int(pitch),
                                                  d_recs_out.data_ptr(), ptr(d_payloads_out), d_origin.data_ptr(),
                                                  d_counts.data_ptr(), stream_ptr))
        else:
            self._check(self.L.dj_partition_joins_self(self.h, int(n), int(n_ranks), int(self_rank), d_recs.data_ptr(),
                                                       ptr(d_payloads), int(pitch), d_recs_out.data_ptr(), ptr(d_payloads_out),
                                                       d_recs_self.data_ptr(), ptr(d_payloads_self), d_origin.data_ptr(),
                                                       d_counts.data_ptr(), stream_ptr))

    # ---- host layer (numpy in, numpy out, synchronous) -----------------------------------------
    def _bufs(self):
        if self._host_bufs is None:
            cc, cp = self.host_cap_conts, self.host_cap_positions
            self._host_bufs = dict(conts=self.empty(cc * CONT.itemsize), lens=self.empty(cp * 4), mws=self.empty(cp * 4),
                                   pay=self.empty(cp * self.stride), res=self.empty(16), st=self.empty(cc * 4))
        return self._host_bufs

    def fork_batch(self, recs):
        recs = np.ascontiguousarray(recs, dtype=FORK_REC)
        n = len(recs)
        if n == 0:
            return np.zeros(0, np.int32)
        d_recs = to_device(recs, self.device)
        d_st = self.empty(n * 4)
        _torch().cuda.synchronize(self.device)
        self.fork_batch_dev(n, d_recs, d_st)
        self.sync()
        return from_device(d_st, np.int32, n)

    def join_batch(self, recs, payloads, now_ms):
        """payloads: uint8 array (n, pitch) with pitch a multiple of 16."""
        recs = np.ascontiguousarray(recs, dtype=JOIN_REC)
        n = len(recs)
        if n == 0:
            return np.zeros(0, np.int32)
        payloads = np.ascontiguousarray(payloads, dtype=np.uint8).reshape(n, -1)
        pitch = payloads.shape[1]
        d_recs = to_device(recs, self.device)
        d_pay = to_device(payloads, self.device)
        d_st = self.empty(n * 4)
        _torch().cuda.synchronize(self.device)
        self.join_batch_dev(n, d_recs, d_pay, pitch, now_ms, d_st)
        self.sync()
        return from_device(d_st, np.int32, n)

    def sweep(self, now_ms, with_payloads=True):
        """-> (conts[CONT], lens u32[n_positions], max_waits u32[...], payloads u8[n_positions, stride], more)"""
        b = self._bufs()
        res = np.zeros(1, dtype=SWEEP_RESULT)
        self.sweep_dev(now_ms, self.host_cap_conts, self.host_cap_positions, b["conts"], b["lens"], b["mws"],
                       b["pay"] if with_payloads else None, b["res"], res)
        n, npos = int(res[0]["n_conts"]), int(res[0]["n_positions"])
        conts = from_device(b["conts"][: max(n, 1) * CONT.itemsize], CONT, n)
        lens = from_device(b["lens"][: max(npos, 1) * 4], np.uint32, npos)
        mws = from_device(b["mws"][: max(npos, 1) * 4], np.uint32, npos)
        pay = (from_device(b["pay"][: max(npos, 1) * self.stride], np.uint8, npos * self.stride).reshape(npos, self.stride)
               if with_payloads else np.zeros((npos, self.stride), np.uint8))
        if with_payloads and self.max_message > self.stride:
            valid = np.zeros(npos, dtype=bool)          # positions of continuations that were gathered
            for c in conts:
                if not int(c["flags"]) & (capi.CONT_VOID | capi.DJ_CONT_NO_METADATA):
                    valid[int(c["pos_base"]):int(c["pos_base"]) + int(c["count"])] = True
            pay = self.resolve_long(np.where(valid, lens, capi.DJ_NIL), pay)
        return conts, lens, mws, pay, bool(res[0]["more"])

    def fetch_long_dev(self, n, d_refs, d_offsets, d_out):
        self._check(self.L.dj_fetch_long(self.h, int(n), d_refs.data_ptr(), d_offsets.data_ptr(), d_out.data_ptr()))

    def resolve_long(self, lens, rows):
        """Host-layer convenience: rows (n, stride) as a sweep / snapshot returned them; positions whose
        length exceeds the stride hold a dj_long_ref.  Returns rows wide enough for the longest message
        with the long messages fetched (dj_fetch_long).  Call before anything else changes the barriers."""
        lens = np.asarray(lens)
        if self.max_message <= self.stride:
            return rows
        idx = np.nonzero((lens != capi.DJ_NIL) & (lens > self.stride))[0]
        if len(idx) == 0:
            return rows
        width = (int(lens[idx].max()) + 15) // 16 * 16
        out = np.zeros((len(lens), max(width, self.stride)), dtype=np.uint8)
        out[:, :self.stride] = rows
        refs = np.ascontiguousarray(rows[idx, :16]).view(capi.LONG_REF).reshape(-1)
        assert (refs["len"] == lens[idx]).all(), "row reference and length disagree"
        sizes = (lens[idx].astype(np.int64) + 15) // 16 * 16
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        d_out = self.empty(int(sizes.sum()))
        self.fetch_long_dev(len(idx), to_device(refs, self.device), to_device(offs, self.device), d_out)
        self.sync()
        flat = d_out.cpu().numpy()
        for j, p in enumerate(idx):
            ln = int(lens[p])
            out[p, :ln] = flat[int(offs[j]):int(offs[j]) + ln]
        return out

    def pop_due(self, now_ms):
        b = self._bufs()
        res = np.zeros(1, dtype=SWEEP_RESULT)
        self.pop_due_dev(now_ms, self.host_cap_conts, b["conts"], b["res"], res)
        n = int(res[0]["n_conts"])
        return from_device(b["conts"][: max(n, 1) * CONT.itemsize], CONT, n), bool(res[0]["more"])

    def ack_batch(self, conts, ok, now_ms):
        conts = np.ascontiguousarray(conts, dtype=CONT)
        n = len(conts)
        if n == 0:
            return np.zeros(0, np.int32)
        d_c = to_device(conts, self.device)
        d_ok = to_device(np.ascontiguousarray(ok, dtype=np.uint8), self.device) if ok is not None else None
        d_st = self.empty(n * 4)
        _torch().cuda.synchronize(self.device)
        self.ack_batch_dev(n, None, d_c, d_ok, now_ms, d_st)
        self.sync()
        return from_device(d_st, np.int32, n)

    def recover(self, now_ms):
        b = self._bufs()
        res = np.zeros(1, dtype=SWEEP_RESULT)
        self.recover_dev(now_ms, self.host_cap_conts, b["conts"], b["st"], b["res"], res)
        n = int(res[0]["n_conts"])
        return (from_device(b["conts"][: max(n, 1) * CONT.itemsize], CONT, n),
                from_device(b["st"][: max(n, 1) * 4], np.int32, n), bool(res[0]["more"]))

    def snapshot_one(self, fp_lo, fp_hi, update_ts=False):
        meta = np.zeros(1, dtype=CONT)
        lens = np.full(self.max_count, capi.DJ_NIL, dtype=np.uint32)
        mws = np.zeros(self.max_count, dtype=np.uint32)
        pay = np.zeros((self.max_count, self.stride), dtype=np.uint8)
        vp = ctypes.c_void_p
        self._check(self.L.dj_snapshot_one(self.h, int(fp_lo), int(fp_hi), 1 if update_ts else 0,
                                           meta.ctypes.data_as(vp), lens.ctypes.data_as(vp), mws.ctypes.data_as(vp),
                                           pay.ctypes.data_as(vp)))
        c = int(meta[0]["count"])
        return meta[0], lens[:c], mws[:c], self.resolve_long(lens[:c], pay[:c])

    def store_in_publish_queue(self, fp_lo, fp_hi, at_ms):
        self._check(self.L.dj_store_in_publish_queue(self.h, int(fp_lo), int(fp_hi), int(at_ms)))

    def queue_scores(self, fp_lo, fp_hi):
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.L.dj_queue_scores(self.h, int(fp_lo), int(fp_hi), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # ---- the store namespace's remaining commands, one key at a time ------------------------------
    def complete_one(self, fp_lo, fp_hi):
        """redis/complete alone -> DJ_ACK_DONE (deleted) or DJ_ACK_DIRTY (a join arrived since the last all-joins)."""
        st = ctypes.c_int32(0)
        self._check(self.L.dj_complete_one(self.h, int(fp_lo), int(fp_hi), ctypes.byref(st)))
        return st.value

    def update_metadata_one(self, fp_lo, fp_hi, count, deadline_ms, retry_publish=None, retry_timeout=None):
        """redis/update-metadata alone: SET of the metadata map -> DJ_ACK_METADATA_SET, or 0 when refused."""
        row = np.zeros(1, dtype=CONT)
        row[0]["fp_lo"], row[0]["fp_hi"], row[0]["count"], row[0]["deadline_ms"] = fp_lo, fp_hi, count, deadline_ms
        row[0]["retry_publish"] = capi.DJ_NIL if retry_publish is None else retry_publish
        row[0]["retry_timeout"] = capi.DJ_NIL if retry_timeout is None else retry_timeout
        st = ctypes.c_int32(0)
        self._check(self.L.dj_update_metadata_one(self.h, row.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st)))
        return st.value

    def store_nearest_in_publish_queue(self, fp_lo, fp_hi, at_ms):
        """redis/store-nearest-in-publish-queue -> 0, or -1 when the id is not in the primary queue (the Lua error)."""
        out = ctypes.c_int32(0)
        self._check(self.L.dj_store_nearest_in_publish_queue(self.h, int(fp_lo), int(fp_hi), int(at_ms), ctypes.byref(out)))
        return out.value

    def unprocessed(self, now_ms):
        """redis/get-unprocessed-publish-messages -> (conts[CONT], more); read-only."""
        b = self._bufs()
        res = np.zeros(1, dtype=SWEEP_RESULT)
        self._check(self.L.dj_unprocessed(self.h, int(now_ms), self.host_cap_conts, b["conts"].data_ptr(), b["res"].data_ptr(),
                                          res.ctypes.data_as(ctypes.c_void_p)))
        n = int(res[0]["n_conts"])
        return from_device(b["conts"][: max(n, 1) * CONT.itemsize], CONT, n), bool(res[0]["more"])

    def set_hot_path(self, mode):
        """-1 automatic, 0 serial, 1 parallel kernels for barriers with long chains; +2 pins the 16-event
        register replay of short chains, 0 / 1 the 8-event one; +4 keeps the separate message-copy pass where
        the link pass could copy the messages itself; +8 never runs the bucketed form of the link and resolve passes,
        +16 runs it wherever it applies (by itself the engine uses it for batches of 196608 to 1310720 events without long
        chains on tables of up to 2^22 slots); 8 and 16 alone leave the other choices automatic (same results in every mode)."""
        self._check(self.L.dj_set_hot_path(self.h, int(mode)))

    def set_sweep_path(self, mode):
        """-1 by table size, 0 every sweep scans the queue words, 1 sweeps drain the ready list that join
        batches fill (same results)."""
        self._check(self.L.dj_set_sweep_path(self.h, int(mode)))

    def set_join_hlen(self, on=True):
        """Join status words also carry HLEN (redis/join's :count, redis.clj:137,157): capi.join_status / capi.join_hlen."""
        self._check(self.L.dj_set_join_hlen(self.h, 1 if on else 0))

    def checkpoint_save(self, path):
        """Write the whole barrier state to a file (waits for the stream)."""
        self._check(self.L.dj_checkpoint_save(self.h, str(path).encode()))

    def checkpoint_load(self, path):
        """Replace this engine's state with a saved one (same capacity, stride and max_count)."""
        self._check(self.L.dj_checkpoint_load(self.h, str(path).encode()))

    def compact(self):
        """Rehash the live barriers into the spare table (drops tombstones); returns the live count."""
        live = ctypes.c_uint64(0)
        self._check(self.L.dj_compact(self.h, ctypes.byref(live)))
        return live.value

    def measure_copy(self, nbytes=1 << 30, iterations=5):
        """GB/s (read + written) of the library's plain 16-bytes-per-lane copy kernel on this GPU."""
        out = ctypes.c_double(0.0)
        self._check(self.L.dj_measure_copy(self.h, int(nbytes), int(iterations), ctypes.byref(out)))
        return out.value

    def profile_enable(self, on=True):
        self._check(self.L.dj_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """-> {kernel name: ms of its most recent launch} for kernels launched since the last read."""
        ms = np.zeros(len(capi.PROF_KERNELS), dtype=np.float32)
        self._check(self.L.dj_profile_read(self.h, ms.ctypes.data_as(ctypes.c_void_p)))
        return {k: float(v) for k, v in zip(capi.PROF_KERNELS, ms) if v >= 0}

    def stats(self):
        st = np.zeros(1, dtype=STATS)
        self._check(self.L.dj_get_stats(self.h, st.ctypes.data_as(ctypes.c_void_p)))
        return {k: int(st[0][k]) for k in STATS.names}
