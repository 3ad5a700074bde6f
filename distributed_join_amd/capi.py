"""ctypes binding of the C ABI in include/dj_engine.h (libdj_engine.so, HIP for gfx950).

There is no CPU implementation behind these symbols: if the shared library is missing
the import fails loudly, and every compute entry point needs a GPU.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DJ_ENGINE_LIB: another build of the same library (A/B timing of two builds in one GPU session)
LIB_PATH = os.environ.get("DJ_ENGINE_LIB") or os.path.join(_HERE, "libdj_engine.so")

# ---- constants of include/dj_engine.h ---------------------------------------------
DJ_OK = 0
DJ_E_INVALID, DJ_E_HIP, DJ_E_NOMEM, DJ_E_CLOCK_RANGE, DJ_E_TOO_LARGE = -1, -2, -3, -4, -5
DJ_NIL = 0xFFFFFFFF
DJ_NOT_QUEUED = 0xFFFFFFFFFFFFFFFF
DJ_JOIN_WAITING, DJ_JOIN_COMPLETE, DJ_JOIN_NOT_FOUND = 1, 2, 3
DJ_JOIN_BAD_POSITION, DJ_JOIN_NEAREST_NOT_QUEUED, DJ_JOIN_TOO_LONG, DJ_JOIN_BAD_MAX_WAIT = 4, 5, 6, 7
DJ_JOIN_NO_SPACE = 8
DJ_FORK_CREATED, DJ_FORK_TABLE_FULL, DJ_FORK_NO_SPACE, DJ_FORK_BAD_COUNT = 1, 2, 3, 4
DJ_ACK_DONE, DJ_ACK_REQUEUED_DIRTY, DJ_ACK_RETRY = 1, 2, 3
DJ_ACK_DROPPED_EXHAUSTED, DJ_ACK_DONE_NO_METADATA = 4, 5
DJ_ACK_DIRTY, DJ_ACK_METADATA_SET = 6, 7
DJ_CONT_TIMEOUT, DJ_CONT_NO_METADATA = 1, 2
DJ_CONT_VOID = CONT_VOID = 0x80000000
DJ_STATIC_JOIN_MAX = 65536
DJ_JOIN_HLEN_SHIFT = 8


def join_status(word):
    """DJ_JOIN_STATUS: the answer code of a join status word."""
    return int(word) & 0xFF


def join_hlen(word):
    """DJ_JOIN_HLEN: HLEN of the barrier's hash after the event (engines with set_join_hlen(True))."""
    return (int(word) & 0xFFFFFFFF) >> DJ_JOIN_HLEN_SHIFT

# ---- record layouts (little-endian, no padding surprises: all fields naturally aligned)
FORK_REC = np.dtype([("fp_lo", "<u8"), ("fp_hi", "<u8"), ("count", "<u4"), ("reserved", "<u4"),
                     ("deadline_ms", "<u8")])
JOIN_REC = np.dtype([("fp_lo", "<u8"), ("fp_hi", "<u8"), ("position", "<u4"), ("len", "<u4"),
                     ("max_wait_s", "<u4"), ("reserved", "<u4")])
CONT = np.dtype([("fp_lo", "<u8"), ("fp_hi", "<u8"), ("deadline_ms", "<u8"), ("publish_at", "<u8"),
                 ("count", "<u4"), ("flags", "<u4"), ("retry_publish", "<u4"), ("retry_timeout", "<u4"),
                 ("pos_base", "<u8"), ("reserved", "<u8")])
LONG_REF = np.dtype([("unit", "<u4"), ("size_class", "<u4"), ("len", "<u4"), ("zero", "<u4")])
SWEEP_RESULT = np.dtype([("n_conts", "<u4"), ("more", "<u4"), ("n_positions", "<u8")])
STATS = np.dtype([("slots", "<u8"), ("live", "<u8"), ("tombstones", "<u8"), ("arena_units_used", "<u8"),
                  ("arena_units_free", "<u8")])
assert FORK_REC.itemsize == 32 and JOIN_REC.itemsize == 32 and CONT.itemsize == 64
assert SWEEP_RESULT.itemsize == 16 and STATS.itemsize == 40


class Config(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32),
                ("capacity_barriers", ctypes.c_uint64), ("arena_bytes", ctypes.c_uint64),
                ("max_count", ctypes.c_uint32), ("payload_stride", ctypes.c_uint32),
                ("max_batch", ctypes.c_uint32), ("max_retry_publish", ctypes.c_uint32),
                ("max_retry_timeout_publish", ctypes.c_uint32), ("max_message_bytes", ctypes.c_uint32),
                ("check_publish_delay_ms", ctypes.c_uint64)]


# every symbol include/dj_engine.h declares
SYMBOLS = ["dj_create", "dj_destroy", "dj_last_error", "dj_sync", "dj_stream", "dj_fingerprint",
           "dj_fingerprint_batch", "dj_fork_batch", "dj_join_batch", "dj_pop_due", "dj_sweep",
           "dj_ack_batch", "dj_recover", "dj_snapshot_one", "dj_store_in_publish_queue",
           "dj_queue_scores", "dj_get_stats", "dj_owner_of", "dj_partition_joins", "dj_profile_enable",
           "dj_profile_read", "dj_compact", "dj_set_hot_path", "dj_checkpoint_save", "dj_checkpoint_load",
           "dj_get_config", "dj_fetch_long", "dj_join_batch_static",
           "dj_set_sweep_path", "dj_route_plan", "dj_route_unique_id", "dj_router_create", "dj_router_destroy",
           "dj_router_last_error", "dj_route_start", "dj_route_advance", "dj_route_next", "dj_route_release",
           "dj_complete_one", "dj_update_metadata_one", "dj_store_nearest_in_publish_queue", "dj_unprocessed",
           "dj_set_join_hlen", "dj_scatter_statuses", "dj_route_status_buffer", "dj_route_return", "dj_route_statuses",
           "dj_measure_copy", "dj_route_agree", "dj_router_get_limits",
           "dj_partition_joins_self"]
# include/dj_batcher.h
BATCHER_SYMBOLS = ["dj_batcher_create", "dj_batcher_create_routed", "dj_batcher_destroy", "dj_batcher_set_clock", "dj_batcher_join",
                   "dj_batcher_fork", "dj_batcher_join_wait", "dj_batcher_fork_wait", "dj_batcher_run",
                   "dj_batcher_get_stats"]
STATUS_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32)
CLOCK_CB = ctypes.CFUNCTYPE(ctypes.c_uint64, ctypes.c_void_p)
TASK_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)


class BatcherConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("max_batch", ctypes.c_uint32), ("max_wait_us", ctypes.c_uint32),
                ("max_message", ctypes.c_uint32)]


class BatcherStats(ctypes.Structure):
    _fields_ = [("batches", ctypes.c_uint64), ("forks", ctypes.c_uint64), ("joins", ctypes.c_uint64),
                ("largest_batch", ctypes.c_uint64), ("tasks", ctypes.c_uint64)]
PROF_KERNELS = ["fork_claim", "fork_apply", "join_link", "join_resolve", "join_complex", "join_copy", "pop", "gather",
                "ack", "recover", "compact"]

_lib = None


def load():
    """Load libdj_engine.so and declare the prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # PyTorch ships its own HIP runtime (same soname).  Load it first so that this library
    # binds to the runtime torch uses: two HIP runtimes in one process do not see the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.dj_create.restype = i32
    L.dj_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.dj_destroy.restype = None
    L.dj_destroy.argtypes = [vp]
    L.dj_last_error.restype = ctypes.c_char_p
    L.dj_last_error.argtypes = [vp]
    L.dj_sync.restype = i32
    L.dj_sync.argtypes = [vp]
    L.dj_stream.restype = vp
    L.dj_stream.argtypes = [vp]
    L.dj_fingerprint.restype = None
    L.dj_fingerprint.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.dj_fingerprint_batch.restype = None
    L.dj_fingerprint_batch.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.dj_fork_batch.restype = i32
    L.dj_fork_batch.argtypes = [vp, u32, vp, vp]
    L.dj_join_batch.restype = i32
    L.dj_join_batch.argtypes = [vp, u32, vp, vp, u64, u64, vp]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_join_batch_static"):
        L.dj_join_batch_static.restype = i32
        L.dj_join_batch_static.argtypes = [vp, u32, vp, vp, u64, u64, vp]
    L.dj_pop_due.restype = i32
    L.dj_pop_due.argtypes = [vp, u64, u32, vp, vp, vp]
    L.dj_sweep.restype = i32
    L.dj_sweep.argtypes = [vp, u64, u32, u64, vp, vp, vp, vp, vp, vp]
    L.dj_ack_batch.restype = i32
    L.dj_ack_batch.argtypes = [vp, u32, vp, vp, vp, u64, vp]
    L.dj_recover.restype = i32
    L.dj_recover.argtypes = [vp, u64, u32, vp, vp, vp, vp]
    L.dj_snapshot_one.restype = i32
    L.dj_snapshot_one.argtypes = [vp, u64, u64, i32, vp, vp, vp, vp]
    L.dj_store_in_publish_queue.restype = i32
    L.dj_store_in_publish_queue.argtypes = [vp, u64, u64, u64]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_complete_one"):
        L.dj_complete_one.restype = i32
        L.dj_complete_one.argtypes = [vp, u64, u64, ctypes.POINTER(ctypes.c_int32)]
        L.dj_update_metadata_one.restype = i32
        L.dj_update_metadata_one.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int32)]
        L.dj_store_nearest_in_publish_queue.restype = i32
        L.dj_store_nearest_in_publish_queue.argtypes = [vp, u64, u64, u64, ctypes.POINTER(ctypes.c_int32)]
        L.dj_unprocessed.restype = i32
        L.dj_unprocessed.argtypes = [vp, u64, u32, vp, vp, vp]
    L.dj_queue_scores.restype = i32
    L.dj_queue_scores.argtypes = [vp, u64, u64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.dj_get_stats.restype = i32
    L.dj_get_stats.argtypes = [vp, vp]
    L.dj_owner_of.restype = u32
    L.dj_owner_of.argtypes = [u64, u64, u32]
    L.dj_partition_joins.restype = i32
    L.dj_partition_joins.argtypes = [vp, u32, u32, vp, vp, u64, vp, vp, vp, vp, vp]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_partition_joins_self"):
        L.dj_partition_joins_self.restype = i32
        L.dj_partition_joins_self.argtypes = [vp, u32, u32, u32, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp]
    L.dj_set_hot_path.restype = i32
    L.dj_set_hot_path.argtypes = [vp, i32]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_set_sweep_path"):     # an older build under A/B lacks it
        L.dj_set_sweep_path.restype = i32
        L.dj_set_sweep_path.argtypes = [vp, i32]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_set_join_hlen"):
        L.dj_set_join_hlen.restype = i32
        L.dj_set_join_hlen.argtypes = [vp, i32]
    L.dj_checkpoint_save.restype = i32
    L.dj_checkpoint_save.argtypes = [vp, ctypes.c_char_p]
    L.dj_checkpoint_load.restype = i32
    L.dj_checkpoint_load.argtypes = [vp, ctypes.c_char_p]
    L.dj_get_config.restype = i32
    L.dj_get_config.argtypes = [vp, vp]
    L.dj_fetch_long.restype = i32
    L.dj_fetch_long.argtypes = [vp, u32, vp, vp, vp]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_route_plan"):
        L.dj_route_plan.restype = u32
        L.dj_route_plan.argtypes = [u32, u32, vp, u64, u32, vp, vp, vp]
        L.dj_route_unique_id.restype = i32
        L.dj_route_unique_id.argtypes = [vp]
        L.dj_router_create.restype = i32
        L.dj_router_create.argtypes = [vp, u32, u32, vp, u32, u32, u64, u32, ctypes.POINTER(vp)]
        L.dj_router_destroy.restype = None
        L.dj_router_destroy.argtypes = [vp]
        L.dj_router_last_error.restype = ctypes.c_char_p
        L.dj_router_last_error.argtypes = [vp]
        L.dj_route_start.restype = i32
        L.dj_route_start.argtypes = [vp, u32, vp, vp, ctypes.POINTER(ctypes.c_int)]
        L.dj_route_advance.restype = i32
        L.dj_route_advance.argtypes = [vp, ctypes.c_int]
        L.dj_route_next.restype = i32
        L.dj_route_next.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u32),
                                    ctypes.POINTER(ctypes.c_int)]
        L.dj_route_release.restype = i32
        L.dj_route_release.argtypes = [vp, ctypes.c_int]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_route_return"):
        L.dj_scatter_statuses.restype = i32
        L.dj_scatter_statuses.argtypes = [vp, u32, vp, vp, vp, vp]
        L.dj_route_status_buffer.restype = i32
        L.dj_route_status_buffer.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
        L.dj_route_return.restype = i32
        L.dj_route_return.argtypes = [vp, ctypes.c_int]
        L.dj_route_statuses.restype = i32
        L.dj_route_statuses.argtypes = [vp, ctypes.c_int, vp, vp]
        L.dj_route_agree.restype = i32
        L.dj_route_agree.argtypes = [vp, u32, vp]
        L.dj_router_get_limits.restype = i32
        L.dj_router_get_limits.argtypes = [vp, vp, vp, vp, vp, vp]
    L.dj_batcher_create.restype = i32
    L.dj_batcher_create.argtypes = [vp, ctypes.POINTER(BatcherConfig), ctypes.POINTER(vp)]
    L.dj_batcher_create_routed.restype = i32
    L.dj_batcher_create_routed.argtypes = [vp, vp, ctypes.POINTER(BatcherConfig), ctypes.POINTER(vp)]
    L.dj_batcher_destroy.restype = None
    L.dj_batcher_destroy.argtypes = [vp]
    L.dj_batcher_set_clock.restype = i32
    L.dj_batcher_set_clock.argtypes = [vp, CLOCK_CB, vp]
    L.dj_batcher_join.restype = i32
    L.dj_batcher_join.argtypes = [vp, vp, vp, STATUS_CB, vp]
    L.dj_batcher_fork.restype = i32
    L.dj_batcher_fork.argtypes = [vp, vp, STATUS_CB, vp]
    L.dj_batcher_join_wait.restype = i32
    L.dj_batcher_join_wait.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_int32)]
    L.dj_batcher_fork_wait.restype = i32
    L.dj_batcher_fork_wait.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int32)]
    L.dj_batcher_run.restype = i32
    L.dj_batcher_run.argtypes = [vp, TASK_CB, vp]
    L.dj_batcher_get_stats.restype = i32
    L.dj_batcher_get_stats.argtypes = [vp, ctypes.POINTER(BatcherStats)]
    if not os.environ.get("DJ_ENGINE_LIB") or hasattr(L, "dj_measure_copy"):
        L.dj_measure_copy.restype = i32
        L.dj_measure_copy.argtypes = [vp, u64, u32, ctypes.POINTER(ctypes.c_double)]
    L.dj_compact.restype = i32
    L.dj_compact.argtypes = [vp, ctypes.POINTER(u64)]
    L.dj_profile_enable.restype = i32
    L.dj_profile_enable.argtypes = [vp, i32]
    L.dj_profile_read.restype = i32
    L.dj_profile_read.argtypes = [vp, vp]
    _lib = L
    return L


def fingerprint(barrier_id):
    """128-bit fingerprint of one barrier-id (str or bytes) -> (fp_lo, fp_hi)."""
    b = barrier_id.encode() if isinstance(barrier_id, str) else bytes(barrier_id)
    lo, hi = ctypes.c_uint64(0), ctypes.c_uint64(0)
    buf = ctypes.create_string_buffer(b, len(b) + 1)
    load().dj_fingerprint(ctypes.cast(buf, ctypes.c_void_p), len(b), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def fingerprint_many(ids):
    """Fingerprints of a sequence of ids -> uint64 array of shape (n, 2) = (fp_lo, fp_hi)."""
    bs = [i.encode() if isinstance(i, str) else bytes(i) for i in ids]
    blob = np.frombuffer(b"".join(bs) or b"\0", dtype=np.uint8).copy()
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        np.cumsum([len(b) for b in bs], out=offs[1:])
    out = np.zeros((len(bs), 2), dtype=np.uint64)
    load().dj_fingerprint_batch(blob.ctypes.data_as(ctypes.c_void_p), offs.ctypes.data_as(ctypes.c_void_p),
                                len(bs), out.ctypes.data_as(ctypes.c_void_p))
    return out


def route_plan(matrix, rank, max_recv, rnd):
    """dj_route_plan: matrix (world x world, [s][r] = records s holds for r) -> (rounds, send_off, send_cnt,
    recv_cnt) of `rank` in round `rnd`."""
    m = np.ascontiguousarray(matrix, dtype=np.uint32)
    w = m.shape[0]
    so, sc, rc = (np.zeros(w, dtype=np.uint32) for _ in range(3))
    p = ctypes.c_void_p
    rounds = load().dj_route_plan(w, int(rank), m.ctypes.data_as(p), int(max_recv), int(rnd), so.ctypes.data_as(p),
                                  sc.ctypes.data_as(p), rc.ctypes.data_as(p))
    return int(rounds), so.tolist(), sc.tolist(), rc.tolist()


def owner_of(fp_lo, fp_hi, n_ranks):
    return int(load().dj_owner_of(int(fp_lo), int(fp_hi), int(n_ranks)))
