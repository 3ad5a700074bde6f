"""Multi-GPU routing: barriers shard by fingerprint over the ranks of one node; a batch of
join (or fork) records is routed to the owning GPUs with one all-to-all-v.

The reference has no counterpart (its replicas share one Redis, reference
src/distributed_join/redis.clj:13-15); the exchange is the only collective on the path —
sweeps, gathers, acks and recovers are GPU-local.

route = partition (HIP kernels, dj_partition_joins: stable, so every owner sees the events of
one source in array order) + counts exchange + two all_to_all_single calls (records,
messages).  The receiver's order is: source rank 0's events, then rank 1's, ...

The router works on its OWN stream with several buffer sets (three by default), so the routing of
batches k+1 and k+2 (its kernels, its collectives and the host's wait for the bucket sizes) runs
while the engine's stream applies batch k; with two batches in flight neither stream waits for
the host:
    h = router.start(recs, payloads, n)     # partition + exchange, asynchronous
    ...                                      # the engine applies the previous batch
    recs_in, pay_in, n_in = router.finish(h) # the engine's stream waits for the data (no host wait)
    engine.join_batch_dev(n_in, recs_in, pay_in, ...)
    router.release(h)                        # after the batch's kernels are queued: buffers may be reused
"""
import os
import sys

import numpy as np

_DEBUG = bool(os.environ.get("DJ_BENCH_DEBUG"))


def _dbg(*a):
    if _DEBUG:
        import time
        print(f"{time.time() % 1000:8.3f} [router rank {os.environ.get('RANK', '0')}]", *a, file=sys.stderr, flush=True)


def owner_of_np(fp_hi, n_ranks):
    """Owner rank of fingerprints (vectorised twin of dj_owner_of): (fp_hi >> 17) % n_ranks."""
    return ((np.asarray(fp_hi, dtype=np.uint64) >> np.uint64(17)) % np.uint64(n_ranks)).astype(np.int64)


def exchange(dist, group, recs_sorted, pay_sorted, send_counts, recv_recs, recv_pay):
    """all-to-all-v of records (rows of recs_sorted) and messages (rows of pay_sorted, optional).
    send_counts: python list, rows for each destination rank.  Returns (n_received, recv_counts).
    Works on any torch.distributed backend (RCCL on GPUs, gloo on CPU tensors)."""
    import torch
    world = dist.get_world_size(group)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=recs_sorted.device)
    rc = torch.empty(world, dtype=torch.int64, device=recs_sorted.device)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.cpu().tolist()]
    return exchange_rows(dist, group, recs_sorted, pay_sorted, send_counts, recv_counts, recv_recs, recv_pay)


def exchange_rows(dist, group, recs_sorted, pay_sorted, send_counts, recv_counts, recv_recs, recv_pay):
    """The data part of exchange(): both sides' bucket sizes are known."""
    n_in, n_out = sum(recv_counts), sum(send_counts)
    if n_in > recv_recs.shape[0]:
        raise RuntimeError(f"router: {n_in} incoming records exceed the receive buffer ({recv_recs.shape[0]})")
    _dbg("records a2a", n_out, "->", n_in)
    dist.all_to_all_single(recv_recs[:n_in], recs_sorted[:n_out], output_split_sizes=recv_counts,
                           input_split_sizes=list(send_counts), group=group)
    _dbg("records done")
    if pay_sorted is not None:
        dist.all_to_all_single(recv_pay[:n_in], pay_sorted[:n_out], output_split_sizes=recv_counts,
                               input_split_sizes=list(send_counts), group=group)
    return n_in, recv_counts


class _Slot:
    """One buffer set."""

    def __init__(self, torch, dev, max_send, max_recv, pitch):
        self.recs_sorted = torch.empty((max_send, 4), dtype=torch.int64, device=dev)
        self.pay_sorted = torch.empty((max_send, pitch), dtype=torch.uint8, device=dev)
        self.origin = torch.empty(max_send, dtype=torch.int32, device=dev)
        self.counts = torch.empty(64, dtype=torch.int32, device=dev)
        self.counts_in = torch.empty(64, dtype=torch.int32, device=dev)
        self.h_counts = torch.empty(128, dtype=torch.int32).pin_memory()   # [0:64] sent, [64:128] received
        self.recv_recs = torch.empty((max_recv, 4), dtype=torch.int64, device=dev)
        self.recv_pay = torch.empty((max_recv, pitch), dtype=torch.uint8, device=dev)
        self.ready = torch.cuda.Event()      # data has arrived (recorded on the router's stream)
        self.free = torch.cuda.Event()       # the engine has consumed it (recorded on the engine's stream)
        self.free_pending = False
        self.n_in = 0
        self.has_pay = False


class Router:
    def __init__(self, engine, group, max_send, max_recv, pitch, depth=3):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.e, self.group, self.dist = engine, group, dist
        self.world = dist.get_world_size(group)
        self.pitch = int(pitch)
        dev = engine.device
        self.stream = torch.cuda.Stream(device=dev)
        self.engine_stream = engine.torch_stream()
        self.slots = [_Slot(torch, dev, max_send, max_recv, self.pitch) for _ in range(max(2, int(depth)))]
        self.turn = 0
        # gloo (tests, rehearsals on one GPU) moves device tensors through worker threads without
        # blocking the caller, and several such exchanges in flight can pair up differently on two
        # ranks; with gloo the rows are therefore staged through host memory, one blocking call at a time
        self.host_staged = dist.get_backend(group) == "gloo"

    def start(self, d_recs, d_payloads, n):
        """Begin routing n records (rows of the (n, 4) int64 tensor d_recs: dj_join_rec or dj_fork_rec)
        with their messages ((n, pitch) uint8, or None).  The inputs must already be complete on the
        device.  Returns a handle for finish()."""
        torch = self.torch
        s = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        with torch.cuda.stream(self.stream):
            if s.free_pending:                    # the engine may still be reading this buffer set
                self.stream.wait_event(s.free)
                s.free_pending = False
            s.has_pay = d_payloads is not None
            self.e.partition_joins_dev(n, self.world, d_recs, d_payloads, self.pitch, s.recs_sorted,
                                       s.pay_sorted if s.has_pay else None, s.origin, s.counts,
                                       stream_ptr=self.stream.cuda_stream)
            if self.host_staged:
                self._exchange_host_staged(s, n)
                return s
            # bucket sizes travel rank to rank on the device; ONE read-back brings both sides' sizes to the host
            w = self.world
            _dbg("partition queued, exchanging bucket sizes")
            self.dist.all_to_all_single(s.counts_in[:w], s.counts[:w], group=self.group)
            s.h_counts[:64].copy_(s.counts, non_blocking=True)
            s.h_counts[64:].copy_(s.counts_in, non_blocking=True)
            self.stream.synchronize()                 # waits for the router's stream only
            hc = s.h_counts.tolist()
            _dbg("bucket sizes", hc[:w], hc[64:64 + w])
            s.n_in, _ = exchange_rows(self.dist, self.group, s.recs_sorted, s.pay_sorted if s.has_pay else None,
                                      hc[:w], hc[64:64 + w], s.recv_recs, s.recv_pay)
            s.ready.record(self.stream)
            _dbg("rows exchanged")
        return s

    def _exchange_host_staged(self, s, n):
        torch, w = self.torch, self.world
        self.stream.synchronize()
        counts = [int(x) for x in s.counts[:w].cpu().tolist()]
        n_out = sum(counts)
        rs = s.recs_sorted[:n_out].cpu()
        ps = s.pay_sorted[:n_out].cpu() if s.has_pay else None
        rr = torch.empty((s.recv_recs.shape[0], 4), dtype=torch.int64)
        rp = torch.empty((s.recv_pay.shape[0], self.pitch), dtype=torch.uint8) if s.has_pay else None
        s.n_in, _ = exchange(self.dist, self.group, rs, ps, counts, rr, rp)
        s.recv_recs[:s.n_in].copy_(rr[:s.n_in], non_blocking=False)
        if s.has_pay:
            s.recv_pay[:s.n_in].copy_(rp[:s.n_in], non_blocking=False)
        s.ready.record(self.stream)

    def finish(self, s):
        """-> (records for this rank, messages for this rank or None, n_received).  The engine's stream
        waits for the exchange; the host does not."""
        self.engine_stream.wait_event(s.ready)
        return s.recv_recs, (s.recv_pay if s.has_pay else None), s.n_in

    def release(self, s):
        """Call after the kernels that read the routed batch have been queued on the engine's stream."""
        s.free.record(self.engine_stream)
        s.free_pending = True

    def route(self, d_recs, d_payloads, n):
        """start + finish in one call (no overlap).  The caller releases nothing: the next start() on
        this buffer set waits for the engine's stream as it is now."""
        self.torch.cuda.current_stream().synchronize()
        s = self.start(d_recs, d_payloads, n)
        out = self.finish(s)
        return out + (s,)
