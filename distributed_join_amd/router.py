"""Multi-GPU routing: barriers shard by fingerprint over the ranks of one node; a batch of
join (or fork) records is routed to the owning GPUs with one all-to-all-v.

The reference has no counterpart (its replicas share one Redis, reference
src/distributed_join/redis.clj:13-15); the exchange is the only collective on the path —
sweeps, gathers, acks and recovers are GPU-local.

route = partition (HIP kernels, dj_partition_joins: stable, so every owner sees the events of
one source in array order) + all-gather of the bucket sizes + all_to_all_single calls (records,
messages).  The receiver's order is: source rank 0's events, then rank 1's, ...

Skew: every rank holds the whole matrix of bucket sizes, so every rank computes the same plan
(dj_route_plan, the function the C router of include/dj_engine.h uses too): a receiver whose incoming
rows exceed its buffer (a hot key owns half of everybody's batch) takes them in several rounds, each a
collective that all ranks enter.  No rank raises while its peers sit in a collective.

The router works on two streams of its OWN with several buffer sets — one for the partition kernels, one for
the collectives — so while the engine's stream applies batch k, batch k+1 is exchanged and batch k+2 is
partitioned (the exchange's kernels occupy a few compute units for a long time: behind the partition on one
stream they would only add to its time).  start() never waits for the device: the host needs the bucket
sizes only when the rows are exchanged (advance() / finish()), one batch later.
    h = router.start(recs, payloads, n)      # partition + bucket sizes, asynchronous
    ...                                       # the engine applies the previous batch
    router.advance(h)                         # optional: exchange the first piece now (sizes are in by now)
    recs_in, pay_in, n_in = router.finish(h)  # the engine's stream waits for the data (no host wait)
    engine.join_batch_dev(n_in, recs_in, pay_in, ...)
    while router.more(h):                     # only under skew beyond the receive buffer
        recs_in, pay_in, n_in = router.next_round(h)
        engine.join_batch_dev(n_in, recs_in, pay_in, ...)
    router.release(h)                         # after the batch's kernels are queued: buffers may be reused
Per-record answers (what the reference's handlers send back, ring.clj:54-69) return to the submitting rank:
    engine.join_batch_dev(n_in, recs_in, pay_in, ..., router.status_buffer(h))    # per piece
    router.return_statuses(h)                 # per piece, a collective: 4 bytes per record over the plan read backwards
    router.statuses(h, d_status_out, stream)  # once, before release: this rank's n answers in its submission order
"""
import os
import sys

import numpy as np

from . import capi

_DEBUG = bool(os.environ.get("DJ_BENCH_DEBUG"))


def _dbg(*a):
    if _DEBUG:
        import time
        print(f"{time.time() % 1000:8.3f} [router rank {os.environ.get('RANK', '0')}]", *a, file=sys.stderr, flush=True)


def owner_of_np(fp_hi, n_ranks):
    """Owner rank of fingerprints (vectorised twin of dj_owner_of): (fp_hi >> 17) % n_ranks."""
    return ((np.asarray(fp_hi, dtype=np.uint64) >> np.uint64(17)) % np.uint64(n_ranks)).astype(np.int64)


def gather_matrix(dist, group, send_counts, device=None):
    """All ranks' bucket sizes: -> numpy (world, world), [s][r] = rows rank s holds for rank r.  Blocking."""
    import torch
    world = dist.get_world_size(group)
    mine = torch.tensor(send_counts, dtype=torch.int64, device=device)
    parts = [torch.empty(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.stack(parts).cpu().numpy().astype(np.uint32)


def exchange_round(dist, group, rank, matrix, max_recv, rnd, recs_sorted, pay_sorted, recv_recs, recv_pay):
    """One round of the planned exchange on any torch.distributed backend (RCCL on GPUs, gloo on CPU
    tensors).  -> (rows received, rounds in all)."""
    import torch
    rounds, send_off, send_cnt, recv_cnt = capi.route_plan(matrix, rank, max_recv, rnd)
    bucket = [int(x) for x in matrix[rank]]
    starts = np.concatenate([[0], np.cumsum(bucket)[:-1]]).tolist()
    n_in, n_out = sum(recv_cnt), sum(send_cnt)
    whole = all(o == 0 and c == b for o, c, b in zip(send_off, send_cnt, bucket))

    def rows(t):
        if whole:
            return t[:n_out]
        return torch.cat([t[st + o:st + o + c] for st, o, c in zip(starts, send_off, send_cnt)])

    _dbg("round", rnd, "of", rounds, ":", n_out, "->", n_in)
    dist.all_to_all_single(recv_recs[:n_in], rows(recs_sorted), output_split_sizes=recv_cnt, input_split_sizes=send_cnt,
                           group=group)
    if pay_sorted is not None:
        dist.all_to_all_single(recv_pay[:n_in], rows(pay_sorted), output_split_sizes=recv_cnt, input_split_sizes=send_cnt,
                               group=group)
    return n_in, rounds


def return_round(dist, group, rank, matrix, max_recv, rnd, status_piece, ret_sorted):
    """The way back of round `rnd`: status_piece[:n_in] holds one answer per row this rank received in that round
    (piece order); every answer goes to the rank its row came from and lands in ret_sorted at the place the row has
    in that rank's recs_sorted.  The same plan as exchange_round, read backwards: one all_to_all_single."""
    import torch
    _, send_off, send_cnt, recv_cnt = capi.route_plan(matrix, rank, max_recv, rnd)
    bucket = [int(x) for x in matrix[rank]]
    starts = np.concatenate([[0], np.cumsum(bucket)[:-1]]).tolist()
    n_in, n_out = sum(recv_cnt), sum(send_cnt)
    whole = all(o == 0 and c == b for o, c, b in zip(send_off, send_cnt, bucket))
    back = ret_sorted[:n_out] if whole else torch.empty(n_out, dtype=ret_sorted.dtype, device=ret_sorted.device)
    dist.all_to_all_single(back, status_piece[:n_in], output_split_sizes=send_cnt, input_split_sizes=recv_cnt, group=group)
    if not whole:
        at = 0
        for st, o, c in zip(starts, send_off, send_cnt):
            ret_sorted[st + o:st + o + c] = back[at:at + c]
            at += c


def exchange(dist, group, recs_sorted, pay_sorted, send_counts, recv_recs, recv_pay):
    """all-to-all-v of records (rows of recs_sorted, grouped by destination rank) and messages (rows of
    pay_sorted, optional) in as many rounds as the receive buffers need; the pieces are handed to `yield`-free
    callers as a list: -> [(n_received, recv_recs copy, recv_pay copy or None), ...] in application order."""
    rank = dist.get_rank(group)
    matrix = gather_matrix(dist, group, send_counts, device=recs_sorted.device)
    out, rnd, rounds = [], 0, 1
    while rnd < rounds:
        n_in, rounds = exchange_round(dist, group, rank, matrix, recv_recs.shape[0], rnd, recs_sorted, pay_sorted,
                                      recv_recs, recv_pay)
        out.append((n_in, recv_recs[:n_in].clone(), recv_pay[:n_in].clone() if pay_sorted is not None else None))
        rnd += 1
    return out


class _Slot:
    """One buffer set."""

    def __init__(self, torch, dev, world, max_send, max_recv, pitch):
        self.recs_sorted = torch.empty((max_send, 4), dtype=torch.int64, device=dev)
        self.pay_sorted = torch.empty((max_send, pitch), dtype=torch.uint8, device=dev)
        self.origin = torch.empty(max_send, dtype=torch.int32, device=dev)
        self.counts = torch.empty(64, dtype=torch.int32, device=dev)
        self.matrix_d = torch.empty(world * world, dtype=torch.int32, device=dev)
        self.matrix_h = torch.empty(world * world, dtype=torch.int32).pin_memory()
        self.recv_recs = torch.empty((max_recv, 4), dtype=torch.int64, device=dev)
        self.recv_pay = torch.empty((max_recv, pitch), dtype=torch.uint8, device=dev)
        self.status_piece = torch.empty(max_recv, dtype=torch.int32, device=dev)    # the engine's answers to the current piece
        self.ret_sorted = torch.empty(max_send, dtype=torch.int32, device=dev)      # answers to this rank's own records, bucket order
        self.returned = torch.cuda.Event()   # the latest piece's answers have left status_piece (router's stream)
        self.home = torch.cuda.Event()       # statuses(): the answers are in the caller's order
        self.return_pending = False
        self.home_pending = False
        self.returned_rounds = 0
        self.n = 0
        self.parted = torch.cuda.Event()     # the partition is done (recorded on the partition stream)
        self.sizes_in = torch.cuda.Event()   # the bucket sizes are in host memory
        self.ready = torch.cuda.Event()      # data has arrived (recorded on the router's stream)
        self.free = torch.cuda.Event()       # the engine has consumed it (recorded on the engine's stream)
        self.free_pending = False
        self.n_in = 0
        self.has_pay = False
        self.matrix = None                   # numpy (world, world) once the sizes were read
        self.round = self.rounds = 0
        self.host = None                     # host-staged form: (records, messages) as CPU tensors


class Router:
    def __init__(self, engine, group, max_send, max_recv, pitch, depth=3):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.e, self.group, self.dist = engine, group, dist
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.pitch = int(pitch)
        self.max_recv = int(max_recv)
        dev = engine.device
        self.stream = torch.cuda.Stream(device=dev)          # partition kernels
        self.comm_stream = torch.cuda.Stream(device=dev)     # bucket sizes and rows between the ranks
        self.engine_stream = engine.torch_stream()
        self.slots = [_Slot(torch, dev, self.world, max_send, max_recv, self.pitch) for _ in range(max(2, int(depth)))]
        self.turn = 0
        # gloo (tests, rehearsals on one GPU) moves device tensors through worker threads without
        # blocking the caller, and several such exchanges in flight can pair up differently on two
        # ranks; with gloo the rows are therefore staged through host memory, one blocking call at a time
        self.host_staged = dist.get_backend(group) == "gloo"

    def start(self, d_recs, d_payloads, n):
        """Begin routing n records (rows of the (n, 4) int64 tensor d_recs: dj_join_rec or dj_fork_rec)
        with their messages ((n, pitch) uint8, or None).  The inputs must already be complete on the
        device.  Queues the partition and the exchange of the bucket sizes; does not wait for them.
        Returns a handle for advance() / finish()."""
        torch = self.torch
        s = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        with torch.cuda.stream(self.stream):
            if s.free_pending:                    # the engine may still be reading this buffer set
                self.stream.wait_event(s.free)
                s.free_pending = False
            if s.home_pending:                    # the scatter of this set's previous batch reads origin[]
                self.stream.wait_event(s.home)
                s.home_pending = False
            s.has_pay = d_payloads is not None
            s.matrix, s.round, s.rounds, s.host = None, 0, 0, None
            s.n, s.returned_rounds, s.return_pending = int(n), 0, False
            self.e.partition_joins_dev(n, self.world, d_recs, d_payloads, self.pitch, s.recs_sorted,
                                       s.pay_sorted if s.has_pay else None, s.origin, s.counts,
                                       stream_ptr=self.stream.cuda_stream)
            if self.host_staged:
                return s
            s.parted.record(self.stream)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(s.parted)
            w = self.world
            self.dist.all_gather_into_tensor(s.matrix_d, s.counts[:w], group=self.group)
            s.matrix_h.copy_(s.matrix_d, non_blocking=True)
            s.sizes_in.record(self.comm_stream)
        return s

    def _plan(self, s):
        if s.matrix is not None:
            return
        w = self.world
        if self.host_staged:
            self.stream.synchronize()
            counts = [int(x) for x in s.counts[:w].cpu().tolist()]
            s.matrix = gather_matrix(self.dist, self.group, counts)
            n_out = sum(counts)
            s.host = (s.recs_sorted[:n_out].cpu(), s.pay_sorted[:n_out].cpu() if s.has_pay else None)
        else:
            s.sizes_in.synchronize()                  # in long ago when the start was issued a batch ahead
            s.matrix = s.matrix_h.numpy().astype(np.uint32).reshape(w, w).copy()
        _dbg("bucket sizes", s.matrix.tolist())

    def _exchange(self, s):
        torch = self.torch
        self._plan(s)
        lane = self.stream if self.host_staged else self.comm_stream
        with torch.cuda.stream(lane):
            if s.round > 0:
                # the receive buffers are reused: the engine's kernels that read the previous piece are queued
                s.free.record(self.engine_stream)
                lane.wait_event(s.free)
            if self.host_staged:
                rr = torch.empty((self.max_recv, 4), dtype=torch.int64)
                rp = torch.empty((self.max_recv, self.pitch), dtype=torch.uint8) if s.has_pay else None
                self.stream.synchronize()
                s.n_in, s.rounds = exchange_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round,
                                                  s.host[0], s.host[1], rr, rp)
                s.recv_recs[:s.n_in].copy_(rr[:s.n_in], non_blocking=False)
                if s.has_pay:
                    s.recv_pay[:s.n_in].copy_(rp[:s.n_in], non_blocking=False)
            else:
                s.n_in, s.rounds = exchange_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round,
                                                  s.recs_sorted, s.pay_sorted if s.has_pay else None, s.recv_recs, s.recv_pay)
            s.ready.record(lane)
        s.round += 1

    def _take(self, s):
        """The engine's stream waits for the piece and, when the previous piece's answers are still leaving the status
        buffer, for that."""
        self.engine_stream.wait_event(s.ready)
        if s.return_pending:
            self.engine_stream.wait_event(s.returned)
            s.return_pending = False

    def status_buffer(self, s):
        """Where the engine call that applies the piece handed out last writes its statuses: an int32 tensor of
        max_recv words (pass it as d_status of join_batch_dev / fork_batch_dev), for return_statuses()."""
        return s.status_piece

    def return_statuses(self, s):
        """After the engine call of the piece handed out last has been queued: every status of the piece goes back to
        the rank its record came from (the reference answers every /v1/join and /v1/fork, ring.clj:54-69).  A
        collective: every rank calls it once per piece, before taking the next piece."""
        torch = self.torch
        assert s.returned_rounds + 1 == s.round, "one return per piece, in order"
        lane = self.stream if self.host_staged else self.comm_stream
        with torch.cuda.stream(lane):
            s.free.record(self.engine_stream)             # behind the engine call that writes the statuses
            lane.wait_event(s.free)
            if self.host_staged:
                lane.synchronize()
                back = torch.empty(s.ret_sorted.shape[0], dtype=torch.int32)
                return_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round - 1, s.status_piece.cpu(), back)
                _, send_off, send_cnt, _ = capi.route_plan(s.matrix, self.rank, self.max_recv, s.round - 1)
                starts = np.concatenate([[0], np.cumsum(s.matrix[self.rank].astype(np.int64))[:-1]]).tolist()
                for st, o, c in zip(starts, send_off, send_cnt):
                    if c:
                        s.ret_sorted[st + o:st + o + c].copy_(back[st + o:st + o + c])
            else:
                return_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round - 1, s.status_piece, s.ret_sorted)
            s.returned.record(lane)
        s.return_pending = True
        s.returned_rounds += 1

    def statuses(self, s, d_status_out, wait_stream=None):
        """After the last piece's return_statuses() and before release(): d_status_out[i] (int32 tensor, n words) = the
        answer to record i of the batch this rank gave start(), in its own submission order.  wait_stream (a torch
        stream) is made to wait for them; without one the call itself waits."""
        torch = self.torch
        assert s.returned_rounds == s.rounds and s.rounds > 0, "every piece's statuses must have returned"
        lane = self.stream if self.host_staged else self.comm_stream
        with torch.cuda.stream(lane):
            self.e._check(self.e.L.dj_scatter_statuses(self.e.h, s.n, s.origin.data_ptr(), s.ret_sorted.data_ptr(),
                                                       d_status_out.data_ptr(), lane.cuda_stream))
            s.home.record(lane)
        s.home_pending = True
        if wait_stream is not None:
            wait_stream.wait_event(s.home)
        else:
            s.home.synchronize()

    def advance(self, s):
        """Exchange the first piece now (the host waits for the bucket sizes if they are not in yet).  Optional:
        finish() does it otherwise.  A collective: every rank calls it at the same point of its sequence."""
        if s.round == 0:
            self._exchange(s)

    def finish(self, s):
        """-> (records for this rank, messages for this rank or None, n_received) of the first piece.  The
        engine's stream waits for the exchange; the host does not (beyond the bucket sizes)."""
        self.advance(s)
        self._take(s)
        return s.recv_recs, (s.recv_pay if s.has_pay else None), s.n_in

    def more(self, s):
        """True while pieces are left (the same answer on every rank)."""
        return s.round < s.rounds

    def next_round(self, s):
        """The next piece, into the same buffers: call after the engine call that consumes the previous
        piece has been queued.  A collective."""
        self._exchange(s)
        self._take(s)
        return s.recv_recs, (s.recv_pay if s.has_pay else None), s.n_in

    def release(self, s):
        """Call after the kernels that read the routed batch have been queued on the engine's stream."""
        s.free.record(self.engine_stream)
        s.free_pending = True

    def route(self, d_recs, d_payloads, n):
        """start + finish in one call (no overlap): -> (records, messages, n_received, handle).  The caller
        queues the engine call that reads them, takes further pieces while more(handle), then release(handle)."""
        self.torch.cuda.current_stream().synchronize()
        s = self.start(d_recs, d_payloads, n)
        out = self.finish(s)
        return out + (s,)
