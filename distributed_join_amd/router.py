"""Multi-GPU routing: barriers shard by fingerprint over the ranks of one node; a batch of
join (or fork) records is routed to the owning GPUs with one all-to-all-v.

The reference has no counterpart (its replicas share one Redis, reference
src/distributed_join/redis.clj:13-15); the exchange is the only collective on the path —
sweeps, gathers, acks and recovers are GPU-local.

route = partition (HIP kernels, dj_partition_joins: stable, so every owner sees the events of
one source in array order) + all-gather of the bucket sizes + all_to_all_single calls (records,
messages).  The receiver's order is: its OWN events first (they never cross the exchange: the partition writes the own
bucket straight into the receive buffers, dj_partition_joins_self), then the other source ranks' in rank order.

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


def piece_layout(matrix, rank, max_recv, rnd):
    """Where round `rnd`'s piece of `rank` lies in its receive buffers (rows): -> (rounds, send_off, send_cnt, recv_cnt,
    piece_at, own_off, own_cnt, remote_at).  A receiver's stream is its OWN rows first (they never cross the exchange:
    the partition puts the own bucket at the front of the receive buffers), then the other sources in rank order; what the
    peers send in a round goes right behind the own bucket's end while own rows are part of the piece, else to the front."""
    rounds, send_off, send_cnt, recv_cnt = capi.route_plan(matrix, rank, max_recv, rnd)
    own = int(matrix[rank][rank])
    own_off, own_cnt = send_off[rank], send_cnt[rank]
    has_own = rnd * max_recv < own
    piece_at = own_off if has_own else 0
    return rounds, send_off, send_cnt, recv_cnt, piece_at, own_off, own_cnt, piece_at + own_cnt


def exchange_round(dist, group, rank, matrix, max_recv, rnd, recs_sorted, pay_sorted, recv_recs, recv_pay, own_in_place=False,
                   layout=None):
    """One round of the planned exchange on any torch.distributed backend (RCCL on GPUs, gloo on CPU
    tensors).  recv_recs / recv_pay hold max_send + max_recv rows.  own_in_place: the partition has already put this
    rank's own bucket at the front of the receive buffers (dj_partition_joins_self); otherwise it is copied there from
    recs_sorted / pay_sorted here.  -> (rows of the piece, rounds in all, row of the receive buffers where the piece starts)."""
    import torch
    rounds, send_off, send_cnt, recv_cnt, piece_at, own_off, own_cnt, remote_at = layout or piece_layout(matrix, rank, max_recv, rnd)
    if own_in_place and len(send_cnt) == 1:           # one rank: the partition's output is the piece
        return own_cnt, rounds, piece_at
    bucket = [int(x) for x in matrix[rank]]
    starts = np.concatenate([[0], np.cumsum(bucket)[:-1]]).tolist()
    if not own_in_place and own_cnt:
        a = starts[rank] + own_off
        recv_recs[own_off:own_off + own_cnt] = recs_sorted[a:a + own_cnt]
        if pay_sorted is not None:
            recv_pay[own_off:own_off + own_cnt] = pay_sorted[a:a + own_cnt]
    out_cnt = [0 if p == rank else c for p, c in enumerate(send_cnt)]
    in_cnt = [0 if p == rank else c for p, c in enumerate(recv_cnt)]
    n_remote, n_out = sum(in_cnt), sum(out_cnt)

    def rows(t):
        parts = [t[st + o:st + o + c] for st, o, c in zip(starts, send_off, out_cnt) if c]
        return torch.cat(parts) if parts else t[:0]

    _dbg("round", rnd, "of", rounds, ":", n_out, "->", n_remote, "+ own", own_cnt)
    if dist.get_world_size(group) > 1:
        dist.all_to_all_single(recv_recs[remote_at:remote_at + n_remote], rows(recs_sorted), output_split_sizes=in_cnt,
                               input_split_sizes=out_cnt, group=group)
        if pay_sorted is not None:
            dist.all_to_all_single(recv_pay[remote_at:remote_at + n_remote], rows(pay_sorted), output_split_sizes=in_cnt,
                                   input_split_sizes=out_cnt, group=group)
    return own_cnt + n_remote, rounds, piece_at


def return_round(dist, group, rank, matrix, max_recv, rnd, status_piece, ret_sorted, layout=None):
    """The way back of round `rnd`: status_piece[:n_in] holds one answer per row of that round's piece (own rows first,
    then the peers' in rank order); every answer goes to the rank its row came from and lands in ret_sorted at the place
    the row has in that rank's recs_sorted.  The same plan as exchange_round, read backwards: one all_to_all_single."""
    import torch
    _, send_off, send_cnt, recv_cnt, _, own_off, own_cnt, _ = layout or piece_layout(matrix, rank, max_recv, rnd)
    bucket = [int(x) for x in matrix[rank]]
    starts = [0]
    for c in bucket[:-1]:
        starts.append(starts[-1] + c)
    if own_cnt:                                       # the answers to this rank's own rows stay here
        a = starts[rank] + own_off
        ret_sorted.narrow(0, a, own_cnt).copy_(status_piece.narrow(0, 0, own_cnt), non_blocking=True)
    out_cnt = [0 if p == rank else c for p, c in enumerate(recv_cnt)]     # answers out = rows that came in
    in_cnt = [0 if p == rank else c for p, c in enumerate(send_cnt)]
    n_out, n_in = sum(out_cnt), sum(in_cnt)
    if dist.get_world_size(group) == 1:
        return
    back = torch.empty(n_in, dtype=ret_sorted.dtype, device=ret_sorted.device)
    dist.all_to_all_single(back, status_piece[own_cnt:own_cnt + n_out], output_split_sizes=in_cnt, input_split_sizes=out_cnt, group=group)
    at = 0
    for st, o, c in zip(starts, send_off, in_cnt):
        if c:
            ret_sorted[st + o:st + o + c] = back[at:at + c]
            at += c


def exchange(dist, group, recs_sorted, pay_sorted, send_counts, recv_recs, recv_pay, max_recv):
    """all-to-all-v of records (rows of recs_sorted, grouped by destination rank) and messages (rows of
    pay_sorted, optional) in as many rounds as receivers with max_recv rows per round need; recv_recs / recv_pay are
    scratch of at least len(recs_sorted) + max_recv rows.  -> [(n_received, records copy, messages copy or None), ...]
    in application order."""
    rank = dist.get_rank(group)
    matrix = gather_matrix(dist, group, send_counts, device=recs_sorted.device)
    out, rnd, rounds = [], 0, 1
    while rnd < rounds:
        n_in, rounds, at = exchange_round(dist, group, rank, matrix, max_recv, rnd, recs_sorted, pay_sorted, recv_recs, recv_pay)
        out.append((n_in, recv_recs[at:at + n_in].clone(), recv_pay[at:at + n_in].clone() if pay_sorted is not None else None))
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
        # this rank's own bucket (written here by the partition) with a round's arrivals right behind it
        self.recv_recs = torch.empty((max_send + max_recv, 4), dtype=torch.int64, device=dev)
        self.recv_pay = torch.empty((max_send + max_recv, pitch), dtype=torch.uint8, device=dev)
        self.piece_at = 0                    # row of the receive buffers where the current piece starts
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
        self.layout = None                   # piece_layout of the round exchanged last


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
        if s.free_pending:                    # the engine may still be reading this buffer set
            self.stream.wait_event(s.free)
            s.free_pending = False
        if s.home_pending:                    # the scatter of this set's previous batch reads origin[]
            self.stream.wait_event(s.home)
            s.home_pending = False
        s.has_pay = d_payloads is not None
        s.matrix, s.round, s.rounds, s.host, s.layout = None, 0, 0, None, None
        s.n, s.returned_rounds, s.return_pending = int(n), 0, False
        self.e.partition_joins_dev(n, self.world, d_recs, d_payloads, self.pitch, s.recs_sorted,
                                   s.pay_sorted if s.has_pay else None, s.origin, s.counts,
                                   stream_ptr=self.stream.cuda_stream, self_rank=self.rank,
                                   d_recs_self=s.recv_recs, d_payloads_self=s.recv_pay if s.has_pay else None)
        if self.host_staged:
            return s
        s.parted.record(self.stream)
        self.comm_stream.wait_event(s.parted)
        w = self.world
        with torch.cuda.stream(self.comm_stream):
            if w == 1:                        # nobody to ask: this rank's sizes are the matrix
                s.matrix_h.copy_(s.counts[:1], non_blocking=True)
            else:
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
            s.host = (s.recs_sorted[:n_out].cpu(), s.pay_sorted[:n_out].cpu() if s.has_pay else None)    # (the own bucket's rows in it are stale: never read)
        else:
            s.sizes_in.synchronize()                  # in long ago when the start was issued a batch ahead
            s.matrix = s.matrix_h.numpy().view(np.uint32).reshape(w, w)      # (pinned memory of this set: stays until the set's next start)
        _dbg("bucket sizes", s.matrix.tolist())

    def _exchange(self, s):
        torch = self.torch
        self._plan(s)
        lane = self.stream if self.host_staged else self.comm_stream
        if s.round > 0:
            # the receive buffers are reused: the engine's kernels that read the previous piece are queued
            s.free.record(self.engine_stream)
            lane.wait_event(s.free)
        s.layout = piece_layout(s.matrix, self.rank, self.max_recv, s.round)
        if self.host_staged:
            with torch.cuda.stream(lane):
                rows = s.recv_recs.shape[0]
                rr = torch.empty((rows, 4), dtype=torch.int64)
                rp = torch.empty((rows, self.pitch), dtype=torch.uint8) if s.has_pay else None
                self.stream.synchronize()
                s.n_in, s.rounds, s.piece_at = exchange_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round,
                                                              s.host[0], s.host[1], rr, rp, own_in_place=True, layout=s.layout)
                # what came from the peers lies behind this round's own rows (which the partition put on the device)
                a, b = s.piece_at + s.layout[6], s.piece_at + s.n_in
                s.recv_recs[a:b].copy_(rr[a:b], non_blocking=False)
                if s.has_pay:
                    s.recv_pay[a:b].copy_(rp[a:b], non_blocking=False)
        elif self.world == 1:
            s.n_in, s.rounds, s.piece_at = s.layout[6], s.layout[0], s.layout[4]     # nothing to exchange
        else:
            with torch.cuda.stream(lane):
                s.n_in, s.rounds, s.piece_at = exchange_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round,
                                                              s.recs_sorted, s.pay_sorted if s.has_pay else None, s.recv_recs,
                                                              s.recv_pay, own_in_place=True, layout=s.layout)
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
        s.free.record(self.engine_stream)                 # behind the engine call that writes the statuses
        lane.wait_event(s.free)
        with torch.cuda.stream(lane):
            if self.host_staged:
                lane.synchronize()
                back = s.ret_sorted.cpu()
                return_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round - 1, s.status_piece.cpu(), back, layout=s.layout)
                s.ret_sorted.copy_(back)
            else:
                return_round(self.dist, self.group, self.rank, s.matrix, self.max_recv, s.round - 1, s.status_piece, s.ret_sorted, layout=s.layout)
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
        return s.recv_recs[s.piece_at:], (s.recv_pay[s.piece_at:] if s.has_pay else None), s.n_in

    def more(self, s):
        """True while pieces are left (the same answer on every rank)."""
        return s.round < s.rounds

    def next_round(self, s):
        """The next piece, into the same buffers: call after the engine call that consumes the previous
        piece has been queued.  A collective."""
        self._exchange(s)
        self._take(s)
        return s.recv_recs[s.piece_at:], (s.recv_pay[s.piece_at:] if s.has_pay else None), s.n_in

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
