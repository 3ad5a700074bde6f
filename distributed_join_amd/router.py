"""Multi-GPU routing: barriers shard by fingerprint over the ranks of one node; a batch of
join (or fork) records is routed to the owning GPUs with one all-to-all-v.

The reference has no counterpart (its replicas share one Redis, reference
src/distributed_join/redis.clj:13-15); the exchange is the only collective on the path —
sweeps, gathers, acks and recovers are GPU-local.

route() = partition (HIP kernels, dj_partition_joins: stable, so every owner sees the
events of one source in array order) + counts exchange + two all_to_all_single calls
(records, messages).  The receiver's order is: source rank 0's events, then rank 1's, ...
"""
import numpy as np


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
    n_in, n_out = sum(recv_counts), sum(send_counts)
    if n_in > recv_recs.shape[0]:
        raise RuntimeError(f"router: {n_in} incoming records exceed the receive buffer ({recv_recs.shape[0]})")
    dist.all_to_all_single(recv_recs[:n_in], recs_sorted[:n_out], output_split_sizes=recv_counts,
                           input_split_sizes=list(send_counts), group=group)
    if pay_sorted is not None:
        dist.all_to_all_single(recv_pay[:n_in], pay_sorted[:n_out], output_split_sizes=recv_counts,
                               input_split_sizes=list(send_counts), group=group)
    return n_in, recv_counts


class Router:
    def __init__(self, engine, group, max_send, max_recv, pitch):
        import torch
        import torch.distributed as dist
        self.e, self.group, self.dist = engine, group, dist
        self.world = dist.get_world_size(group)
        self.pitch = int(pitch)
        dev = engine.device
        self.recs_sorted = torch.empty((max_send, 4), dtype=torch.int64, device=dev)
        self.pay_sorted = torch.empty((max_send, self.pitch), dtype=torch.uint8, device=dev)
        self.origin = torch.empty(max_send, dtype=torch.int32, device=dev)
        self.counts = torch.empty(64, dtype=torch.int32, device=dev)
        self.recv_recs = torch.empty((max_recv, 4), dtype=torch.int64, device=dev)
        self.recv_pay = torch.empty((max_recv, self.pitch), dtype=torch.uint8, device=dev)

    def route(self, d_recs, d_payloads, n):
        """d_recs: (n, 4) int64 rows = 32-byte records whose first 16 bytes are the fingerprint
        (dj_join_rec or dj_fork_rec); d_payloads: (n, pitch) uint8 or None.
        -> (records for this rank, messages for this rank, n_received)."""
        import torch
        torch.cuda.current_stream().synchronize()
        self.e.partition_joins_dev(n, self.world, d_recs, d_payloads, self.pitch, self.recs_sorted,
                                   self.pay_sorted if d_payloads is not None else None, self.origin, self.counts)
        self.e.sync()
        send_counts = [int(x) for x in self.counts[:self.world].cpu().tolist()]
        n_in, _ = exchange(self.dist, self.group, self.recs_sorted, self.pay_sorted if d_payloads is not None else None,
                           send_counts, self.recv_recs, self.recv_pay)
        torch.cuda.current_stream().synchronize()
        return self.recv_recs, (self.recv_pay if d_payloads is not None else None), n_in
