"""The N>1 path with two (and eight) PROCESSES that each drive an engine on the GPU (all on device 0: the box
has one): every rank routes its event stream with the Router (HIP partition kernels, dj_partition_joins; the
buckets cross ranks with the planned all-to-all-v over gloo, staged through host memory here because
two ranks of one RCCL communicator cannot share a device), and each rank applies what it owns.  One
barrier receives as many events again as all the others together, so its owner takes its rows in several
rounds (the receive buffers hold MAX_RECV rows).  The union of what the two engines fire, and what the hot
barrier holds, must be what the oracle has after the whole stream in the router's order (for every owner: its
own events first, then the other ranks' in rank order).  Every request's answer (WAITING, COMPLETE, 404 for an id nobody forked, 400 for a
position beyond the count) is computed by the owner's engine and returns to the rank that submitted the request
(Router.return_statuses / statuses): each rank's answers, in its own submission order, must be the oracle's for
that rank's part of the stream (ring.clj:54-69, core.clj:52-58, redis.clj:152-159)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, P = 6, 64
T0 = 1_700_000_000_000
MAX_RECV = 8192
GHOSTS = 5            # ids behind the hot barrier that nobody forks


def shape(world):
    """-> (barriers per rank, index of the hot barrier: positions 0..C-2 only, it never completes)"""
    nb = 3000 if world <= 2 else 1000
    return nb, world * nb


def gen_stream(rank, world):
    NB, HOT = shape(world)
    rng = np.random.default_rng(500 + rank)
    ev = rng.permutation(NB * C)
    ev = np.concatenate([ev, ev[rng.integers(0, NB * C, 300)]])      # replays: last writer wins
    b = ev // C + rank * NB
    pos = ev % C
    hot_at = np.sort(rng.integers(0, len(b) + 1, len(b)))
    b = np.insert(b, hot_at, HOT)
    pos = np.insert(pos, hot_at, rng.integers(0, C - 1, len(hot_at)))
    bad = rng.integers(0, len(b), 400)
    b[bad[:200]] = HOT + 1 + rng.integers(0, GHOSTS, 200)          # 404, redis.clj:152
    pos[bad[200:]] = C + rng.integers(0, 4, 200)                   # 400, redis.clj:153
    from tests.parity import payload_bytes
    return b, pos, payload_bytes(b, pos, np.arange(len(b)) + 11 * rank, P)


def worker(rank, world, port, out_dir):
    WORLD = world
    NB, HOT = shape(world)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine
    from distributed_join_amd.router import Router

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    dev = torch.device("cuda", 0)
    e = Engine(capacity_barriers=WORLD * NB + 1, arena_bytes=64 << 20, max_count=C, payload_stride=P, max_batch=1 << 16,
               host_cap_conts=1 << 14, host_cap_positions=1 << 17)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB + 1 + GHOSTS)]
    fps = capi.fingerprint_many(ids)
    router = Router(e, dist.group.WORLD, max_send=1 << 16, max_recv=MAX_RECV, pitch=P)
    rounds = []

    def routed(recs, pay, apply):
        """Route a structured FORK_REC / JOIN_REC array; apply(records, messages, n, d_status) runs once per piece and
        writes the piece's statuses.  -> this rank's answers in its submission order."""
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev).view(torch.int64).reshape(len(recs), 4)
        d_pay = torch.from_numpy(pay).to(dev) if pay is not None else None
        d_answers = torch.full((len(recs),), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        h = router.start(d_recs, d_pay, len(recs))
        r_recs, r_pay, n_in = router.finish(h)
        k = 1
        apply(r_recs, r_pay, n_in, router.status_buffer(h))
        router.return_statuses(h)
        while router.more(h):
            r_recs, r_pay, n_in = router.next_round(h)
            apply(r_recs, r_pay, n_in, router.status_buffer(h))
            router.return_statuses(h)
            k += 1
        router.statuses(h, d_answers)
        router.release(h)
        rounds.append(k)
        return d_answers.cpu().numpy()

    mine = np.arange(rank * NB, (rank + 1) * NB + (1 if rank == WORLD - 1 else 0))
    frecs = np.zeros(len(mine), dtype=capi.FORK_REC)
    frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[mine, 0], fps[mine, 1], C, T0 + 3_600_000
    def apply_forks(r_recs, r_pay, n, d_st):
        e.fork_batch_dev(n, r_recs, d_st)

    assert (routed(frecs, None, apply_forks) == capi.DJ_FORK_CREATED).all()
    b, pos, pay = gen_stream(rank, world)
    jrecs = np.zeros(len(b), dtype=capi.JOIN_REC)
    jrecs["fp_lo"], jrecs["fp_hi"], jrecs["position"], jrecs["len"] = fps[b, 0], fps[b, 1], pos, P

    def apply_joins(r_recs, r_pay, n, d_st):
        got = r_recs[:min(n, 200)].cpu().numpy().view(capi.JOIN_REC).reshape(-1)
        assert all(capi.owner_of(int(r["fp_lo"]), int(r["fp_hi"]), WORLD) == rank for r in got)
        e.join_batch_dev(n, r_recs, r_pay, P, T0 + 5, d_st)

    answers = routed(jrecs, pay, apply_joins)
    np.save(os.path.join(out_dir, f"answers_{rank}.npy"), answers)
    e.sync()
    by_fp = {(int(lo), int(hi)): k for k, (lo, hi) in enumerate(fps)}
    fired = {"rounds": rounds[-1]}
    if capi.owner_of(int(fps[HOT, 0]), int(fps[HOT, 1]), WORLD) == rank:
        _, hl, _, hp = e.snapshot_one(fps[HOT, 0], fps[HOT, 1])
        fired["hot"] = [bytes(hp[p, :int(hl[p])]) if int(hl[p]) != capi.DJ_NIL else None for p in range(C)]
    conts, lens, _, payo, more = e.sweep(T0 + 10)
    assert not more
    for c in conts:
        pb = int(c["pos_base"])
        fired[by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))]] = (
            int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    np.save(os.path.join(out_dir, f"fired_{rank}.npy"), np.array([fired], dtype=object), allow_pickle=True)
    del router
    e.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("WORLD", [2, 8])
def test_engine_processes_fire_what_one_store_fires(tmp_path, oracle_lib, WORLD):
    """Two ranks, and the eight of a full node (all on the one GPU of the box)."""
    import torch.multiprocessing as mp
    from distributed_join_amd import capi
    NB, HOT = shape(WORLD)
    port = 31500 + os.getpid() % 2000 + WORLD
    mp.spawn(worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    fired = {}
    for r in range(WORLD):
        part = np.load(tmp_path / f"fired_{r}.npy", allow_pickle=True)[0]
        assert part.pop("rounds") >= 2, "the hot barrier's owner needs several rounds, and so every rank takes part in them"
        assert not (fired.keys() & part.keys()), "a barrier fired on two ranks"
        fired.update(part)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB + 1 + GHOSTS)]
    o = oracle_lib.Oracle(ids, max_count=C, payload_stride=P)
    o.fork_batch(np.arange(WORLD * NB + 1), np.full(WORLD * NB + 1, C), np.full(WORLD * NB + 1, T0 + 3_600_000))
    # the same requests on one store, in the router's order: an owner's own events first, then the other ranks'
    from distributed_join_amd.router import owner_of_np
    from tests.parity import routed_order
    owners_all = owner_of_np(capi.fingerprint_many(ids)[:, 1], WORLD)
    streams = [gen_stream(r, WORLD) for r in range(WORLD)]
    want_st = [np.zeros(len(st[0]), dtype=np.int32) for st in streams]
    for src, idx in routed_order([owners_all[st[0]] for st in streams]):
        b, pos, pay = streams[src]
        want_st[src][idx] = o.join_batch(b[idx], pos[idx], np.full(len(idx), P), np.zeros(len(idx)), pay[idx], T0 + 5)
    seen = set()
    for r in range(WORLD):
        got = np.load(tmp_path / f"answers_{r}.npy")
        assert got.tolist() == want_st[r].tolist(), f"rank {r}: the answers that came home differ from one store's"
        seen |= set(got.tolist())
    assert seen == {capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE, capi.DJ_JOIN_NOT_FOUND, capi.DJ_JOIN_BAD_POSITION}
    _, hl, _, hp = o.snapshot_one(HOT)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=1 << 14, cap_positions=1 << 17)
    want = {"hot": [bytes(hp[p, :int(hl[p])]) if int(hl[p]) != 0xFFFFFFFF else None for p in range(C)]}
    for c in conts:
        pb = int(c["pos_base"])
        want[int(c["id_idx"])] = (int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    assert len(want) > WORLD * NB * 0.8          # (a refused request can leave its barrier incomplete)
    assert fired == want
