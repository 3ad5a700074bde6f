"""The N>1 path on CPU: two gloo ranks route a join stream to the owning rank with the router's
all-to-all-v (distributed_join_amd.router.exchange) and apply it to a per-rank store; the union
of what the ranks fire must equal what one store fires on the whole stream.

Half of every rank's events go to ONE barrier (a hot key): its owner receives far more rows than its
receive buffer holds (RECV_CAP), so the exchange takes several rounds, planned identically on both ranks.

Every event's answer (WAITING / COMPLETE / 404 for an id nobody forked / 400 for a position beyond the count) returns
to the rank that submitted it over the same plan read backwards (router.return_round), and each rank's answers, in its own
submission order, must be what one store answers for that rank's part of the concatenated stream (the reference
answers every request: ring.clj:54-69, core.clj:52-58, redis.clj:152-159).

The HIP partition kernel (dj_partition_joins) is replaced here by a numpy stable partition with
the same contract; the per-rank store is the CPU oracle (this is a test of the routing layer,
the GPU parity tests cover the engine)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2
NB, C, P = 600, 5, 64
T0 = 1_700_000_000_000
HOT = WORLD * NB      # an extra barrier that receives as many events again as all others (positions 0..C-2: never complete)
GHOSTS = 7            # ids behind HOT that nobody forks: joins to them answer 404
RECV_CAP = 1000       # rows per round a rank can take: the hot key's owner needs several rounds


def gen_stream(rank):
    """Rank r generates the events of barriers [r*NB, (r+1)*NB), shuffled, with a few replays."""
    rng = np.random.default_rng(100 + rank)
    ev = rng.permutation(NB * C)
    ev = np.concatenate([ev, ev[rng.integers(0, NB * C, 40)]])      # at-least-once replays
    b = ev // C + rank * NB
    pos = ev % C
    hot_at = np.sort(rng.integers(0, len(b) + 1, len(b)))          # hot events sprinkled through the stream
    b = np.insert(b, hot_at, HOT)
    pos = np.insert(pos, hot_at, rng.integers(0, C - 1, len(hot_at)))
    bad = rng.integers(0, len(b), 120)                             # some requests are refused
    b[bad[:60]] = HOT + 1 + rng.integers(0, GHOSTS, 60)            # unknown barrier: 404 (redis.clj:152)
    pos[bad[60:]] = C + rng.integers(0, 3, 60)                     # position >= count: 400 (redis.clj:153)
    from tests.parity import payload_bytes
    pay = payload_bytes(b, pos, np.arange(len(b)) + 7 * rank, P)
    return b, pos, pay


def capi_fps(ids):
    from distributed_join_amd import capi
    return capi.fingerprint_many(ids)


def stable_partition(owners, world):
    order = np.argsort(owners, kind="stable")
    counts = np.bincount(owners, minlength=world)
    return order, counts.tolist()


def worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from distributed_join_amd import capi
    from distributed_join_amd.router import exchange_round, gather_matrix, owner_of_np, return_round
    from oracle import oracle_py

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB + 1 + GHOSTS)]
    fps = capi.fingerprint_many(ids)
    owners_all = owner_of_np(fps[:, 1], WORLD)
    assert all(capi.owner_of(int(lo), int(hi), WORLD) == int(o) for (lo, hi), o in zip(fps[:64], owners_all[:64]))
    o = oracle_py.Oracle(ids, max_count=C, payload_stride=P)
    rounds = []

    def route(recs, pay, apply):
        """Partition, exchange in rounds, apply(records, messages) -> statuses per piece, answers back per round.
        -> this rank's answers in its submission order."""
        own = owners_all[recs[:, 0]]
        order, counts = stable_partition(own, WORLD)
        rs = torch.from_numpy(recs[order].copy())
        ps = torch.from_numpy(pay[order].copy()) if pay is not None else None
        rr = torch.zeros((len(recs) + RECV_CAP, recs.shape[1]), dtype=torch.int64)      # own bucket + a round's arrivals
        rp = torch.zeros((len(recs) + RECV_CAP, P), dtype=torch.uint8) if pay is not None else None
        matrix = gather_matrix(dist, dist.group.WORLD, counts)
        ret_sorted = torch.full((len(recs),), -1, dtype=torch.int32)
        rnd, n_rounds = 0, 1
        while rnd < n_rounds:
            n_in, n_rounds, at = exchange_round(dist, dist.group.WORLD, rank, matrix, RECV_CAP, rnd, rs, ps, rr, rp)
            assert n_in <= RECV_CAP
            got = rr[at:at + n_in].numpy()
            assert (owners_all[got[:, 0]] == rank).all()                # only my barriers arrive
            st = apply(got, rp[at:at + n_in].numpy() if pay is not None else None)
            return_round(dist, dist.group.WORLD, rank, matrix, RECV_CAP, rnd, torch.from_numpy(np.asarray(st, dtype=np.int32)), ret_sorted)
            rnd += 1
        rounds.append(n_rounds)
        answers = np.full(len(recs), -1, dtype=np.int32)
        answers[order] = ret_sorted.numpy()                            # origin[j] = order[j]
        return answers

    # forks: every rank forks its own id range, routed to the owners
    mine = np.arange(rank * NB, (rank + 1) * NB + (1 if rank == WORLD - 1 else 0))       # the last rank also forks the hot one
    frecs = np.stack([mine, np.full(len(mine), C)], axis=1).astype(np.int64)
    fork_answers = route(frecs, None, lambda fr, _: o.fork_batch(fr[:, 0], fr[:, 1], np.full(len(fr), T0 + 3_600_000)))
    assert (fork_answers == capi.DJ_FORK_CREATED).all()
    b, pos, pay = gen_stream(rank)
    join_answers = route(np.stack([b, pos], axis=1).astype(np.int64), pay,
                         lambda jr, jp: o.join_batch(jr[:, 0], jr[:, 1], np.full(len(jr), P), np.zeros(len(jr)), jp, T0 + 5))
    assert rounds[-1] >= 2, "the hot key's owner must have needed more than one round (on both ranks the same number)"
    np.save(os.path.join(out_dir, f"answers_{rank}.npy"), join_answers)
    hot_mine = owners_all[HOT] == rank
    hm, hl, _, hp = o.snapshot_one(HOT) if hot_mine else (None, None, None, None)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=4096, cap_positions=4096 * C)
    fired = {}
    if hot_mine:         # what the hot barrier holds position by position (last writer in the global order)
        fired["hot"] = [bytes(hp[p, :int(hl[p])]) if int(hl[p]) != 0xFFFFFFFF else None for p in range(C)]
    for c in conts:
        pb = int(c["pos_base"])
        fired[int(c["id_idx"])] = (int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    np.save(os.path.join(out_dir, f"fired_{rank}.npy"), np.array([fired], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_fire_what_one_store_fires(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    fired = {}
    for r in range(WORLD):
        part = np.load(tmp_path / f"fired_{r}.npy", allow_pickle=True)[0]
        assert not (fired.keys() & part.keys()), "a barrier fired on two ranks"
        fired.update(part)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB + 1 + GHOSTS)]
    o = oracle_lib.Oracle(ids, max_count=C, payload_stride=P)
    o.fork_batch(np.arange(WORLD * NB + 1), np.full(WORLD * NB + 1, C), np.full(WORLD * NB + 1, T0 + 3_600_000))
    # the same requests on one store, in the router's order: an owner's own events first, then the other ranks'
    from distributed_join_amd.router import owner_of_np
    from tests.parity import routed_order
    owners_all = owner_of_np(capi_fps(ids)[:, 1], WORLD)
    streams = [gen_stream(r) for r in range(WORLD)]
    want = [np.zeros(len(st[0]), dtype=np.int32) for st in streams]
    for src, idx in routed_order([owners_all[st[0]] for st in streams]):
        b, pos, pay = streams[src]
        want[src][idx] = o.join_batch(b[idx], pos[idx], np.full(len(idx), P), np.zeros(len(idx)), pay[idx], T0 + 5)
    seen = set()
    for r in range(WORLD):
        # every request's answer came home to the rank that sent it, in that rank's own order
        got = np.load(tmp_path / f"answers_{r}.npy")
        assert got.tolist() == want[r].tolist(), f"rank {r}: answers differ from one store's"
        seen |= set(got.tolist())
    assert seen == {1, 2, 3, 4}, "the streams hold WAITING, COMPLETE, 404 and 400 answers"
    hm, hl, _, hp = o.snapshot_one(HOT)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=4096, cap_positions=4096 * C)
    want = {"hot": [bytes(hp[p, :int(hl[p])]) if int(hl[p]) != 0xFFFFFFFF else None for p in range(C)]}
    for c in conts:
        pb = int(c["pos_base"])
        want[int(c["id_idx"])] = (int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    assert len(want) > WORLD * NB * 0.8         # most barriers fire (a refused request can leave one incomplete), plus the hot one's snapshot
    assert fired == want
