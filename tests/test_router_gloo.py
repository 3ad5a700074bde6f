"""The N>1 path on CPU: two gloo ranks route a join stream to the owning rank with the router's
all-to-all-v (distributed_join_amd.router.exchange) and apply it to a per-rank store; the union
of what the ranks fire must equal what one store fires on the whole stream.

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


def gen_stream(rank):
    """Rank r generates the events of barriers [r*NB, (r+1)*NB), shuffled, with a few replays."""
    rng = np.random.default_rng(100 + rank)
    ev = rng.permutation(NB * C)
    ev = np.concatenate([ev, ev[rng.integers(0, NB * C, 40)]])      # at-least-once replays
    b = ev // C + rank * NB
    pos = ev % C
    from tests.parity import payload_bytes
    pay = payload_bytes(b, pos, np.arange(len(ev)) + 7 * rank, P)
    return b, pos, pay


def stable_partition(owners, world):
    order = np.argsort(owners, kind="stable")
    counts = np.bincount(owners, minlength=world)
    return order, counts.tolist()


def worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from distributed_join_amd import capi
    from distributed_join_amd.router import exchange, owner_of_np
    from oracle import oracle_py

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB)]
    fps = capi.fingerprint_many(ids)
    owners_all = owner_of_np(fps[:, 1], WORLD)
    assert all(capi.owner_of(int(lo), int(hi), WORLD) == int(o) for (lo, hi), o in zip(fps[:64], owners_all[:64]))
    o = oracle_py.Oracle(ids, max_count=C, payload_stride=P)

    def route(recs, pay):
        own = owners_all[recs[:, 0]]
        order, counts = stable_partition(own, WORLD)
        rs = torch.from_numpy(recs[order].copy())
        ps = torch.from_numpy(pay[order].copy()) if pay is not None else None
        rr = torch.zeros((len(recs) * WORLD + 64, recs.shape[1]), dtype=torch.int64)
        rp = torch.zeros((len(recs) * WORLD + 64, P), dtype=torch.uint8) if pay is not None else None
        n_in, rc = exchange(dist, dist.group.WORLD, rs, ps, counts, rr, rp)
        assert n_in == sum(rc)
        got = rr[:n_in].numpy()
        assert (owners_all[got[:, 0]] == rank).all()                # only my barriers arrive
        return got, (rp[:n_in].numpy() if pay is not None else None)

    # forks: every rank forks its own id range, routed to the owners
    mine = np.arange(rank * NB, (rank + 1) * NB)
    frecs = np.stack([mine, np.full(NB, C)], axis=1).astype(np.int64)
    fr, _ = route(frecs, None)
    o.fork_batch(fr[:, 0], fr[:, 1], np.full(len(fr), T0 + 3_600_000))
    b, pos, pay = gen_stream(rank)
    jr, jp = route(np.stack([b, pos], axis=1).astype(np.int64), pay)
    o.join_batch(jr[:, 0], jr[:, 1], np.full(len(jr), P), np.zeros(len(jr)), jp, T0 + 5)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=4096, cap_positions=4096 * C)
    fired = {}
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
    # the same global stream on one store: rank 0's events first, then rank 1's
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB)]
    o = oracle_lib.Oracle(ids, max_count=C, payload_stride=P)
    o.fork_batch(np.arange(WORLD * NB), np.full(WORLD * NB, C), np.full(WORLD * NB, T0 + 3_600_000))
    for r in range(WORLD):
        b, pos, pay = gen_stream(r)
        o.join_batch(b, pos, np.full(len(b), P), np.zeros(len(b)), pay, T0 + 5)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=4096, cap_positions=4096 * C)
    want = {}
    for c in conts:
        pb = int(c["pos_base"])
        want[int(c["id_idx"])] = (int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    assert len(want) == WORLD * NB
    assert fired == want
