"""The N>1 path with two PROCESSES that each drive an engine on the GPU (both on device 0: the box
has one): every rank partitions its event stream with the HIP kernels (dj_partition_joins),
the buckets cross ranks with the router's all-to-all-v over gloo (staged through host memory
here, because two ranks of one RCCL communicator cannot share a device), and each rank applies
what it owns.  The union of what the two engines fire must be what the oracle fires on the whole
stream in the router's global order (rank 0's events, then rank 1's)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2
NB, C, P = 3000, 6, 64
T0 = 1_700_000_000_000


def gen_stream(rank):
    rng = np.random.default_rng(500 + rank)
    ev = rng.permutation(NB * C)
    ev = np.concatenate([ev, ev[rng.integers(0, NB * C, 300)]])      # replays: last writer wins
    b = ev // C + rank * NB
    pos = ev % C
    from tests.parity import payload_bytes
    return b, pos, payload_bytes(b, pos, np.arange(len(ev)) + 11 * rank, P)


def worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine
    from distributed_join_amd.router import exchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    dev = torch.device("cuda", 0)
    e = Engine(capacity_barriers=WORLD * NB, arena_bytes=64 << 20, max_count=C, payload_stride=P, max_batch=1 << 16,
               host_cap_conts=1 << 14, host_cap_positions=1 << 17)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB)]
    fps = capi.fingerprint_many(ids)

    def route(recs, pay):
        """recs: structured FORK_REC / JOIN_REC array; -> (records owned by this rank, their messages)"""
        n = len(recs)
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
        d_pay = torch.from_numpy(pay).to(dev) if pay is not None else None
        d_ro = torch.empty_like(d_recs)
        d_po = torch.empty_like(d_pay) if pay is not None else None
        d_origin = torch.empty(n, dtype=torch.int32, device=dev)
        d_counts = torch.zeros(64, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.partition_joins_dev(n, WORLD, d_recs, d_pay, P, d_ro, d_po, d_origin, d_counts)
        e.sync()
        counts = d_counts.cpu().tolist()[:WORLD]
        rs = d_ro.cpu().view(torch.int64).reshape(n, 4)
        ps = d_po.cpu() if pay is not None else None
        rr = torch.zeros((n * WORLD + 64, 4), dtype=torch.int64)
        rp = torch.zeros((n * WORLD + 64, P), dtype=torch.uint8) if pay is not None else None
        n_in, rc = exchange(dist, dist.group.WORLD, rs, ps, counts, rr, rp)
        return rr[:n_in].contiguous(), (rp[:n_in].contiguous() if pay is not None else None), n_in

    mine = np.arange(rank * NB, (rank + 1) * NB)
    frecs = np.zeros(NB, dtype=capi.FORK_REC)
    frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[mine, 0], fps[mine, 1], C, T0 + 3_600_000
    fr, _, n_f = route(frecs, None)
    st = e.fork_batch(fr.numpy().view(capi.FORK_REC).reshape(-1))
    assert (st == capi.DJ_FORK_CREATED).all()
    b, pos, pay = gen_stream(rank)
    jrecs = np.zeros(len(b), dtype=capi.JOIN_REC)
    jrecs["fp_lo"], jrecs["fp_hi"], jrecs["position"], jrecs["len"] = fps[b, 0], fps[b, 1], pos, P
    jr, jp, n_j = route(jrecs, pay)
    got = jr.numpy().view(capi.JOIN_REC).reshape(-1)
    assert all(capi.owner_of(int(r["fp_lo"]), int(r["fp_hi"]), WORLD) == rank for r in got[:200])
    d_st = e.empty(n_j * 4)
    e.join_batch_dev(n_j, jr.to(dev), jp.to(dev), P, T0 + 5, d_st)
    e.sync()
    conts, lens, _, payo, more = e.sweep(T0 + 10)
    assert not more
    by_fp = {(int(lo), int(hi)): k for k, (lo, hi) in enumerate(fps)}
    fired = {}
    for c in conts:
        pb = int(c["pos_base"])
        fired[by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))]] = (
            int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    np.save(os.path.join(out_dir, f"fired_{rank}.npy"), np.array([fired], dtype=object), allow_pickle=True)
    e.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_engine_processes_fire_what_one_store_fires(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    fired = {}
    for r in range(WORLD):
        part = np.load(tmp_path / f"fired_{r}.npy", allow_pickle=True)[0]
        assert not (fired.keys() & part.keys()), "a barrier fired on two ranks"
        fired.update(part)
    ids = [f"j{b:x}~t{b:x}" for b in range(WORLD * NB)]
    o = oracle_lib.Oracle(ids, max_count=C, payload_stride=P)
    o.fork_batch(np.arange(WORLD * NB), np.full(WORLD * NB, C), np.full(WORLD * NB, T0 + 3_600_000))
    for r in range(WORLD):
        b, pos, pay = gen_stream(r)
        o.join_batch(b, pos, np.full(len(b), P), np.zeros(len(b)), pay, T0 + 5)
    conts, lens, mws, payo = o.sweep(T0 + 10, cap_conts=1 << 14, cap_positions=1 << 17)
    want = {}
    for c in conts:
        pb = int(c["pos_base"])
        want[int(c["id_idx"])] = (int(c["flags"]), [bytes(payo[pb + p, :int(lens[pb + p])]) for p in range(C)])
    assert len(want) == WORLD * NB
    assert fired == want
