#!/usr/bin/env python3
"""Throughput bench of the barrier hot path (BASELINE.json metric: join events/sec,
continuations fired/sec, % of HBM roofline).

Workload (BASELINE.json configs[1], per GPU): 1 M barriers x count 8, 64-byte messages,
8 M joins in uniformly shuffled order, no timeouts.  One STEP = one batch of 1 M join
events applied (dj_join_batch), one sweep that pops and gathers every completed barrier
(dj_sweep) and the ack of all of them (dj_ack_batch); every 8th step starts a new job with
the fork of its 1 M barriers (dj_fork_batch), inside the timed region.  All inputs are
resident in HBM before the timed region.

N > 1 (one process per GPU, launched by torch.distributed.run): barriers shard by
fingerprint, every rank generates a uniform slice of the global event stream, events are
routed to their owner with an all-to-all over RCCL, then applied locally (weak scaling:
per-GPU work is fixed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak
HBM_ACHIEVABLE_GBPS = 6290.0   # what a float4 copy reaches on this chip (hardware guide: 79 % of the spec peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--barriers", type=int, default=1 << 20, help="barriers per GPU per job")
    ap.add_argument("--count", type=int, default=8)
    ap.add_argument("--payload", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-timed-ms", type=float, default=200.0,
                    help="the K steps are repeated (whole multiples of K) until the timed region is at least this long")
    ap.add_argument("--config3", type=int, default=-1,
                    help="also run BASELINE configs[2] (Zipf counts, 30 %% of the barriers expire) with this many barriers and "
                         "report it under 'config3' (default: 10 M at N=1 with default sizes, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--hot-frac", type=float, default=0.0,
                    help="experiment: retarget this fraction of the joins to ONE barrier (hot key, replays)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="functional rehearsal of the N>1 path on a box with ONE GPU: all ranks share device 0 and the "
                         "collectives go over gloo (RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--max-message", type=int, default=0,
                    help="experiment: create the engine with messages up to this many bytes accepted (longer than a row: "
                         "blocks of their own); the bench's messages stay inline, the long-message kernels run")
    ap.add_argument("--max-count", type=int, default=0,
                    help="experiment: the engine's :fork-count-max (default: the job's count)")
    ap.add_argument("--join-max-wait", type=int, default=0,
                    help="experiment: every join carries this max-wait in seconds (store-nearest path)")
    ap.add_argument("--zipf-frac", type=float, default=0.0,
                    help="experiment: this fraction of the joins goes to the first 1000 barriers with Zipf(1.2) weights "
                         "(many hot barriers at once; they never complete)")
    ap.add_argument("--recv-cap", type=int, default=0,
                    help="experiment: rows a rank's receive buffers hold (default: 2 x batch); smaller values make the exchange "
                         "take several rounds")
    ap.add_argument("--capacity", type=int, default=0,
                    help="experiment: the engine's capacity_barriers (table slots = next power of two >= 2x), default: --barriers; "
                         "a large value gives a table that no cache holds while the batches stay at --barriers events")
    ap.add_argument("--hot-path", type=int, default=-1, help="experiment: dj_set_hot_path mode (8: never the bucketed join passes, 16: wherever they apply)")
    ap.add_argument("--route", action="store_true",
                    help="run the multi-GPU routing path (partition + all-to-all) even with one rank")
    return ap.parse_args()


DEBUG = bool(os.environ.get("DJ_BENCH_DEBUG"))


def dbg(*a):
    if DEBUG:
        print(f"{time.time() % 1000:8.3f} [rank {os.environ.get('RANK', '0')}]", *a, file=sys.stderr, flush=True)


def make_ids(lo, hi):
    return [f"j{b:x}~t{b:x}" for b in range(lo, hi)]


def cpu_baseline(count, payload):
    """The CPU oracle (a sequential restatement of the reference's core.clj + redis.clj, in-process:
    no Redis round trips, so an upper bound on the reference) on a bounded sample of the workload."""
    from oracle import oracle_py
    nb = 1 << 19          # half of the GPU job: about 10 s of single-thread work
    ids = make_ids(0, nb)
    o = oracle_py.Oracle(ids, max_count=count, payload_stride=payload)
    rng = np.random.default_rng(2)
    ev = rng.permutation(nb * count)
    pay = rng.integers(0, 256, (nb, payload), dtype=np.uint8)
    now = 1_700_000_000_000
    t = time.perf_counter()
    o.fork_batch(np.arange(nb), np.full(nb, count), np.full(nb, now + 3_600_000))
    fired = 0
    for k in range(count):
        now += 1
        part = ev[k * nb:(k + 1) * nb]
        o.join_batch(part // count, part % count, np.full(nb, payload), np.zeros(nb), pay, now)
        conts, _, _, _ = o.sweep(now, cap_conts=nb, cap_positions=nb * count)
        fired += len(conts)
        o.ack_batch(conts, None, now)
    dt = time.perf_counter() - t
    assert fired == nb
    out = {"value": nb * count / dt, "unit": "join events/s", "cores": 1, "kind": "port",
           "continuations_per_s": nb / dt,
           "sample": f"{nb} barriers x count {count}, {payload} B messages, {count} batches of {nb} joins, "
                     "sweep + ack per batch, single thread, in-process (no Redis round trips)"}
    del o
    out["sharded"] = cpu_baseline_sharded(count, payload)
    return out


def cpu_baseline_sharded(count, payload):
    """The same op-log on every host core this process may use: barriers are hash-sharded over T
    independent oracle instances (barrier b -> shard b % T), one thread each (the oracle's C calls
    run without the interpreter lock).  The op-log is partitioned before the clock starts."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py
    T = max(1, min(16, len(os.sched_getaffinity(0))))
    nb = (1 << 20) // T * T       # the GPU job's size
    per = nb // T
    ids = make_ids(0, nb)
    shards = [oracle_py.Oracle(ids[t::T], max_count=count, payload_stride=payload) for t in range(T)]
    rng = np.random.default_rng(3)
    ev = rng.permutation(nb * count)
    pay = rng.integers(0, 256, (2 * per, payload), dtype=np.uint8)
    plan = []      # plan[t][k] = (barrier index inside the shard, position) of batch k
    for t in range(T):
        mine = []
        for k in range(count):
            part = ev[k * nb:(k + 1) * nb]
            b = part // count
            sel = part[b % T == t]
            mine.append(((sel // count) // T, sel % count))
        plan.append(mine)
    now0 = 1_700_000_000_000

    def run(t):
        o, now, fired = shards[t], now0, 0
        o.fork_batch(np.arange(per), np.full(per, count), np.full(per, now + 3_600_000))
        for k in range(count):
            now += 1
            b, p = plan[t][k]
            m = len(b)
            o.join_batch(b, p, np.full(m, payload), np.zeros(m), pay[:m], now)
            conts, _, _, _ = o.sweep(now, cap_conts=per, cap_positions=per * count)
            fired += len(conts)
            o.ack_batch(conts, None, now)
        return fired

    with ThreadPoolExecutor(T) as ex:
        t0 = time.perf_counter()
        fired = sum(ex.map(run, range(T)))
        dt = time.perf_counter() - t0
    assert fired == nb
    return {"value": nb * count / dt, "unit": "join events/s", "cores": T, "kind": "port",
            "continuations_per_s": nb / dt,
            "sample": f"{nb} barriers x count {count} hash-sharded over {T} oracle instances / threads, "
                      f"{count} batches, sweep + ack per batch"}


def run_config3(torch, capi, Engine, dev, local, nb):
    """BASELINE configs[2] on one GPU: nb barriers with counts Zipf(1) on [2, 64]; 30 % of them receive only a
    strict subset of their positions and carry a deadline inside the run, so they leave through the timeout
    sweep; batches of 2^22 joins in shuffled order, the clock advances with every batch; one sweep (pop +
    gather) and the ack of everything popped per batch.  Timed: the fork of all barriers, every batch, and the
    sweeps after the last batch that catch the remaining deadlines.  All inputs are generated in HBM beforehand."""
    P, batch = 64, 1 << 22
    T0 = 1_700_000_000_000
    t_gen = time.perf_counter()
    rng = np.random.default_rng(3)
    fps = torch.from_numpy(capi.fingerprint_many([f"j{b:x}~t{b:x}" for b in range(nb)]).view(np.int64)).to(dev)
    k = np.arange(2, 65)
    counts = torch.from_numpy(rng.choice(k, size=nb, p=(1.0 / k) / (1.0 / k).sum())).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    offsets = torch.cumsum(counts, 0) - counts
    cells = int(counts.sum())
    expire = torch.rand(nb, device=dev, generator=g) < 0.3
    cell_b = torch.repeat_interleave(torch.arange(nb, device=dev), counts)
    cell_p = torch.arange(cells, device=dev) - offsets[cell_b]
    # an expiring barrier never gets one (hashed) position and each of its others with probability 1/2
    never = (cell_b * 2654435761 >> 7) % counts[cell_b]
    sent = ~expire[cell_b] | ((cell_p != never) & (torch.rand(cells, device=dev, generator=g) < 0.5))
    ev = torch.nonzero(sent).reshape(-1)
    ev = ev[torch.randperm(len(ev), device=dev, generator=g)]
    n_events = len(ev)
    n_batches = (n_events + batch - 1) // batch
    deadline = torch.where(expire, T0 + 11 + 2 * torch.randint(0, n_batches, (nb,), device=dev, generator=g),
                           torch.full((nb,), T0 + 30 * 86_400_000, device=dev))
    frecs = torch.empty((nb, 4), dtype=torch.int64, device=dev)
    frecs[:, 0], frecs[:, 1], frecs[:, 2], frecs[:, 3] = fps[:, 0], fps[:, 1], counts, deadline
    recs = torch.empty((n_events, 4), dtype=torch.int64, device=dev)
    jb, jp = cell_b[ev], cell_p[ev]
    recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = fps[jb, 0], fps[jb, 1], jp | (P << 32), 0
    del cell_b, cell_p, never, sent, ev, jb, jp, fps, offsets
    pay = torch.randint(0, 256, (batch, P), dtype=torch.uint8, device=dev, generator=g)     # the same bytes for every batch
    slots_pow2 = 1 << (2 * nb - 1).bit_length()
    eng = Engine(capacity_barriers=nb, arena_bytes=int(cells * 1.6 * (P + 8)) + nb * 128 + (256 << 20), max_count=64,
                 payload_stride=P, max_batch=max(nb, batch), device=local)
    # most barriers complete in the last batches (a barrier is complete when the LAST of its shuffled positions
    # arrives), so one sweep may pop a large share of them: buffers for all of them, no sweep is ever cut short
    cap_c = nb + 1024
    cap_p = cells + 64
    d_conts, d_lens, d_out = eng.empty(cap_c * 64), eng.empty(cap_p * 4), eng.empty(cap_p * P)
    d_res = torch.zeros((n_batches + 16, 4), dtype=torch.int32, device=dev)
    d_st = eng.empty(max(nb, batch) * 4)
    d_ack = eng.empty(cap_c * 4)
    d_fst = eng.empty(nb * 4)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen

    def job(profile):
        acc = {}
        sweeps = 0

        def read():
            if profile:
                for name, ms in eng.profile_read().items():
                    acc.setdefault(name, []).append(ms)

        eng.fork_batch_dev(nb, frecs, d_fst)
        read()
        for k in range(n_batches + 8):
            now = T0 + 2 * (k + 1)
            if k < n_batches:
                lo = k * batch
                n = min(batch, n_events - lo)
                eng.join_batch_dev(n, recs[lo:lo + n], pay, P, now, d_st)
            eng.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_out, d_res[k])
            eng.ack_batch_dev(cap_c, d_res[k], d_conts, None, now, d_ack)
            sweeps += 1
            read()
        return acc, sweeps

    job(False)                                     # warm-up: a whole job (the table is empty again afterwards)
    eng.sync()
    st = d_fst[:nb * 4].view(torch.int32)
    assert bool((st == capi.DJ_FORK_CREATED).all()), f"config 3: forks refused: {torch.unique(st, return_counts=True)}"
    assert int(d_res[:, 0].to(torch.int64).sum()) == nb and not int(d_res[:, 1].sum()), \
        f"config 3 warm-up: {int(d_res[:, 0].to(torch.int64).sum())} of {nb} barriers fired, more flags {d_res[:, 1].tolist()}"
    d_res.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, sweeps = job(False)
    eng.sync()
    dt = time.perf_counter() - t0
    fired = int(d_res[:, 0].to(torch.int64).sum())
    assert fired == nb, (fired, nb)
    eng.profile_enable(True)
    acc, _ = job(True)
    eng.profile_enable(False)
    kern = {name: {"avg_ms": round(sum(v) / len(v), 4), "launches": len(v), "total_ms": round(sum(v), 3)} for name, v in acc.items()}
    n_timeout = int(expire.sum())
    # the join kernels of this job against their per-event figure 112 + 2P, as roofline.join_path does for configs[1]
    jt = sum(kern[k]["total_ms"] for k in ("join_link", "join_resolve", "join_complex", "join_copy") if k in kern)
    jp = {"alg_bytes_per_event": 112 + 2 * P, "ms_per_job": round(jt, 3),
          "achieved_GBps": round((112 + 2 * P) * n_events / (jt * 1e-3) / 1e9, 1) if jt else None,
          "frac": round((112 + 2 * P) * n_events / (jt * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if jt else None}
    out = {"workload": f"configs[2]: {nb} barriers, counts Zipf(1) on [2, 64] ({cells} positions), 30 % ({n_timeout}) expire "
                       f"through the timeout sweep with a strict subset of their positions, {P} B messages, {n_batches} batches "
                       f"of {batch} joins + sweep/gather + ack each, clock +2 ms per batch, fork of all barriers included",
           "value": n_events / dt, "unit": "join events/s", "join_events": n_events, "job_ms": dt * 1e3,
           "continuations_per_s": nb / dt, "continuations_fired": fired, "complete": nb - n_timeout, "timeout": n_timeout,
           "sweeps": sweeps, "table_slots": slots_pow2, "generation_s": round(gen_s, 1), "join_path": jp, "kernels": kern}
    eng.close()
    return out


def main():
    a = parse()
    # the one JSON line is all that may appear on stdout: libraries that write there (RCCL prints its version banner when
    # the first communicator comes up) are sent to stderr for the length of the run
    # (a caller that has put its own file object in sys.stdout gets the line there)
    sys.stdout.flush()
    json_fd = None
    if sys.stdout is sys.__stdout__:
        json_fd = os.dup(1)
        os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    routed = world > 1 or a.route
    if world > 1 and a.rehearse_on_one_gpu:
        local = 0
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif a.route:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29400 + os.getpid() % 500}", rank=0, world_size=1,
                                device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dbg("process group up, device", local)

    nb, c, P = a.barriers, a.count, a.payload
    batch = nb                      # events per step per GPU
    steps_per_job = c
    stride = (P + 63) // 64 * 64
    pitch = (P + 15) // 16 * 16

    # ---- synthetic job, resident in HBM ------------------------------------------------
    # global barrier space: world * nb ids; this rank OWNS the ids whose fingerprint maps to it
    # and GENERATES events for the id range [rank*nb, (rank+1)*nb)
    ids = make_ids(rank * nb, (rank + 1) * nb)
    fps = torch.from_numpy(capi.fingerprint_many(ids).view(np.int64)).to(dev)       # (nb, 2)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    perm = torch.randperm(nb * c, device=dev, generator=g)
    b_of, p_of = perm // c, perm % c
    if a.hot_frac > 0:
        # barrier 0 is the hot key: it keeps receiving replays of positions 0..c-2 and never completes
        hot = torch.rand(nb * c, device=dev, generator=g) < a.hot_frac
        b_of = torch.where(hot, torch.zeros_like(b_of), b_of)
        p_of = torch.where(b_of == 0, p_of % (c - 1), p_of)
    if a.zipf_frac > 0:
        wz = 1.0 / torch.arange(1, 1001, dtype=torch.float64, device=dev) ** 1.2
        pick = torch.multinomial(wz / wz.sum(), nb * c, replacement=True, generator=g)
        sel = torch.rand(nb * c, device=dev, generator=g) < a.zipf_frac
        b_of = torch.where(sel, pick, b_of)
        p_of = torch.where(b_of < 1000, p_of % (c - 1), p_of)
    recs = torch.empty((nb * c, 4), dtype=torch.int64, device=dev)                 # dj_join_rec as 4 x i64
    recs[:, 0] = fps[b_of, 0]
    recs[:, 1] = fps[b_of, 1]
    recs[:, 2] = p_of | (P << 32)          # position | len << 32
    recs[:, 3] = a.join_max_wait            # max_wait_s, reserved
    payloads = torch.randint(0, 256, (nb * c, pitch), dtype=torch.uint8, device=dev, generator=g)
    fork_recs = torch.empty((nb, 4), dtype=torch.int64, device=dev)                # dj_fork_rec
    fork_recs[:, 0] = fps[:, 0]
    fork_recs[:, 1] = fps[:, 1]
    fork_recs[:, 2] = c
    fork_recs[:, 3] = 1_700_000_000_000 + 30 * 86_400_000     # deadline far in the future: no timeouts
    del perm, b_of, p_of

    # ---- engine ------------------------------------------------------------------------------
    slab_bytes = 64 * (1 + (8 * (1 << (c - 1).bit_length()) + 63) // 64) + (1 << (c - 1).bit_length()) * stride
    recv_cap = batch if world == 1 else int(batch * 1.25) + 4096       # hash imbalance headroom
    # capacity nb -> 2 * nb slots at every N (a rank owns nb +- 0.3 % of the global barrier space)
    eng = Engine(capacity_barriers=max(nb, a.capacity),
                 arena_bytes=int(nb * (1.0 if world == 1 else 1.3) * (slab_bytes + 64)) + (64 << 20),
                 max_count=max(c, a.max_count),
                 payload_stride=stride, max_batch=max(recv_cap, nb * 2 if world > 1 else nb), device=local,
                 max_message_bytes=a.max_message)
    if a.hot_path >= 0:
        eng.set_hot_path(a.hot_path)
    cap_conts = int(nb * (1.0 if world == 1 else 1.3)) + 1024
    cap_pos = cap_conts * c
    d_conts = eng.empty(cap_conts * capi.CONT.itemsize)
    d_lens = eng.empty(cap_pos * 4)
    d_pay_out = eng.empty(cap_pos * stride)
    # one dj_sweep_result (16 bytes) per step: the sweep writes it, the ack reads its n_conts, the host sums
    # the column after the clock has stopped
    res_cap = a.warmup + a.steps + 4 * steps_per_job + 8
    d_res_all = torch.zeros((res_cap, 4), dtype=torch.int32, device=dev)
    res_i = [0]
    d_status = eng.empty(max(recv_cap, nb * 2) * 4)
    d_ack_status = eng.empty(cap_conts * 4)
    d_status_home = torch.empty(max(batch, nb), dtype=torch.int32, device=dev)     # routed mode: the answers to this rank's own records
    es = eng.torch_stream()

    router = None
    if routed:
        from distributed_join_amd.router import Router
        router = Router(eng, dist.group.WORLD, max_send=batch, max_recv=a.recv_cap or max(recv_cap, nb * 2), pitch=pitch, depth=4)

    dbg("engine and router ready")
    now = [1_700_000_000_000]
    pending = []         # routing handles of the next batches (up to two), started while the current one is applied
    pending_fork = []    # routing handle of the next job's forks, started during the last step of the current job

    def start_route(k):
        lo = (k % steps_per_job) * batch
        return router.start(recs[lo:lo + batch], payloads[lo:lo + batch], batch)

    def step(k, prefetch=2):
        """One step: (fork at job start) + join batch + sweep + ack."""
        nonlocal d_res_all
        s = k % steps_per_job
        now[0] += 1
        dbg("step", k)
        if s == 0:
            if router is None:
                # the hot key (barrier 0) is forked once and stays: re-forking it would clear what the engine has
                # learnt about it, which is not what a barrier under a replay storm looks like
                lo_f = (1 if a.hot_frac > 0 else 1000 if a.zipf_frac > 0 else 0) if k >= steps_per_job else 0
                eng.fork_batch_dev(nb - lo_f, fork_recs[lo_f:], d_status)
            else:
                # every record's answer goes back to the rank that submitted it (the reference answers every request):
                # the engine writes a piece's statuses into the router's buffer, the router returns them per piece and
                # puts them into this rank's submission order at the end (d_status_home)
                h = pending_fork.pop(0) if pending_fork else router.start(fork_recs, None, nb)
                r_recs, _, n_in = router.finish(h)
                eng.fork_batch_dev(n_in, r_recs, router.status_buffer(h))
                router.return_statuses(h)
                while router.more(h):           # a rank that owns more than its buffer holds takes the rest in rounds
                    r_recs, _, n_in = router.next_round(h)
                    eng.fork_batch_dev(n_in, r_recs, router.status_buffer(h))
                    router.return_statuses(h)
                router.statuses(h, d_status_home, wait_stream=router.comm_stream)
                router.release(h)
        lo = s * batch
        if router is None:
            eng.join_batch_dev(batch, recs[lo:lo + batch], payloads[lo:lo + batch], pitch, now[0], d_status)
        else:
            h = pending.pop(0) if pending else start_route(k)
            r_recs, r_pay, n_in = router.finish(h)
            eng.join_batch_dev(n_in, r_recs, r_pay, pitch, now[0], router.status_buffer(h))
            router.return_statuses(h)
            while router.more(h):               # a hot key's owner takes its rows in several rounds (all ranks take part)
                r_recs, r_pay, n_in = router.next_round(h)
                eng.join_batch_dev(n_in, r_recs, r_pay, pitch, now[0], router.status_buffer(h))
                router.return_statuses(h)
            router.statuses(h, d_status_home, wait_stream=router.comm_stream)
            router.release(h)
        d_res = d_res_all[res_i[0] % d_res_all.shape[0]]
        res_i[0] += 1
        eng.sweep_dev(now[0], cap_conts, cap_pos, d_conts, d_lens, None, d_pay_out, d_res)
        eng.ack_batch_dev(cap_conts, d_res, d_conts, None, now[0], d_ack_status)
        if router is not None:
            # the whole step is queued on the engine's stream; now the host can afford to wait for the
            # next batches' bucket sizes (the routing runs on the router's stream meanwhile).  Two batches
            # ahead: the exchange of batch k+1 was started a step ago, so the engine never waits for the
            # host.  The next job's forks are routed during the current job's last step.
            # the batch after this one was started a step ago: its bucket sizes are in, its rows can cross now (queued
            # before the next start, whose all-gather of sizes waits for its partition on the same stream)
            if pending:
                router.advance(pending[0])
            for ahead in range(1, prefetch + 1):
                if len(pending) < ahead:
                    pending.append(start_route(k + len(pending) + 1))
            if pending:
                router.advance(pending[0])      # (already done unless the pipeline was empty)
            if prefetch and s == steps_per_job - 1 and not pending_fork:
                pending_fork.append(router.start(fork_recs, None, nb))
        if a.hot_frac > 0 or a.zipf_frac > 0:
            eng.sync()      # a host that reads each batch's statuses; also lets the engine see the previous
                            # batch's longest chain in time to pick the parallel hot-barrier kernels

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        eng.sync()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    # warmup: whole jobs so the timed region starts at a job boundary with an empty table
    warm = (a.warmup + steps_per_job - 1) // steps_per_job * steps_per_job if a.warmup else 0
    for k in range(warm):
        step(k, prefetch=min(2, warm - 1 - k))
    barrier()
    torch.cuda.synchronize()

    # how many times the K steps are repeated: the timed region should last at least --min-timed-ms (a few
    # milliseconds of kernels would be inside the noise of one box); decided from a trial job, the same on every rank
    t_trial = time.perf_counter()
    for k in range(steps_per_job):
        step(warm + k, prefetch=min(2, steps_per_job - 1 - k))
    barrier()
    trial = torch.tensor([(time.perf_counter() - t_trial) / steps_per_job], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(trial, op=dist.ReduceOp.MAX)
    warm += steps_per_job
    repeats = max(1, int(np.ceil(1.1 * a.min_timed_ms * 1e-3 / (float(trial.item()) * a.steps))))
    timed_steps = repeats * a.steps
    res_cap_needed = timed_steps + 8
    if res_cap_needed > d_res_all.shape[0]:
        d_res_all = torch.zeros((res_cap_needed, 4), dtype=torch.int32, device=dev)
        res_cap = res_cap_needed
    res_i[0] = 0

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    res_first = res_i[0]
    t0 = time.perf_counter()
    ev0.record(es)
    for k in range(timed_steps):
        # step numbers go on from the warm-up's (whole jobs): a barrier forked once stays forked once
        step(warm + k, prefetch=min(2, timed_steps - 1 - k))
    ev1.record(es)
    issue = time.perf_counter() - t0      # when the host had queued the last step: close to `wall` = the host is the limit
    barrier()
    wall = time.perf_counter() - t0
    elapsed = torch.tensor([wall], dtype=torch.float64, device=dev)
    fired = d_res_all[res_first:res_i[0], 0].to(torch.int64).sum().reshape(1)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(fired)
    elapsed_s = float(elapsed.item())
    gpu_ms = ev0.elapsed_time(ev1)

    # ---- per-kernel timing leg (after the timed region): one more job with HIP events around
    # every kernel on the engine's stream ------------------------------------------------------------
    kernels = {}
    roofline = None
    acc = {}
    if not a.no_profile:
        # every rank runs these steps (the routed path is a collective); only rank 0 times its kernels
        done = timed_steps
        # finish the job in flight so the profiled job starts on an empty table
        while done % steps_per_job:
            step(warm + done, prefetch=0)
            done += 1
        eng.sync()
        if rank == 0:
            eng.profile_enable(True)
        for k in range(steps_per_job):
            step(warm + done + k, prefetch=0)        # not the first job: a hot key, if any, is not forked again
            if rank == 0:
                for name, ms in eng.profile_read().items():
                    acc.setdefault(name, []).append(ms)
        if rank == 0:
            eng.profile_enable(False)
    if acc:
        fired_job = nb   # per rank at N=1; at N>1 about nb
        units = {   # algorithmic bytes per launch: per-unit figure x units.  Per event: 32 B header + 64 B table line
            # (link), 16 B presence update (resolve), message in + out (copy); per fired barrier 96 + 2cP;
            # per inserted barrier 160; per slot scanned 16
            # (when the link pass copies the messages itself there is no copy launch and the 2P go to the link)
            "join_link": (96 if "join_copy" in acc else 96 + 2 * P, batch), "join_resolve": (16, batch), "join_copy": (2 * P, batch),
            "join_complex": (0, 0), "pop": (16, eng.stats()["slots"]), "gather": (96 + 2 * c * P, fired_job / steps_per_job),
            "fork_claim": (96, nb), "fork_apply": (64, nb), "ack": (128, fired_job / steps_per_job),
            "compact": (2 * 72, eng.stats()["slots"])}     # table hygiene: every slot + queue word read and rewritten
        for name, v in acc.items():
            avg = sum(v) / len(v)
            per, n_units = units.get(name, (0, 0))
            gbps = per * n_units / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            kernels[name] = {"avg_ms": round(avg, 4), "launches": len(v), "total_ms": round(sum(v), 4),
                             "alg_bytes_per_launch": int(per * n_units), "achieved_GBps": round(gbps, 1)}
        dom = max((k for k in kernels if kernels[k]["alg_bytes_per_launch"] > 0), key=lambda k: kernels[k]["total_ms"])
        traffic = traffic_detail = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and (nb, c, P, world) == (1 << 20, 8, 64, 1) and not routed and a.hot_frac == 0 and not a.max_message and not a.max_count and not a.join_max_wait and not a.zipf_frac and not a.capacity and a.hot_path < 0:
            # the counters were collected on this workload
            with open(tf) as f:
                traffic_detail = json.load(f).get(dom)
            traffic = traffic_detail.get("hbm_bytes") if traffic_detail else None     # HBM bytes per launch
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(kernels[dom]["achieved_GBps"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": (None if traffic is None else
                                       "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                                       "collected in an earlier run (" + str((traffic_detail or {}).get("collected", "unlabelled collection")) +
                                       "), average per launch; not measured in this run"),
                    "avg_launch_ms": kernels[dom]["avg_ms"],
                    # what the memory actually moved for this kernel (counter bytes of the earlier collection over this
                    # run's launch time): bytes the algorithm does not need count here, so this is load on the HBM, not credit
                    "traffic_GBps": None if traffic is None else round(traffic / (kernels[dom]["avg_ms"] * 1e-3) / 1e9, 1),
                    "traffic_frac_of_peak": None if traffic is None else round(traffic / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "alg_bytes_per_launch": kernels[dom]["alg_bytes_per_launch"], "traffic_detail": traffic_detail}
        # measured streaming ceiling of this GPU beside the spec peak: the library's own 16-bytes-per-lane HIP copy kernel
        # over 1 GiB (dj_measure_copy), and what the hardware guide measured for such a copy on this chip
        roofline["measured_copy_GBps"] = round(eng.measure_copy(1 << 30, 5), 1) if hasattr(eng.L, "dj_measure_copy") else None    # (an older build under A/B lacks it)
        roofline["measured_copy_kernel"] = "dj_measure_copy: one uint4 load and store per lane, the grid covers the buffer (the fastest of the shapes in tools/copy_sweep.hip), 1 GiB, read + written bytes"
        roofline["peak_achievable"] = HBM_ACHIEVABLE_GBPS
        # the whole join path (link + resolve + copy) against its per-event figure 112 + 2P
        jt = sum(kernels[k]["avg_ms"] for k in ("join_link", "join_resolve", "join_complex", "join_copy") if k in kernels)
        roofline["join_path"] = {"alg_bytes_per_event": 112 + 2 * P, "ms_per_batch": round(jt, 4),
                                 "achieved_GBps": round((112 + 2 * P) * batch / (jt * 1e-3) / 1e9, 1),
                                 "frac": round((112 + 2 * P) * batch / (jt * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

    if roofline is not None:
        # the whole step against the end-to-end figure of one join event on a fire-everything stream:
        # 112 + 4P + 256/c bytes (join path 112 + 2P, gather 2P + 96/c, insert 160/c)
        step_bytes = 112 + 4 * P + 256.0 / c
        step_gbps = step_bytes * batch / (elapsed_s / timed_steps) / 1e9
        roofline["step"] = {"alg_bytes_per_event": round(step_bytes, 1), "ms_per_step": round(elapsed_s * 1e3 / timed_steps, 4),
                            "achieved_GBps": round(step_gbps, 1), "frac": round(step_gbps / HBM_PEAK_GBPS, 4),
                            "frac_of_achievable": round(step_gbps / HBM_ACHIEVABLE_GBPS, 4)}
        # the metric's "% of HBM roofline" is the whole step's; roofline.frac above is the dominant kernel's own
        roofline["frac_step"] = roofline["step"]["frac"]
        if "join_path" in roofline:
            roofline["frac_join_path"] = roofline["join_path"]["frac"]

    c3 = None
    n_c3 = a.config3 if a.config3 >= 0 else (10_000_000 if (world == 1 and not routed and (nb, c, P) == (1 << 20, 8, 64)
                                                            and not a.hot_frac and not a.zipf_frac) else 0)
    if n_c3 and world == 1:
        eng.close()
        eng = None
        del payloads, recs, fork_recs, d_conts, d_lens, d_pay_out
        torch.cuda.empty_cache()
        c3 = run_config3(torch, capi, Engine, dev, local, n_c3)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(c, P)

    if rank == 0:
        total_events = timed_steps * batch * world
        out = {
            "metric": "join events/sec + continuations fired/sec at 1/2/4/8 GPUs; % HBM roofline",
            "value": total_events / elapsed_s, "unit": "join events/s", "n_gpus": world, "steps": a.steps,
            "warmup": warm, "ms_per_step": elapsed_s * 1e3 / timed_steps, "higher_is_better": True,
            "repeats": repeats, "timed_steps": timed_steps, "timed_region_ms": elapsed_s * 1e3,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64 (integer and byte work)", "data": "synthetic",
            "config": {"workload": f"configs[1]: {nb} barriers x count {c} per GPU, {P} B messages, uniform arrival, "
                                   f"no timeouts; step = {batch} joins + sweep/gather + ack, fork of {nb} barriers every "
                                   f"{steps_per_job} steps", "barriers_per_gpu": nb, "count": c, "payload_bytes": P,
                       "batch": batch,
                       "sharding": "none" if not routed else f"fingerprint hash over {world} GPU(s), partition + all-to-all"
                                   + (" (REHEARSAL: all ranks on one GPU, gloo)" if a.rehearse_on_one_gpu else "")},
            "continuations_per_s": float(fired.item()) / elapsed_s,
            "continuations_fired": int(fired.item()),
            "gpu_stream_ms": gpu_ms, "host_issue_ms_per_step": issue * 1e3 / timed_steps,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu, "config3": c3,
        }
        sys.stdout.flush()
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
    if routed:
        dist.destroy_process_group()
    if eng is not None:
        eng.close()


if __name__ == "__main__":
    main()
