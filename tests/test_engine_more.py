"""More GPU checks through the C ABI: output caps of the sweep, the routing partition kernel,
BASELINE config #3's shape (Zipf counts, 30 % expire through the timeout sweep) against the
oracle, and config #2 at full size through size-independent properties."""
import os

import numpy as np
import pytest

from distributed_join_amd import capi
from tests.parity import ParityDriver, payload_bytes

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000


def _engine(sweep_mode=1, **kw):
    """sweep_mode 1: sweeps drain the ready list whatever the table size (the op-log tests of
    test_engine_parity.py run both forms); -1 leaves the choice to the engine."""
    from distributed_join_amd.engine import Engine
    e = Engine(**kw)
    e.set_sweep_path(sweep_mode)
    return e


def test_sweep_caps_leave_the_rest_queued(oracle_lib):
    """A sweep pops only what fits the caller's buffers and reports `more`; repeating it drains the
    queue, and the union equals the oracle's single pop."""
    nb = 300
    ids = [f"cap{b}~t" for b in range(nb)]
    counts = np.array([1 + b % 6 for b in range(nb)])
    e = _engine(capacity_barriers=nb, arena_bytes=32 << 20, max_count=8, payload_stride=64, max_batch=4096,
                host_cap_conts=64, host_cap_positions=100)
    o = oracle_lib.Oracle(ids, max_count=8, payload_stride=64)
    d = ParityDriver(e, o, ids)
    try:
        d.fork(np.arange(nb), counts, np.full(nb, T0 + 1000))
        jb = np.arange(nb)
        d.join(jb, np.zeros(nb, np.int64), np.full(nb, 64), np.zeros(nb, np.int64), payload_bytes(jb, jb * 0, jb * 0, 64), T0 + 5)
        oc, ol, om, op_ = o.sweep(T0 + 2000)             # everything timed out
        want = d._canon_oracle(oc, ol, om, op_)
        got, rounds = {}, 0
        while True:
            conts, lens, mws, pay, more = e.sweep(T0 + 2000)
            assert len(conts) <= 64 and len(lens) <= 100
            part = d._canon_engine(conts, lens, mws, pay)
            assert not (got.keys() & part.keys())
            got.update(part)
            rounds += 1
            if not more:
                break
            assert rounds < 100
        assert rounds > 1
        assert got == want
        assert all(v[0][2] == capi.DJ_CONT_TIMEOUT for v in got.values())
    finally:
        e.close()


@pytest.mark.parametrize("n_ranks,n,pitch", [(1, 100_003, 48), (2, 100_003, 48), (3, 100_003, 48), (8, 100_003, 48),
                                            (8, 1, 64), (5, 63, 16), (8, 1025, 64), (64, 70_001, 32), (7, 5_000, 1024),
                                            (4, 3_001, 2064), (8, 1 << 17, 64), (6, 4_097, 0), (64, (1 << 18) - 5, 16), (8, (1 << 20) + 77, 16),
                                            (3, (1 << 21) - 1029, 0)])
def test_partition_kernel_is_a_stable_partition_by_owner(n_ranks, n, pitch):
    """Rows of any width (pitch 0: records only), batches that end inside a tile, a wave or on a tile's edge, as many
    owners as a wave has lanes, and batches large enough for a block to take several tiles."""
    import torch
    from distributed_join_amd.router import owner_of_np
    rng = np.random.default_rng(n_ranks + n)
    e = _engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=1 << 21)
    try:
        recs = np.zeros(n, dtype=capi.JOIN_REC)
        recs["fp_lo"] = rng.integers(2, 1 << 62, n, dtype=np.uint64)
        recs["fp_hi"] = rng.integers(1, 1 << 62, n, dtype=np.uint64)
        recs["position"] = np.arange(n) % 1000
        recs["len"] = 48
        pay = rng.integers(0, 256, (n, max(pitch, 16)), dtype=np.uint8)
        dev = e.device
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
        d_pay = torch.from_numpy(pay).to(dev)
        d_ro, d_po = torch.empty_like(d_recs), torch.zeros_like(d_pay)
        d_origin = torch.empty(n, dtype=torch.int32, device=dev)
        d_counts = torch.zeros(64, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if pitch:
            e.partition_joins_dev(n, n_ranks, d_recs, d_pay, pitch, d_ro, d_po, d_origin, d_counts)
        else:
            e.partition_joins_dev(n, n_ranks, d_recs, None, 16, d_ro, None, d_origin, d_counts)
        e.sync()
        owners = owner_of_np(recs["fp_hi"], n_ranks)
        assert all(capi.owner_of(int(r["fp_lo"]), int(r["fp_hi"]), n_ranks) == int(w) for r, w in zip(recs[:200], owners[:200]))
        order = np.argsort(owners, kind="stable")
        np.testing.assert_array_equal(d_counts.cpu().numpy()[:n_ranks], np.bincount(owners, minlength=n_ranks))
        np.testing.assert_array_equal(d_origin.cpu().numpy(), order.astype(np.int32))
        assert bytes(d_ro.cpu().numpy()) == recs[order].tobytes()
        if pitch:
            np.testing.assert_array_equal(d_po.cpu().numpy(), pay[order])
    finally:
        e.close()


def zipf_counts(rng, n, lo=2, hi=64):
    k = np.arange(lo, hi + 1)
    p = 1.0 / k
    return rng.choice(k, size=n, p=p / p.sum())


def test_config3_shape_scaled(oracle_lib):
    """BASELINE config #3 scaled: counts Zipf(1) over [2, 64], 30 % of the barriers get a strict
    subset of their positions and a deadline inside the run; the clock advances 1 ms per batch.
    Expect 70 % COMPLETE + 30 % TIMEOUT continuations with exact presence, bit-exact with the oracle."""
    rng = np.random.default_rng(3)
    nb = 6000
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)]
    counts = zipf_counts(rng, nb)
    expire = rng.random(nb) < 0.3
    e = _engine(capacity_barriers=nb, arena_bytes=256 << 20, max_count=64, payload_stride=64, max_batch=1 << 17,
                host_cap_conts=1 << 14, host_cap_positions=1 << 19)
    o = oracle_lib.Oracle(ids, max_count=64, payload_stride=64)
    d = ParityDriver(e, o, ids)
    try:
        ev_b, ev_p = [], []
        for b in range(nb):
            pos = np.arange(counts[b])
            if expire[b]:
                pos = rng.permutation(pos)[:rng.integers(0, counts[b])]      # strict subset, maybe empty
            ev_b.append(np.full(len(pos), b))
            ev_p.append(pos)
        ev_b, ev_p = np.concatenate(ev_b), np.concatenate(ev_p)
        perm = rng.permutation(len(ev_b))
        ev_b, ev_p = ev_b[perm], ev_p[perm]
        batches = 20
        now = T0
        # deadlines fall on odd milliseconds inside the run, sweeps on even ones: a barrier popped at
        # now == deadline would NOT be a timeout (is-timeout? is strict, core.clj:86-89)
        deadline = np.where(expire, T0 + 11 + 2 * rng.integers(0, batches, nb), T0 + 3_600_000)
        d.fork(np.arange(nb), counts, deadline)
        n_complete = n_timeout = 0
        per = (len(ev_b) + batches - 1) // batches
        for k in range(batches + 15):
            now += 2
            sl = slice(k * per, (k + 1) * per)
            if len(ev_b[sl]):
                jb, jp = ev_b[sl], ev_p[sl]
                d.join(jb, jp, np.full(len(jb), 64), np.zeros(len(jb), np.int64), payload_bytes(jb, jp, jb * 0, 64), now)
            ec, oc = d.sweep(now)
            for c in ec:
                if int(c["flags"]) & capi.DJ_CONT_TIMEOUT:
                    n_timeout += 1
                else:
                    n_complete += 1
            d.ack(ec, oc, lambda k_: True, now)
        assert n_complete == int((~expire).sum())
        assert n_timeout == int(expire.sum())
        assert e.stats()["live"] == 0 and o.live_barriers() == 0
    finally:
        e.close()


@pytest.mark.parametrize("uneven", [0, 200_000])
def test_config2_full_size_properties(uneven):
    """BASELINE config #2 at full size (1 M barriers x count 8, 64 B messages, 8 M joins, batches of
    1 M): every barrier fires exactly once, COMPLETE, with every byte where the generator put it.
    uneven: batches of 1 M + and - that many events in turn (what a rank of a routed job receives is never exactly 2^20)."""
    import torch
    nb, c, P = 1 << 20, 8, 64
    rng = np.random.default_rng(22)
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)]
    fps = capi.fingerprint_many(ids)
    e = _engine(capacity_barriers=nb, arena_bytes=nb * 640 + (64 << 20), max_count=c, payload_stride=P, max_batch=nb + uneven)
    dev = e.device
    try:
        frecs = np.zeros(nb, dtype=capi.FORK_REC)
        frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[:, 0], fps[:, 1], c, T0 + 3_600_000
        st = e.fork_batch(frecs)
        assert (st == capi.DJ_FORK_CREATED).all()
        ev = rng.permutation(nb * c)
        cap_c, cap_p = nb, nb * c
        d_conts, d_lens, d_pay, d_res = e.empty(cap_c * 64), e.empty(cap_p * 4), e.empty(cap_p * P), e.empty(16)
        d_ack = e.empty(cap_c * 4)
        fired = np.zeros(nb, dtype=np.int32)
        lookup = {}
        order = np.lexsort((fps[:, 1], fps[:, 0]))
        sorted_lo, sorted_hi = fps[order, 0], fps[order, 1]
        total_complete = 0
        cuts = [0]
        for k in range(c):
            cuts.append(cuts[-1] + nb + (uneven if k % 2 == 0 else -uneven))
        assert cuts[-1] == nb * c
        for k in range(c):
            now = T0 + 1 + k
            part = ev[cuts[k]:cuts[k + 1]]
            jb, jp = part // c, part % c
            recs = np.zeros(len(part), dtype=capi.JOIN_REC)
            recs["fp_lo"], recs["fp_hi"], recs["position"], recs["len"] = fps[jb, 0], fps[jb, 1], jp, P
            st = e.join_batch(recs, payload_bytes(jb, jp, jb * 0, P), now)
            assert np.isin(st, [capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE]).all()
            total_complete += int((st == capi.DJ_JOIN_COMPLETE).sum())
            res = np.zeros(1, dtype=capi.SWEEP_RESULT)
            e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_pay, d_res, res)
            n, npos = int(res[0]["n_conts"]), int(res[0]["n_positions"])
            assert not res[0]["more"] and npos == n * c
            conts = d_conts[:n * 64].cpu().numpy().view(capi.CONT)
            assert (conts["flags"] == 0).all() and (conts["count"] == c).all()
            # which barrier is each continuation: binary search on the sorted fingerprints
            at = np.searchsorted(sorted_lo, conts["fp_lo"])
            assert (sorted_lo[at] == conts["fp_lo"]).all() and (sorted_hi[at] == conts["fp_hi"]).all()
            bidx = order[at]
            np.add.at(fired, bidx, 1)
            lens = d_lens[:npos * 4].cpu().numpy().view(np.uint32)
            assert (lens == P).all()
            got = d_pay[:npos * P].cpu().numpy().reshape(n, c, P)
            pb = (conts["pos_base"]).astype(np.int64)
            assert sorted(pb.tolist()) == list(range(0, npos, c))
            got = d_pay[:npos * P].cpu().numpy().reshape(npos, P)
            rows = (pb[:, None] + np.arange(c)[None, :]).reshape(-1)
            want = payload_bytes(np.repeat(bidx, c), np.tile(np.arange(c), n), np.zeros(n * c), P)
            assert np.array_equal(got[rows], want), "a gathered byte differs from the generator"
            e.ack_batch_dev(n, None, d_conts, None, now, d_ack)
            e.sync()
            assert (d_ack[:n * 4].cpu().numpy().view(np.int32) == capi.DJ_ACK_DONE).all()
        assert (fired == 1).all()
        assert total_complete == nb
        s = e.stats()
        assert s["live"] == 0 and s["tombstones"] == nb
    finally:
        e.close()


def test_router_single_rank_roundtrip():
    """The multi-GPU routing path with one RCCL rank: partition kernels + all_to_all_single must hand
    every record (and its message) to the only owner, in array order."""
    import os
    import torch
    import torch.distributed as dist
    from distributed_join_amd.router import Router
    n, pitch = 50_000, 64
    rng = np.random.default_rng(9)
    e = _engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=1 << 16)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29600 + os.getpid() % 300}", rank=0, world_size=1,
                            device_id=e.device)
    try:
        recs = torch.from_numpy(rng.integers(2, 1 << 62, (n, 4), dtype=np.int64)).to(e.device)
        pay = torch.from_numpy(rng.integers(0, 256, (n, pitch), dtype=np.uint8)).to(e.device)
        r = Router(e, dist.group.WORLD, max_send=n, max_recv=n, pitch=pitch)
        rr, rp, n_in, h = r.route(recs, pay, n)
        e.sync()
        torch.cuda.synchronize()
        assert n_in == n
        assert torch.equal(rr[:n], recs) and torch.equal(rp[:n], pay)
        r.release(h)
        # overlapped form: two batches in flight on the router's own stream
        h1 = r.start(recs[:1000], None, 1000)
        h2 = r.start(recs[1000:3000], pay[1000:3000], 2000)
        rr1, rp1, n1 = r.finish(h1)
        rr2, rp2, n2 = r.finish(h2)
        e.sync()
        torch.cuda.synchronize()
        assert n1 == 1000 and rp1 is None and torch.equal(rr1[:1000], recs[:1000])
        assert n2 == 2000 and torch.equal(rr2[:2000], recs[1000:3000]) and torch.equal(rp2[:2000], pay[1000:3000])
        r.release(h1)
        r.release(h2)
        h3 = r.start(recs[:500], pay[:500], 500)          # reuses the first buffer set after its release
        rr3, rp3, n3 = r.finish(h3)
        e.sync()
        torch.cuda.synchronize()
        assert n3 == 500 and torch.equal(rr3[:500], recs[:500]) and torch.equal(rp3[:500], pay[:500])
    finally:
        dist.destroy_process_group()
        e.close()


def test_config5_shape_scaled(oracle_lib):
    """BASELINE config #5 scaled: join-ids drawn Zipf(1.2) over 1 K ids, 1 % of the events on one
    hot barrier, 10 % of the joins replayed byte-identically, 1 % re-sent with different bytes (last
    writer wins), some replays landing between a sweep and its ack (at-least-once: the barrier fires
    again), 1 KB messages."""
    rng = np.random.default_rng(5)
    n_join_ids, n_tx, c, P = 1000, 4, 16, 1024
    nb = n_join_ids * n_tx
    ids = [f"j{b // n_tx:x}~t{b % n_tx:x}" for b in range(nb)]
    e = _engine(capacity_barriers=nb, arena_bytes=256 << 20, max_count=c, payload_stride=P, max_batch=1 << 16,
                host_cap_conts=1 << 13, host_cap_positions=1 << 17)
    o = oracle_lib.Oracle(ids, max_count=c, payload_stride=P)
    d = ParityDriver(e, o, ids)
    try:
        now = T0
        d.fork(np.arange(nb), np.full(nb, c), np.full(nb, T0 + 3_600_000))
        w = 1.0 / np.arange(1, n_join_ids + 1) ** 1.2
        w /= w.sum()
        hot = 7
        history_b, history_p, history_v = [], [], []
        fired_twice = 0
        pending = None
        version = 0
        seen_statuses = set()
        for rnd in range(30):
            now += 3
            n = 4000
            jb = rng.choice(n_join_ids, size=n, p=w) * n_tx + rng.integers(0, n_tx, n)
            jb[rng.random(n) < 0.01] = hot
            jp = rng.integers(0, c, n)
            ver = np.full(n, version)
            version += 1
            if history_b:
                hb, hp, hv = np.concatenate(history_b), np.concatenate(history_p), np.concatenate(history_v)
                rep = rng.random(n) < 0.10                 # byte-identical replay of an earlier join
                pick = rng.integers(0, len(hb), n)
                jb[rep], jp[rep], ver[rep] = hb[pick[rep]], hp[pick[rep]], hv[pick[rep]]
                chg = rng.random(n) < 0.01                 # same (barrier, position), different bytes
                jb[chg], jp[chg] = hb[pick[chg]], hp[pick[chg]]
                ver[chg] = version + 1000
            history_b.append(jb.copy()); history_p.append(jp.copy()); history_v.append(ver.copy())
            st = d.join(jb, jp, np.full(n, P), np.zeros(n, np.int64), payload_bytes(jb, jp, ver, P), now)
            seen_statuses |= set(st.tolist())
            if pending is not None:
                # these joins landed between the previous sweep and its ack
                ec, oc = pending
                out = d.ack(ec, oc, lambda k: True, now + 1)
                fired_twice += sum(1 for v in out.values() if v == capi.DJ_ACK_REQUEUED_DIRTY)
            pending = d.sweep(now + 2)
        d.ack(pending[0], pending[1], lambda k: True, now + 3)
        assert fired_twice > 0, "no replay landed between a sweep and its ack"
        assert {capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE, capi.DJ_JOIN_NOT_FOUND} <= seen_statuses
        d.check_all_snapshots(range(0, nb, 7))
    finally:
        e.close()


def test_compaction_under_key_churn(oracle_lib):
    """Unique barrier-ids keep arriving (every transaction has its own id), so deleted barriers pile
    up as tombstones; the fork path compacts the table by itself.  State, in-flight continuations
    (their slot hints go stale) and parity with the oracle must survive it."""
    rng = np.random.default_rng(11)
    rounds, per = 40, 100
    ids = [f"churn{b:x}~tx{b:x}" for b in range(rounds * per)]
    e = _engine(capacity_barriers=256, arena_bytes=16 << 20, max_count=4, payload_stride=64, max_batch=4096,
                host_cap_conts=4096, host_cap_positions=16384)
    o = oracle_lib.Oracle(ids, max_count=4, payload_stride=64)
    d = ParityDriver(e, o, ids)
    try:
        assert e.stats()["slots"] == 1024
        now, pending, compactions, last_tombs = T0, None, 0, 0
        for r in range(rounds):
            now += 10
            fb = np.arange(r * per, (r + 1) * per)
            d.fork(fb, np.full(per, 2), np.full(per, now + 25))      # the partial ones time out 2-3 rounds later
            tombs = e.stats()["tombstones"]
            if tombs < last_tombs:
                compactions += 1
            last_tombs = tombs
            if pending is not None:                                   # acked after a possible compaction
                d.ack(pending[0], pending[1], lambda k: k % 9 != 0, now + 1)
            done = fb[:90]
            jb = np.concatenate([done, done, fb[90:]])
            jp = np.concatenate([np.zeros(90), np.ones(90), np.zeros(10)]).astype(np.int64)
            order = rng.permutation(len(jb))
            jb, jp = jb[order], jp[order]
            d.join(jb, jp, np.full(len(jb), 64), np.zeros(len(jb), np.int64), payload_bytes(jb, jp, jb * 0, 64), now + 2)
            pending = d.sweep(now + 3)
            last_tombs = e.stats()["tombstones"]
        d.ack(pending[0], pending[1], lambda k: True, now + 4)
        assert compactions >= 2, "the table never compacted"
        # (three in a row: a compaction moves the live slots into the other table and must leave the one it came from
        # all-empty, or barriers deleted long ago come back as live keys two compactions later)
        for _ in range(3):
            live = e.compact()
            s = e.stats()
            assert s["tombstones"] == 0 and s["live"] == live == o.live_barriers()
        d.check_all_snapshots(range(0, len(ids), 13))
    finally:
        e.close()


def test_max_wait_beyond_the_engine_limit_is_refused():
    """A max-wait that does not fit the link word (>= 2^29 s, ~17 years) is answered with its own
    status and leaves the barrier untouched; the largest accepted value behaves like any other."""
    e = _engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=4, payload_stride=64, max_batch=64)
    lo, hi = capi.fingerprint("j1~t1")
    fk = np.zeros(1, dtype=capi.FORK_REC)
    fk[0] = (lo, hi, 2, 0, T0 + 5000)
    assert e.fork_batch(fk).tolist() == [capi.DJ_FORK_CREATED]
    recs = np.zeros(2, dtype=capi.JOIN_REC)
    recs[0] = (lo, hi, 0, 3, 1 << 29, 0)
    recs[1] = (lo, hi, 1, 3, (1 << 29) - 1, 0)
    st = e.join_batch(recs, np.zeros((2, 64), np.uint8), T0 + 1)
    assert st.tolist() == [capi.DJ_JOIN_BAD_MAX_WAIT, capi.DJ_JOIN_WAITING]
    meta, lens, mws, _ = e.snapshot_one(lo, hi)
    assert lens.tolist()[:2] == [capi.DJ_NIL, 3] and mws.tolist()[1] == (1 << 29) - 1
    e.close()


def test_config4_one_rank_share_scaled():
    """BASELINE config #4's shape (count 16, 256 B messages, events generated for the whole barrier
    space, owner = hash mod 8, routed by partition) from the point of view of ONE of the 8 ranks,
    scaled to 2^18 global barriers: every batch goes through the 8-way partition kernels, the rank
    applies its own bucket straight from the partition's output buffers.  Its barriers (and only
    they) fire exactly once, complete, with every byte where the generator put it."""
    import torch
    from distributed_join_amd.router import owner_of_np
    nbg, c, P, ranks, me = 1 << 18, 16, 256, 8, 3
    rng = np.random.default_rng(44)
    ids = [f"job{b % 1000}~tx{b:x}" for b in range(nbg)]
    fps = capi.fingerprint_many(ids)
    owner = owner_of_np(fps[:, 1], ranks)
    mine = np.nonzero(owner == me)[0]
    nb = len(mine)
    assert abs(nb - nbg / ranks) < 0.05 * nbg / ranks            # the hash spreads the barriers evenly
    batch = 1 << 18
    e = _engine(capacity_barriers=nb, arena_bytes=nb * (16 * P + 256) + (64 << 20), max_count=c, payload_stride=P,
                max_batch=batch)
    dev = e.device
    try:
        frecs = np.zeros(nb, dtype=capi.FORK_REC)
        frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[mine, 0], fps[mine, 1], c, T0 + 3_600_000
        assert (e.fork_batch(frecs) == capi.DJ_FORK_CREATED).all()
        ev = rng.permutation(nbg * c)
        d_ro = torch.empty(batch * 32, dtype=torch.uint8, device=dev)
        d_po = torch.empty((batch, P), dtype=torch.uint8, device=dev)
        d_origin = torch.empty(batch, dtype=torch.int32, device=dev)
        d_counts = torch.zeros(64, dtype=torch.int32, device=dev)
        d_st = e.empty(batch * 4)
        cap_c, cap_p = nb, nb * c
        d_conts, d_lens, d_pay, d_res, d_ack = e.empty(cap_c * 64), e.empty(cap_p * 4), e.empty(cap_p * P), e.empty(16), e.empty(cap_c * 4)
        pos_of = {int(b): k for k, b in enumerate(mine)}
        by_fp = {(int(a), int(b)): int(g) for g, (a, b) in zip(mine, fps[mine])}
        fired = np.zeros(nb, dtype=np.int32)
        applied = 0
        for k in range(nbg * c // batch):
            now = T0 + 1 + k
            part = ev[k * batch:(k + 1) * batch]
            jb, jp = part // c, part % c
            recs = np.zeros(batch, dtype=capi.JOIN_REC)
            recs["fp_lo"], recs["fp_hi"], recs["position"], recs["len"] = fps[jb, 0], fps[jb, 1], jp, P
            d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
            d_in = torch.from_numpy(payload_bytes(jb, jp, jb * 0, P)).to(dev)
            torch.cuda.synchronize()
            e.partition_joins_dev(batch, ranks, d_recs, d_in, P, d_ro, d_po, d_origin, d_counts)
            e.sync()
            counts = d_counts.cpu().numpy()[:ranks]
            np.testing.assert_array_equal(counts, np.bincount(owner[jb], minlength=ranks))
            lo, n_me = int(counts[:me].sum()), int(counts[me])
            applied += n_me
            if n_me:
                e.join_batch_dev(n_me, d_ro[lo * 32:], d_po[lo:], P, now, d_st)
                e.sync()
                st = d_st[:n_me * 4].cpu().numpy().view(np.int32)
                assert np.isin(st, [capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE]).all()
            res = np.zeros(1, dtype=capi.SWEEP_RESULT)
            e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_pay, d_res, res)
            n, npos = int(res[0]["n_conts"]), int(res[0]["n_positions"])
            assert not res[0]["more"] and npos == n * c
            if n == 0:
                continue
            conts = d_conts[:n * 64].cpu().numpy().view(capi.CONT)
            assert (conts["flags"] == 0).all() and (conts["count"] == c).all()
            gidx = np.array([by_fp[(int(a), int(b))] for a, b in zip(conts["fp_lo"], conts["fp_hi"])])
            np.add.at(fired, [pos_of[int(g)] for g in gidx], 1)
            assert (d_lens[:npos * 4].cpu().numpy().view(np.uint32) == P).all()
            got = d_pay[:npos * P].cpu().numpy().reshape(npos, P)
            rows = (conts["pos_base"].astype(np.int64)[:, None] + np.arange(c)[None, :]).reshape(-1)
            want = payload_bytes(np.repeat(gidx, c), np.tile(np.arange(c), n), np.zeros(n * c), P)
            assert np.array_equal(got[rows], want), "a gathered byte differs from the generator"
            e.ack_batch_dev(n, None, d_conts, None, now, d_ack)
            e.sync()
            assert (d_ack[:n * 4].cpu().numpy().view(np.int32) == capi.DJ_ACK_DONE).all()
        assert applied == nb * c
        assert (fired == 1).all()
        s = e.stats()
        assert s["live"] == 0 and s["tombstones"] == nb
    finally:
        e.close()


def _splitmix64_t(torch, x):
    """tests.parity.splitmix64 on int64 tensors (two's-complement wrap-around, logical shifts)."""
    def shr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    x = x + (-7046029254386353131)               # 0x9E3779B97F4A7C15
    x = (x ^ shr(x, 30)) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    x = (x ^ shr(x, 27)) * (-7723592293110705685)  # 0x94D049BB133111EB
    return x ^ shr(x, 31)


def _payload_bytes_t(torch, barrier, pos, pitch):
    """tests.parity.payload_bytes (version 0) on the device -> uint8 (n, pitch)."""
    base = _splitmix64_t(torch, barrier * 1024 + pos)
    k = torch.arange(pitch // 8, dtype=torch.int64, device=barrier.device) * 0x100000001B3
    return _splitmix64_t(torch, base[:, None] + k[None, :]).view(torch.uint8).reshape(len(barrier), pitch)


def test_config3_full_size_properties():
    """BASELINE config #3 at full size: 10 M barriers, counts Zipf(1) over [2, 64], 30 % of the barriers get a
    strict subset of their positions and a deadline inside the run, the clock advances per batch.  Checked
    without the oracle, from what the generator knows: every barrier fires exactly once; the 70 % fire COMPLETE
    and the 30 % TIMEOUT; a continuation holds exactly the positions that were sent before its deadline, each
    with the generator's bytes; joins that arrive after their barrier fired are answered NOT_FOUND."""
    import time
    import torch
    t_start = time.time()
    nb, P, batch = 10_000_000, 64, 1 << 22
    rng = np.random.default_rng(3)
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)]
    fps_np = capi.fingerprint_many(ids).view(np.int64)
    del ids
    counts_np = zipf_counts(rng, nb)
    e = _engine(capacity_barriers=nb, arena_bytes=26 << 30, max_count=64, payload_stride=P, max_batch=nb)
    dev = e.device
    NIL = 0xFFFFFFFF
    try:
        assert np.array_equal(_payload_bytes_t(torch, torch.arange(5, device=dev), torch.arange(5, device=dev) * 3, P).cpu().numpy(),
                              payload_bytes(np.arange(5), np.arange(5) * 3, np.zeros(5), P))
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        fps = torch.from_numpy(fps_np).to(dev)
        counts = torch.from_numpy(counts_np).to(dev)
        offsets = torch.cumsum(counts, 0) - counts                     # first (barrier, position) cell of each barrier
        cells = int(counts.sum())
        expire = torch.rand(nb, device=dev, generator=g) < 0.3
        # (barrier, position) cells; an expiring barrier never gets position hash(b) % count, the others with p = 1/2
        cell_b = torch.repeat_interleave(torch.arange(nb, device=dev), counts)
        cell_p = torch.arange(cells, device=dev) - offsets[cell_b]
        h = _splitmix64_t(torch, cell_b * 64 + cell_p + (1 << 50))
        never = (_splitmix64_t(torch, cell_b + (1 << 51)) & 0xFFFF) % counts[cell_b]
        sent = ~expire[cell_b] | ((cell_p != never) & ((h & 1) == 1))
        del h, never
        ev_cell = torch.nonzero(sent).reshape(-1)
        ev_cell = ev_cell[torch.randperm(len(ev_cell), device=dev, generator=g)]
        n_events = len(ev_cell)
        n_batches = (n_events + batch - 1) // batch
        t_of = lambda k: T0 + 2 * (k + 1)                              # joins and sweep of batch k; even milliseconds
        # deadlines on odd milliseconds: popped at the first sweep after them, strictly late (core.clj:86-89)
        deadline = torch.where(expire, T0 + 11 + 2 * torch.randint(0, n_batches, (nb,), device=dev, generator=g),
                               torch.full((nb,), T0 + 3_600_000, device=dev))
        # expected presence of every cell: sent, and applied before the sweep that pops the barrier (the first
        # one at or after the deadline, i.e. at deadline + 1; a batch's joins run before its sweep)
        arrive = torch.full((cells,), 1 << 60, dtype=torch.int64, device=dev)
        arrive[ev_cell] = T0 + 2 * (torch.arange(n_events, device=dev) // batch + 1)
        present = arrive <= deadline[cell_b] + 1
        late_events = int((sent & ~present).sum())
        del arrive, sent

        frecs = torch.empty((nb, 4), dtype=torch.int64, device=dev)
        frecs[:, 0], frecs[:, 1], frecs[:, 2], frecs[:, 3] = fps[:, 0], fps[:, 1], counts, deadline
        d_status = torch.empty(nb, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.fork_batch_dev(nb, frecs, d_status)
        e.sync()
        assert bool((d_status == capi.DJ_FORK_CREATED).all())
        del frecs

        order = torch.argsort(fps[:, 0])
        sorted_lo = fps[order, 0].contiguous()
        cap_c, cap_p = 1 << 21, 40_000_000
        d_conts = torch.empty((cap_c, 8), dtype=torch.int64, device=dev)
        d_lens = torch.empty(cap_p, dtype=torch.int32, device=dev)
        d_pay = torch.empty((cap_p, P), dtype=torch.uint8, device=dev)
        d_res = torch.empty(16, dtype=torch.uint8, device=dev)
        d_ack = torch.empty(cap_c, dtype=torch.int32, device=dev)
        fired = torch.zeros(nb, dtype=torch.int32, device=dev)
        n_complete = n_timeout = n_not_found = n_status_complete = 0
        print(f"\n  generated {n_events} events over {nb} barriers in {time.time() - t_start:.1f} s", flush=True)

        def sweep_and_check(now):
            nonlocal n_complete, n_timeout
            while True:
                res = np.zeros(1, dtype=capi.SWEEP_RESULT)
                e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_pay, d_res, res)
                n, npos = int(res[0]["n_conts"]), int(res[0]["n_positions"])
                if n:
                    c = d_conts[:n]
                    # rows of barriers that did not fit under cap_positions are void; they stay queued
                    void = ((c[:, 4] >> 32) & capi.DJ_CONT_VOID) != 0
                    assert bool(void.any()) <= bool(res[0]["more"])
                    c = c[~void].contiguous()
                    rows_acked = n
                    n = len(c)
                if n:
                    at = torch.searchsorted(sorted_lo, c[:, 0].contiguous()).clamp_(max=nb - 1)
                    b = order[at]
                    assert bool((fps[b, 0] == c[:, 0]).all()) and bool((fps[b, 1] == c[:, 1]).all())
                    fired.index_add_(0, b, torch.ones(n, dtype=torch.int32, device=dev))
                    cnt = c[:, 4] & 0xFFFFFFFF
                    flags = c[:, 4] >> 32
                    assert bool((cnt == counts[b]).all())
                    assert bool(((flags & capi.DJ_CONT_TIMEOUT) != 0).eq(expire[b]).all())
                    assert bool((c[:, 2] == deadline[b]).all())
                    tot = int(cnt.sum())
                    assert tot <= npos and (tot == npos or bool(void.any()))
                    n_timeout += int(expire[b].sum())
                    n_complete += n - int(expire[b].sum())
                    # positions of the continuations -> their cells
                    k_of = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
                    p_of = torch.arange(tot, device=dev) - (torch.cumsum(cnt, 0) - cnt)[k_of]
                    row = c[k_of, 6] + p_of                             # pos_base + position
                    cell = offsets[b][k_of] + p_of
                    want_len = torch.where(present[cell], P, NIL - (1 << 32)).to(torch.int32)
                    assert bool((d_lens[row] == want_len).all()), "presence differs from what was sent in time"
                    here = present[cell]
                    want = _payload_bytes_t(torch, b[k_of][here], p_of[here], P)
                    assert bool((d_pay[row[here]] == want).all()), "a gathered byte differs from the generator"
                    torch.cuda.synchronize()
                    e.ack_batch_dev(rows_acked, None, d_conts, None, now, d_ack)
                    e.sync()
                    assert bool((d_ack[:rows_acked][~void] == capi.DJ_ACK_DONE).all()) and bool((d_ack[:rows_acked][void] == 0).all())
                if not res[0]["more"]:
                    return

        d_jst = torch.empty(batch, dtype=torch.int32, device=dev)
        for k in range(n_batches):
            now = t_of(k)
            cell = ev_cell[k * batch:(k + 1) * batch]
            n = len(cell)
            jb, jp = cell_b[cell], cell_p[cell]
            recs = torch.empty((n, 4), dtype=torch.int64, device=dev)
            recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = fps[jb, 0], fps[jb, 1], jp | (P << 32), 0
            pay = _payload_bytes_t(torch, jb, jp, P)
            torch.cuda.synchronize()
            e.join_batch_dev(n, recs, pay, P, now, d_jst)
            e.sync()
            st = d_jst[:n]
            n_not_found += int((st == capi.DJ_JOIN_NOT_FOUND).sum())
            n_status_complete += int((st == capi.DJ_JOIN_COMPLETE).sum())
            assert bool(((st == capi.DJ_JOIN_WAITING) | (st == capi.DJ_JOIN_COMPLETE) | (st == capi.DJ_JOIN_NOT_FOUND)).all())
            sweep_and_check(now)
        for k in range(n_batches, n_batches + 8):                     # the deadlines after the last batch
            sweep_and_check(t_of(k))
        print(f"  {n_batches} batches, {n_complete} complete + {n_timeout} timeout, {time.time() - t_start:.1f} s", flush=True)
        assert bool((fired == 1).all())
        assert n_complete == nb - int(expire.sum()) and n_timeout == int(expire.sum())
        assert n_status_complete == n_complete          # one join completes each barrier that is not starved
        assert n_not_found == late_events
        assert e.stats()["live"] == 0
    finally:
        e.close()


def test_config5_large_properties():
    """BASELINE config #5's event mix at one GPU's scale, checked from the generator's own arithmetic (no
    oracle at this size): 1 M barriers of count 16 with 1 KB messages whose join-ids are drawn Zipf(1.2) over
    1 K ids, 10 % of the joins replayed byte-identically, 1 % re-sent with different bytes, and 1 % of all
    events aimed at four hot barriers of count 64 that can only complete in the last batch (tens of thousands
    of overwrites each, the long-chain kernels at full batch size).  Expected: every barrier fires once, in the
    batch in which its last missing position arrives; each position carries the bytes of the LAST event that
    reached it up to that batch (HSET semantics, redis.clj:135); later events are answered NOT_FOUND."""
    import time
    import torch
    t_start = time.time()
    nb, c, P, batch = 1 << 20, 16, 1024, 1 << 20
    n_hot, c_hot = 4, 64
    rng = np.random.default_rng(55)
    w = 1.0 / np.arange(1, 1001) ** 1.2
    jid = rng.choice(1000, size=nb + n_hot, p=w / w.sum())
    ids = [f"j{int(j):x}~t{b:x}" for b, j in enumerate(jid)]
    fps_np = capi.fingerprint_many(ids).view(np.int64)
    assert len({(int(a), int(b)) for a, b in fps_np[:1000]}) == 1000
    del ids
    e = _engine(capacity_barriers=nb + n_hot, arena_bytes=(nb * (c * P + 256)) + (256 << 20), max_count=c_hot,
                payload_stride=P, max_batch=batch)
    dev = e.device
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(55)
        fps = torch.from_numpy(fps_np).to(dev)
        counts = torch.full((nb + n_hot,), c, dtype=torch.int64, device=dev)
        counts[nb:] = c_hot
        offsets = torch.cumsum(counts, 0) - counts
        cells = int(counts.sum())
        cell_b = torch.repeat_interleave(torch.arange(nb + n_hot, device=dev), counts)
        cell_p = torch.arange(cells, device=dev) - offsets[cell_b]
        # base events: every cell once, except the last position of the hot barriers (appended at the very end)
        is_hot_last = (cell_b >= nb) & (cell_p == c_hot - 1)
        base = torch.nonzero(~is_hot_last).reshape(-1)
        n_base = len(base)
        n_rep, n_chg, n_hotev = n_base // 10, n_base // 100, n_base // 100
        rep = base[torch.randint(0, n_base, (n_rep,), device=dev, generator=g)]           # byte-identical replays
        chg = base[torch.randint(0, n_base, (n_chg,), device=dev, generator=g)]           # same cell, other bytes
        hot_cell = offsets[nb + torch.randint(0, n_hot, (n_hotev,), device=dev, generator=g)] + \
            torch.randint(0, c_hot - 1, (n_hotev,), device=dev, generator=g)             # overwrites on the hot barriers
        ev_cell = torch.cat([base, rep, chg, hot_cell])
        fresh = torch.cat([torch.zeros(n_base + n_rep, dtype=torch.bool, device=dev),
                           torch.ones(n_chg + n_hotev, dtype=torch.bool, device=dev)])   # carries bytes of its own
        perm = torch.randperm(len(ev_cell), device=dev, generator=g)
        ev_cell, fresh = ev_cell[perm], fresh[perm]
        ev_cell = torch.cat([ev_cell, torch.nonzero(is_hot_last).reshape(-1)])            # the hot barriers complete last
        fresh = torch.cat([fresh, torch.zeros(n_hot, dtype=torch.bool, device=dev)])
        n_events = len(ev_cell)
        idx = torch.arange(n_events, device=dev)
        ev_ver = torch.where(fresh, idx + 1, torch.zeros_like(idx))                       # version 0 = the base bytes
        ev_batch = idx // batch
        n_batches = int(ev_batch[-1]) + 1
        # expectations
        first_batch = torch.full((cells,), 1 << 30, dtype=torch.int64, device=dev).scatter_reduce(0, ev_cell, ev_batch, "amin")
        fire_batch = torch.zeros(nb + n_hot, dtype=torch.int64, device=dev).scatter_reduce(0, cell_b, first_batch, "amax")
        assert int(fire_batch.max()) == n_batches - 1
        applied = ev_batch <= fire_batch[cell_b[ev_cell]]
        last_idx = torch.full((cells,), -1, dtype=torch.int64, device=dev).scatter_reduce(
            0, ev_cell[applied], idx[applied], "amax")
        want_ver = ev_ver[last_idx]
        del first_batch, last_idx, perm, base, rep, chg, hot_cell

        frecs = torch.empty((nb + n_hot, 4), dtype=torch.int64, device=dev)
        frecs[:, 0], frecs[:, 1], frecs[:, 2], frecs[:, 3] = fps[:, 0], fps[:, 1], counts, T0 + 3_600_000
        d_status = torch.empty(batch + n_hot, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.fork_batch_dev(batch, frecs[:batch], d_status)
        e.fork_batch_dev(n_hot, frecs[batch:], d_status[batch:])
        e.sync()
        assert bool((d_status == capi.DJ_FORK_CREATED).all())
        order = torch.argsort(fps[:, 0])
        sorted_lo = fps[order, 0].contiguous()
        cap_c, cap_p = 1 << 20, 6_000_000
        d_conts = torch.empty((cap_c, 8), dtype=torch.int64, device=dev)
        d_lens = torch.empty(cap_p, dtype=torch.int32, device=dev)
        d_pay = torch.empty((cap_p, P), dtype=torch.uint8, device=dev)
        d_res = torch.empty(16, dtype=torch.uint8, device=dev)
        d_ack = torch.empty(cap_c, dtype=torch.int32, device=dev)
        d_jst = torch.empty(batch, dtype=torch.int32, device=dev)
        fired = torch.zeros(nb + n_hot, dtype=torch.int32, device=dev)
        n_not_found = n_complete_status = 0
        print(f"\n  generated {n_events} events in {n_batches} batches, {time.time() - t_start:.1f} s", flush=True)

        def payload(b, p, ver):
            base_ = _splitmix64_t(torch, b * 1024 + p + (ver << 40))
            k = torch.arange(P // 8, dtype=torch.int64, device=dev) * 0x100000001B3
            return _splitmix64_t(torch, base_[:, None] + k[None, :]).view(torch.uint8).reshape(len(b), P)

        assert np.array_equal(payload(torch.arange(4, device=dev), torch.arange(4, device=dev), torch.arange(4, device=dev) * 99999).cpu().numpy(),
                              payload_bytes(np.arange(4), np.arange(4), np.arange(4) * 99999, P))

        for k in range(n_batches):
            now = T0 + 2 * (k + 1)
            sl = slice(k * batch, min((k + 1) * batch, n_events))
            cell = ev_cell[sl]
            n = len(cell)
            jb, jp = cell_b[cell], cell_p[cell]
            recs = torch.empty((n, 4), dtype=torch.int64, device=dev)
            recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = fps[jb, 0], fps[jb, 1], jp | (P << 32), 0
            pay = payload(jb, jp, ev_ver[sl])
            torch.cuda.synchronize()
            e.join_batch_dev(n, recs, pay, P, now, d_jst)
            e.sync()
            st = d_jst[:n]
            want_nf = ~applied[sl]
            assert bool(((st == capi.DJ_JOIN_NOT_FOUND) == want_nf).all()), "NOT_FOUND answers differ from the expected late events"
            assert bool(((st == capi.DJ_JOIN_WAITING) | (st == capi.DJ_JOIN_COMPLETE) | want_nf).all())
            n_not_found += int(want_nf.sum())
            n_complete_status += int((st == capi.DJ_JOIN_COMPLETE).sum())
            del recs, pay
            while True:
                res = np.zeros(1, dtype=capi.SWEEP_RESULT)
                e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_pay, d_res, res)
                rows = int(res[0]["n_conts"])
                if rows:
                    cc = d_conts[:rows]
                    void = ((cc[:, 4] >> 32) & capi.DJ_CONT_VOID) != 0
                    cc = cc[~void].contiguous()
                    m = len(cc)
                    at = torch.searchsorted(sorted_lo, cc[:, 0].contiguous()).clamp_(max=nb + n_hot - 1)
                    b = order[at]
                    assert bool((fps[b, 0] == cc[:, 0]).all()) and bool((fps[b, 1] == cc[:, 1]).all())
                    fired.index_add_(0, b, torch.ones(m, dtype=torch.int32, device=dev))
                    assert bool((fire_batch[b] == k).all()), "a barrier fired in another batch than the one that completed it"
                    cnt = cc[:, 4] & 0xFFFFFFFF
                    assert bool((cnt == counts[b]).all()) and bool(((cc[:, 4] >> 32) == 0).all())
                    tot = int(cnt.sum())
                    k_of = torch.repeat_interleave(torch.arange(m, device=dev), cnt)
                    p_of = torch.arange(tot, device=dev) - (torch.cumsum(cnt, 0) - cnt)[k_of]
                    row = cc[k_of, 6] + p_of
                    cellx = offsets[b][k_of] + p_of
                    assert bool((d_lens[row] == P).all()), "a fired barrier misses a position"
                    for lo in range(0, tot, 1 << 20):      # in pieces: 1 KB rows
                        s2 = slice(lo, lo + (1 << 20))
                        want = payload(b[k_of[s2]], p_of[s2], want_ver[cellx[s2]])
                        assert bool((d_pay[row[s2]] == want).all()), "a position does not hold its last writer's bytes"
                    torch.cuda.synchronize()
                    e.ack_batch_dev(rows, None, d_conts, None, now, d_ack)
                    e.sync()
                    assert bool((d_ack[:rows][~void] == capi.DJ_ACK_DONE).all())
                if not res[0]["more"]:
                    break
        print(f"  all batches checked, {time.time() - t_start:.1f} s", flush=True)
        assert bool((fired == 1).all())
        assert n_complete_status >= nb + n_hot      # the completing join and every applied replay after it in its batch
        assert n_not_found > 0
        assert e.stats()["live"] == 0
    finally:
        e.close()


def test_config4_one_rank_full_size_properties():
    """BASELINE config #4 from the point of view of one of its eight GPUs, at full size: 12.5 M barriers of
    count 16 with 256-byte messages, 200 M joins in uniformly shuffled order, batches of 4 M.  Checked from the
    generator's arithmetic: every barrier fires exactly once, COMPLETE, in the batch that brings its last
    position, with every byte where the generator put it.  (Routing is covered at a smaller size by
    test_config4_one_rank_share_scaled; here the events are already the rank's own.)"""
    import time
    import torch
    t_start = time.time()
    nb, c, P, batch = 12_500_000, 16, 256, 1 << 22
    ids = [f"j{b % 1000:x}~t{b:x}" for b in range(nb)]
    fps_np = capi.fingerprint_many(ids).view(np.int64)
    del ids
    e = _engine(capacity_barriers=nb, arena_bytes=nb * (c * P + 192) + (256 << 20), max_count=c, payload_stride=P,
                max_batch=batch)
    dev = e.device
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(4)
        fps = torch.from_numpy(fps_np).to(dev)
        n_events = nb * c
        ev_cell = torch.randperm(n_events, device=dev, generator=g)          # cell = barrier * c + position
        n_batches = (n_events + batch - 1) // batch
        ev_batch = torch.arange(n_events, device=dev) // batch
        fire_batch = torch.zeros(nb, dtype=torch.int64, device=dev).scatter_reduce(0, ev_cell // c, ev_batch, "amax")
        del ev_batch
        d_status = torch.empty(batch, dtype=torch.int32, device=dev)
        frecs = torch.empty((batch, 4), dtype=torch.int64, device=dev)
        for lo in range(0, nb, batch):                                       # forks in pieces of one batch
            m = min(batch, nb - lo)
            frecs[:m, 0], frecs[:m, 1], frecs[:m, 2], frecs[:m, 3] = fps[lo:lo + m, 0], fps[lo:lo + m, 1], c, T0 + 3_600_000
            torch.cuda.synchronize()
            e.fork_batch_dev(m, frecs, d_status)
            e.sync()
            assert bool((d_status[:m] == capi.DJ_FORK_CREATED).all())
        del frecs
        order = torch.argsort(fps[:, 0])
        sorted_lo = fps[order, 0].contiguous()
        cap_c, cap_p = 1 << 21, 1 << 24
        d_conts = torch.empty((cap_c, 8), dtype=torch.int64, device=dev)
        d_lens = torch.empty(cap_p, dtype=torch.int32, device=dev)
        d_pay = torch.empty((cap_p, P), dtype=torch.uint8, device=dev)
        d_res = torch.empty(16, dtype=torch.uint8, device=dev)
        d_ack = torch.empty(cap_c, dtype=torch.int32, device=dev)
        fired = torch.zeros(nb, dtype=torch.int32, device=dev)
        n_complete_status = 0
        print(f"\n  {nb} barriers forked, {n_events} events in {n_batches} batches, {time.time() - t_start:.1f} s", flush=True)
        for k in range(n_batches):
            now = T0 + 2 * (k + 1)
            cell = ev_cell[k * batch:(k + 1) * batch]
            n = len(cell)
            jb, jp = cell // c, cell % c
            recs = torch.empty((n, 4), dtype=torch.int64, device=dev)
            recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = fps[jb, 0], fps[jb, 1], jp | (P << 32), 0
            pay = _payload_bytes_t(torch, jb, jp, P)
            torch.cuda.synchronize()
            e.join_batch_dev(n, recs, pay, P, now, d_status)
            e.sync()
            st = d_status[:n]
            assert bool(((st == capi.DJ_JOIN_WAITING) | (st == capi.DJ_JOIN_COMPLETE)).all())
            n_complete_status += int((st == capi.DJ_JOIN_COMPLETE).sum())
            del recs, pay
            while True:
                res = np.zeros(1, dtype=capi.SWEEP_RESULT)
                e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_pay, d_res, res)
                rows = int(res[0]["n_conts"])
                if rows:
                    cc = d_conts[:rows]
                    void = ((cc[:, 4] >> 32) & capi.DJ_CONT_VOID) != 0
                    cc = cc[~void].contiguous()
                    m = len(cc)
                    at = torch.searchsorted(sorted_lo, cc[:, 0].contiguous()).clamp_(max=nb - 1)
                    b = order[at]
                    assert bool((fps[b, 0] == cc[:, 0]).all()) and bool((fps[b, 1] == cc[:, 1]).all())
                    fired.index_add_(0, b, torch.ones(m, dtype=torch.int32, device=dev))
                    assert bool((fire_batch[b] == k).all()), "a barrier fired in another batch than the one that completed it"
                    assert bool((cc[:, 4] == c).all())                       # count 16, no flags
                    row = (cc[:, 6][:, None] + torch.arange(c, device=dev)[None, :]).reshape(-1)
                    assert bool((d_lens[row] == P).all())
                    for lo in range(0, m, 1 << 16):                           # in pieces: 256-byte rows
                        bb = b[lo:lo + (1 << 16)]
                        want = _payload_bytes_t(torch, bb.repeat_interleave(c), torch.arange(c, device=dev).repeat(len(bb)), P)
                        assert bool((d_pay[row[lo * c:(lo + (1 << 16)) * c]] == want).all()), "a gathered byte differs from the generator"
                    torch.cuda.synchronize()
                    e.ack_batch_dev(rows, None, d_conts, None, now, d_ack)
                    e.sync()
                    assert bool((d_ack[:rows][~void] == capi.DJ_ACK_DONE).all())
                if not res[0]["more"]:
                    break
        print(f"  all batches checked, {time.time() - t_start:.1f} s", flush=True)
        assert bool((fired == 1).all())
        assert n_complete_status == nb
        assert e.stats()["live"] == 0
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0])
def test_replay_storm_at_full_batch_size(hot_mode):
    """Half of a 1 M-event batch on ONE barrier (count 64, positions 0..62, so it cannot complete) that the
    engine has never seen hot, the other half spread over 128 K barriers of count 8; then the same again with
    other bytes (by now the barrier is flagged, and left to itself the engine has the parallel kernels on;
    pinned to the serial kernels, both batches take the chain walk that gives up and scans the batch), then the
    missing position.  Checked from the generator's arithmetic: every storm event is answered WAITING, the
    barrier's positions hold the bytes of the LAST event in array order, exactly those of the 128 K barriers
    complete whose eight positions were all in the batch, and the last join completes the storm barrier."""
    import torch
    nb, c, c_hot, P, n = 1 << 17, 8, 64, 64, 1 << 20
    ids = [f"q{b:x}~t{b:x}" for b in range(nb + 1)]
    fps_np = capi.fingerprint_many(ids).view(np.int64)
    e = _engine(capacity_barriers=nb + 1, arena_bytes=nb * (c * P + 128) + (64 << 20), max_count=c_hot,
                payload_stride=P, max_batch=n)
    e.set_hot_path(hot_mode)
    dev = e.device
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(77)
        fps = torch.from_numpy(fps_np).to(dev)
        frecs = torch.empty((nb + 1, 4), dtype=torch.int64, device=dev)
        frecs[:, 0], frecs[:, 1], frecs[:, 2], frecs[:, 3] = fps[:, 0], fps[:, 1], c, T0 + 3_600_000
        frecs[nb, 2] = c_hot
        d_status = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.fork_batch_dev(nb + 1, frecs, d_status)
        e.sync()
        assert bool((d_status[:nb + 1] == capi.DJ_FORK_CREATED).all())
        half = n // 2
        for rnd in range(2):
            # the ordinary half: half of the 1 M (barrier, position) cells, each at most once per batch
            cells = torch.randperm(nb * c, device=dev, generator=g)[:half]
            b_of = torch.cat([cells // c, torch.full((half,), nb, dtype=torch.int64, device=dev)])
            p_of = torch.cat([cells % c, torch.randint(0, c_hot - 1, (half,), device=dev, generator=g)])
            perm = torch.randperm(n, device=dev, generator=g)
            b_of, p_of = b_of[perm], p_of[perm]
            recs = torch.empty((n, 4), dtype=torch.int64, device=dev)
            recs[:, 0], recs[:, 1], recs[:, 2], recs[:, 3] = fps[b_of, 0], fps[b_of, 1], p_of | (P << 32), 0
            pay = torch.randint(0, 256, (n, P), dtype=torch.uint8, device=dev, generator=g)
            torch.cuda.synchronize()        # the engine has a stream of its own: the inputs must be complete
            e.join_batch_dev(n, recs, pay, P, T0 + 1 + rnd, d_status)
            e.sync()
            storm = b_of == nb
            vals, cnts = torch.unique(d_status[storm], return_counts=True)
            assert vals.tolist() == [capi.DJ_JOIN_WAITING], (rnd, vals.tolist(), cnts.tolist())
            if rnd == 0:
                per_barrier = torch.bincount(b_of[~storm], minlength=nb)
                assert int((d_status[~storm] == capi.DJ_JOIN_COMPLETE).sum()) == int((per_barrier == c).sum())
                assert bool(((d_status[~storm] == capi.DJ_JOIN_WAITING) | (d_status[~storm] == capi.DJ_JOIN_COMPLETE)).all())
            idx = torch.arange(n, device=dev)
            last = torch.full((c_hot,), -1, dtype=torch.int64, device=dev).scatter_reduce(0, p_of[storm], idx[storm], "amax")
            meta, lens, mws, got = e.snapshot_one(int(fps_np[nb, 0]) & (2**64 - 1), int(fps_np[nb, 1]) & (2**64 - 1))
            assert int(meta["count"]) == c_hot
            assert bool((last[:c_hot - 1] >= 0).all())
            assert list(lens[:c_hot - 1]) == [P] * (c_hot - 1) and lens[c_hot - 1] == capi.DJ_NIL
            want = pay[last[:c_hot - 1]].cpu().numpy()
            assert np.array_equal(np.asarray(got[:c_hot - 1]), want)
        # the missing position completes it
        one = torch.empty((1, 4), dtype=torch.int64, device=dev)
        one[0, 0], one[0, 1], one[0, 2], one[0, 3] = fps[nb, 0], fps[nb, 1], (c_hot - 1) | (P << 32), 0
        torch.cuda.synchronize()
        e.join_batch_dev(1, one, pay[:1], P, T0 + 5, d_status)
        e.sync()
        assert int(d_status[0]) == capi.DJ_JOIN_COMPLETE
    finally:
        e.close()


def test_native_router_single_rank_roundtrip():
    """The C router (dj_router_*: RCCL bound at run time, bucket sizes all-gathered, planned pieces) with one rank
    and receive buffers smaller than the batch: every record and message comes back, in array order, over
    several pieces, with the first piece sent ahead (dj_route_advance) or not."""
    import ctypes
    import torch
    L = capi.load()
    n, pitch, cap = 50_000, 64, 16_384
    rng = np.random.default_rng(19)
    e = _engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=1 << 16)
    recs = torch.from_numpy(rng.integers(2, 1 << 62, (n, 4), dtype=np.int64)).to(e.device)
    pay = torch.from_numpy(rng.integers(0, 256, (n, pitch), dtype=np.uint8)).to(e.device)
    torch.cuda.synchronize()
    uid = (ctypes.c_uint8 * 128)()
    assert L.dj_route_unique_id(uid) == 0, L.dj_router_last_error(None)
    r = ctypes.c_void_p()
    assert L.dj_router_create(e.h, 1, 0, uid, n, cap, pitch, 2, ctypes.byref(r)) == 0, L.dj_router_last_error(None)
    try:
        for with_pay, ahead in ((True, False), (False, True), (True, True)):
            h = ctypes.c_int(-1)
            assert L.dj_route_start(r, n, recs.data_ptr(), pay.data_ptr() if with_pay else None, ctypes.byref(h)) == 0
            if ahead:
                assert L.dj_route_advance(r, h.value) == 0, L.dj_router_last_error(r)
                assert L.dj_route_advance(r, h.value) == 0        # a second call changes nothing
            got_r, got_p, pieces, more = [], [], 0, ctypes.c_int(1)
            while more.value:
                pr, pp, pn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
                assert L.dj_route_next(r, h.value, ctypes.byref(pr), ctypes.byref(pp), ctypes.byref(pn), ctypes.byref(more)) == 0, \
                    L.dj_router_last_error(r)
                assert 0 < pn.value <= cap and bool(pp.value) == with_pay
                # the engine's stream waits for the piece: once that stream has drained the piece is there; copy it
                # out before the next one overwrites it
                e.sync()
                out_r = torch.empty((pn.value, 4), dtype=torch.int64, device=e.device)
                assert capi_memcpy(out_r, pr.value, pn.value * 32)
                got_r.append(out_r)
                if with_pay:
                    out_p = torch.empty((pn.value, pitch), dtype=torch.uint8, device=e.device)
                    assert capi_memcpy(out_p, pp.value, pn.value * pitch)
                    got_p.append(out_p)
                pieces += 1
            assert L.dj_route_release(r, h.value) == 0
            assert pieces == -(-n // cap)
            assert torch.equal(torch.cat(got_r), recs)
            if with_pay:
                assert torch.equal(torch.cat(got_p), pay)
        # more batches in flight than buffer sets: refused, not overwritten
        h1, h2, h3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert L.dj_route_start(r, 100, recs.data_ptr(), None, ctypes.byref(h1)) == 0
        assert L.dj_route_start(r, 100, recs.data_ptr(), None, ctypes.byref(h2)) == 0
        assert L.dj_route_start(r, 100, recs.data_ptr(), None, ctypes.byref(h3)) != 0
        assert b"in use" in L.dj_router_last_error(r)
    finally:
        L.dj_router_destroy(r)
        e.close()


def capi_memcpy(dst_tensor, src_ptr, nbytes):
    """device-to-device copy from a raw pointer into a tensor (synchronous)."""
    import ctypes
    import torch
    hip = ctypes.CDLL(None)
    fn = getattr(hip, "hipMemcpy", None)
    if fn is None:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        fn = hip.hipMemcpy
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return fn(dst_tensor.data_ptr(), src_ptr, nbytes, 3) == 0      # hipMemcpyDeviceToDevice


@pytest.mark.gpu
def test_fork_batches_with_and_without_contested_barriers():
    """The highest record of a barrier in a fork batch wins (redis.clj:97-103), whichever of its records made the
    slot; a batch of new barriers only (no record finds its barrier in the table) takes the pass without the
    winner word and must leave the same table."""
    from distributed_join_amd import capi
    from distributed_join_amd.engine import FORK_REC
    e = _engine(capacity_barriers=1 << 16, arena_bytes=1 << 30, max_count=999, payload_stride=64, max_batch=1 << 15)
    rng = np.random.default_rng(5)

    def recs_for(ids, counts):
        r = np.zeros(len(ids), dtype=FORK_REC)
        r["fp_lo"] = ids * 2 + 2
        r["fp_hi"] = ids + 77
        r["count"] = counts
        r["deadline_ms"] = 1_700_000_000_000 + ids
        return r

    def check(ids, want):
        for b, c in zip(ids.tolist(), want.tolist()):
            meta, lens, _, _ = e.snapshot_one(b * 2 + 2, b + 77)
            assert int(meta["count"]) == c and len(lens) == c and (lens == capi.DJ_NIL).all()

    # 1. new barriers only
    ids = np.arange(20000, dtype=np.uint64)
    counts = rng.integers(1, 40, len(ids)).astype(np.uint32)
    st = e.fork_batch(recs_for(ids, counts))
    assert (st == capi.DJ_FORK_CREATED).all()
    pick = rng.choice(len(ids), 300, replace=False)
    check(ids[pick], counts[pick])
    # 2. new barriers, each named two to four times in the batch: the last record's count stays
    base = np.arange(30000, 36000, dtype=np.uint64)
    reps = rng.integers(2, 5, len(base))
    ids2 = np.repeat(base, reps)
    order = rng.permutation(len(ids2))
    ids2 = ids2[order]
    counts2 = rng.integers(1, 40, len(ids2)).astype(np.uint32)
    st = e.fork_batch(recs_for(ids2, counts2))
    assert (st == capi.DJ_FORK_CREATED).all()
    last = {}
    for b, c in zip(ids2.tolist(), counts2.tolist()):
        last[b] = c
    pick = rng.choice(base, 300, replace=False)
    check(pick, np.array([last[int(b)] for b in pick]))
    # 3. one known barrier among new ones
    ids3 = np.concatenate([np.arange(40000, 41000, dtype=np.uint64), ids[:1]])
    counts3 = np.full(len(ids3), 7, np.uint32)
    st = e.fork_batch(recs_for(ids3, counts3))
    assert (st == capi.DJ_FORK_CREATED).all()
    check(ids3[-3:], counts3[-3:])
    check(ids[1:4], counts[1:4])
    # 4. and a batch of new ones again after a contested one
    ids4 = np.arange(50000, 52000, dtype=np.uint64)
    counts4 = rng.integers(1, 999, len(ids4)).astype(np.uint32)
    e.fork_batch(recs_for(ids4, counts4))
    pick = rng.choice(len(ids4), 100, replace=False)
    check(ids4[pick], counts4[pick])
    assert e.stats()["live"] == 20000 + 6000 + 1000 + 2000
    e.close()


@pytest.mark.gpu
def test_fork_winner_at_a_million_records():
    """One fork batch of 2^20 records over 600 000 barriers, 200 000 of which are live from an earlier batch and many
    named twice or more inside the batch, each record with a count of its own: the last record of a barrier decides
    (redis.clj:97-103).  Checked for every barrier through the positions a join batch accepts: count - 1 is taken,
    count is refused."""
    from distributed_join_amd import capi
    from distributed_join_amd.engine import FORK_REC, JOIN_REC
    nb, n = 600_000, 1 << 20
    e = _engine(capacity_barriers=nb, arena_bytes=3 << 30, max_count=64, payload_stride=64, max_batch=n)
    rng = np.random.default_rng(77)
    fp = capi.fingerprint_many([f"w{b:x}~t" for b in range(nb)])

    def fork(bidx, counts):
        r = np.zeros(len(bidx), dtype=FORK_REC)
        r["fp_lo"], r["fp_hi"] = fp[bidx, 0], fp[bidx, 1]
        r["count"] = counts
        r["deadline_ms"] = 1_700_000_000_000 + 3_600_000
        st = e.fork_batch(r)
        assert (st == capi.DJ_FORK_CREATED).all()

    old = np.arange(200_000)
    fork(old, rng.integers(1, 65, len(old)))
    # every barrier at least once, the rest of the 2^20 records are repeats
    bidx = np.concatenate([np.arange(nb), rng.integers(0, nb, n - nb)])
    rng.shuffle(bidx)
    counts = rng.integers(1, 65, n).astype(np.uint32)
    fork(bidx, counts)
    final = np.zeros(nb, np.uint32)
    final[bidx] = counts                      # numpy keeps the last assignment of a repeated index, as the engine must
    for pos, want in ((final - 1, (capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE)), (final, (capi.DJ_JOIN_BAD_POSITION,))):
        j = np.zeros(nb, dtype=JOIN_REC)
        j["fp_lo"], j["fp_hi"] = fp[:, 0], fp[:, 1]
        j["position"], j["len"] = pos, 64
        st = e.join_batch(j, np.zeros((nb, 64), np.uint8), 1_700_000_000_000)
        bad = np.nonzero(~np.isin(st, want))[0]
        assert len(bad) == 0, (bad[:10], st[bad[:10]], final[bad[:10]])
    assert e.stats()["live"] == nb
    e.close()
