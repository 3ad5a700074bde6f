"""Bit-exact parity of the HIP engine with the CPU oracle on seeded random op-logs.

Each op-log mixes forks (re-forks included), joins with duplicates inside one batch (last
writer wins, redis.clj:135), joins with max-wait, unknown ids and out-of-range positions,
sweeps at a moving virtual clock, acks with failures (retry counters, exhaustion), joins
that land between a sweep and its ack (dirty complete, redis.clj:177-188), recover passes,
acks that name a barrier twice (continuations of two pops acked together), and failed publishes whose
metadata snapshot carries another :count than the barrier has by then (re-forked meanwhile): the
snapshot's count is written back over messages stored under the other one (core.clj:107-111).
"""
import numpy as np
import pytest

from distributed_join_amd import capi
from tests.parity import ParityDriver, payload_bytes

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000
SWEEP_MODE = [0]


@pytest.fixture(autouse=True, params=[0, 1], ids=["scan", "readylist"])
def sweep_mode(request):
    """Every op-log runs twice: sweeps that scan the queue words, and sweeps that drain the ready list."""
    SWEEP_MODE[0] = request.param
    yield request.param


def make_pair(oracle_lib, ids, stride, max_count=999, capacity=None, arena=256 << 20, max_batch=1 << 16, delay=2000,
              hot_mode=-1, max_message=0):
    from distributed_join_amd.engine import Engine
    e = Engine(capacity_barriers=capacity or max(len(ids), 64), arena_bytes=arena, max_count=max_count,
               payload_stride=stride, max_batch=max_batch, check_publish_delay_ms=delay, host_cap_conts=1 << 16,
               host_cap_positions=1 << 20, max_message_bytes=max_message)
    e.set_hot_path(hot_mode)     # -1 automatic, 0 serial walk, 1 parallel kernels for long chains
    e.set_sweep_path(SWEEP_MODE[0])
    # the oracle keeps whole messages: its row width is the longest message the engine accepts
    o = oracle_lib.Oracle(ids, max_count=max_count, payload_stride=max(stride, max_message), check_publish_delay_ms=delay)
    return e, o, ParityDriver(e, o, ids)


def run_oplog(d, rng, n_barriers, counts, stride, rounds, batch, hot=None, var_len=False, unknown=0, on_round=None,
              long_to=None, recount=None, fail_every=5):
    """counts[b] is the count barrier b is forked with; a re-fork keeps it unless recount=(lo, hi) draws a new one
    for every fork record (a failed publish that was in flight over such a re-fork then writes the older count
    back, core.clj:107-111).  on_round(r) runs after round r (continuations may be in flight at that moment).
    long_to: longest accepted message when the engine takes messages beyond its inline row (40 % of the messages
    are then longer than a row).  One publish in fail_every fails."""
    now = T0
    forked = np.zeros(n_barriers, dtype=bool)
    version = 0
    inflight = []   # (engine conts, oracle conts) not yet acked
    for r in range(rounds):
        now += int(rng.integers(1, 400))
        # forks: a random subset, some of them re-forks; duplicates inside the batch allowed
        nf = int(rng.integers(1, max(2, n_barriers // 3)))
        fb = rng.integers(0, n_barriers, nf)
        mw = rng.integers(1, 4, nf)
        if recount:
            newc = rng.integers(recount[0], recount[1] + 1, nf)
            d.fork(fb, newc, now + 1000 * mw)
            counts[fb] = newc                     # of several records of a barrier the last stays, here as there
        else:
            d.fork(fb, counts[fb], now + 1000 * mw)
        forked[fb] = True
        for _ in range(int(rng.integers(1, 4))):
            now += int(rng.integers(0, 300))
            n = int(rng.integers(1, batch))
            jb = rng.integers(0, n_barriers + unknown, n)
            if hot is not None:
                jb[rng.random(n) < 0.3] = hot
            known = jb < n_barriers
            cnt = np.where(known, counts[np.minimum(jb, n_barriers - 1)], 4)
            pos = (rng.random(n) * cnt).astype(np.int64)
            pos[rng.random(n) < 0.02] += 1000          # out of range
            version += 1
            top = long_to if long_to else stride
            pitch = top + 16 if (var_len or long_to) else stride
            pay = payload_bytes(jb, pos % 1024, np.full(n, version) + np.arange(n), pitch)
            ln = rng.integers(0, stride + 1, n) if (var_len or long_to) else np.full(n, stride)
            if long_to:
                big = rng.random(n) < 0.4
                ln[big] = rng.integers(stride + 1, long_to + 1, int(big.sum()))
            if var_len or long_to:
                ln[rng.random(n) < 0.01] = top + 5      # too long
            jmw = np.where(rng.random(n) < 0.2, rng.integers(1, 5, n), 0)
            d.join(jb, pos, ln, jmw, pay, now)
            if rng.random() < 0.7:
                now += int(rng.integers(0, 1500))
                inflight.append(d.sweep(now))
            if inflight and rng.random() < 0.6:
                ec, oc = inflight.pop(0)
                if inflight and rng.random() < 0.4:
                    # the continuations of two pops acked in one go: a barrier that was popped, re-queued by a
                    # recover pass or a failed publish and popped again is then in the list twice
                    ec2, oc2 = inflight.pop(0)
                    ec, oc = np.concatenate([ec, ec2]), np.concatenate([oc, oc2])
                now += int(rng.integers(0, 50))
                d.ack(ec, oc, lambda k: (k * 7 + r) % fail_every != 0, now)
        if rng.random() < 0.5:
            now += int(rng.integers(0, 3000))
            d.recover(now)
        if on_round is not None:
            on_round(r)
    while inflight:
        ec, oc = inflight.pop(0)
        d.ack(ec, oc, lambda k: True, now)
    now += 10_000
    ec, oc = d.sweep(now)
    d.ack(ec, oc, lambda k: k % 2 == 0, now)
    d.check_all_snapshots()


@pytest.mark.parametrize("hot_mode", [-1, 1, 2, 3, 4, 6, 8, 16])
def test_small_counts_many_duplicates(oracle_lib, hot_mode):
    rng = np.random.default_rng(1)
    nb = 40
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)] + [f"ghost{k}~x" for k in range(5)]
    counts = rng.integers(1, 9, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=25, batch=400, unknown=5)
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0, 2, 4, 16])
def test_thousands_of_barriers_with_ten_events_a_batch(oracle_lib, hot_mode):
    """About ten events per barrier and batch on thousands of barriers: chains of 9..16 events, which the
    engine (left to itself) moves from the long-chain kernels to the 16-event register replay after the first
    batch; the pinned modes run the same stream through each form alone."""
    rng = np.random.default_rng(21)
    nb = 6000
    ids = [f"mid{b:x}~t{b:x}" for b in range(nb)]
    counts = rng.integers(2, 17, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode, max_batch=1 << 17)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=5, batch=60000)
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode,n", [(-1, 1_200_000), (16, 1_200_000), (16, 400_000), (16, 700_000),
                                        (16, 1_500_000), (-1, 2_000_000), (16, 3_000_000)])
def test_batch_a_little_above_a_million_events(oracle_lib, hot_mode, n):
    """1.2 M events in one batch on 300 000 barriers (a rank of a routed job with 2^20 events per rank receives a
    little more or less than 2^20): the bucketed passes hold up to 1280 events per bucket on average.  Statuses
    element by element, then the stored state of a sample of the barriers.  The smaller batches make tiles of two
    and three rounds in the first pass (1.2 M: 512 tiles of three).  Above 1024 x 1280 events a bucket holds the
    keys of two or four sub-ranges of its slots, which the second pass takes one after the other (1.5 M and 2 M:
    two, 3 M: four)."""
    rng = np.random.default_rng(77)
    nb = 300_000
    ids = [f"big{b:x}~t{b:x}" for b in range(nb)]
    counts = rng.integers(2, 10, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, max_count=16, arena=1 << 30, max_batch=1 << (22 if n > 1 << 21 else 21),
                        hot_mode=hot_mode)
    try:
        now = T0 + 5
        for lo in range(0, nb, 1 << 16):
            idx = np.arange(lo, min(lo + (1 << 16), nb))
            d.fork(idx, counts[idx], np.full(len(idx), now + 60_000))
        for r in range(2):
            now += 7
            jb = rng.integers(0, nb + 1000, n) % nb
            pos = (rng.random(n) * counts[jb]).astype(np.int64)
            pos[rng.random(n) < 0.01] += 1000          # out of range
            jmw = np.where(rng.random(n) < 0.1, rng.integers(1, 5, n), 0)
            d.join(jb, pos, np.full(n, 64), jmw, payload_bytes(jb, pos % 1024, np.full(n, r + 1) + np.arange(n), 64), now)
        d.check_all_snapshots(which=rng.choice(nb, 2000, replace=False))
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 16])
def test_max_waits_short_and_long_in_one_batch(oracle_lib, hot_mode):
    """Max-waits of seconds, hours and weeks in the same batches (the bucketed passes carry a short one in the event's
    key and read a long one from its record): answers element by element, then sweeps at the dates the short ones
    fall due, at the hours' and at the weeks'."""
    rng = np.random.default_rng(313)
    nb = 5000
    ids = [f"mw{b:x}~t{b:x}" for b in range(nb)]
    counts = rng.integers(2, 9, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode, max_batch=1 << 17)
    try:
        now = T0 + 11
        d.fork(np.arange(nb), counts, np.full(nb, now + 40 * 24 * 3_600_000))
        choices = np.array([0, 0, 2, 3, 16383, 16384, 20_000, 5_000_000])
        for r in range(3):
            now += 40
            n = 30_000
            jb = rng.integers(0, nb, n)
            pos = (rng.random(n) * counts[jb]).astype(np.int64)
            pos[pos == counts[jb] - 1] = 0                   # the last position stays open: nothing completes
            jmw = choices[rng.integers(0, len(choices), n)]
            d.join(jb, pos, np.full(n, 64), jmw, payload_bytes(jb, pos, np.full(n, r + 1) + np.arange(n), 64), now)
        for at in (now + 2_000, now + 3_000, now + 16_383_000, now + 16_384_000 + 80, now + 20_000_000 + 80, now + 5_000_000_000 + 80):
            ec, oc = d.sweep(at)
            d.ack(ec, oc, lambda k: True, at)
        d.check_all_snapshots(range(0, nb, 50))
    finally:
        e.close()


@pytest.mark.parametrize("hot_events", [2300, 6000])
def test_hot_barrier_inside_a_batch_of_fat_buckets(oracle_lib, hot_events):
    """1.5 M events in the bucketed form (two sub-ranges per bucket), one barrier with thousands of them and no earlier
    batch to warn of it.  2300: its sub-range passes the 2048 keys a block holds while its bucket stays within 4096 (that
    sub-range's events are chained on their slots for the long-chain pass, every other sub-range replays as usual);
    6000: the bucket itself overflows and the whole batch falls back to chains.  Statuses element by element, then the
    hot barrier's and a sample of the others' stored state."""
    rng = np.random.default_rng(91)
    nb, n = 300_000, 1_500_000
    ids = [f"fat{b:x}~t{b:x}" for b in range(nb)]
    counts = rng.integers(2, 10, nb)
    counts[5] = 9
    e, o, d = make_pair(oracle_lib, ids, 64, max_count=16, arena=1 << 30, max_batch=1 << 21, hot_mode=16)
    try:
        now = T0 + 5
        d.fork(np.arange(nb), counts, np.full(nb, now + 60_000))
        for r in range(2):
            now += 7
            jb = rng.integers(0, nb, n)
            jb[rng.choice(n, hot_events, replace=False)] = 5
            pos = (rng.random(n) * counts[jb]).astype(np.int64)
            pos[jb == 5] = rng.integers(0, 8, int((jb == 5).sum()))      # position 8 stays open: the barrier never completes
            jmw = np.where(rng.random(n) < 0.1, rng.integers(1, 5, n), 0)
            d.join(jb, pos, np.full(n, 64), jmw, payload_bytes(jb, pos % 1024, np.full(n, r + 1) + np.arange(n), 64), now)
        d.check_all_snapshots(which=np.concatenate([[5], rng.choice(nb, 1000, replace=False)]))
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0, 1, 4, 16])
def test_hot_barrier_long_chains(oracle_lib, hot_mode):
    rng = np.random.default_rng(2)
    nb = 30
    ids = [f"hot{b}~t{b}" for b in range(nb)]
    counts = rng.integers(2, 64, nb)
    counts[3] = 64
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=20, batch=3000, hot=3)
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0, 16])
def test_chain_longer_than_one_lane_walks(oracle_lib, hot_mode):
    """A third of a 30 000-event batch on one barrier that nobody knew to be hot: the serial long-chain
    kernel gives the walk up after 256 links (batch size / 512, at least 256) and scans the batch instead (every batch with the serial
    kernels pinned, the first batch when the engine picks)."""
    rng = np.random.default_rng(31)
    nb = 40
    ids = [f"storm{b}~t{b}" for b in range(nb)]
    counts = rng.integers(2, 64, nb)
    counts[5] = 33
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode, max_batch=1 << 15)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=5, batch=30001, hot=5)
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode", [-1, 1, 16])
def test_many_hot_barriers_at_once(oracle_lib, hot_mode):
    """Every barrier receives hundreds of events per batch, so all of them get flagged hot: more hot barriers
    than one block of the link pass keeps lists for (64), which exercises the block-local lists, their
    overflow into direct exchanges and the parallel long-chain kernels with many registered barriers."""
    rng = np.random.default_rng(12)
    nb = 90
    ids = [f"mh{b}~t{b}" for b in range(nb)]
    counts = rng.integers(2, 65, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode, max_batch=1 << 17)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=6, batch=60000)
    finally:
        e.close()


@pytest.mark.parametrize("lo,hi,hot_mode", [(1, 12, -1), (1, 12, 4), (5, 8, -1), (40, 90, -1), (2, 300, -1), (2, 300, 1), (55, 70, 4), (1, 12, 16), (40, 90, 16), (2, 300, 16)])
def test_failed_publish_writes_an_older_count_back(oracle_lib, lo, hi, hot_mode):
    """Barriers are re-forked with another count while continuations of their earlier lives are in flight, and every
    second publish fails: the write-back sets a count below or above the one the stored messages arrived under
    (presence word and bitmap forms, slabs that have to grow; counts 5..8 share one slab size, so there the count
    goes down and up over stored messages without the slab ever changing)."""
    rng = np.random.default_rng(100 + lo + hi + hot_mode)
    nb = 60
    counts = rng.integers(lo, hi + 1, nb)
    ids = [f"rc-{lo}-{hi}-{i}" for i in range(nb)]
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode)
    run_oplog(d, rng, nb, counts, 64, rounds=40, batch=300, recount=(lo, hi), fail_every=2, hot=3 if hot_mode == 1 else None)
    e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0, 4, 16])
def test_count_written_back_below_stored_messages_same_slab_size(oracle_lib, hot_mode):
    """core.clj:107-111 step by step.  A barrier forked with count 5 times out and its publish is in flight; it is
    forked again with count 8 and gets positions 5 and 6; the publish fails and writes count 5 back (same slab size:
    8 positions), then the barrier is popped again.  Redis now holds 2 fields that reads no longer show, and HLEN
    counts them: of a batch with positions 0, 1, 2, 3 the third event answers COMPLETE (HLEN 5 = count), the fourth
    makes HLEN 6 and waits again, and its max-wait finds the barrier queued by the third."""
    ids = ["back~1", "other~1"]
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode, delay=2000)
    now = T0
    d.fork([0], [5], [now + 1000])
    now += 1500                                       # past the deadline
    ec1, oc1 = d.sweep(now)                           # the timeout publish of count 5, in flight
    assert len(oc1) == 1
    d.fork([0], [8], [now + 60_000])
    pay = payload_bytes(np.array([0, 0]), np.array([5, 6]), np.array([1, 2]), 64)
    assert d.join([0, 0], [5, 6], [64, 64], [0, 0], pay, now).tolist() == [capi.DJ_JOIN_WAITING] * 2
    now += 10
    assert list(d.ack(ec1, oc1, lambda k: False, now).values()) == [capi.DJ_ACK_RETRY]     # count 5 is back, queued at now
    ec2, oc2 = d.sweep(now)                           # popped again: in no primary queue
    assert len(oc2) == 1 and int(oc2[0]["count"]) == 5
    now += 10
    pay = payload_bytes(np.zeros(4, np.int64), np.arange(4), np.arange(4) + 10, 64)
    st = d.join([0, 0, 0, 0], [0, 1, 2, 3], [64] * 4, [0, 0, 0, 1], pay, now)
    assert st.tolist() == [capi.DJ_JOIN_WAITING, capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE, capi.DJ_JOIN_WAITING]
    # a position at or beyond the count is refused although its message is still stored
    assert d.join([0], [5], [64], [0], pay[:1], now).tolist() == [capi.DJ_JOIN_BAD_POSITION]
    ec3, oc3 = d.sweep(now)                           # queued at now by the third event: 5 positions, 4 of them present
    assert len(oc3) == 1
    d.ack(ec2, oc2, lambda k: True, now)              # the publish of the second pop succeeds, but joins came after
    d.ack(ec3, oc3, lambda k: True, now)              # it was read (dirty complete, redis.clj:177-188)
    now += 5000
    ec4, oc4 = d.sweep(now)
    d.ack(ec4, oc4, lambda k: True, now)
    d.check_all_snapshots()
    e.close()


@pytest.mark.parametrize("hot_mode", [-1, 0, 1, 4, 16])
def test_large_counts_multiword_bitmaps(oracle_lib, hot_mode):
    rng = np.random.default_rng(3)
    nb = 12
    ids = [f"big{b}~t{b}" for b in range(nb)]
    counts = np.array([65, 128, 129, 200, 999, 64, 63, 500, 70, 1, 2, 998])
    e, o, d = make_pair(oracle_lib, ids, 64, hot_mode=hot_mode)
    try:
        run_oplog(d, rng, nb, counts, 64, rounds=15, batch=4000, hot=4)
    finally:
        e.close()


def test_variable_length_messages(oracle_lib):
    rng = np.random.default_rng(4)
    nb = 50
    ids = [f"v{b}~t{b}" for b in range(nb)]
    counts = rng.integers(1, 20, nb)
    e, o, d = make_pair(oracle_lib, ids, 256)
    try:
        run_oplog(d, rng, nb, counts, 256, rounds=15, batch=500, var_len=True)
    finally:
        e.close()


def test_kilobyte_messages(oracle_lib):
    rng = np.random.default_rng(5)
    nb = 30
    ids = [f"k{b}~t{b}" for b in range(nb)]
    counts = rng.integers(1, 17, nb)
    e, o, d = make_pair(oracle_lib, ids, 1024)
    try:
        run_oplog(d, rng, nb, counts, 1024, rounds=10, batch=300, var_len=True)
    finally:
        e.close()


def test_config2_shape_scaled(oracle_lib):
    """BASELINE config #2 scaled down: N barriers x count 8, 64-byte messages, uniformly shuffled
    arrival in 8 batches, one sweep per batch, every ack ok: N complete continuations."""
    rng = np.random.default_rng(6)
    nb, c = 20000, 8
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)]
    e, o, d = make_pair(oracle_lib, ids, 64, max_count=8, capacity=nb)
    try:
        now = T0
        d.fork(np.arange(nb), np.full(nb, c), np.full(nb, now + 3_600_000))
        ev = rng.permutation(nb * c)
        fired = 0
        for k in range(8):
            now += 1
            part = ev[k * nb:(k + 1) * nb]
            jb, pos = part // c, part % c
            st = d.join(jb, pos, np.full(nb, 64), np.zeros(nb, np.int64), payload_bytes(jb, pos, np.zeros(nb), 64), now)
            ec, oc = d.sweep(now)
            assert all(int(x["flags"]) == 0 for x in ec)
            assert len(ec) == int((st == 2).sum())
            fired += len(ec)
            d.ack(ec, oc, lambda k_: True, now)
        assert fired == nb
        assert e.stats()["live"] == 0 and o.live_barriers() == 0
    finally:
        e.close()


def test_checkpoint_restore_mid_oplog(oracle_lib, tmp_path):
    """Every third round the engine is saved to a file, destroyed, and a NEW engine restored from the
    file carries on (with continuations in flight, tombstones, freed slabs and retry counters in the
    table); the oracle runs through uninterrupted.  Everything must stay bit-exact."""
    from distributed_join_amd.engine import Engine
    rng = np.random.default_rng(77)
    nb, stride = 300, 64
    ids = [f"ck{b:x}~t{b % 7}" for b in range(nb)]
    counts = rng.integers(1, 80, nb)          # includes multi-word bitmaps (count > 64)
    kw = dict(capacity_barriers=nb, arena_bytes=64 << 20, max_count=999, payload_stride=stride, max_batch=1 << 16,
              check_publish_delay_ms=2000, host_cap_conts=1 << 16, host_cap_positions=1 << 20)
    e, o, d = make_pair(oracle_lib, ids, stride, capacity=nb, arena=64 << 20)
    path = tmp_path / "engine.ckpt"
    swaps = []

    def swap(r):
        if r % 3 != 2:
            return
        d.e.checkpoint_save(path)
        before = d.e.stats()
        d.e.close()
        d.e = Engine(**kw)
        d.e.checkpoint_load(path)
        assert d.e.stats() == before
        swaps.append(r)

    run_oplog(d, rng, nb, counts, stride, rounds=14, batch=500, on_round=swap)
    assert len(swaps) == 4
    d.e.close()


def test_checkpoint_load_rejects_other_geometry(tmp_path):
    from distributed_join_amd.engine import Engine
    a = Engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=64)
    path = tmp_path / "a.ckpt"
    a.checkpoint_save(path)
    a.close()
    b = Engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=128, max_batch=64)
    with pytest.raises(RuntimeError, match="checkpoint"):
        b.checkpoint_load(path)
    with pytest.raises(RuntimeError, match="checkpoint"):
        b.checkpoint_load(tmp_path / "missing.ckpt")
    junk = tmp_path / "junk.ckpt"
    junk.write_bytes(b"not a checkpoint" * 100)
    with pytest.raises(RuntimeError, match="checkpoint"):
        b.checkpoint_load(junk)
    b.close()


def test_damaged_checkpoint_is_refused_before_the_engine_is_touched(oracle_lib, tmp_path):
    """A truncated file, a flipped byte and a free list that names a unit outside the arena are all refused,
    and the engine that tried to load them still holds exactly what it held (compared with the oracle).  A save
    replaces its target only when the new file is complete: no .tmp file stays behind."""
    import os
    rng = np.random.default_rng(12)
    nb = 60
    ids = [f"ck{b}~t" for b in range(nb)]
    counts = rng.integers(1, 9, nb)
    e, o, d = make_pair(oracle_lib, ids, 64, capacity=nb)
    try:
        d.fork(np.arange(nb), counts, np.full(nb, T0 + 60_000))
        jb = rng.integers(0, nb, 300)
        pos = (rng.random(300) * counts[jb]).astype(np.int64)
        d.join(jb, pos, np.full(300, 64), np.zeros(300, np.int64), payload_bytes(jb, pos, np.arange(300), 64), T0 + 5)
        ec, oc = d.sweep(T0 + 6)
        d.ack(ec, oc, lambda k: True, T0 + 7)          # frees slabs: the free lists are not empty
        good = tmp_path / "good.ckpt"
        e.checkpoint_save(good)
        assert sorted(os.listdir(tmp_path)) == ["good.ckpt"]
        raw = good.read_bytes()
        for name, data in (("short.ckpt", raw[:len(raw) - 100]), ("long.ckpt", raw + b"x"),
                           ("flip.ckpt", raw[:4000] + bytes([raw[4000] ^ 0x40]) + raw[4001:])):
            bad = tmp_path / name
            bad.write_bytes(data)
            with pytest.raises(RuntimeError, match="truncated or damaged"):
                e.checkpoint_load(bad)
            d.check_all_snapshots()
        # the state goes on after the refusals, and the good file still loads
        d.join(jb[:50], pos[:50], np.full(50, 64), np.zeros(50, np.int64), payload_bytes(jb[:50], pos[:50], np.arange(50) + 7, 64), T0 + 9)
        e.checkpoint_save(good)                         # over the old one
        e.checkpoint_load(good)
        d.check_all_snapshots()
    finally:
        e.close()


@pytest.mark.parametrize("hot_mode,recount", [(-1, None), (1, None), (-1, (1, 90))])
def test_messages_longer_than_a_row(oracle_lib, tmp_path, hot_mode, recount):
    """Engine with 64-byte inline rows that accepts messages up to 2 KiB: 40 % of the messages take a
    block of their own.  Overwrites (same and other size class, long over short and back), re-forks,
    failed publishes with retries, drops, recover, and a save / restore into a new engine every
    fourth round.  At the end every barrier is completed and deleted: every arena unit ever handed out
    must be back in a free list (nothing leaked, nothing freed twice).  With recount the barriers are re-forked with
    other counts and every second publish fails: long messages then also sit at positions beyond a count that a
    failed publish wrote back, and move with their barrier when its slab has to grow."""
    from distributed_join_amd.engine import Engine
    rng = np.random.default_rng(99)
    nb, stride, M = 250, 64, 2048
    ids = [f"long{b}~t" for b in range(nb)]
    counts = rng.integers(1, 24, nb)
    counts[:4] = [70, 130, 1, 64]
    e, o, d = make_pair(oracle_lib, ids, stride, capacity=nb, arena=64 << 20, max_message=M, hot_mode=hot_mode)
    kw = dict(capacity_barriers=nb, arena_bytes=64 << 20, max_count=999, payload_stride=stride, max_batch=1 << 16,
              check_publish_delay_ms=2000, host_cap_conts=1 << 16, host_cap_positions=1 << 20, max_message_bytes=M)
    path = tmp_path / "long.ckpt"

    def swap(r):
        if r % 4 != 3:
            return
        d.e.checkpoint_save(path)
        before = d.e.stats()
        d.e.close()
        d.e = Engine(**kw)
        d.e.set_hot_path(hot_mode)
        d.e.checkpoint_load(path)
        assert d.e.stats() == before

    # with the parallel long-chain kernels forced on, a third of the joins go to barrier 1 (count 130)
    run_oplog(d, rng, nb, counts, stride, rounds=12, batch=400, long_to=M, on_round=swap, hot=1 if hot_mode == 1 else None,
              recount=recount, fail_every=2 if recount else 5)
    # wind down: fork everything again (resets what is left), complete every barrier, publish, ack
    now = T0 + 10_000_000
    d.fork(np.arange(nb), counts, np.full(nb, now + 50_000))
    for b0 in range(0, nb, 50):
        bb = np.repeat(np.arange(b0, min(b0 + 50, nb)), counts[b0:b0 + 50])
        pp = np.concatenate([np.arange(c) for c in counts[b0:b0 + 50]])
        ln = rng.integers(0, M + 1, len(bb))
        d.join(bb, pp, ln, np.zeros(len(bb), np.int64), payload_bytes(bb, pp, np.full(len(bb), 7777), M + 16), now)
    ec, oc = d.sweep(now)
    assert len(ec) == nb
    out = d.ack(ec, oc, lambda k: True, now)
    assert set(out.values()) == {capi.DJ_ACK_DONE}
    s = d.e.stats()
    assert s["live"] == 0
    assert s["arena_units_free"] == s["arena_units_used"] - 1, s       # unit 0 is reserved
    d.e.close()


def test_messages_of_the_reference_limit(oracle_lib):
    """The reference's production limit is 262144 characters per message (config/prod/config.edn:38,
    store.clj:24-26), i.e. up to 786432 UTF-8 bytes: an engine configured for it takes messages of exactly
    262144 and 786432 bytes (and every size class between), overwrites them by shorter and longer ones, hands
    them back byte for byte, and gives the blocks back when the barriers go."""
    from distributed_join_amd.store import engine_message_bytes
    rng = np.random.default_rng(262144)
    M = engine_message_bytes()
    assert M == 786432
    nb, stride = 6, 256
    ids = [f"big{b}~t" for b in range(nb)]
    counts = np.array([1, 2, 3, 2, 1, 2])
    e, o, d = make_pair(oracle_lib, ids, stride, max_count=8, capacity=64, arena=96 << 20, max_message=M, max_batch=64)
    d.oracle_caps = dict(cap_conts=16, cap_positions=64)      # the oracle's sweep rows are M bytes wide
    try:
        now = T0
        d.fork(np.arange(nb), counts, np.full(nb, now + 60_000))
        pitch = M + 16
        sizes = [262144, M, 262145, 65, 256, 257, 100_000, 524288, 0, M - 1]
        for rnd in range(3):
            now += 7
            jb = rng.integers(0, nb, len(sizes))
            pos = (rng.random(len(sizes)) * counts[jb]).astype(np.int64)
            ln = np.array(sizes if rnd != 1 else sizes[::-1])
            if rnd == 2:
                ln[3] = M + 1                       # one byte too long: refused, the position keeps what it had
            pay = payload_bytes(jb, pos, np.full(len(sizes), rnd) + np.arange(len(sizes)), pitch)
            st = d.join(jb, pos, ln, np.zeros(len(sizes), np.int64), pay, now)
            assert (st[ln > M] == capi.DJ_JOIN_TOO_LONG).all()
            d.check_all_snapshots()
        # fill what is missing with long messages, fire everything
        now += 7
        bb = np.repeat(np.arange(nb), counts)
        pp = np.concatenate([np.arange(c) for c in counts])
        ln = rng.integers(262144, M + 1, len(bb))
        d.join(bb, pp, ln, np.zeros(len(bb), np.int64), payload_bytes(bb, pp, np.full(len(bb), 99), pitch), now)
        ec, oc = d.sweep(now)
        assert len(ec) == nb
        assert set(d.ack(ec, oc, lambda k: True, now).values()) == {capi.DJ_ACK_DONE}
        s = e.stats()
        assert s["live"] == 0 and s["arena_units_free"] == s["arena_units_used"] - 1, s
    finally:
        e.close()
