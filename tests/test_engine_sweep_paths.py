"""The two forms of the sweep's pop (drain of the ready list / scan of the queue words, picked on the device)
must pop exactly what redis/all-publish-messages pops (redis.clj:283-289: every member with score <= now),
whatever happened between the join batch that completed a barrier and the sweep: a clock that goes back, a
ready list that overflowed, output caps that leave barriers queued, a re-fork, a table compaction."""
import numpy as np
import pytest

from distributed_join_amd import capi
from tests.parity import ParityDriver

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000


MODE = [1]


@pytest.fixture(autouse=True, params=[1, 0], ids=["readylist", "scan"])
def sweep_mode(request):
    MODE[0] = request.param
    yield request.param


def _pair(oracle_lib, n, count=2, max_batch=1 << 12, cap_conts=1 << 14, capacity=None):
    from distributed_join_amd.engine import Engine
    ids = [f"sw{k:x}~t{k:x}" for k in range(n)]
    e = Engine(capacity_barriers=capacity or max(n, 64), arena_bytes=64 << 20, max_count=8, payload_stride=64,
               max_batch=max_batch, host_cap_conts=cap_conts, host_cap_positions=cap_conts * 8)
    e.set_sweep_path(MODE[0])
    o = oracle_lib.Oracle(ids, max_count=8, payload_stride=64)
    d = ParityDriver(e, o, ids)
    for lo in range(0, n, max_batch):
        idx = np.arange(lo, min(lo + max_batch, n))
        d.fork(idx, np.full(len(idx), count), np.full(len(idx), T0 + 3_600_000))
    return e, o, d


def _complete(d, which, count, now):
    for p in range(count):
        st = d.join(which, np.full(len(which), p), np.full(len(which), 64), np.zeros(len(which)),
                    np.full((len(which), 64), p + 1, np.uint8), now)
    assert (st == capi.DJ_JOIN_COMPLETE).all()


def test_sweep_with_an_earlier_clock_leaves_completed_barriers_queued(oracle_lib):
    e, o, d = _pair(oracle_lib, 300)
    _complete(d, np.arange(0, 200), 2, T0 + 100)
    ec, oc = d.sweep(T0 + 50)                   # nothing is due at that time
    assert len(ec) == 0
    d.check_all_snapshots(range(0, 4))          # still queued at T0 + 100
    ec, oc = d.sweep(T0 + 99)
    assert len(ec) == 0
    ec, oc = d.sweep(T0 + 100)                  # inclusive (redis.clj:286)
    assert len(ec) == 200
    d.ack(ec, oc, lambda k: True, T0 + 101)
    _complete(d, np.arange(200, 300), 2, T0 + 102)
    ec, oc = d.sweep(T0 + 102)
    assert len(ec) == 100
    d.ack(ec, oc, lambda k: k % 3 != 0, T0 + 103)     # some publishes fail: re-queued at T0 + 103
    ec, oc = d.sweep(T0 + 103)
    assert len(ec) == 33
    e.close()


def test_drained_list_entry_that_is_not_due_yet_is_popped_by_the_next_sweep(oracle_lib):
    """The drain form proper: a first sweep clears the "list incomplete" state that set_sweep_path leaves, so the sweeps
    that follow drain the list and read region bounds.  A barrier completed at T0 + 100 and met by a sweep whose clock
    is earlier must come out of the first sweep at or after T0 + 100 (ZRANGEBYSCORE 0 now, redis.clj:286), not at its
    fork deadline."""
    e, o, d = _pair(oracle_lib, 300)
    ec, oc = d.sweep(T0 + 1)                    # a full scan: bounds recomputed, the list is complete from here on
    assert len(ec) == 0
    _complete(d, np.arange(0, 200), 2, T0 + 100)
    ec, oc = d.sweep(T0 + 50)                   # the list's entries are not due: it is emptied all the same
    assert len(ec) == 0
    ec, oc = d.sweep(T0 + 99)
    assert len(ec) == 0
    ec, oc = d.sweep(T0 + 100)
    assert len(ec) == 200
    d.ack(ec, oc, lambda k: True, T0 + 101)
    # the same with two clocks that leapfrog: every completion must be popped by the first sweep at or after its date
    for r in range(4):
        which = np.arange(200 + 25 * r, 225 + 25 * r)
        _complete(d, which, 2, T0 + 200 + 10 * r)
        ec, oc = d.sweep(T0 + 195 + 10 * r)     # five milliseconds behind the joins' clock
        assert len(ec) == (25 if r else 0)      # the previous round's barriers
        d.ack(ec, oc, lambda k: True, T0 + 196 + 10 * r)
    ec, oc = d.sweep(T0 + 300)
    assert len(ec) == 25
    e.close()


def test_ready_list_overflow_falls_back_to_the_scan(oracle_lib):
    # max_batch 256: the list's shards hold 64 slots each and a batch of 256 events uses four of them
    n = 3000
    e, o, d = _pair(oracle_lib, n, count=1, max_batch=256)
    for lo in range(0, n, 256):                 # twelve batches without a sweep between them
        idx = np.arange(lo, min(lo + 256, n))
        _complete(d, idx, 1, T0 + 5 + lo // 256)
    ec, oc = d.sweep(T0 + 3)
    assert len(ec) == 0
    ec, oc = d.sweep(T0 + 10)                   # batches 0..5 are due
    assert len(ec) == 6 * 256
    ec2, oc2 = d.sweep(T0 + 10_000)
    assert len(ec2) == n - 6 * 256
    d.ack(ec, oc, lambda k: True, T0 + 10_001)
    d.ack(ec2, oc2, lambda k: True, T0 + 10_001)
    assert e.stats()["live"] == 0
    e.close()


def test_output_caps_leave_the_rest_for_the_next_sweeps(oracle_lib):
    n = 1000
    e, o, d = _pair(oracle_lib, n, count=2, cap_conts=300)
    _complete(d, np.arange(n), 2, T0 + 7)
    want, _, _, _ = o.sweep(T0 + 8)
    assert len(want) == n
    got = {}
    for attempt in range(6):
        conts, lens, mws, pay, more = e.sweep(T0 + 8)
        for c in conts:
            if int(c["flags"]) & capi.CONT_VOID:
                continue
            key = (int(c["fp_lo"]), int(c["fp_hi"]))
            assert key not in got, "barrier popped twice"
            b = int(c["pos_base"])
            assert lens[b:b + 2].tolist() == [64, 64] and pay[b, 0] == 1 and pay[b + 1, 0] == 2
            got[key] = int(c["publish_at"])
        if not more:
            break
    assert len(got) == n and set(got.values()) == {T0 + 7}
    assert {d.by_fp[k] for k in got} == {int(c["id_idx"]) for c in want}
    e.close()


def test_refork_and_compaction_between_completion_and_sweep(oracle_lib):
    n = 600
    e, o, d = _pair(oracle_lib, n, count=2, capacity=400)
    _complete(d, np.arange(0, 400), 2, T0 + 20)
    # half of the completed barriers are forked again before anybody swept: queued at their new deadline only
    d.fork(np.arange(0, 200), np.full(200, 2), np.full(200, T0 + 500))
    live = e.compact()                           # slots move: the ready list names slots of the old table
    assert live == n
    ec, oc = d.sweep(T0 + 20)
    assert len(ec) == 200
    d.ack(ec, oc, lambda k: True, T0 + 21)
    _complete(d, np.arange(0, 100), 2, T0 + 30)  # completed again after the re-fork
    ec, oc = d.sweep(T0 + 400)
    assert len(ec) == 100
    ec2, oc2 = d.sweep(T0 + 500)                 # the re-forked rest times out (partial: nothing joined)
    assert len(ec2) == 100
    d.ack(ec, oc, lambda k: True, T0 + 501)
    d.ack(ec2, oc2, lambda k: True, T0 + 501)
    d.check_all_snapshots(range(0, n, 37))
    e.close()


def test_dates_spread_over_a_large_sparse_table(oracle_lib, tmp_path):
    """Deadlines, max-wait dates and re-queues in a table of 8192 regions of which most are empty: a sweep that reads
    only the regions whose bound has been reached must still pop exactly what is due, after re-forks to earlier
    deadlines, failed publishes (re-queued at now), a switch of the sweep form in mid-run, a compaction and a restore
    from a checkpoint.  The op-log is the parity suite's; what differs is the table: 3000 barriers in 524 288 slots."""
    from distributed_join_amd.engine import Engine
    from tests import test_engine_parity as P
    rng = np.random.default_rng(4242)
    nb = 3000
    ids = [f"sp{k:x}~t{k:x}" for k in range(nb)]
    counts = rng.integers(1, 9, nb)
    kw = dict(capacity_barriers=262_144, arena_bytes=64 << 20, max_count=999, payload_stride=64, max_batch=1 << 16,
              check_publish_delay_ms=2000, host_cap_conts=1 << 16, host_cap_positions=1 << 20)
    e = Engine(**kw)
    e.set_sweep_path(MODE[0])
    o = oracle_lib.Oracle(ids, max_count=999, payload_stride=64, check_publish_delay_ms=2000)
    d = ParityDriver(e, o, ids)
    path = tmp_path / "sparse.ckpt"

    def between(r):
        if r == 5:
            d.e.set_sweep_path(0)                 # a few rounds with plain scans: nobody keeps the regions' bounds
        if r == 8:
            d.e.set_sweep_path(MODE[0])
        if r == 11:
            d.e.compact()
        if r == 14:
            d.e.checkpoint_save(path)
            d.e.close()
            d.e = Engine(**kw)
            d.e.set_sweep_path(MODE[0])
            d.e.checkpoint_load(path)

    P.SWEEP_MODE[0] = MODE[0]
    P.run_oplog(d, rng, nb, counts, 64, rounds=20, batch=1500, on_round=between, fail_every=3)
    d.e.close()
