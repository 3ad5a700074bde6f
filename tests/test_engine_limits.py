"""Limits and refusals of the C ABI on the GPU: empty batches, argument checks, the clock range,
record limits (store.clj:32-34), a full table and an exhausted arena.  A refusal must leave the
engine usable and the state untouched."""
import numpy as np
import pytest

from distributed_join_amd import capi
from distributed_join_amd.engine import EngineError

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000


def _engine(**kw):
    from distributed_join_amd.engine import Engine
    base = dict(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=256)
    base.update(kw)
    return Engine(**base)


def _forks(ids, count, deadline=T0 + 60_000):
    fps = capi.fingerprint_many(ids)
    r = np.zeros(len(ids), dtype=capi.FORK_REC)
    r["fp_lo"], r["fp_hi"], r["count"], r["deadline_ms"] = fps[:, 0], fps[:, 1], count, deadline
    return r


def _joins(ids, pos, length=4, mw=0):
    fps = capi.fingerprint_many(ids)
    r = np.zeros(len(ids), dtype=capi.JOIN_REC)
    r["fp_lo"], r["fp_hi"], r["position"], r["len"], r["max_wait_s"] = fps[:, 0], fps[:, 1], pos, length, mw
    return r


def test_empty_batches_and_empty_engine():
    e = _engine()
    assert len(e.fork_batch(np.zeros(0, dtype=capi.FORK_REC))) == 0
    assert len(e.join_batch(np.zeros(0, dtype=capi.JOIN_REC), np.zeros((0, 64), np.uint8), T0)) == 0
    conts, lens, mws, pay, more = e.sweep(T0 + 10**9)
    assert len(conts) == 0 and len(lens) == 0 and not more
    conts, st, _ = e.recover(T0 + 10**9)
    assert len(conts) == 0
    assert e.stats()["live"] == 0
    meta, lens, _, _ = e.snapshot_one(*capi.fingerprint("nobody~1"))
    assert int(meta["flags"]) & capi.DJ_CONT_NO_METADATA and len(lens) == 0
    assert e.queue_scores(*capi.fingerprint("nobody~1")) == (capi.DJ_NOT_QUEUED, capi.DJ_NOT_QUEUED)
    e.close()


def test_clock_outside_the_13_digit_range_is_refused():
    e = _engine()
    assert e.fork_batch(_forks(["a~1"], 2)).tolist() == [capi.DJ_FORK_CREATED]
    for bad in (0, 999_999_999_999, 9_000_000_000_000):
        with pytest.raises(EngineError, match="now_ms"):
            e.join_batch(_joins(["a~1"], 0), np.zeros((1, 64), np.uint8), bad)
        with pytest.raises(EngineError, match="now_ms"):
            e.sweep(bad)
    # still usable, and the refused joins left nothing behind
    assert e.join_batch(_joins(["a~1"], 0), np.zeros((1, 64), np.uint8), T0).tolist() == [capi.DJ_JOIN_WAITING]
    _, lens, _, _ = e.snapshot_one(*capi.fingerprint("a~1"))
    assert lens.tolist() == [4, capi.DJ_NIL]
    e.close()


def test_batch_larger_than_max_batch_is_refused():
    e = _engine(max_batch=16)
    ids = [f"b~{k}" for k in range(17)]
    with pytest.raises(EngineError, match="max_batch"):
        e.fork_batch(_forks(ids, 2))
    assert e.stats()["live"] == 0
    assert (e.fork_batch(_forks(ids[:16], 2)) == capi.DJ_FORK_CREATED).all()
    with pytest.raises(EngineError, match="max_batch"):
        e.join_batch(_joins(ids, 0), np.zeros((17, 64), np.uint8), T0)
    e.close()


def test_record_limits():
    e = _engine(max_count=8)
    ids = ["c~0", "c~1", "c~2", "c~3"]
    st = e.fork_batch(_forks(ids, np.array([0, 9, 8, 1])))
    assert st.tolist() == [capi.DJ_FORK_BAD_COUNT, capi.DJ_FORK_BAD_COUNT, capi.DJ_FORK_CREATED, capi.DJ_FORK_CREATED]
    assert e.stats()["live"] == 2
    j = _joins(["c~2", "c~2", "c~3", "c~0"], np.array([7, 8, 0, 0]), length=np.array([64, 4, 65, 4]))
    st = e.join_batch(j, np.zeros((4, 80), np.uint8), T0)
    assert st.tolist() == [capi.DJ_JOIN_WAITING, capi.DJ_JOIN_BAD_POSITION, capi.DJ_JOIN_TOO_LONG, capi.DJ_JOIN_NOT_FOUND]
    # a message longer than the row pitch handed over is refused even when the engine could hold it
    st = e.join_batch(_joins(["c~2"], 0, length=40), np.zeros((1, 32), np.uint8), T0)
    assert st.tolist() == [capi.DJ_JOIN_TOO_LONG]
    e.close()


def test_full_table_and_exhausted_arena_are_reported_per_record():
    # 1024 slots (the minimum); slabs of count 8 x 64 B = 9 units: the arena below holds 40 of them
    e = _engine(capacity_barriers=64, arena_bytes=64 * (1 + 9 * 40), max_count=8, payload_stride=64, max_batch=2048)
    ids = [f"full~{k}" for k in range(60)]
    st = e.fork_batch(_forks(ids, 8))
    assert (st == capi.DJ_FORK_CREATED).sum() == 40 and (st == capi.DJ_FORK_NO_SPACE).sum() == 20
    s = e.stats()
    assert s["live"] == 40
    ok_ids = [i for i, v in zip(ids, st) if v == capi.DJ_FORK_CREATED]
    no_ids = [i for i, v in zip(ids, st) if v == capi.DJ_FORK_NO_SPACE]
    # refused barriers do not exist; accepted ones work
    assert (e.join_batch(_joins(no_ids, 0), np.zeros((len(no_ids), 64), np.uint8), T0) == capi.DJ_JOIN_NOT_FOUND).all()
    assert (e.join_batch(_joins(ok_ids, 0), np.zeros((len(ok_ids), 64), np.uint8), T0) == capi.DJ_JOIN_WAITING).all()
    # deleting barriers gives their slabs back: the refused ones fit afterwards
    conts, *_ = e.sweep(T0 + 120_000)
    assert len(conts) == 40
    assert (e.ack_batch(conts, None, T0 + 120_000) == capi.DJ_ACK_DONE).all()
    assert (e.fork_batch(_forks(no_ids, 8)) == capi.DJ_FORK_CREATED).all()
    e.close()

    # table: count-1 barriers (1 unit each... posmeta + payload = 2 units), plenty of arena, 1024 slots
    e = _engine(capacity_barriers=64, arena_bytes=8 << 20, max_count=1, payload_stride=64, max_batch=2048)
    ids = [f"slot~{k}" for k in range(1100)]
    st = e.fork_batch(_forks(ids, 1))
    assert (st == capi.DJ_FORK_CREATED).sum() == 1024 and (st == capi.DJ_FORK_TABLE_FULL).sum() == 76
    assert e.stats()["live"] == 1024
    # lookups of absent keys terminate on a completely full table
    assert e.join_batch(_joins(["absent~1"], 0), np.zeros((1, 64), np.uint8), T0).tolist() == [capi.DJ_JOIN_NOT_FOUND]
    e.close()


def test_mass_delete_beyond_capacity_barriers_keeps_slabs_apart():
    """More live barriers than capacity_barriers (the table takes them while slots are left) are all deleted
    at once: every slab must come back through the free lists exactly once, so that the next generation of
    barriers gets a slab each and keeps its own bytes."""
    e = _engine(capacity_barriers=64, arena_bytes=8 << 20, max_count=1, payload_stride=64, max_batch=2048,
                host_cap_conts=2048, host_cap_positions=4096)
    ids = [f"gen~{k}" for k in range(1024)]
    used = None
    for gen in range(3):
        assert (e.fork_batch(_forks(ids, 1)) == capi.DJ_FORK_CREATED).all()
        pay = np.zeros((1024, 64), np.uint8)
        pay[:, :8] = (np.arange(1024, dtype=np.uint64) * 2654435761 + gen).view(np.uint8).reshape(1024, 8)
        assert (e.join_batch(_joins(ids, 0, length=8), pay, T0 + gen) == capi.DJ_JOIN_COMPLETE).all()
        conts, lens, _, out, more = e.sweep(T0 + gen)
        assert len(conts) == 1024 and not more
        fps = capi.fingerprint_many(ids)
        index = {(int(a), int(b)): k for k, (a, b) in enumerate(fps)}
        for c in conts:
            k = index[(int(c["fp_lo"]), int(c["fp_hi"]))]
            assert bytes(out[int(c["pos_base"]), :8]) == bytes(pay[k, :8]), f"generation {gen}: barrier {k} holds another's bytes"
        assert (e.ack_batch(conts, None, T0 + gen) == capi.DJ_ACK_DONE).all()
        st = e.stats()
        assert st["live"] == 0
        # nothing leaks: what the bump pointer handed out sits in the free lists, and does not grow per generation
        assert st["arena_units_free"] == st["arena_units_used"] - 1
        used = used or st["arena_units_used"]
        assert st["arena_units_used"] == used
    e.close()


def test_misaligned_or_null_pointers_are_refused():
    import torch
    e = _engine()
    buf = torch.zeros(4096, dtype=torch.uint8, device=e.device)
    st = e.empty(64)
    with pytest.raises(EngineError, match="align"):
        e.join_batch_dev(1, buf[8:], buf[1024:], 64, T0, st)
    with pytest.raises(EngineError, match="pitch"):
        e.join_batch_dev(1, buf, buf[1024:], 40, T0, st)
    with pytest.raises(EngineError, match="misaligned"):
        e.fork_batch_dev(1, buf[4:], st)
    e.close()


def test_configurations_outside_the_limits_are_refused():
    for kw, what in [(dict(capacity_barriers=0), "capacity_barriers"), (dict(capacity_barriers=1 << 63), "capacity_barriers"),
                     (dict(max_count=1000), "max_count"), (dict(payload_stride=96), "payload_stride"),
                     (dict(max_batch=0), "max_batch"), (dict(arena_bytes=64), "arena_bytes")]:
        with pytest.raises(EngineError, match=what):
            _engine(**kw)


def test_long_messages_when_the_arena_runs_out():
    """Messages beyond the inline row need a block each; when the arena cannot give one the event is
    answered NO_SPACE and its position holds an empty message (the barrier stays consistent and still
    completes).  Blocks come back when the barrier is deleted."""
    # slabs of count 4: 1 + 4 units; 8 barriers = 40 units; blocks of 4 KiB = 64 units each: room for 3
    e = _engine(capacity_barriers=64, arena_bytes=64 * (1 + 40 + 3 * 64 + 10), max_count=4, payload_stride=64,
                max_message_bytes=4096, max_batch=256)
    ids = [f"big~{k}" for k in range(8)]
    assert (e.fork_batch(_forks(ids, 4)) == capi.DJ_FORK_CREATED).all()
    msg = np.arange(5 * 4096, dtype=np.uint32).view(np.uint8)[:5 * 4096].reshape(5, 4096).copy()
    st = e.join_batch(_joins(ids[:5], 0, length=4096), msg, T0)
    assert sorted(st.tolist()) == [capi.DJ_JOIN_WAITING] * 3 + [capi.DJ_JOIN_NO_SPACE] * 2
    for k in range(5):
        meta, lens, _, pay = e.snapshot_one(*capi.fingerprint(ids[k]))
        if st[k] == capi.DJ_JOIN_WAITING:
            assert int(lens[0]) == 4096 and bytes(pay[0][:4096]) == msg[k].tobytes()
        else:
            assert int(lens[0]) == 0
    # a short message over a long one gives the block back at once: the next long message fits
    kept = [k for k in range(5) if st[k] == capi.DJ_JOIN_WAITING]
    assert e.join_batch(_joins([ids[kept[0]]], 0, length=8), np.zeros((1, 16), np.uint8), T0 + 1).tolist() == [capi.DJ_JOIN_WAITING]
    assert e.join_batch(_joins([ids[6]], 1, length=4000), msg[:1], T0 + 2).tolist() == [capi.DJ_JOIN_WAITING]
    _, lens, _, pay = e.snapshot_one(*capi.fingerprint(ids[6]))
    assert lens.tolist() == [capi.DJ_NIL, 4000, capi.DJ_NIL, capi.DJ_NIL] and bytes(pay[1][:4000]) == msg[0].tobytes()[:4000]
    # delete everything: all units are free again
    conts, *_ = e.sweep(T0 + 120_000)
    assert len(conts) == 8
    assert (e.ack_batch(conts, None, T0 + 120_000) == capi.DJ_ACK_DONE).all()
    s = e.stats()
    assert s["live"] == 0 and s["arena_units_free"] == s["arena_units_used"] - 1
    e.close()


@pytest.mark.gpu
def test_hot_path_modes_accepted_and_refused():
    """dj_set_hot_path: -1, the pinned combinations 0..7, and 8 / 16 (+ the same low bits) for the bucketed join passes;
    anything else is refused and leaves the engine as it was."""
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine, EngineError
    e = Engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=1024)
    try:
        for mode in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 23, -1):
            e.set_hot_path(mode)
        for mode in (-2, 24, 100):
            with pytest.raises(EngineError):
                e.set_hot_path(mode)
        # still usable
        rec = np.zeros(1, dtype=capi.FORK_REC)
        rec[0]["fp_lo"], rec[0]["fp_hi"] = capi.fingerprint("m~1")
        rec[0]["count"], rec[0]["deadline_ms"] = 2, 1_700_000_100_000
        assert int(e.fork_batch(rec)[0]) == capi.DJ_FORK_CREATED
    finally:
        e.close()
