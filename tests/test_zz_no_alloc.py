"""dj_join_batch_static (work arrays in the library's static device array, nothing allocated by the call)
against dj_join_batch on the same inputs: per-event statuses, popped continuations with their messages and
final snapshots must be equal bit for bit.  Two calls in a row on the same GPU, the second with a smaller
input (the array keeps the first call's contents), then a third after the engines were swept."""
import numpy as np
import pytest

from distributed_join_amd import capi
from distributed_join_amd.engine import EngineError, from_device, to_device
from tests.parity import payload_bytes

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000
NB, COUNT, PITCH = 6000, 8, 64


def _engine(max_batch=1 << 17):
    from distributed_join_amd.engine import Engine
    return Engine(capacity_barriers=NB, arena_bytes=64 << 20, max_count=64, payload_stride=64, max_batch=max_batch,
                  host_cap_conts=NB + 8, host_cap_positions=(NB + 8) * 64)


def _events(rng, n, version):
    """A mixed batch: uniform joins, overwrites of the same position inside the batch, one barrier with
    hundreds of events (long chain), unknown ids, positions out of range, joins with a max-wait, ragged
    messages."""
    b = rng.integers(0, NB, n)
    pos = rng.integers(0, COUNT, n)
    hot = rng.random(n) < 0.05
    b[hot] = 7                                   # barrier 7 has count 64: never completes, long chain
    pos[hot] = rng.integers(0, 40, int(hot.sum()))
    pos[rng.random(n) < 0.01] = COUNT + 3        # 400 (barrier 7: a valid position)
    length = np.where(rng.random(n) < 0.2, rng.integers(0, PITCH + 1, n), PITCH)
    mw = np.where(rng.random(n) < 0.1, rng.integers(1, 500, n), 0)
    unknown = rng.random(n) < 0.01
    return b, pos, length.astype(np.uint32), mw.astype(np.uint32), unknown, payload_bytes(b, pos, np.full(n, version), PITCH)


def _recs(fps, b, pos, length, mw, unknown):
    r = np.zeros(len(b), dtype=capi.JOIN_REC)
    r["fp_lo"], r["fp_hi"] = fps[b, 0], fps[b, 1]
    r["fp_hi"][unknown] ^= np.uint64(0x5555)     # nobody forked these
    r["position"], r["len"], r["max_wait_s"] = pos, length, mw
    return r


def _sweep_canon(e, now):
    conts, lens, mws, pay, more = e.sweep(now)
    assert not more
    out = {}
    for c in conts:
        base, cnt = int(c["pos_base"]), int(c["count"])
        out[(int(c["fp_lo"]), int(c["fp_hi"]))] = (
            cnt, int(c["flags"]), int(c["publish_at"]), lens[base:base + cnt].tobytes(), mws[base:base + cnt].tobytes(),
            b"".join(bytes(pay[base + p, :int(lens[base + p])]) for p in range(cnt) if int(lens[base + p]) != capi.DJ_NIL))
    return conts, out


def test_static_join_matches_join_batch_two_calls_in_a_row():
    ids = [f"s{k:x}~t{k:x}" for k in range(NB)]
    fps = capi.fingerprint_many(ids)
    forks = np.zeros(NB, dtype=capi.FORK_REC)
    forks["fp_lo"], forks["fp_hi"], forks["count"], forks["deadline_ms"] = fps[:, 0], fps[:, 1], COUNT, T0 + 3_600_000
    forks["count"][7] = 64
    ref, sta = _engine(), _engine()              # reference path / static path, same GPU
    for e in (ref, sta):
        assert (e.fork_batch(forks) == capi.DJ_FORK_CREATED).all()
    rng = np.random.default_rng(77)
    now = T0
    # second call smaller than the first, third larger again after a sweep + ack
    for call, n in enumerate((capi.DJ_STATIC_JOIN_MAX, 9001, 30000)):
        now += 5
        b, pos, length, mw, unknown, pay = _events(rng, n, call)
        recs = _recs(fps, b, pos, length, mw, unknown)
        want = ref.join_batch(recs, pay, now)
        d_recs, d_pay, d_st = to_device(recs, sta.device), to_device(pay, sta.device), sta.empty(n * 4)
        sta.sync()
        import torch
        torch.cuda.synchronize()
        sta.join_batch_static_dev(n, d_recs, d_pay, PITCH, now, d_st)
        sta.sync()
        got = from_device(d_st, np.int32, n)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, f"call {call}: statuses differ at {bad[:8]}: static {got[bad[:8]]} plain {want[bad[:8]]}"
        assert {int(s) for s in np.unique(want)} >= {capi.DJ_JOIN_WAITING, capi.DJ_JOIN_COMPLETE, capi.DJ_JOIN_NOT_FOUND,
                                                     capi.DJ_JOIN_BAD_POSITION}
        if call == 1:
            rc, rcanon = _sweep_canon(ref, now)
            sc, scanon = _sweep_canon(sta, now)
            assert len(rcanon) > 300 and rcanon == scanon, "popped continuations differ"
            assert (ref.ack_batch(rc, None, now) == capi.DJ_ACK_DONE).all()
            assert (sta.ack_batch(sc, None, now) == capi.DJ_ACK_DONE).all()
    _, rcanon = _sweep_canon(ref, now + 3_600_001)      # everything that is left, as timeouts
    _, scanon = _sweep_canon(sta, now + 3_600_001)
    assert len(rcanon) > 100 and rcanon == scanon
    # the hot barrier (count 64, never complete) position by position
    rm, rl, rw, rp = ref.snapshot_one(*fps[7])
    sm, sl, sw, sp = sta.snapshot_one(*fps[7])
    assert rl.tobytes() == sl.tobytes() and rw.tobytes() == sw.tobytes()
    # (a row's bytes beyond its message's length are not part of the answer: an empty message leaves its row alone)
    for p in range(64):
        k = 0 if int(rl[p]) == capi.DJ_NIL else int(rl[p])
        assert bytes(rp[p][:k]) == bytes(sp[p][:k]), p
    ref.close()
    sta.close()


def _forked_pair(prefix, static_max_batch=1 << 17):
    """(fingerprints, engine for dj_join_batch, engine for dj_join_batch_static), both with the same NB barriers."""
    ids = [f"{prefix}{k:x}~t{k:x}" for k in range(NB)]
    fps = capi.fingerprint_many(ids)
    forks = np.zeros(NB, dtype=capi.FORK_REC)
    forks["fp_lo"], forks["fp_hi"], forks["count"], forks["deadline_ms"] = fps[:, 0], fps[:, 1], COUNT, T0 + 3_600_000
    forks["count"][7] = 64
    ref, sta = _engine(), _engine(static_max_batch)
    for e in (ref, sta):
        for lo in range(0, NB, 4096):             # (an engine with a small max_batch takes its forks in pieces)
            assert (e.fork_batch(forks[lo:lo + 4096]) == capi.DJ_FORK_CREATED).all()
    return fps, ref, sta


def _static_call_equals_plain(fps, ref, sta, rng, n, version, now):
    import torch
    b, pos, length, mw, unknown, pay = _events(rng, n, version)
    recs = _recs(fps, b, pos, length, mw, unknown)
    want = ref.join_batch(recs, pay, now)
    d_recs, d_pay, d_st = to_device(recs, sta.device), to_device(pay, sta.device), sta.empty(n * 4)
    torch.cuda.synchronize()                      # nothing of another engine's call is still running on this GPU
    sta.join_batch_static_dev(n, d_recs, d_pay, PITCH, now, d_st)
    sta.sync()
    got = from_device(d_st, np.int32, n)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"statuses differ at {bad[:8]}: static {got[bad[:8]]} plain {want[bad[:8]]}"


def test_static_join_on_an_engine_whose_own_batches_are_smaller():
    """The static path brings its own work arrays: its limit is DJ_STATIC_JOIN_MAX, whatever max_batch the engine's own
    arrays were sized for (here 4096 events against calls of 30 000 and 65 536)."""
    fps, ref, sta = _forked_pair("m", static_max_batch=4096)
    rng = np.random.default_rng(5)
    now = T0
    for call, n in enumerate((30000, capi.DJ_STATIC_JOIN_MAX, 777)):
        now += 5
        _static_call_equals_plain(fps, ref, sta, rng, n, call, now)
    _, rcanon = _sweep_canon(ref, now)
    _, scanon = _sweep_canon(sta, now)
    assert len(rcanon) > 300 and rcanon == scanon, "popped continuations differ"
    with pytest.raises(EngineError, match="exceeds max_batch=4096"):
        sta.join_batch(np.zeros(5000, dtype=capi.JOIN_REC), np.zeros((5000, PITCH), np.uint8), now)   # the plain path keeps its own limit
    ref.close()
    sta.close()


def test_two_engines_on_one_gpu_take_turns_on_the_static_array():
    """The array belongs to the GPU, not to an engine: two engines that call one after the other (never at the same time)
    each find the other's leftovers in it, larger and smaller than their own batch."""
    fa, ref_a, sta_a = _forked_pair("a")
    fb, ref_b, sta_b = _forked_pair("b")
    rng = np.random.default_rng(6)
    now = T0
    turns = [(fa, ref_a, sta_a, 40000), (fb, ref_b, sta_b, 9000), (fa, ref_a, sta_a, 3000), (fb, ref_b, sta_b, 65000),
             (fa, ref_a, sta_a, 20000)]
    for call, (fps, ref, sta, n) in enumerate(turns):
        now += 5
        _static_call_equals_plain(fps, ref, sta, rng, n, call, now)
    for ref, sta in ((ref_a, sta_a), (ref_b, sta_b)):
        _, rcanon = _sweep_canon(ref, now + 3_600_001)
        _, scanon = _sweep_canon(sta, now + 3_600_001)
        assert len(rcanon) > 1000 and rcanon == scanon
        ref.close()
        sta.close()


def test_static_join_refuses_more_than_its_limit():
    e = _engine()
    n = capi.DJ_STATIC_JOIN_MAX + 1
    recs, pay, st = e.empty(n * 32), e.empty(n * PITCH), e.empty(n * 4)
    with pytest.raises(EngineError, match="exceeds DJ_STATIC_JOIN_MAX=65536"):
        e.join_batch_static_dev(n, recs, pay, PITCH, T0, st)
    e.close()
