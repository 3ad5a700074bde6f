"""The per-request front (include/dj_batcher.h) on the GPU: many threads, one request each at a
time, answers identical to what the oracle gives for the same requests in the order of each
barrier's own thread."""
import threading

import numpy as np
import pytest

from distributed_join_amd import capi

pytestmark = pytest.mark.gpu
T0 = 1_700_000_000_000


def _pair(**kw):
    from distributed_join_amd.batcher import Batcher
    from distributed_join_amd.engine import Engine
    e = Engine(capacity_barriers=4096, arena_bytes=64 << 20, max_count=16, payload_stride=64, max_batch=1 << 14)
    return e, Batcher(e, **kw)


def test_threads_fork_and_join_through_the_batcher():
    """16 threads own disjoint barriers and send fork, then the joins of every position, one blocking
    request at a time.  Per barrier the requests are sequential, so the answers are fixed (core.clj:52-58);
    the sweep (run on the drain thread) returns every message where it belongs."""
    e, b = _pair(max_batch=256, max_wait_us=200)
    clock = [T0 + 1]
    b.set_clock(lambda: clock[0])
    nthreads, per, c = 16, 24, 5
    ids = [f"req{t}~b{k}" for t in range(nthreads) for k in range(per)]
    fps = capi.fingerprint_many(ids)
    got = {}
    errors = []

    def msg(g, p):
        return bytes([(g * 7 + p * 13 + i) % 251 for i in range(10 + p)])

    def worker(t):
        try:
            rng = np.random.default_rng(t)
            for k in range(per):
                g = t * per + k
                fp = (int(fps[g, 0]), int(fps[g, 1]))
                st = [b.fork_wait(fp, c, T0 + 60_000)]
                for p in rng.permutation(c):
                    st.append(b.join_wait(fp, int(p), msg(g, int(p))))
                got[g] = st
        except BaseException as ex:      # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors
    for g in range(len(ids)):
        assert got[g] == [capi.DJ_FORK_CREATED] + [capi.DJ_JOIN_WAITING] * (c - 1) + [capi.DJ_JOIN_COMPLETE], (g, got[g])

    def sweep(now):
        conts, lens, _, pay, more = e.sweep(now)
        assert not more
        return conts, lens, pay

    conts, lens, pay = b.run(sweep)
    assert len(conts) == len(ids) and (conts["flags"] == 0).all()
    by_fp = {(int(a), int(bb)): g for g, (a, bb) in enumerate(fps)}
    for cont in conts:
        g = by_fp[(int(cont["fp_lo"]), int(cont["fp_hi"]))]
        base = int(cont["pos_base"])
        for p in range(c):
            m = msg(g, p)
            assert int(lens[base + p]) == len(m)
            assert bytes(pay[base + p][:len(m)]) == m
    s = b.stats()
    assert s["forks"] == len(ids) and s["joins"] == len(ids) * c and s["tasks"] == 1
    assert s["largest_batch"] > 1          # requests of different threads did share batches
    b.close()
    e.close()


def test_async_answers_virtual_clock_and_limits():
    e, b = _pair(max_batch=64, max_wait_us=50, max_message=32)
    clock = [T0 + 5]
    b.set_clock(lambda: clock[0])
    fp = capi.fingerprint("solo~1")
    assert b.fork_wait(fp, 3, T0 + 10_000) == capi.DJ_FORK_CREATED
    done = threading.Event()
    seen = []

    def on_status(st):
        seen.append(st)
        if len(seen) == 3:
            done.set()

    b.join_async(fp, 0, b"abc", on_status, max_wait=2)       # store-nearest: queued at now + 2000
    b.join_async(fp, 7, b"x", on_status)                      # position out of range
    b.join_async(capi.fingerprint("nobody~0"), 0, b"x", on_status)
    assert done.wait(10)
    assert sorted(seen) == sorted([capi.DJ_JOIN_WAITING, capi.DJ_JOIN_BAD_POSITION, capi.DJ_JOIN_NOT_FOUND])
    assert b.run(lambda now: e.queue_scores(*fp)) == (T0 + 5 + 2000, T0 + 5 + 2000)
    assert b.join_wait(fp, 1, b"y" * 33) == capi.DJ_JOIN_TOO_LONG     # longer than the staging rows
    assert b.join_wait(fp, 1, b"") == capi.DJ_JOIN_WAITING            # empty message is a message
    clock[0] = T0 + 9
    assert b.join_wait(fp, 2, b"z" * 32) == capi.DJ_JOIN_COMPLETE
    assert b.run(lambda now: (now, e.queue_scores(*fp))) == (T0 + 9, (T0 + 9, T0 + 9))
    with pytest.raises(ZeroDivisionError):
        b.run(lambda now: 1 // 0)
    b.close()
    e.close()


def test_long_messages_through_the_batcher():
    from distributed_join_amd.batcher import Batcher
    from distributed_join_amd.engine import Engine
    e = Engine(capacity_barriers=256, arena_bytes=16 << 20, max_count=4, payload_stride=64, max_batch=1024,
               max_message_bytes=3000)
    b = Batcher(e, max_batch=64, max_wait_us=50)          # staging rows as wide as the engine's longest message
    clock = [T0 + 1]
    b.set_clock(lambda: clock[0])
    fp = capi.fingerprint("big~1")
    assert b.fork_wait(fp, 2, T0 + 10_000) == capi.DJ_FORK_CREATED
    long_msg = bytes((i * 31 + 7) % 256 for i in range(3000))
    assert b.join_wait(fp, 1, long_msg) == capi.DJ_JOIN_WAITING
    assert b.join_wait(fp, 0, b"x" * 3001) == capi.DJ_JOIN_TOO_LONG
    assert b.join_wait(fp, 0, b"short") == capi.DJ_JOIN_COMPLETE

    def sweep(now):
        conts, lens, _, pay, more = e.sweep(now)
        return conts, lens, pay

    conts, lens, pay = b.run(sweep)
    assert len(conts) == 1 and lens.tolist() == [5, 3000]
    assert bytes(pay[0][:5]) == b"short" and bytes(pay[1][:3000]) == long_msg
    b.close()
    e.close()


def test_requests_of_one_thread_keep_their_order_inside_a_window():
    """One thread stages, without waiting for any answer: join (no such barrier yet), fork, two joins, a re-fork
    of the same id with another count, one join.  All five land in one window (the linger is long); the engine
    applies forks and joins as separate batches, so the window must be cut where a fork follows a join of its
    barrier: the answers and the final state are those of the requests taken one by one."""
    from distributed_join_amd.batcher import Batcher
    from distributed_join_amd.engine import Engine
    e = Engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=256)
    b = Batcher(e, max_batch=64, max_wait_us=200_000)
    b.set_clock(lambda: T0 + 5)
    fp = capi.fingerprint("order~1")
    other = capi.fingerprint("order~2")
    got = []
    try:
        b.join_async(fp, 0, b"early", lambda st: got.append(("j0", st)))
        b.fork_async(fp, 2, T0 + 60_000, lambda st: got.append(("f1", st)))
        b.fork_async(other, 1, T0 + 60_000, lambda st: got.append(("fo", st)))
        b.join_async(fp, 0, b"a", lambda st: got.append(("j1", st)))
        b.join_async(fp, 1, b"b", lambda st: got.append(("j2", st)))
        b.fork_async(fp, 3, T0 + 60_000, lambda st: got.append(("f2", st)))
        b.join_async(fp, 2, b"c", lambda st: got.append(("j3", st)))
        b.join_async(other, 0, b"o", lambda st: got.append(("jo", st)))
        b.run(lambda now: 0)                       # returns after everything staged before it was applied
        ans = dict(got)
        assert ans == {"j0": capi.DJ_JOIN_NOT_FOUND, "f1": capi.DJ_FORK_CREATED, "fo": capi.DJ_FORK_CREATED,
                       "j1": capi.DJ_JOIN_WAITING, "j2": capi.DJ_JOIN_COMPLETE, "f2": capi.DJ_FORK_CREATED,
                       "j3": capi.DJ_JOIN_WAITING, "jo": capi.DJ_JOIN_COMPLETE}, ans
        # per kind, answers come in staging order
        assert [k for k, _ in got if k[0] == "j" and k != "jo"] == ["j0", "j1", "j2", "j3"]
        meta, lens, _, pay = e.snapshot_one(*fp)
        assert int(meta["count"]) == 3 and lens.tolist() == [capi.DJ_NIL, capi.DJ_NIL, 1] and bytes(pay[2, :1]) == b"c"
        assert b.stats()["batches"] >= 5           # three phases: join | fork + joins | fork + join
    finally:
        b.close()
        e.close()
