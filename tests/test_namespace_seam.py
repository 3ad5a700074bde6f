"""The reference's own granularity: core.clj restated function by function (core_unchanged.Core) over the
thirteen functions of distributed-join.redis.

CPU: Core over the oracle's command-level functions reproduces the reference's known-answer tests, which pins
the restatement of core.clj itself.  GPU: the same scenarios and seeded random request streams with Core over
the engine's single-key entry points (redis_namespace.EngineRedis), against the checker."""
import random

import pytest

from tests.namespace_view import CoreService, NamespaceView
from tests.scenario_runner import FakeDestination, load_scenarios, run_scenario

T0, SCENARIOS = load_scenarios()


def oracle_view(oracle_lib, env, clock):
    o = oracle_lib.Oracle([], max_count=999, payload_stride=256, check_publish_delay_ms=env["check-publish-delay-ms"],
                          max_retry_publish=env["max-retry-post-publish"],
                          max_retry_timeout_publish=env["max-retry-post-timeout-publish"])
    r = oracle_lib.OracleRedis(o, clock, env)
    none = lambda v: None if v == oracle_lib.NOT_QUEUED else v
    return NamespaceView(r, lambda i: tuple(none(v) for v in o.queue_scores(r._idx(i))))


def engine_view(env, clock, engines):
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine
    from distributed_join_amd.redis_namespace import EngineRedis
    e = Engine(capacity_barriers=256, arena_bytes=64 << 20, max_count=999, payload_stride=256, max_batch=4096,
               check_publish_delay_ms=env["check-publish-delay-ms"], max_retry_publish=env["max-retry-post-publish"],
               max_retry_timeout_publish=env["max-retry-post-timeout-publish"], host_cap_conts=1024, host_cap_positions=8192)
    engines.append(e)
    r = EngineRedis(e, clock, env)
    none = lambda v: None if v == capi.DJ_NOT_QUEUED else v
    return NamespaceView(r, lambda i: tuple(none(v) for v in e.queue_scores(*capi.fingerprint(i))))


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_core_clj_over_the_oracle_namespace_matches_reference_assertions(sc, oracle_lib):
    run_scenario(sc, T0, lambda env, clock: oracle_view(oracle_lib, env, clock), CoreService)


@pytest.mark.gpu
@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_core_clj_over_the_engine_namespace_matches_reference_assertions(sc):
    engines = []
    try:
        run_scenario(sc, T0, lambda env, clock: engine_view(env, clock, engines), CoreService)
    finally:
        for e in engines:
            e.close()


def request_stream(seed, n_steps):
    """A seeded stream of requests and scheduler ticks over few barriers: forks and re-forks with other counts,
    joins with and without max-wait (replays, overwrites, bad positions, unknown ids), queue entries without
    metadata, publisher polls whose POSTs partly fail and during which further joins arrive, recover passes, a
    clock that crosses deadlines and the recover delay."""
    rng = random.Random(seed)
    ids = [(f"j{k}", f"t{k % 3}") for k in range(6)]

    def a_join():
        j, x = rng.choice(ids)
        req = {"barrier-id": f"{j}~{x}" if rng.random() < 0.95 else "nobody~knows",
               "position": rng.choice((0, 0, 1, 1, 2, 2, 3, 4, 5, 69, 70)),
               "message": rng.choice(("", "m", "grüße", "x" * 200)) + str(rng.randrange(3))}
        if rng.random() < 0.3:
            req["max-wait"] = rng.choice((1, 2, 30))
        return req

    t = 0
    for _ in range(n_steps):
        t += rng.choice((1, 1, 1, 2, 7, 50, 300, 300, 2500, 11000))
        j, x = rng.choice(ids)
        roll = rng.random()
        if roll < 0.10:
            req = {"join-id": j, "transaction-id": x, "count": rng.choice((1, 2, 3, 3, 5, 70)),
                   "destination": "http://d/" + rng.choice(("ok", "ok", "bad/x")), "max-wait": rng.choice((2, 5, 20, 60))}
            if rng.random() < 0.6:
                req["timeout-destination"] = "http://d/" + rng.choice(("late", "bad/late"))
            yield t, "fork", req
        elif roll < 0.72:
            yield t, "join", a_join()
        elif roll < 0.78:
            yield t, "status", f"{j}~{x}"
        elif roll < 0.80:
            yield t, "orphan", rng.choice(("ghost~1", f"{j}~{x}"))
        elif roll < 0.96:
            yield t, "publish", [a_join() for _ in range(rng.choice((0, 0, 1, 2)))]
        else:
            yield t, "recover", None


def run_stream(view, seed, n_steps):
    """-> everything observable: every answer, every POST (url, bytes), what the publisher did per barrier."""
    clock = [T0]
    view.redis.clock = lambda: clock[0]
    dest = FakeDestination()
    svc = CoreService(view, lambda: clock[0], dest.post, env=view.redis.env)
    log = []
    for t, kind, arg in request_stream(seed, n_steps):
        clock[0] = T0 + t
        if kind == "fork":
            log.append(("fork", svc.fork(arg)))
        elif kind == "join":
            log.append(("join", svc.join(arg)))
        elif kind == "status":
            log.append(("status", svc.status(arg)))
        elif kind == "orphan":
            view.store_in_publish_queue(arg, clock[0])            # a queue entry that may have no metadata (core.clj:159-161)
        elif kind == "publish":
            dest.requests.clear()
            during = list(arg)
            answers = []
            # joins that arrive while the first POST is on its way (core.clj:119-130: complete must notice them)
            dest.hook = lambda: answers.extend(svc.join(during.pop()) for _ in range(len(during)))
            res = svc.process_publish_messages()
            dest.hook = None
            log.append(("publish", None if res is None else [type(r).__name__ for r in res], list(svc.last_ack_statuses),
                        [(r["url"], r["barrier-id"], r["messages"]) for r in dest.requests], answers))
        else:
            log.append(("recover", [(c.barrier_id, s) for c, s in svc.recover_unprocessed_publish_messages()]))
    ids = sorted({f"j{k}~t{k % 3}" for k in range(6)} | {"ghost~1"})
    log.append(("end", [(i, view.queue_scores(i), view.redis.join_metadata(i)) for i in ids]))
    return log


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_request_streams_engine_namespace_against_oracle_namespace(seed, oracle_lib):
    from distributed_join_amd.store import DEFAULT_ENV
    env = dict(DEFAULT_ENV, **{"check-publish-delay-ms": 10000, "join-max-wait": 6000})
    engines = []
    try:
        want = run_stream(oracle_view(oracle_lib, env, None), seed, 600)
        got = run_stream(engine_view(env, None, engines), seed, 600)
        for k, (w, g) in enumerate(zip(want, got)):
            assert w == g, f"seed {seed} step {k}: oracle {w} engine {g}"
        assert len(want) == len(got)
    finally:
        for e in engines:
            e.close()


def command_stream(seed, n_steps):
    """Namespace functions called directly, in orders core.clj never produces: metadata written with any count
    while messages are stored (larger, smaller, for ids that have none), completes at any time, queue dates moved
    by hand, pops and backup listings in between."""
    rng = random.Random(1000 + seed)
    ids = [(f"c{k}", "x") for k in range(5)]
    t = 0
    for _ in range(n_steps):
        t += rng.choice((1, 1, 3, 40, 900, 12000))
        j, x = rng.choice(ids)
        bid = f"{j}~{x}"
        roll = rng.random()
        if roll < 0.12:
            yield t, "fork", {"join-id": j, "transaction-id": x, "count": rng.choice((1, 2, 4, 9, 65, 130)),
                              "destination": "http://d/ok", "max-wait": rng.choice((1, 30, 600))}
        elif roll < 0.50:
            req = {"barrier-id": bid, "position": rng.choice((0, 1, 2, 3, 5, 8, 64, 100, 129)), "message": "m%d" % rng.randrange(50)}
            if rng.random() < 0.25:
                req["max-wait"] = rng.choice((1, 50))
            yield t, "join", req
        elif roll < 0.62:
            yield t, "update_metadata", {"join-id": j, "transaction-id": x, "count": rng.choice((1, 2, 3, 4, 9, 16, 65, 130, 200)),
                                         "timeout-secs": T0 + t + rng.choice((-5000, 10, 5000)), "destination": "http://d/ok",
                                         **({"publish-retry-cnt": rng.randrange(4)} if rng.random() < 0.5 else {}),
                                         **({"timeout-publish-retry-cnt": rng.randrange(4)} if rng.random() < 0.3 else {})}
        elif roll < 0.70:
            yield t, "complete", bid
        elif roll < 0.78:
            yield t, "all_joins", bid
        elif roll < 0.83:
            yield t, "store_in_publish_queue", (bid, T0 + t + rng.choice((-1, 0, 700)))
        elif roll < 0.90:
            yield t, "store_nearest_in_publish_queue", (bid, T0 + t + rng.choice((1, 500, 100000)))
        elif roll < 0.96:
            yield t, "all_publish_messages", None
        else:
            yield t, "get_unprocessed_publish_messages", None


def run_commands(view, seed, n_steps):
    from distributed_join_amd.redis_namespace import StoreError
    clock = [T0]
    r = view.redis
    r.clock = lambda: clock[0]
    core = CoreService(view, lambda: clock[0], None, env=r.env).core
    ids = [f"c{k}~x" for k in range(5)]
    log, hlens = [], []
    for t, kind, arg in command_stream(seed, n_steps):
        clock[0] = T0 + t
        try:
            if kind == "fork":
                # redis/fork itself, then what core.clj:30 calls next (the engine's fork has applied it already)
                ts = clock[0] + 1000 * arg["max-wait"]
                out = r.fork(dict(arg, **{"timeout-secs": ts}))
                r.store_in_publish_queue(out, ts)
            elif kind == "join":
                # redis/join itself: {:metadata m :count HLEN} (redis.clj:137,157), HLEN past the metadata's count included;
                # then the calls core.clj:52-58 makes next (the engine's join has applied them already)
                resp = r.join(arg)
                out = ["join", resp["count"], resp["metadata"]]
                hlens.append((resp["count"], (resp["metadata"] or {}).get("count")))
                if core.all_there(arg["barrier-id"], resp["metadata"], resp["count"]):
                    r.store_in_publish_queue(arg["barrier-id"], clock[0])
                    out.append("complete")
                elif arg.get("max-wait"):
                    try:
                        out.append(r.store_nearest_in_publish_queue(arg["barrier-id"], clock[0] + 1000 * arg["max-wait"]))
                    except StoreError as ex:
                        out.append(("error", str(ex), sorted(ex.data)))
            elif kind == "all_joins":
                md = r.join_metadata(arg)
                out = r.all_joins(arg, md["count"] if md else None)
            elif isinstance(arg, tuple):
                out = getattr(r, kind)(*arg)
            elif arg is None:
                out = getattr(r, kind)()
            else:
                out = getattr(r, kind)(arg)
        except StoreError as ex:
            out = ("error", str(ex), sorted(ex.data))
        # everything a caller could see afterwards
        state = []
        for i in ids:
            md = r.join_metadata(i)
            state.append((i, md, view.queue_scores(i), None if md is None else r.all_joins_status(i, md["count"])))
        log.append((kind, arg, out, state))
    log.append(("hlen", hlens))
    return log


seen_past_count = []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_namespace_functions_called_directly_against_the_oracle(seed, oracle_lib):
    from distributed_join_amd.store import DEFAULT_ENV
    env = dict(DEFAULT_ENV)
    engines = []
    try:
        want = run_commands(oracle_view(oracle_lib, env, None), seed, 300)
        got = run_commands(engine_view(env, None, engines), seed, 300)
        for k, (w, g) in enumerate(zip(want, got)):
            assert w == g, f"seed {seed} step {k}:\noracle {w}\nengine {g}"
        seen_past_count.append(any(h > c for h, c in want[-1][1] if c is not None))
        if seed == 5:
            assert any(seen_past_count), "no stream had a join answer HLEN above the metadata's count"
    finally:
        for e in engines:
            e.close()


def _poll_that_outlasts_its_cap(make_service, store_or_view, clock):
    """core.clj:183-191: a poll whose POSTs take longer than :publish-all-complete-timeout-ms ends in a timeout for
    the scheduler, after every popped barrier was dealt with (completed here: the next poll finds nothing)."""
    dest = FakeDestination()
    svc = make_service(store_or_view, lambda: clock[0], dest.post)
    assert svc.fork({"join-id": "slow", "transaction-id": "1", "count": 1, "destination": "http://d/ok"})[0] == 201
    assert svc.join({"barrier-id": "slow~1", "position": 0, "message": "m"})[0] == 202
    dest.hook = lambda: clock.__setitem__(0, clock[0] + 30001)
    with pytest.raises(TimeoutError):
        svc.process_publish_messages()
    assert [r["barrier-id"] for r in dest.requests] == ["slow~1"]
    dest.hook = None
    assert svc.process_publish_messages() is None
    assert svc.status("slow~1")[0] == 404


def test_poll_cap_on_the_oracle_at_both_seams(oracle_lib):
    from distributed_join_amd.core import ForkJoinService
    from distributed_join_amd.store import DEFAULT_ENV
    clock = [T0]
    _poll_that_outlasts_its_cap(CoreService, oracle_view(oracle_lib, dict(DEFAULT_ENV), lambda: clock[0]), clock)
    clock = [T0]
    _poll_that_outlasts_its_cap(ForkJoinService, oracle_lib.OracleStore(oracle_lib.Oracle([], max_count=999, payload_stride=256)), clock)


@pytest.mark.gpu
def test_poll_cap_on_the_engine_at_both_seams():
    from distributed_join_amd.core import ForkJoinService
    from distributed_join_amd.store import DEFAULT_ENV, EngineStore
    engines = []
    try:
        clock = [T0]
        _poll_that_outlasts_its_cap(CoreService, engine_view(dict(DEFAULT_ENV), lambda: clock[0], engines), clock)
        clock = [T0]
        view = engine_view(dict(DEFAULT_ENV), None, engines)
        view.redis.e.set_join_hlen(False)          # the batch seam reads plain status codes
        _poll_that_outlasts_its_cap(ForkJoinService, EngineStore(view.redis.e), clock)
    finally:
        for e in engines:
            e.close()
