"""The HIP engine against the reference's own known-answer tests (tests/golden/scenarios.json),
through the C ABI and the host mirror — the same scenarios that pin the oracle."""
import pytest

from tests.scenario_runner import load_scenarios, run_scenario

T0, SCENARIOS = load_scenarios()
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_engine_matches_reference_assertions(sc):
    from distributed_join_amd.engine import Engine
    from distributed_join_amd.store import EngineStore
    engines = []

    def make_store(env):
        e = Engine(capacity_barriers=256, arena_bytes=64 << 20, max_count=999, payload_stride=256, max_batch=4096,
                   check_publish_delay_ms=env["check-publish-delay-ms"],
                   max_retry_publish=env["max-retry-post-publish"],
                   max_retry_timeout_publish=env["max-retry-post-timeout-publish"],
                   host_cap_conts=1024, host_cap_positions=8192)
        engines.append(e)
        return EngineStore(e)
    try:
        run_scenario(sc, T0, make_store)
    finally:
        for e in engines:
            e.close()


def test_publish_body_bytes_from_a_sweep():
    """The EDN body the destination receives, byte for byte, for a barrier popped by its deadline with one
    position missing, an empty message and a message that needs the printer's escapes and UTF-8
    (publish.clj:9-11, core.clj:165-166; the format follows Clojure's printer, no reference fixture holds a
    body: parity unpinned)."""
    from distributed_join_amd.core import ForkJoinService
    from distributed_join_amd.engine import Engine
    from distributed_join_amd.store import EngineStore
    e = Engine(capacity_barriers=64, arena_bytes=1 << 20, max_count=8, payload_stride=64, max_batch=256)
    clock = [1_700_000_000_000]
    got = []

    def post(url, body, content_type):
        got.append((url, body, content_type))
        return {"status": 201}

    svc = ForkJoinService(EngineStore(e), lambda: clock[0], post)
    try:
        st, bid = svc.fork({"join-id": "wire", "transaction-id": "t1", "count": 4, "destination": "http://d/ok",
                            "timeout-destination": "http://d/late", "max-wait": 5})
        assert (st, bid) == (201, "wire~t1")
        for pos, msg in ((0, 'say "hi"\\now\n\ttab'), (2, ""), (3, "grüße ☃ 雪")):
            assert svc.join({"barrier-id": bid, "position": pos, "message": msg})[0] == 202
        clock[0] += 5001                      # past the deadline: goes to the timeout destination
        assert svc.process_publish_messages() == [{"status": 201}]
        assert got == [("http://d/late",
                        '{:barrier-id "wire~t1", :messages ("say \\"hi\\"\\\\now\\n\\ttab" nil "" "grüße ☃ 雪")}'.encode("utf-8"),
                        "application/edn")]
        assert svc.status(bid)[0] == 404
    finally:
        e.close()
