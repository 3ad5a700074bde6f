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
