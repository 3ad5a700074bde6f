"""Pins the CPU oracle to the reference's own known-answer tests (tests/golden/scenarios.json)."""
import pytest

from tests.scenario_runner import load_scenarios, run_scenario

T0, SCENARIOS = load_scenarios()


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_oracle_matches_reference_assertions(sc, oracle_lib):
    def make_store(env):
        o = oracle_lib.Oracle([], max_count=999, payload_stride=256,
                              check_publish_delay_ms=env["check-publish-delay-ms"],
                              max_retry_publish=env["max-retry-post-publish"],
                              max_retry_timeout_publish=env["max-retry-post-timeout-publish"])
        return oracle_lib.OracleStore(o)
    run_scenario(sc, T0, make_store)
