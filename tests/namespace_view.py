"""Lets tests/scenario_runner.py drive the fine-grained seam: core.clj restated function by function
(distributed_join_amd/core_unchanged.py) over a store namespace with the reference's thirteen functions
(redis_namespace.EngineRedis on the engine, oracle_py.OracleRedis on the checker)."""
from types import SimpleNamespace

from distributed_join_amd import capi
from distributed_join_amd.core_unchanged import Core

CODES = {"DONE": capi.DJ_ACK_DONE, "REQUEUED_DIRTY": capi.DJ_ACK_REQUEUED_DIRTY, "RETRY": capi.DJ_ACK_RETRY,
         "DROPPED_EXHAUSTED": capi.DJ_ACK_DROPPED_EXHAUSTED, "DONE_NO_METADATA": capi.DJ_ACK_DONE_NO_METADATA}


class NamespaceView:
    """The runner's direct store steps (pop, queue scores, metadata) on a namespace object."""

    def __init__(self, redis, both_scores):
        self.redis = redis
        self._both = both_scores            # barrier-id -> (primary score or None, backup score or None)

    def pop_due(self, now_ms):
        return [SimpleNamespace(barrier_id=i) for i in self.redis.all_publish_messages()]

    def queue_scores(self, barrier_id):
        return self._both(barrier_id)

    def snapshot(self, barrier_id):
        md = self.redis.join_metadata(barrier_id)
        return SimpleNamespace(no_metadata=md is None, retry_publish=(md or {}).get("publish-retry-cnt"),
                               retry_timeout=(md or {}).get("timeout-publish-retry-cnt"), count=(md or {}).get("count"))

    def store_in_publish_queue(self, barrier_id, at_ms):
        self.redis.store_in_publish_queue(barrier_id, at_ms)


class CoreService:
    """core_unchanged.Core with the answers the runner reads from core.ForkJoinService."""

    def __init__(self, view, clock, post, env=None):
        self.core = Core(view.redis, clock, post, env=env)
        self.last_ack_statuses = []

    def fork(self, req):
        return self.core.fork(req)

    def join(self, req):
        return self.core.join(req)

    def status(self, barrier_id):
        return self.core.status(barrier_id)

    def process_publish_messages(self):
        res = self.core.process_publish_messages()
        self.last_ack_statuses = [CODES[how] for _, how in self.core.trace]
        return res

    def recover_unprocessed_publish_messages(self):
        return [(SimpleNamespace(barrier_id=i), CODES[how]) for i, how in self.core.recover_unprocessed_publish_messages()]
