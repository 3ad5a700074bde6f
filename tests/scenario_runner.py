"""Runs the golden scenarios (tests/golden/scenarios.json) against a ForkJoinService.

The same runner drives the service over the CPU oracle (checker pinned to the reference's
own assertions) and over the HIP engine (the product), so both are held to the identical
known answers.
"""
import json
import os

from distributed_join_amd import capi, edn
from distributed_join_amd.core import ForkJoinService, PublishError

HERE = os.path.dirname(os.path.abspath(__file__))
ACK_NAMES = {capi.DJ_ACK_DONE: "DONE", capi.DJ_ACK_REQUEUED_DIRTY: "REQUEUED_DIRTY", capi.DJ_ACK_RETRY: "RETRY",
             capi.DJ_ACK_DROPPED_EXHAUSTED: "DROPPED_EXHAUSTED", capi.DJ_ACK_DONE_NO_METADATA: "DONE_NO_METADATA"}


def load_scenarios():
    with open(os.path.join(HERE, "golden", "scenarios.json")) as f:
        doc = json.load(f)
    return doc["t0"], doc["scenarios"]


class FakeDestination:
    """Stands in for the embedded Jetty of the reference tests (embedded_jetty.clj:30-59)."""

    def __init__(self):
        self.requests = []
        self.bodies = []
        self.hook = None

    def post(self, url, body, content_type):
        # what the destination receives: EDN bytes (publish.clj:9-11); it reads them like any EDN consumer would
        assert isinstance(body, bytes) and content_type == "application/edn"
        barrier_id, messages = edn.decode(body)
        assert edn.encode(barrier_id, messages) == body
        self.bodies.append(body)
        failed = url is None or "/bad/" in url
        self.requests.append({"url": url, "barrier-id": barrier_id, "messages": messages, "failed": failed})
        if self.hook:
            self.hook()
        if failed:
            raise PublishError("status: 404")
        return {"status": 201}


def run_scenario(sc, t0, make_store, make_service=ForkJoinService):
    """make_store(env) or make_store(env, clock) -> store object (EngineStore or OracleStore, or a namespace view
    from tests/namespace_view.py); make_service(store, clock, post, env=env) -> the request logic over it."""
    env = sc["env"]
    clock = [t0]
    dest = FakeDestination()
    try:
        store = make_store(env, lambda: clock[0])
    except TypeError:
        store = make_store(env)
    svc = make_service(store, lambda: clock[0], dest.post, env=env)

    def do(step):
        clock[0] = t0 + step["t"]
        exp = step.get("expect", {})
        kind = step["do"]
        ctx = f"{sc['name']} step {step}"
        if kind == "fork":
            st, body = svc.fork(step["req"])
            assert st == exp["status"], ctx
            if "body" in exp:
                assert body == exp["body"], ctx
        elif kind == "join":
            st, body = svc.join(step["req"])
            assert st == exp["status"], f"{ctx}: got {st} {body}"
            if "body" in exp:
                assert body == exp["body"], ctx
            if "body_contains" in exp:
                assert exp["body_contains"] in body, f"{ctx}: got {body}"
        elif kind == "status":
            st, body = svc.status(step["id"])
            assert st == exp["status"], f"{ctx}: got {st} {body}"
            if "count" in exp:
                assert body["count"] == exp["count"], ctx
            if "joins" in exp:
                got = [None if j is None else j["message"] for j in body["joins"]]
                assert got == exp["joins"], f"{ctx}: got {got}"
                for p, j in enumerate(body["joins"]):
                    if j is not None:
                        assert j["position"] == p and j["barrier-id"] == step["id"], ctx
        elif kind == "process_publish":
            dest.requests.clear()
            dest.bodies.clear()
            nested = step.get("during_post")

            def hook():
                for s in nested or []:
                    do(s)
                if "ack_t" in step:
                    clock[0] = t0 + step["ack_t"]
            dest.hook = hook if (nested or "ack_t" in step) else None
            res = svc.process_publish_messages()
            dest.hook = None
            if exp.get("result") == "nil":
                assert res is None, f"{ctx}: got {res}"
                return
            assert res, f"{ctx}: expected work, got {res}"
            if "posts" in exp:
                key = lambda r: r["barrier-id"]
                want = sorted(({"failed": False, **p} for p in exp["posts"]), key=key)
                assert sorted(dest.requests, key=key) == want, f"{ctx}: got {dest.requests}"
                # and the bytes on the wire, for the scenarios' plain messages: "x" prints as itself
                assert sorted(dest.bodies) == sorted(
                    ('{:barrier-id "%s", :messages (%s)}' % (p["barrier-id"], " ".join(
                        "nil" if m is None else '"%s"' % m for m in p["messages"]))).encode() for p in exp["posts"]), ctx
            if "errors" in exp:
                assert sum(isinstance(r, Exception) for r in res) == exp["errors"], f"{ctx}: got {res}"
            if "illegal_state" in exp:
                assert sum(isinstance(r, LookupError) for r in res) == exp["illegal_state"], f"{ctx}: got {res}"
            if "ack" in exp:
                assert [ACK_NAMES[s] for s in svc.last_ack_statuses] == exp["ack"], f"{ctx}: got {svc.last_ack_statuses}"
        elif kind == "pop":
            got = sorted(c.barrier_id for c in store.pop_due(clock[0]))
            assert got == sorted(exp["ids"]), f"{ctx}: got {got}"
        elif kind == "recover":
            out = svc.recover_unprocessed_publish_messages()
            assert sorted(c.barrier_id for c, _ in out) == sorted(exp["ids"]), f"{ctx}: got {out}"
            if "status" in exp:
                assert [ACK_NAMES[s] for _, s in out] == exp["status"], f"{ctx}: got {out}"
        elif kind == "publish_date":
            pa, _ = store.queue_scores(step["id"])
            assert pa is not None, ctx
            if exp.get("gt_now"):
                assert pa > clock[0], ctx
            if "le_now_plus_s" in exp:
                assert pa <= clock[0] + 1000 * exp["le_now_plus_s"], ctx
            if "equals" in exp:
                assert pa == exp["equals"], f"{ctx}: got {pa}"
        elif kind == "metadata":
            snap = store.snapshot(step["id"])
            assert not snap.no_metadata, ctx
            assert snap.retry_timeout == exp["timeout-publish-retry-cnt"], f"{ctx}: got {snap}"
            assert snap.retry_publish == exp["publish-retry-cnt"], f"{ctx}: got {snap}"
        elif kind == "store_in_publish_queue":
            store.store_in_publish_queue(step["id"], clock[0])
        elif kind == "queue_scores":
            pa, ba = store.queue_scores(step["id"])
            assert pa == exp["primary"] and ba == exp["backup"], f"{ctx}: got {(pa, ba)}"
        else:
            raise ValueError(f"unknown step {kind}")

    for step in sc["steps"]:
        do(step)
