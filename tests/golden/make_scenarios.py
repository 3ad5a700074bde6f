#!/usr/bin/env python3
"""Writes tests/golden/scenarios.json: the reference's own known-answer tests, transcribed
with a virtual clock.

Provenance: every scenario restates one test of the reference tree (cited as file:line,
relative to the reference root) — its steps and the assertions that test makes.  The
expectations are DERIVED BY INSPECTION of those tests and of core.clj / redis.clj; they
are not captured output, because the reference (Clojure + an external Redis) cannot run
in the build environment.  Real sleeps of the reference tests become clock offsets
("t", milliseconds after T0); the embedded Jetty destinations become a fake POST target
that records requests and fails for URLs containing "/bad/" (request_util.clj:65-68).

Run:  python tests/golden/make_scenarios.py
"""
import itertools
import json
import os

T0 = 1_700_000_000_000
ROOT = "http://localhost:3001"
DEST = ROOT + "/v1/publish"
TDEST = ROOT + "/v1/publish/timeout"
# config/test/config.edn
ENV = {"fork-max-wait": 500, "join-max-wait": 500, "check-publish-delay-ms": 2000,
       "max-retry-post-publish": 3, "max-retry-post-timeout-publish": 3}
PAUSE_EXTRA = 100                     # request_util.clj:13-16
TIMEOUT_PAUSE = 1000 + PAUSE_EXTRA    # publish_or_timeout_test.clj:22-24 (fork max-wait 1 s)
REPROCESS_PAUSE = 2000 + PAUSE_EXTRA  # publish_or_timeout_test.clj:26-27


def fork_req(n, count, max_wait=1, dest=DEST, tdest=TDEST):
    r = {"join-id": f"a{n}", "transaction-id": f"b{n}", "count": count, "destination": dest,
         "timeout-destination": tdest}
    if max_wait is not None:
        r["max-wait"] = max_wait
    return r


def bid(n):
    return f"a{n}~b{n}"


def join_req(n, pos=0, max_wait=None, msg=None):
    r = {"barrier-id": bid(n), "position": pos,
         "message": msg if msg is not None else f"Join for barrier-id {bid(n)} and position {pos}"}
    if max_wait is not None:
        r["max-wait"] = max_wait
    return r


def jmsg(n, pos=0):
    return f"Join for barrier-id {bid(n)} and position {pos}"


S = []


def scenario(name, source, steps):
    S.append({"name": name, "source": source, "provenance": "derived-by-inspection", "env": ENV, "steps": steps})


# ---- fork-join-test.sh --------------------------------------------------------------------
scenario("smoke_script", "fork-join-test.sh:4-37", [
    {"t": 0, "do": "status", "id": "a~b", "expect": {"status": 404}},
    {"t": 10, "do": "fork", "req": {"count": 3, "destination": "http://localhost:3000/v1/publish", "join-id": "a",
                                     "transaction-id": "b",
                                     "timeout-destination": "http://localhost:3000/v1/publish/timeout", "max-wait": 10},
     "expect": {"status": 201, "body": "a~b"}},
    {"t": 20, "do": "status", "id": "a~b", "expect": {"status": 200, "count": 3, "joins": [None, None, None]}},
    {"t": 30, "do": "join", "req": {"barrier-id": "a~b", "position": 0, "message": "fffffff"},
     "expect": {"status": 202, "body": "join received for a~b, waiting on more chunks."}},
    {"t": 40, "do": "join", "req": {"barrier-id": "a~b", "position": 1, "message": "fffffff"},
     "expect": {"status": 202, "body": "join received for a~b, waiting on more chunks."}},
    {"t": 50, "do": "status", "id": "a~b", "expect": {"status": 200, "joins": ["fffffff", "fffffff", None]}},
    {"t": 60, "do": "join", "req": {"barrier-id": "a~b", "position": 2, "message": "fffffff"},
     "expect": {"status": 202, "body": "join received for a~b, request will be completed."}},
    {"t": 1000, "do": "process_publish",
     "expect": {"posts": [{"url": "http://localhost:3000/v1/publish", "barrier-id": "a~b",
                           "messages": ["fffffff", "fffffff", "fffffff"]}]}},
    {"t": 2070, "do": "status", "id": "a~b", "expect": {"status": 404}},
])

# ---- publish_or_timeout_test.clj ---------------------------------------------------------
scenario("process_publish_no_data", "test/distributed_join/publish_or_timeout_test.clj:70-72", [
    {"t": 0, "do": "process_publish", "expect": {"result": "nil"}},
])

scenario("process_publish_not_all_data_there", "test/distributed_join/publish_or_timeout_test.clj:74-77", [
    {"t": 0, "do": "fork", "req": fork_req(1, 3), "expect": {"status": 201, "body": bid(1)}},
    {"t": 5, "do": "join", "req": join_req(1), "expect": {"status": 202}},
    {"t": 10, "do": "process_publish", "expect": {"result": "nil"}},
])

scenario("process_publish_all_data_there", "test/distributed_join/publish_or_timeout_test.clj:79-83", [
    {"t": 0, "do": "fork", "req": fork_req(2, 1), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": join_req(2), "expect": {"status": 202, "body_contains": "request will be completed"}},
    {"t": 10, "do": "process_publish", "expect": {"posts": [{"url": DEST, "barrier-id": bid(2), "messages": [jmsg(2)]}]}},
])

scenario("process_timeout_data_wait_for_timeout", "test/distributed_join/publish_or_timeout_test.clj:85-90", [
    {"t": 0, "do": "fork", "req": fork_req(3, 3), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": join_req(3), "expect": {"status": 202, "body_contains": "waiting on more chunks"}},
    {"t": 5 + TIMEOUT_PAUSE, "do": "process_publish",
     "expect": {"posts": [{"url": TDEST, "barrier-id": bid(3), "messages": [jmsg(3), None, None]}]}},
])

scenario("process_timeout_data_all_joins_were_already_processed",
         "test/distributed_join/publish_or_timeout_test.clj:92-97", [
    {"t": 0, "do": "fork", "req": fork_req(4, 1), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": join_req(4), "expect": {"status": 202, "body_contains": "request will be completed"}},
    {"t": 5 + TIMEOUT_PAUSE, "do": "process_publish",
     "expect": {"posts": [{"url": TDEST, "barrier-id": bid(4), "messages": [jmsg(4)]}]}},
])

scenario("process_publish_all_data_there_but_does_not_publish",
         "test/distributed_join/publish_or_timeout_test.clj:99-110", [
    {"t": 0, "do": "fork", "req": fork_req(5, 1), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": join_req(5), "expect": {"status": 202}},
    {"t": 10, "do": "pop", "expect": {"ids": [bid(5)]}},
    {"t": 10 + REPROCESS_PAUSE, "do": "pop", "expect": {"ids": []}},
    {"t": 15 + REPROCESS_PAUSE, "do": "recover", "expect": {"ids": [bid(5)]}},
    {"t": 20 + REPROCESS_PAUSE, "do": "pop", "expect": {"ids": [bid(5)]}},
])

scenario("fail_to_process_timeout_in_time", "test/distributed_join/publish_or_timeout_test.clj:112-128", [
    {"t": 0, "do": "fork", "req": fork_req(6, 3), "expect": {"status": 201}},
    {"t": TIMEOUT_PAUSE, "do": "pop", "expect": {"ids": [bid(6)]}},
    {"t": TIMEOUT_PAUSE + 5, "do": "recover", "expect": {"ids": []}},
    {"t": TIMEOUT_PAUSE + 10, "do": "pop", "expect": {"ids": []}},
    {"t": TIMEOUT_PAUSE + 10 + REPROCESS_PAUSE, "do": "recover", "expect": {"ids": [bid(6)]}},
    {"t": TIMEOUT_PAUSE + 15 + REPROCESS_PAUSE, "do": "pop", "expect": {"ids": [bid(6)]}},
    {"t": TIMEOUT_PAUSE + 20 + REPROCESS_PAUSE, "do": "metadata", "id": bid(6),
     "expect": {"timeout-publish-retry-cnt": 1, "publish-retry-cnt": None}},
])

BAD = ROOT + "/bad/destination"
TBAD = ROOT + "/bad/timeout/destination"


def bad_publishes(n, t0, num, url, expect_msgs):
    # test-several-bad-publishes, publish_or_timeout_test.clj:130-137: every response is a 404 error
    return [{"t": t0 + 10 * k, "do": "process_publish",
             "expect": {"errors": 1, "posts": [{"url": url, "barrier-id": bid(n), "messages": expect_msgs, "failed": True}]}}
            for k in range(num)]


scenario("invalid_destinations_will_exhaust_retries", "test/distributed_join/publish_or_timeout_test.clj:149-155",
         [{"t": 0, "do": "fork", "req": fork_req(7, 1, dest=BAD), "expect": {"status": 201}},
          {"t": 5, "do": "join", "req": join_req(7), "expect": {"status": 202}}]
         + bad_publishes(7, 10, 4, BAD, [jmsg(7)])
         + [{"t": 60, "do": "process_publish", "expect": {"result": "nil"}},       # validate-all-published :139-141
            {"t": 70, "do": "status", "id": bid(7), "expect": {"status": 404}}])

scenario("invalid_destinations_will_exhaust_retries_for_timeout_too",
         "test/distributed_join/publish_or_timeout_test.clj:157-164",
         [{"t": 0, "do": "fork", "req": fork_req(8, 3, tdest=TBAD), "expect": {"status": 201}},
          {"t": 5, "do": "join", "req": join_req(8), "expect": {"status": 202}}]
         + bad_publishes(8, 5 + TIMEOUT_PAUSE, 4, TBAD, [jmsg(8), None, None])
         + [{"t": 100 + TIMEOUT_PAUSE, "do": "process_publish", "expect": {"result": "nil"}}])

scenario("invalid_destinations_will_exhaust_retries_independently",
         "test/distributed_join/publish_or_timeout_test.clj:166-178",
         [{"t": 0, "do": "fork", "req": fork_req(9, 1, dest=BAD, tdest=TBAD), "expect": {"status": 201}},
          {"t": 5, "do": "join", "req": join_req(9), "expect": {"status": 202}}]
         + bad_publishes(9, 10, 2, BAD, [jmsg(9)])
         + [{"t": 40, "do": "metadata", "id": bid(9), "expect": {"publish-retry-cnt": 2, "timeout-publish-retry-cnt": None}}]
         + bad_publishes(9, 40 + TIMEOUT_PAUSE, 4, TBAD, [jmsg(9)])
         + [{"t": 100 + TIMEOUT_PAUSE, "do": "process_publish", "expect": {"result": "nil"}}])

# fork-and-two-joins-random-max-wait, publish_or_timeout_test.clj:180-199: all 27 selections of {2,3,4}^3
steps = []
for k, (f, j1, j2) in enumerate(itertools.product([2, 3, 4], repeat=3)):
    n = 100 + k
    t = 50 * k
    steps += [
        {"t": t, "do": "fork", "req": fork_req(n, 3, max_wait=f), "expect": {"status": 201}},
        {"t": t + 5, "do": "join", "req": join_req(n, max_wait=j1), "expect": {"status": 202}},
        {"t": t + 10, "do": "join", "req": join_req(n, max_wait=j2), "expect": {"status": 202}},
        # (is (> publish-date now)) (is (<= publish-date (now-with-offset min-offset)))
        {"t": t + 15, "do": "publish_date", "id": bid(n),
         "expect": {"gt_now": True, "le_now_plus_s": min(f, j1, j2),
                    "equals": T0 + min(t + 1000 * f, t + 5 + 1000 * j1, t + 10 + 1000 * j2)}},
    ]
scenario("fork_and_two_joins_random_max_wait", "test/distributed_join/publish_or_timeout_test.clj:180-199", steps)

scenario("fail_complete_when_join_arrives_during_publish",
         "test/distributed_join/publish_or_timeout_test.clj:201-220", [
    {"t": 0, "do": "fork", "req": fork_req(10, 1, max_wait=2, dest=ROOT + "/v1/delayed/echo?delay=1000"),
     "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": join_req(10), "expect": {"status": 202}},
    # the publish starts, and a second (identical) join sneaks in 500 ms into the 1000 ms POST
    {"t": 10, "do": "process_publish", "during_post": [{"t": 510, "do": "join", "req": join_req(10), "expect": {"status": 202}}],
     "ack_t": 1010,
     "expect": {"posts": [{"url": ROOT + "/v1/delayed/echo?delay=1000", "barrier-id": bid(10), "messages": [jmsg(10)]}],
                "ack": ["REQUEUED_DIRTY"]}},
    # (is (empty? publish-msg-resp1)) is taken while the publish is still running: the second join
    # re-queued the barrier at t=510, but the reference reads the queue before that join lands only
    # by timing; the order-independent assertion is the second one:
    {"t": 1120, "do": "pop", "expect": {"ids": [bid(10)]}},
])

scenario("no_metadata_for_barrier_id_in_publish_queue", "test/distributed_join/publish_or_timeout_test.clj:228-233", [
    {"t": 0, "do": "store_in_publish_queue", "id": "fake-barrier-id"},
    {"t": 5, "do": "process_publish", "expect": {"illegal_state": 1, "posts": []}},
    {"t": 10, "do": "pop", "expect": {"ids": []}},
    {"t": 10 + REPROCESS_PAUSE, "do": "queue_scores", "id": "fake-barrier-id", "expect": {"primary": None, "backup": None}},
])

scenario("no_metadata_for_barrier_id_in_backup_publish_queue",
         "test/distributed_join/publish_or_timeout_test.clj:235-242", [
    {"t": 0, "do": "store_in_publish_queue", "id": "fake-barrier-id"},
    {"t": 5, "do": "pop", "expect": {"ids": ["fake-barrier-id"]}},
    {"t": 5 + REPROCESS_PAUSE, "do": "recover", "expect": {"ids": ["fake-barrier-id"], "status": ["DONE_NO_METADATA"]}},
    {"t": 10 + REPROCESS_PAUSE, "do": "pop", "expect": {"ids": []}},
    {"t": 15 + REPROCESS_PAUSE, "do": "queue_scores", "id": "fake-barrier-id", "expect": {"primary": None, "backup": None}},
])

# ---- ring_test.clj -------------------------------------------------------------------------
CNT = 7   # derived-names, ring_test.clj:16-23


def ring_fork(tid, count=CNT):
    return {"join-id": f"a{tid}", "transaction-id": f"b{tid}", "count": count, "destination": DEST,
            "timeout-destination": TDEST}


def rid(tid):
    return f"a{tid}~b{tid}"


scenario("repeated_fork_calls_with_different_parameters", "test/distributed_join/ring_test.clj:182-198", [
    {"t": 0, "do": "fork", "req": ring_fork("12", CNT + 1), "expect": {"status": 201, "body": rid("12")}},
    {"t": 2, "do": "join", "req": {"barrier-id": rid("12"), "position": CNT, "message": "early"}, "expect": {"status": 202}},
    {"t": 5, "do": "fork", "req": ring_fork("12", CNT), "expect": {"status": 201, "body": rid("12")}},
    {"t": 10, "do": "status", "id": rid("12"), "expect": {"status": 200, "count": CNT, "joins": [None] * CNT}},
])

scenario("valid_join_calls_border_cases", "test/distributed_join/ring_test.clj:202-223", [
    {"t": 0, "do": "fork", "req": ring_fork("21"), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": {"barrier-id": rid("21"), "position": 0, "message": "msg210"}, "expect": {"status": 202}},
    {"t": 10, "do": "join", "req": {"barrier-id": rid("21"), "position": CNT // 2, "message": f"msg21{CNT // 2}"},
     "expect": {"status": 202}},
    {"t": 15, "do": "join", "req": {"barrier-id": rid("21"), "position": CNT - 1, "message": f"msg21{CNT - 1}"},
     "expect": {"status": 202}},
    {"t": 20, "do": "status", "id": rid("21"),
     "expect": {"status": 200, "count": CNT,
                "joins": ["msg210", None, None, f"msg21{CNT // 2}", None, None, f"msg21{CNT - 1}"]}},
])

scenario("call_that_overrides_an_existing_position", "test/distributed_join/ring_test.clj:225-240", [
    {"t": 0, "do": "fork", "req": ring_fork("22"), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": {"barrier-id": rid("22"), "position": CNT - 1, "message": "msg221"}, "expect": {"status": 202}},
    {"t": 10, "do": "join", "req": {"barrier-id": rid("22"), "position": CNT - 1, "message": "msg222"}, "expect": {"status": 202}},
    {"t": 15, "do": "status", "id": rid("22"), "expect": {"status": 200, "joins": [None] * (CNT - 1) + ["msg222"]}},
])

scenario("valid_status_call", "test/distributed_join/ring_test.clj:244-257", [
    {"t": 0, "do": "fork", "req": ring_fork("31"), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": {"barrier-id": rid("31"), "position": CNT // 2, "message": "msg31"}, "expect": {"status": 202}},
    {"t": 10, "do": "status", "id": rid("31"),
     "expect": {"status": 200, "joins": [None, None, None, "msg31", None, None, None]}},
])

order = [4, 0, 6, 2, 5, 1, 3]   # one fixed shuffle of (range cnt), ring_test.clj:265
steps = [{"t": 0, "do": "fork", "req": ring_fork("32"), "expect": {"status": 201}}]
for k, pos in enumerate(order):
    steps.append({"t": 5 + 10 * k, "do": "status", "id": rid("32"), "expect": {"status": 200}})
    steps.append({"t": 10 + 10 * k, "do": "join", "req": {"barrier-id": rid("32"), "position": pos, "message": f"msg32{pos}"},
                  "expect": {"status": 202,
                             "body_contains": "request will be completed" if k == CNT - 1 else "waiting on more chunks"}})
steps.append({"t": 100, "do": "process_publish",
              "expect": {"posts": [{"url": DEST, "barrier-id": rid("32"), "messages": [f"msg32{p}" for p in range(CNT)]}]}})
steps.append({"t": 110, "do": "status", "id": rid("32"), "expect": {"status": 404}})
scenario("successive_status_until_all_joins_received", "test/distributed_join/ring_test.clj:259-274", steps)

# join validation rows that reach the store, ring_test.clj:99-117 (the others are schema failures,
# covered by tests/test_validation.py)
scenario("join_interface_rows", "test/distributed_join/ring_test.clj:99-117,149-165", [
    {"t": 0, "do": "fork", "req": ring_fork("02"), "expect": {"status": 201}},
    {"t": 5, "do": "join", "req": {"barrier-id": "XXX-XXX", "position": CNT // 2, "message": "msg02"}, "expect": {"status": 404}},
    {"t": 10, "do": "join", "req": {"barrier-id": rid("02"), "position": CNT, "message": "msg02"}, "expect": {"status": 400}},
    {"t": 15, "do": "status", "id": rid("02"), "expect": {"status": 200, "joins": [None] * CNT}},
    {"t": 20, "do": "join", "req": {"barrier-id": rid("02"), "position": CNT // 2, "message": ""}, "expect": {"status": 202}},
    {"t": 25, "do": "status", "id": rid("02"), "expect": {"status": 200, "joins": [None, None, None, "", None, None, None]}},
])

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenarios.json")
    with open(out, "w") as f:
        json.dump({"t0": T0, "scenarios": S}, f, indent=1)
    print(f"wrote {len(S)} scenarios to {out}")
