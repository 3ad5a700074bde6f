"""Host mirror of the reference's request logic (reference src/distributed_join/core.clj).

ForkJoinService keeps the function names, argument meaning and answers of core.clj
(fork, join, status, process-publish-messages, recover-unprocessed-publish-messages) and
of the HTTP mapping in http_handling.clj:21-36, on top of a store object that plays the
role of the namespace distributed-join.redis (store.EngineStore over the HIP engine).
Time is injected (`clock`, epoch milliseconds), as is the continuation POST (`post`), so
the reference's timing-sensitive tests become deterministic scenarios.

What stays on the host, as in the reference: id strings, destination URLs, validation.
"""
from . import capi, edn
from .store import DEFAULT_ENV, validate_fork, validate_join

# HTTP answers (http_handling.clj:27-36)
OK, CREATED, ACCEPTED, BAD_REQUEST, NOT_FOUND, INTERNAL_ERROR = 200, 201, 202, 400, 404, 500


class PublishError(Exception):
    """What publish/post raises for a failed or timed-out POST (publish.clj:6-15)."""


class PollTimeout(TimeoutError):
    """core.clj:183-191: one poll of the publisher took longer than :publish-all-complete-timeout-ms.  The
    reference's deferred/timeout! fails the scheduler's wait and nothing else: the publishes that were under way
    run to their end, so by the time this is raised every popped barrier has been acked or re-queued."""


class ForkJoinService:
    def __init__(self, store, clock, post, env=None):
        self.store = store
        self.clock = clock            # () -> epoch ms   (time_util.clj:3-4)
        self.post = post              # (url, body bytes, content type) -> response, raises on failure (publish.clj:6-15)
        self.env = dict(DEFAULT_ENV)
        if env:
            self.env.update(env)
        self.forks = {}               # barrier-id -> Fork request (the host half of "<id>:metadata")

    # core/fork, core.clj:24-33
    def fork(self, request):
        err = validate_fork(request, self.env)
        if err:
            return BAD_REQUEST, err
        now = self.clock()
        max_wait = request.get("max-wait") or self.env["fork-max-wait"]          # core.clj:18-19
        timeout_secs = now + 1000 * max_wait                                      # time_util.clj:6-10
        barrier_id = f"{request['join-id']}~{request['transaction-id']}"          # redis.clj:42-45
        st = self.store.fork(barrier_id, request["count"], timeout_secs)
        if st != capi.DJ_FORK_CREATED:
            return INTERNAL_ERROR, f"engine limit reached ({st})"
        self.forks[barrier_id] = dict(request)
        return CREATED, barrier_id

    # core/join, core.clj:44-58
    def join(self, request):
        err = validate_join(request, self.env)
        if err:
            return BAD_REQUEST, err
        bid = request["barrier-id"]
        st = self.store.join(bid, request["position"], request["message"].encode(), request.get("max-wait"),
                             self.clock())
        if st == capi.DJ_JOIN_COMPLETE:
            return ACCEPTED, f"join received for {bid}, request will be completed."
        if st == capi.DJ_JOIN_WAITING:
            return ACCEPTED, f"join received for {bid}, waiting on more chunks."
        if st == capi.DJ_JOIN_NOT_FOUND:
            return NOT_FOUND, "barrier-id does not exist"
        if st == capi.DJ_JOIN_BAD_POSITION:
            return BAD_REQUEST, "position is greater than total"
        if st == capi.DJ_JOIN_TOO_LONG:
            return BAD_REQUEST, f"Message length cannot exceed {self.env['join-message-len-max']}"
        if st == capi.DJ_JOIN_BAD_MAX_WAIT:     # unreachable behind validate_join (store.clj:27-28), kept for direct callers
            return BAD_REQUEST, f"max-wait must be positive and less than or equal to {self.env['join-max-wait']}"
        return INTERNAL_ERROR, "Database failure(s)"                              # redis.clj:17-21

    # core/status, core.clj:60-68 (+ the length guard of ring.clj:78-79)
    def status(self, barrier_id):
        if not isinstance(barrier_id, str):
            return BAD_REQUEST, "barrier-id missing"
        if len(barrier_id) > self.env["barrier-id-len-max"]:
            return BAD_REQUEST, f"barrier-id exceeds {self.env['barrier-id-len-max']} in length"
        snap = self.store.snapshot(barrier_id, update_ts=False)
        if snap.no_metadata:
            return NOT_FOUND, "barrier-id does not exist"
        body = dict(self.forks.get(barrier_id, {"count": snap.count}))            # select-schema Fork, core.clj:66
        joins = []
        for p, m in enumerate(snap.messages):
            if m is None:
                joins.append(None)
            else:
                j = {"barrier-id": barrier_id, "position": p, "message": m.decode()}
                if snap.max_waits[p]:
                    j["max-wait"] = snap.max_waits[p]
                joins.append(j)
        body["joins"] = joins
        return OK, body

    def _destination(self, cont):
        # select-destination-url, core.clj:91-95
        fork = self.forks.get(cont.barrier_id)
        if fork is None:
            return None
        return fork.get("timeout-destination") if cont.timeout else fork.get("destination")

    # core/process-publish-messages, core.clj:180-191.  Returns None when nothing was due,
    # else one entry per popped barrier: the POST response or the exception it ended with.
    def process_publish_messages(self):
        now = started = self.clock()
        conts = self.store.sweep(now)
        if not conts:
            return None
        results, oks = [], []
        for c in conts:
            if c.no_metadata:
                # core.clj:159-161: nothing to publish, the fork is completed as failed
                results.append(LookupError(f"Metadata for barrier-id {c.barrier_id} is missing, so publish cannot "
                                           "happen. The fork will be completed as failed."))
                oks.append(False)
                continue
            # {:barrier-id id :messages (map :message joins)} (core.clj:165-166) as the EDN bytes aleph sends for
            # :form-params with :content-type :application/edn (publish.clj:9-11)
            body = edn.encode(c.barrier_id, [None if m is None else m.decode() for m in c.messages])
            try:
                results.append(self.post(self._destination(c), body, edn.CONTENT_TYPE))
                oks.append(True)
            except Exception as ex:      # core.clj:167,170: any failure goes to attempt-retry-publish
                results.append(ex)
                oks.append(False)
        statuses = self.store.ack(conts, oks, self.clock())
        for c, st in zip(conts, statuses):
            if st in (capi.DJ_ACK_DONE, capi.DJ_ACK_DROPPED_EXHAUSTED, capi.DJ_ACK_DONE_NO_METADATA):
                self.forks.pop(c.barrier_id, None)
        self.last_ack_statuses = statuses
        if self.clock() - started > self.env["publish-all-complete-timeout-ms"]:      # core.clj:183-191
            raise PollTimeout("Timeout while processing publish messages")
        return results

    # core/recover-unprocessed-publish-messages, core.clj:193-201
    def recover_unprocessed_publish_messages(self):
        out = self.store.recover(self.clock())
        for c, st in out:
            if st in (capi.DJ_ACK_DROPPED_EXHAUSTED, capi.DJ_ACK_DONE_NO_METADATA):
                self.forks.pop(c.barrier_id, None)
        return out
