"""core.clj as it stands, over a store namespace with the reference's thirteen functions.

Every function of reference src/distributed_join/core.clj is restated here one for one, in its own order of
store calls, so that running the reference's test scenarios through this class and redis_namespace.EngineRedis
shows what the Clojure service does when ONLY its namespace distributed-join.redis is replaced (core.clj, ring.clj,
publish.clj and the scheduler untouched).  core.ForkJoinService is the other host: one engine batch per request.

Deferreds are plain values here, error deferreds exceptions (redis_namespace.StoreError carries the ex-data);
`respond` (http_handling.clj:21-36) turns them into HTTP answers.  The clock and the POST are injected.
"""
from . import edn
from .redis_namespace import StoreError, err
from .store import DEFAULT_ENV, validate_fork, validate_join

OK, CREATED, ACCEPTED, BAD_REQUEST, NOT_FOUND, INTERNAL_ERROR = 200, 201, 202, 400, 404, 500


def respond(thunk):
    """http-handling/respond: success -> its status, :reason not-found -> 404, bad-data -> 400, else 500."""
    try:
        return thunk()
    except StoreError as ex:
        reason = ex.data.get("reason")
        if reason == "not-found":
            return NOT_FOUND, ex.data["message"]
        if reason == "bad-data":
            return BAD_REQUEST, ex.data["message"]
        return INTERNAL_ERROR, str(ex)


class Core:
    def __init__(self, redis, clock, post, env=None):
        self.redis = redis
        self.clock = clock                                   # now-milliseconds, time_util.clj:3-4
        self.post = post                                     # publish/post, publish.clj:6-15
        self.env = dict(DEFAULT_ENV)
        if env:
            self.env.update(env)
        self.trace = []                                      # (barrier-id, what the publisher did with it), for tests

    def _now_with_offset(self, secs):                        # time_util.clj:6-10
        return None if secs is None else self.clock() + 1000 * secs

    # core.clj:24-33
    def fork(self, request):
        bad = validate_fork(request, self.env)               # the schema coercion of ring.clj:54-60
        if bad:
            return BAD_REQUEST, bad

        def go():
            timeout_secs = self._now_with_offset(request.get("max-wait") or self.env["fork-max-wait"])   # core.clj:18-19,28
            barrier_id = self.redis.fork(dict(request, **{"timeout-secs": timeout_secs}))                # core.clj:29
            self.redis.store_in_publish_queue(barrier_id, timeout_secs)                                  # core.clj:30
            return CREATED, barrier_id
        return respond(go)

    # core.clj:36-38
    @staticmethod
    def all_there(barrier_id, metadata, join_count):
        c = (metadata or {}).get("count")
        return None if c is None else join_count == c

    # core.clj:40-42
    def _join_with_max_wait(self, request):
        date = self._now_with_offset(request.get("max-wait"))
        if date is not None:
            return self.redis.store_nearest_in_publish_queue(request["barrier-id"], date)
        return None

    # core.clj:44-58
    def join(self, request):
        bad = validate_join(request, self.env)               # ring.clj:62-68
        if bad:
            return BAD_REQUEST, bad

        def go():
            bid = request["barrier-id"]
            join_resp = self.redis.join(request)                                                         # core.clj:49
            if self.all_there(bid, join_resp["metadata"], join_resp["count"]):
                self.redis.store_in_publish_queue(bid, self.clock())                                     # core.clj:54
                return ACCEPTED, f"join received for {bid}, request will be completed."
            self._join_with_max_wait(request)                                                            # core.clj:57
            return ACCEPTED, f"join received for {bid}, waiting on more chunks."
        return respond(go)

    # core.clj:60-68 (+ the guards of ring.clj:76-80)
    def status(self, barrier_id):
        if not isinstance(barrier_id, str):
            return BAD_REQUEST, "barrier-id missing"
        if len(barrier_id) > self.env["barrier-id-len-max"]:
            return BAD_REQUEST, f"barrier-id exceeds {self.env['barrier-id-len-max']} in length"

        def go():
            metadata = self.redis.join_metadata(barrier_id)
            if metadata is None:
                raise err("not-found", "barrier-id does not exist")
            joins = self.redis.all_joins_status(barrier_id, metadata["count"])
            fork_keys = ("join-id", "transaction-id", "count", "destination", "timeout-destination", "max-wait")
            body = {k: metadata[k] for k in fork_keys if k in metadata}                                  # select-schema Fork
            body["joins"] = joins
            return OK, body
        return respond(go)

    # core.clj:86-89
    def _is_timeout(self, metadata):
        return self.clock() > metadata["timeout-secs"]

    # core.clj:91-95
    def _select_destination_url(self, metadata):
        if not metadata:
            return None
        return metadata.get("timeout-destination") if self._is_timeout(metadata) else metadata.get("destination")

    # core.clj:107-111
    def _update_to_retry_publish(self, barrier_id, metadata, timeout, publish_retry_cnt):
        key = "timeout-publish-retry-cnt" if timeout else "publish-retry-cnt"                            # core.clj:97-100
        self.redis.update_metadata(dict(metadata, **{key: publish_retry_cnt}))
        return self.redis.store_in_publish_queue(barrier_id, self.clock())

    # core.clj:113-117
    def _reset_publish(self, barrier_id):
        metadata = self.redis.join_metadata(barrier_id) or {}
        dropped = {k: v for k, v in metadata.items() if k not in ("timeout-publish-retry-cnt", "publish-retry-cnt")}
        return self._update_to_retry_publish(barrier_id, dropped, False, 0)

    # core.clj:119-130
    def _complete(self, barrier_id, published, publish_resp, how):
        try:
            self.redis.complete(barrier_id)
            self.trace.append((barrier_id, how))
            return publish_resp
        except StoreError as ex:
            if not ex.data.get("updated-join-before-complete"):
                raise
            self._reset_publish(barrier_id)                                                              # core.clj:127-129
            self.trace.append((barrier_id, "REQUEUED_DIRTY"))
            return publish_resp

    # core.clj:132-155
    def _attempt_retry_publish(self, barrier_id, metadata, error):
        if metadata is None:                                                                             # core.clj:134-136
            return self._complete(barrier_id, False, LookupError(
                f"Metadata for barrier-id {barrier_id} is missing, so retry of publish cannot happen. "
                "The fork will be completed as failed."), "DONE_NO_METADATA")
        timeout = self._is_timeout(metadata)
        cnt = metadata.get("timeout-publish-retry-cnt" if timeout else "publish-retry-cnt") or 0         # core.clj:138-139
        cap = self.env["max-retry-post-timeout-publish" if timeout else "max-retry-post-publish"]        # core.clj:102-105,141
        if cnt >= cap:
            return self._complete(barrier_id, False, error, "DROPPED_EXHAUSTED")                         # core.clj:142-147
        self._update_to_retry_publish(barrier_id, metadata, timeout, cnt + 1)                            # core.clj:151
        self.trace.append((barrier_id, "RETRY"))
        return error

    def _attempt_retry_publish_2(self, barrier_id, error):                                               # core.clj:153-155
        return self._attempt_retry_publish(barrier_id, self.redis.join_metadata(barrier_id), error)

    # core.clj:157-170
    def _publish_message_to_destination(self, barrier_id, metadata, destination):
        if metadata is None:                                                                             # core.clj:159-161
            return self._complete(barrier_id, False, LookupError(
                f"Metadata for barrier-id {barrier_id} is missing, so publish cannot happen. "
                "The fork will be completed as failed."), "DONE_NO_METADATA")
        try:
            joins = self.redis.all_joins(barrier_id, metadata["count"])                                  # core.clj:163
            body = edn.encode(barrier_id, [None if j is None else j["message"] for j in joins])          # core.clj:165-166
            try:
                publish_resp = self.post(destination, body, edn.CONTENT_TYPE)
            except Exception as ex:                                                                      # core.clj:164,167
                return self._attempt_retry_publish(barrier_id, metadata, ex)
            return self._complete(barrier_id, True, publish_resp, "DONE")                                # core.clj:168
        except StoreError as ex:                                                                         # core.clj:170
            return self._attempt_retry_publish(barrier_id, metadata, ex)

    # core.clj:173-178
    def _publish_message(self, barrier_id):
        meta = self.redis.join_metadata(barrier_id)
        return self._publish_message_to_destination(barrier_id, meta, self._select_destination_url(meta))

    # core.clj:180-191.  deferred/timeout! fails the scheduler's wait after :publish-all-complete-timeout-ms and
    # cancels nothing: the publishes under way run to their end, then the poll answers with the timeout.
    def process_publish_messages(self):
        self.trace = []
        started = self.clock()
        ids = self.redis.all_publish_messages()
        if not ids:
            return None
        out = [self._publish_message(i) for i in ids]
        if self.clock() - started > self.env["publish-all-complete-timeout-ms"]:
            raise TimeoutError("Timeout while processing publish messages, cancelling all publish messages commands")
        return out

    # core.clj:193-201
    def recover_unprocessed_publish_messages(self):
        self.trace = []
        ids = self.redis.get_unprocessed_publish_messages()
        for i in ids:
            self._attempt_retry_publish_2(i, TimeoutError(
                f"Timeout waiting to publish barrier id: {i} - adding back into publish queue."))
        return list(self.trace)
