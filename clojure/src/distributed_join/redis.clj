(ns distributed-join.redis
  "Drop-in for the reference's namespace of the same name (reference src/distributed_join/redis.clj): the same
  thirteen public functions, arguments, return shapes and failures, with the barrier state on the MI355X engine
  (include/dj_engine.h, bound with JNA in distributed-join.engine-store) instead of a Redis server.  Put this
  file in place of the reference's redis.clj and core.clj, ring.clj, publish.clj and the scheduler run as they
  are.

  NOT COMPILED OR RUN by this repository's tests: the build image has no JVM.  The tested twin, call for call, is
  distributed_join_amd/redis_namespace.py under distributed_join_amd/core_unchanged.py (core.clj restated
  function by function): tests/test_namespace_seam.py runs the reference's own test scenarios and random
  request streams through that pair on the GPU.

  Two requests do more here than their Redis commands, because the engine applies a request as one record:
    fork  also queues the barrier at (:timeout-secs record), which core.clj:30 asks for in its next call;
    join  also applies the queue update core.clj:52-58 asks for next (now when complete, else the nearer of the
          current date and now + max-wait).
  The calls core.clj then makes (store-in-publish-queue, store-nearest-in-publish-queue) find the state they
  would have produced and leave it as it is.  :count of join's answer is HLEN of the barrier's hash after the event
  (redis.clj:137,157), which the engine reports in the event's status word (dj_set_join_hlen).

  With :engine-ranks above 1 (a node of several GPUs) requests go through the routed batchers of engine-store and the
  single-key commands to the owner's engine (es/engine-of); all-publish-messages and
  get-unprocessed-publish-messages below poll rank 0 only — a node needs them once per rank, with that rank's engine
  and buffers (not written out: nothing here can be run).

  Id strings and the Fork maps (URLs) stay in this JVM, keyed by fingerprint; counts, deadlines, retry counters,
  messages and both queues live in HBM."
  (:require [clojure.tools.logging :as log]
            [config.core :refer [env]]
            [manifold.deferred :as deferred]
            [distributed-join.http-handling :refer [err] :as handling]
            [distributed-join.time-util :refer :all]
            [distributed-join.engine-store :as es])
  (:import (com.sun.jna Memory Pointer)))

(defn- database-failure [cause]                                         ; redis.clj:17-21
  (deferred/error-deferred (ex-info "Database failure(s)" {:causes [cause]})))

(defn- ids-in-redis-order
  "The barrier ids of n dj_cont rows (64 bytes each) ordered as ZRANGEBYSCORE orders them: by score
  (publish_at, offset 24), then by member (redis.clj:286,293).  Rows marked void are skipped; a fingerprint this
  JVM has no id for (queued by nobody it knows) is skipped with a warning."
  [^Memory rows n]
  (->> (range n)
       (remove (fn [i] (neg? (.getInt rows (+ (* 64 i) 36)))))          ; DJ_CONT_VOID
       (keep (fn [i]
               (let [o (* 64 i)
                     fp [(.getLong rows o) (.getLong rows (+ o 8))]
                     id (:barrier-id (.get es/forks fp))]
                 (if id
                   [(.getLong rows (+ o 24)) id]
                   (log/warn "queue entry of an unknown fingerprint" fp)))))
       (sort)
       (mapv second)))

;; redis.clj:90-104
(defn fork [record]
  (es/fork record))

;; redis.clj:106-111
(defn update-metadata [metadata]
  (deferred/future
    (let [barrier-id (es/barrier-id-of metadata)
          [lo hi :as fp] (es/fingerprint barrier-id)
          row (doto (Memory. 64)                                         ; struct dj_cont
                (.clear)
                (.setLong 0 lo) (.setLong 8 hi)
                (.setLong 16 (long (:timeout-secs metadata)))
                (.setInt 32 (int (:count metadata)))
                (.setInt 40 (int (or (:publish-retry-cnt metadata) -1)))          ; DJ_NIL = key absent
                (.setInt 44 (int (or (:timeout-publish-retry-cnt metadata) -1))))
          status (Memory. 4)]
      (es/check (es/native-int es/dj "dj_update_metadata_one" (es/engine-of fp) row status) "dj_update_metadata_one")
      (when-not (= 7 (.getInt status 0))                                 ; DJ_ACK_METADATA_SET
        (throw (ex-info "Database failure(s)" {:causes [(str "metadata refused: " (.getInt status 0))]})))
      (.put es/forks fp (-> metadata
                            (dissoc :publish-retry-cnt :timeout-publish-retry-cnt)
                            (assoc :barrier-id barrier-id)))
      (str barrier-id ":metadata"))))

(declare join-metadata)

;; redis/join answers HLEN (redis.clj:137,157): the engine puts it into every join's status word from here on
(defonce ^:private hlen-on
  (delay (doseq [{:keys [engine]} @es/node]
           (es/check (es/native-int es/dj "dj_set_join_hlen" engine (int 1)) "dj_set_join_hlen"))))

;; redis.clj:143-159
(defn join [record]
  @hlen-on
  (deferred/let-flow [word (es/join-word record)
                      failure (es/join-failure (bit-and (int word) 0xFF))   ; an error deferred stops the flow here
                      metadata (join-metadata (:barrier-id record))]
    {:metadata metadata
     :count    (unsigned-bit-shift-right (bit-and (long word) 0xFFFFFFFF) 8)}))   ; DJ_JOIN_HLEN: HLEN after this HSET

;; redis.clj:191-205
(defn complete [barrier-id]
  (log/info "Completing for barrier-id" barrier-id)
  (deferred/future
    (let [[lo hi :as fp] (es/fingerprint barrier-id)
          status (Memory. 4)]
      (es/check (es/native-int es/dj "dj_complete_one" (es/engine-of fp) (long lo) (long hi) status) "dj_complete_one")
      (if (= 6 (.getInt status 0))                                       ; DJ_ACK_DIRTY
        (throw (ex-info "Join was updated before fork could be completed" {:updated-join-before-complete true}))
        (do (.remove es/forks fp) 1)))))

(defn- all-joins-cond-update-timestamp [barrier-id count update-timestamp?]
  (deferred/chain (es/snapshot barrier-id update-timestamp?)
                  (fn [snap]
                    (when (some? count)                                  ; as-array, redis.clj:76-80
                      (let [joins (vec (:joins snap))]
                        (map #(get joins %) (range count)))))))

;; redis.clj:242-243
(defn all-joins [barrier-id count]
  (all-joins-cond-update-timestamp barrier-id count true))

;; redis.clj:245-246
(defn all-joins-status [barrier-id count]
  (all-joins-cond-update-timestamp barrier-id count false))

;; redis.clj:248-253
(defn join-metadata [barrier-id]
  (deferred/chain (es/snapshot barrier-id false)
                  (fn [snap]
                    (log/debug "In join-metadata with barrier-id:" barrier-id " and metadata:" (:metadata snap))
                    (:metadata snap))))

;; redis.clj:255-259
(defn store-in-publish-queue [barrier-id publish-at]
  (deferred/future
    (let [[lo hi :as fp] (es/fingerprint barrier-id)]
      (.putIfAbsent es/forks fp {:barrier-id barrier-id})                ; an id may be queued without metadata
      (es/check (es/native-int es/dj "dj_store_in_publish_queue" (es/engine-of fp) (long lo) (long hi) (long publish-at))
                "dj_store_in_publish_queue")
      (list publish-at))))

;; redis.clj:277-281
(defn store-nearest-in-publish-queue [barrier-id publish-at]
  (deferred/chain
    (deferred/future
      (let [[lo hi :as fp] (es/fingerprint barrier-id)
            result (Memory. 4)]
        (es/check (es/native-int es/dj "dj_store_nearest_in_publish_queue" (es/engine-of fp) (long lo) (long hi)
                                 (long publish-at) result)
                  "dj_store_nearest_in_publish_queue")
        (.getInt result 0)))
    (fn [resp]
      (if (neg? resp)
        ;; ZSCORE of an absent member is Lua false, and `false > string` raises (redis.clj:268-269)
        (database-failure "attempt to compare boolean with string")
        (cons publish-at [resp])))))

;; redis.clj:283-289
(defn all-publish-messages []
  (let [now-milliseconds (now-milliseconds)]
    (log/info "Calling all-publish-messages to see if there is anything that is ready to be published...")
    (deferred/future
      (locking es/publisher-lock
        (let [{:keys [conts result]} @es/buffers
              h-result (Memory. 16)]
          (es/check (es/native-int es/dj "dj_pop_due" @es/engine (long now-milliseconds) (int es/sweep-cap)
                                   conts result h-result)
                    "dj_pop_due")
          (let [n (.getInt h-result 0)]
            (ids-in-redis-order (es/to-host conts (* 64 n)) n)))))))

;; redis.clj:291-293
(defn get-unprocessed-publish-messages []
  (let [now-milliseconds (now-milliseconds)]
    (deferred/future
      (locking es/publisher-lock
        (let [{:keys [conts result]} @es/buffers
              h-result (Memory. 16)]
          (es/check (es/native-int es/dj "dj_unprocessed" @es/engine (long now-milliseconds) (int es/sweep-cap)
                                   conts result h-result)
                    "dj_unprocessed")
          (let [n (.getInt h-result 0)]
            (ids-in-redis-order (es/to-host conts (* 64 n)) n)))))))

;; redis.clj:295-297
(defn get-publish-date [barrier-id]
  (es/get-publish-date barrier-id))

;; redis.clj:299-300
(defn shutdown []
  (es/shutdown))
