(ns distributed-join.engine-store
  "Barrier state on the MI355X engine instead of Redis: the JVM side of include/dj_engine.h and
  include/dj_batcher.h, bound with JNA.

  NOT COMPILED OR RUN by this repository's tests: the build image has no JVM.  The same mapping is
  exercised in Python (distributed_join_amd/store.py + core.py over the same C ABI).

  It takes the place of the namespace distributed-join.redis (reference
  src/distributed_join/redis.clj).  The engine applies the decisions that core.clj takes between two
  Redis calls (all-there?, is-timeout?, the retry counters) on the device, so the functions here are
  one per request instead of one per Redis command:

    fork     = redis/fork + store-in-publish-queue                       (core.clj:24-33)
    join     = redis/join + all-there? + store-(nearest-)in-publish-queue (core.clj:44-58)
    snapshot = join-metadata + all-joins-status                          (core.clj:60-68)
    sweep    = all-publish-messages + join-metadata + all-joins          (core.clj:173-188)
    ack      = complete / reset-publish / attempt-retry-publish          (core.clj:107-155)
    recover  = get-unprocessed-publish-messages + attempt-retry-publish  (core.clj:193-201)

  As in the reference every function returns a manifold deferred, and user-visible failures are
  error deferreds built by http-handling/err (redis.clj:152-153).  Request threads block inside
  dj_batcher_*_wait on a manifold future thread, the way wcar-async* blocks on a Redis connection
  (redis.clj:23-29); the batcher's drain thread turns what has arrived into one device batch."
  (:require [clojure.tools.logging :as log]
            [config.core :refer [env]]
            [manifold.deferred :as deferred]
            [distributed-join.http-handling :refer [err] :as handling])
  (:import (com.sun.jna NativeLibrary Function Memory Pointer)
           (com.sun.jna.ptr PointerByReference)
           (java.nio.charset StandardCharsets)
           (java.util.concurrent ConcurrentHashMap)))

;; ---- native libraries -------------------------------------------------------------------------

(def ^NativeLibrary dj (NativeLibrary/getInstance "dj_engine"))
(def ^:private ^NativeLibrary hip (NativeLibrary/getInstance "amdhip64"))

(defn native-int
  "Calls an int-returning C function.  Arguments must already have their C width:
  (int x) for 32-bit parameters, (long x) for 64-bit ones, Pointer / byte[] for pointers."
  [^NativeLibrary lib ^String fname & args]
  (.invokeInt ^Function (.getFunction lib fname) (to-array args)))

(defn check [rc what]
  (when (neg? rc)
    ;; redis.clj:17-21: internal failures surface as ex-info with :causes -> HTTP 500
    (throw (ex-info "Database failure(s)" {:causes [(str what " failed with " rc)]})))
  rc)

(defn use-device
  "HIP's current device is per thread; manifold future threads come and go."
  []
  (check (- (native-int hip "hipSetDevice" (int (or (:engine-device env) 0)))) "hipSetDevice"))

(defn device-alloc ^Pointer [^long bytes]
  (use-device)
  (let [ref (PointerByReference.)]
    (check (- (native-int hip "hipMalloc" ref (long bytes))) "hipMalloc")
    (.getValue ref)))

(defn to-host ^Memory [^Pointer device ^long bytes]
  (use-device)
  (let [m (Memory. (max bytes 1))]
    (check (- (native-int hip "hipMemcpy" m device (long bytes) (int 2))) "hipMemcpy")   ; device to host
    m))

(defn to-device [^Pointer device ^Memory host ^long bytes]
  (use-device)
  (check (- (native-int hip "hipMemcpy" device host (long bytes) (int 1))) "hipMemcpy"))  ; host to device

;; ---- engine and batcher -----------------------------------------------------------------------

(def payload-stride (long (or (:engine-payload-stride env) 256)))
(def max-count (long (:fork-count-max env)))                 ; config.edn:33
(def max-message (long (:join-message-len-max env)))         ; config.edn:38, in characters (store.clj:24-26)
;; the engine stores UTF-8 bytes: at most three per UTF-16 code unit of a String
(def ^:private max-message-bytes (long (min (* 3 max-message) (bit-shift-left 1 20))))
(def sweep-cap (long (or (:engine-sweep-cap env) 65536)))    ; continuations per poll

(defn- make-config ^Memory []
  ;; struct dj_config, 56 bytes (include/dj_engine.h)
  (doto (Memory. 56)
    (.setInt 0 (int 56))
    (.setInt 4 (int (or (:engine-device env) 0)))
    (.setLong 8 (long (or (:engine-capacity-barriers env) (bit-shift-left 1 24))))
    (.setLong 16 (long (or (:engine-arena-bytes env) (bit-shift-left 1 36))))
    (.setInt 24 (int max-count))
    (.setInt 28 (int payload-stride))
    (.setInt 32 (int (or (:engine-max-batch env) (bit-shift-left 1 20))))
    (.setInt 36 (int (:max-retry-post-publish env)))                   ; config.edn:24
    (.setInt 40 (int (:max-retry-post-timeout-publish env)))           ; config.edn:25
    (.setInt 44 (int max-message-bytes))
    (.setLong 48 (long (:check-publish-delay-ms env)))))               ; config.edn:16

(def engine-ranks
  "Engines of this node (one per GPU, device ordinals 0 .. n-1).  With more than one, barriers shard by fingerprint
  (dj_owner_of), every rank has a router and a ROUTED batcher (dj_batcher_create_routed), a request enters at the
  batcher of any rank and its answer comes back to the thread that waits for it; single-key commands go to the
  owner's engine directly, the publisher polls every rank.  (Like everything in this file: written against the
  C ABI, never compiled here.)"
  (long (or (:engine-ranks env) 1)))

(defn- create-engine [device]
  (let [ref (PointerByReference.)]
    (check (native-int dj "dj_create" (doto (make-config) (.setInt 4 (int device))) ref) "dj_create")
    (.getValue ref)))

(defn- batcher-config ^Memory []
  (doto (Memory. 16)                                                   ; struct dj_batcher_config
    (.setInt 0 (int 16))
    (.setInt 4 (int (or (:engine-batcher-max-batch env) 65536)))
    (.setInt 8 (int (or (:engine-batcher-linger-us env) 0)))
    (.setInt 12 (int max-message-bytes))))

(defonce node
  ;; a vector of {:device d :engine e :router r :batcher b}, one per rank; router nil with a single rank
  (delay
    (if (= engine-ranks 1)
      (let [e (create-engine (or (:engine-device env) 0))
            ref (PointerByReference.)]
        (check (native-int dj "dj_batcher_create" e (batcher-config) ref) "dj_batcher_create")
        [{:device (or (:engine-device env) 0) :engine e :router nil :batcher (.getValue ref)}])
      (let [engines (mapv create-engine (range engine-ranks))
            id (Memory. 128)
            pitch (long (* 16 (quot (+ (max max-message-bytes 1) 15) 16)))      ; the batcher's staging rows
            batch (int (or (:engine-batcher-max-batch env) 65536))
            _ (check (native-int dj "dj_route_unique_id" id) "dj_route_unique_id")
            ;; dj_router_create is a collective over the ranks: one thread per rank
            routers (->> (range engine-ranks)
                         (mapv (fn [r] (future
                                         (let [ref (PointerByReference.)]
                                           (check (native-int dj "dj_router_create" (nth engines r) (int engine-ranks) (int r) id
                                                              batch (int (or (:engine-max-batch env) (bit-shift-left 1 20)))
                                                              (long pitch) (int 3) ref)
                                                  "dj_router_create")
                                           (.getValue ref)))))
                         (mapv deref))]
        (mapv (fn [r]
                (let [ref (PointerByReference.)]
                  (check (native-int dj "dj_batcher_create_routed" (nth engines r) (nth routers r) (batcher-config) ref)
                         "dj_batcher_create_routed")
                  {:device r :engine (nth engines r) :router (nth routers r) :batcher (.getValue ref)}))
              (range engine-ranks))))))

(defonce engine (delay (:engine (first @node))))      ; the one engine of a single-rank node
(defonce batcher (delay (:batcher (first @node))))

(defn owner-rank [[lo hi]]
  (if (= engine-ranks 1)
    0
    (.invokeInt ^Function (.getFunction dj "dj_owner_of") (to-array [(long lo) (long hi) (int engine-ranks)]))))

(defn engine-of
  "The engine that holds the barrier with this fingerprint (single-key commands go there directly)."
  [fp]
  (:engine (nth @node (owner-rank fp))))

(defn request-batcher
  "The batcher a request thread stages its request in: any rank's will do (the routed batchers bring the answer
  back); threads are spread over the ranks by their id."
  []
  (:batcher (nth @node (mod (.getId (Thread/currentThread)) engine-ranks))))

;; device buffers of the publisher (one sweep / ack / recover at a time: the two scheduler threads of
;; ring.clj:113-114 take this lock around their poll)
(defonce publisher-lock (Object.))
(defonce ^:private last-sweep-rows (atom 0))       ; rows the last sweep wrote, unused (void) ones included
(defonce buffers
  (delay (let [positions (* sweep-cap max-count)]
           {:conts    (device-alloc (* 64 sweep-cap))
            :lens     (device-alloc (* 4 positions))
            :waits    (device-alloc (* 4 positions))
            :payloads (device-alloc (* payload-stride positions))
            :ok       (device-alloc sweep-cap)
            :status   (device-alloc (* 4 sweep-cap))
            :result   (device-alloc 16)})))

;; ---- what stays on the JVM: id strings and the Fork maps (URLs), keyed by fingerprint -----------

(defonce ^ConcurrentHashMap forks (ConcurrentHashMap.))      ; [fp-lo fp-hi] -> Fork map with :timeout-secs

(defn utf8 ^bytes [^String s] (.getBytes s StandardCharsets/UTF_8))

(defn fingerprint [^String barrier-id]
  (let [b (utf8 barrier-id) out (Memory. 16)]
    (.invokeVoid ^Function (.getFunction dj "dj_fingerprint")
                 (to-array [b (long (alength b)) out (.share out 8)]))
    [(.getLong out 0) (.getLong out 8)]))

(defn barrier-id-of [record]
  (str (:join-id record) "~" (:transaction-id record)))                ; make-key, redis.clj:42-45

;; ---- requests ------------------------------------------------------------------------------------

(defn fork
  "record: the Fork map with :timeout-secs added (core.clj:21-22,29).  Deferred of the barrier id."
  [record]
  (deferred/future
    (let [barrier-id (barrier-id-of record)
          [lo hi :as fp] (fingerprint barrier-id)
          rec (doto (Memory. 32)                                       ; struct dj_fork_rec
                (.setLong 0 lo) (.setLong 8 hi)
                (.setInt 16 (int (:count record))) (.setInt 20 (int 0))
                (.setLong 24 (long (:timeout-secs record))))
          status (Memory. 4)
          previous (.put forks fp (assoc record :barrier-id barrier-id))]   ; before the engine can pop it
      (check (native-int dj "dj_batcher_fork_wait" (request-batcher) rec status) "dj_batcher_fork_wait")
      (if (= 1 (.getInt status 0))                                     ; DJ_FORK_CREATED
        barrier-id
        (do (if previous (.put forks fp previous) (.remove forks fp))    ; a refused fork leaves the barrier as it was
            (throw (ex-info "Database failure(s)" {:causes [(str "engine limit reached: " (.getInt status 0))]})))))))

(defn join-word
  "record: the Join map (store.clj:19-28).  Deferred of the engine's status word of the event: DJ_JOIN_* in the low
  byte and, once dj_set_join_hlen is on (the namespace distributed-join.redis turns it on), HLEN of the barrier's
  hash after the event above it (DJ_JOIN_HLEN_SHIFT = 8)."
  [record]
  (deferred/future
    (let [[lo hi] (fingerprint (:barrier-id record))
          message (utf8 (:message record))
          rec (doto (Memory. 32)                                       ; struct dj_join_rec
                (.setLong 0 lo) (.setLong 8 hi)
                (.setInt 16 (int (:position record)))
                (.setInt 20 (int (alength message)))
                (.setInt 24 (int (or (:max-wait record) 0)))
                (.setInt 28 (int 0)))
          status (Memory. 4)]
      (check (native-int dj "dj_batcher_join_wait" (request-batcher) rec message status) "dj_batcher_join_wait")
      (.getInt status 0))))

(defn join-status
  "Deferred of the engine's DJ_JOIN_* status of the event (DJ_JOIN_STATUS of its status word)."
  [record]
  (deferred/chain (join-word record) #(bit-and (int %) 0xFF)))

(defn join-failure
  "The error deferred of a refused join (the texts of redis.clj:152-153), nil for a status that stored the message."
  [status]
  (case (int status)
    (1 2 5) nil
    3 (err ::handling/not-found "barrier-id does not exist")
    4 (err ::handling/bad-data "position is greater than total")
    6 (err ::handling/bad-data (str "Message length cannot exceed " max-message))
    (deferred/error-deferred (ex-info "Database failure(s)" {:causes [(str "join status " status)]}))))

(defn join
  "record: the Join map (store.clj:19-28).  Deferred of {:complete? bool}; 404 / 400 as error
  deferreds with the texts of redis.clj:152-153."
  [record]
  (deferred/chain
    (join-status record)
    (fn [status]
      (or (join-failure status)
          (case (int status)
            1 {:complete? false}                                       ; DJ_JOIN_WAITING
            2 {:complete? true}                                        ; DJ_JOIN_COMPLETE
            ;; DJ_JOIN_NEAREST_NOT_QUEUED: the Lua compare of redis.clj:268-269 raised after the message was stored
            (deferred/error-deferred (ex-info "Database failure(s)" {:causes [(str "join status " status)]})))))))

(defn read-messages
  "Positions 0..count-1 of one barrier from host copies of the lens / max-waits / payload arrays:
  a vector of Join-shaped maps, nil where the position is missing (as-array, redis.clj:76-80)."
  [barrier-id ^Memory lens ^Memory waits ^Memory payloads ^long base ^long cnt]
  (mapv (fn [p]
          (let [i (+ base p)
                len (.getInt lens (* 4 i))]
            (when-not (= len -1)                                       ; DJ_NIL
              (let [wait (.getInt waits (* 4 i))]
                (cond-> {:barrier-id barrier-id :position p
                         :message (String. (.getByteArray payloads (long (* payload-stride i)) (int len)) StandardCharsets/UTF_8)}
                  (pos? wait) (assoc :max-wait wait))))))
        (range cnt)))

(defn- metadata-of
  "The metadata map of redis/join-metadata: the Fork map this JVM holds for the id, with the fields the engine
  owns read from the dj_cont at h-meta (:count, :timeout-secs, the two retry counters when they are set)."
  [fp ^Memory h-meta]
  (let [retry-publish (.getInt h-meta 40)
        retry-timeout (.getInt h-meta 44)]
    (cond-> (assoc (or (.get forks fp) {})
                   :count (.getInt h-meta 32)
                   :timeout-secs (.getLong h-meta 16))
      (not= -1 retry-publish) (assoc :publish-retry-cnt retry-publish)           ; DJ_NIL
      (not= -1 retry-timeout) (assoc :timeout-publish-retry-cnt retry-timeout))))

(defn snapshot
  "Deferred of {:metadata map :joins [join-or-nil ...]} or nil when the id has no metadata
  (join-metadata + all-joins-status, redis.clj:245-253).  update-ts? true is redis/all-joins
  (redis.clj:242-243): join-ts-read := join-ts."
  ([barrier-id] (snapshot barrier-id false))
  ([barrier-id update-ts?]
   (deferred/future
     (locking publisher-lock
       (let [[lo hi :as fp] (fingerprint barrier-id)
             meta (Memory. 64)
             lens (Memory. (* 4 max-count))
             waits (Memory. (* 4 max-count))
             payloads (Memory. (* payload-stride max-count))]
         (check (native-int dj "dj_snapshot_one" (engine-of fp) (long lo) (long hi) (int (if update-ts? 1 0)) meta lens waits payloads)
                "dj_snapshot_one")
         (when (zero? (bit-and 2 (.getInt meta 36)))                    ; DJ_CONT_NO_METADATA
           {:metadata (metadata-of fp meta)
            :joins    (read-messages barrier-id lens waits payloads 0 (.getInt meta 32))}))))))

;; ---- publisher -------------------------------------------------------------------------------------
;; sweep / ack / recover below work on ONE engine (@engine, rank 0 of the node; its device buffers are `buffers`).
;; With :engine-ranks above 1 every rank needs its own publisher loop over its own engine and buffers (sweeps, acks and
;; recovers are GPU-local): run these three with `engine` and `buffers` rebound per rank — not written out here.

(defn sweep
  "One poll of the publisher (core.clj:180-188 up to the POST).  Deferred of a vector of
  {:barrier-id :metadata :timeout? :messages [...] :index row}; :metadata is nil for an id that was
  queued without metadata (core.clj:159-161; its :barrier-id is nil too when this JVM never saw the fork: the
  engine keeps fingerprints, not id strings).  Call ack with the same vector before the next sweep."
  [now-ms]
  (deferred/future
    (locking publisher-lock
      (let [{:keys [conts lens waits payloads result]} @buffers
            h-result (Memory. 16)]
        (check (native-int dj "dj_sweep" @engine (long now-ms) (int sweep-cap) (long (* sweep-cap max-count))
                           conts lens waits payloads result h-result)
               "dj_sweep")
        (let [n (.getInt h-result 0)
              positions (.getLong h-result 8)
              h-conts (to-host conts (* 64 n))
              h-lens (to-host lens (* 4 positions))
              h-waits (to-host waits (* 4 positions))
              h-pay (to-host payloads (* payload-stride positions))]
          (reset! last-sweep-rows n)
          (into []
                (comp
                 (remove (fn [i] (neg? (.getInt h-conts (+ (* 64 i) 36)))))   ; DJ_CONT_VOID (bit 31): row not used
                 (map (fn [i]
                  (let [o (* 64 i)
                        fp [(.getLong h-conts o) (.getLong h-conts (+ o 8))]
                        flags (.getInt h-conts (+ o 36))
                        fork (.get forks fp)
                        id (:barrier-id fork)]
                    {:index i
                     :barrier-id id
                     :metadata (when (zero? (bit-and 2 flags)) fork)
                     :timeout? (pos? (bit-and 1 flags))                ; DJ_CONT_TIMEOUT, core.clj:86-89
                     :messages (mapv :message (read-messages id h-lens h-waits h-pay
                                                             (.getLong h-conts (+ o 48)) (.getInt h-conts (+ o 32))))
                     :fingerprint fp}))))
                (range n)))))))

(def ^:private ack-status
  {1 :done 2 :requeued-dirty 3 :retry 4 :dropped-exhausted 5 :done-no-metadata})

(defn ack
  "published?: one boolean per continuation of the last sweep, in its order.  Deferred of one keyword
  per continuation (complete / reset-publish / attempt-retry-publish, core.clj:107-155)."
  [continuations published? now-ms]
  (deferred/future
    (locking publisher-lock
      (let [{:keys [conts ok status]} @buffers
            n @last-sweep-rows
            h-ok (Memory. (max n 1))]
        (.clear h-ok)
        (doseq [[c p] (map vector continuations published?)]
          (.setByte h-ok (:index c) (byte (if p 1 0))))
        (to-device ok h-ok n)
        (check (native-int dj "dj_ack_batch" @engine (int n) Pointer/NULL conts ok (long now-ms) status) "dj_ack_batch")
        (check (native-int dj "dj_sync" @engine) "dj_sync")
        (let [h-status (to-host status (* 4 n))]
          (mapv (fn [c]
                  (let [s (ack-status (.getInt h-status (* 4 (:index c))))]
                    (when (#{:done :dropped-exhausted :done-no-metadata} s)
                      (.remove forks (:fingerprint c)))
                    s))
                continuations))))))

(defn recover
  "core/recover-unprocessed-publish-messages (core.clj:193-201).  Deferred of [barrier-id outcome] pairs."
  [now-ms]
  (deferred/future
    (locking publisher-lock
      (let [{:keys [conts status result]} @buffers
            h-result (Memory. 16)]
        (check (native-int dj "dj_recover" @engine (long now-ms) (int sweep-cap) conts status result h-result) "dj_recover")
        (let [n (min (.getInt h-result 0) sweep-cap)
              h-conts (to-host conts (* 64 n))
              h-status (to-host status (* 4 n))]
          (mapv (fn [i]
                  (let [fp [(.getLong h-conts (* 64 i)) (.getLong h-conts (+ (* 64 i) 8))]
                        s (ack-status (.getInt h-status (* 4 i)))
                        id (:barrier-id (.get forks fp))]
                    (log/info "Recovering publish queue entry for barrier id:" id)
                    (when (#{:dropped-exhausted :done-no-metadata} s)
                      (.remove forks fp))
                    [id s]))
                (range n)))))))

(defn get-publish-date
  "redis/get-publish-date (redis.clj:295-297): the primary-queue score, nil when not queued."
  [barrier-id]
  (deferred/future
    (let [[lo hi :as fp] (fingerprint barrier-id) out (Memory. 16)]
      (check (native-int dj "dj_queue_scores" (engine-of fp) (long lo) (long hi) out (.share out 8)) "dj_queue_scores")
      (let [at (.getLong out 0)]
        (when-not (= at -1) at)))))                                    ; DJ_NOT_QUEUED

(defn shutdown []                                                       ; redis/shutdown, redis.clj:299-300
  (log/info "Shutting down the barrier engine...")
  (when (realized? node)
    ;; a routed batcher's destroy is a collective (its drain thread leaves when every rank has been asked to stop)
    (->> @node
         (mapv (fn [{:keys [batcher]}] (future (.invokeVoid ^Function (.getFunction dj "dj_batcher_destroy") (to-array [batcher])))))
         (run! deref))
    (doseq [{:keys [router]} @node :when router]
      (.invokeVoid ^Function (.getFunction dj "dj_router_destroy") (to-array [router])))
    (doseq [{:keys [engine device]} @node]
      (when-let [path (:engine-checkpoint-path env)]
        (check (native-int dj "dj_checkpoint_save" engine ^String (if (= engine-ranks 1) path (str path "." device))) "dj_checkpoint_save"))
      (.invokeVoid ^Function (.getFunction dj "dj_destroy") (to-array [engine])))))
