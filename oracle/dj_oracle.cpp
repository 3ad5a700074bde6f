// dj_oracle.cpp — CPU oracle.  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see dj_oracle.h).
//
// A literal, sequential restatement of the reference's barrier logic.  File:line
// citations are relative to the reference tree.  The store half mirrors the Redis data
// model key by key:
//   "<id>:metadata"      -> metadata   (redis.clj:51-57)
//   "<id>"               -> joins      (hash position -> join record, redis.clj:42-47)
//   "<id>:join-ts"       -> join_ts    (redis.clj:59-65)
//   "<id>:join-ts-read"  -> join_ts_read (redis.clj:67-73)
//   "publish-queue" / "publish-queue-backup" -> two sorted sets (redis.clj:36-37)
// Parity is pinned by tests/golden/scenarios.json (the reference tests' own assertions);
// the reference itself cannot run in this environment.
#include "dj_oracle.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

using std::optional;
using std::string;

// The metadata map stored by redis/fork: the Fork request plus :timeout-secs
// (core.clj:21-22) and, later, the two optional retry counters (core.clj:97-111).
struct Metadata {
    uint32_t count = 0;
    uint64_t timeout_secs = 0;         // absolute epoch ms despite the name (time_util.clj:6-10)
    optional<uint32_t> publish_retry_cnt;
    optional<uint32_t> timeout_publish_retry_cnt;
};

// The join record stored per position: the whole Join request (redis.clj:135,155).
struct JoinRecord {
    string message;
    optional<uint32_t> max_wait;
};

// A Redis sorted set: members ordered by (score, member bytes).
struct ZSet {
    std::unordered_map<string, uint64_t> score;
    std::set<std::pair<uint64_t, string>> order;
    void zadd(const string& m, uint64_t s) {
        auto it = score.find(m);
        if (it != score.end()) {
            order.erase({it->second, m});
            it->second = s;
        } else {
            score.emplace(m, s);
        }
        order.insert({s, m});
    }
    void zrem(const string& m) {
        auto it = score.find(m);
        if (it == score.end()) return;
        order.erase({it->second, m});
        score.erase(it);
    }
    optional<uint64_t> zscore(const string& m) const {
        auto it = score.find(m);
        if (it == score.end()) return std::nullopt;
        return it->second;
    }
    // ZRANGEBYSCORE key 0 max (inclusive)
    std::vector<string> zrangebyscore(uint64_t max) const {
        std::vector<string> r;
        for (auto& p : order) {
            if (p.first > max) break;
            r.push_back(p.second);
        }
        return r;
    }
    void zremrangebyscore(uint64_t max) {
        while (!order.empty() && order.begin()->first <= max) {
            score.erase(order.begin()->second);
            order.erase(order.begin());
        }
    }
};

// How Redis renders an integral score and how carmine renders a Long argument: plain
// decimal digits.  The Lua script compares these as STRINGS (redis.clj:268-269).
string decimal(uint64_t v) {
    char b[32];
    snprintf(b, sizeof b, "%llu", (unsigned long long)v);
    return b;
}

enum class CompleteResult { Deleted, UpdatedJoinBeforeComplete };

}  // namespace

struct djo {
    // config (config/*/config.edn)
    uint64_t check_publish_delay_ms;
    uint32_t max_retry_post_publish;
    uint32_t max_retry_post_timeout_publish;
    uint32_t fork_count_max;
    uint32_t join_message_len_max;

    // the "Redis" keyspace
    std::unordered_map<string, Metadata> metadata;
    std::unordered_map<string, std::map<uint32_t, JoinRecord>> joins;
    std::unordered_map<string, uint64_t> join_ts, join_ts_read;
    ZSet publish_queue, publish_queue_backup;

    // id table for the array interface
    std::vector<string> ids;
    std::unordered_map<string, uint32_t> id_index;

    // ------------------------------------------------------------------ redis.clj
    // redis/fork, redis.clj:90-104 (one MULTI)
    void redis_fork(const string& key, const Metadata& record) {
        metadata[key] = record;            // SET <id>:metadata
        joins.erase(key);                  // DEL <id>
        join_ts.erase(key);                // DEL <id>:join-ts
        join_ts_read.erase(key);           // DEL <id>:join-ts-read
        publish_queue.zrem(key);           // ZREM publish-queue
        publish_queue_backup.zrem(key);    // ZREM publish-queue-backup
    }
    // redis/update-metadata, redis.clj:106-111
    void redis_update_metadata(const string& key, const Metadata& m) { metadata[key] = m; }
    // redis/join-metadata, redis.clj:248-253
    optional<Metadata> redis_join_metadata(const string& key) const {
        auto it = metadata.find(key);
        if (it == metadata.end()) return std::nullopt;
        return it->second;
    }
    // redis/join, redis.clj:143-159.  Returns a DJO_JOIN_* error code or 0 with *hlen set.
    int redis_join(const string& key, uint32_t pos, const JoinRecord& record, uint64_t now_ts,
                   Metadata* md_out, uint32_t* hlen) {
        auto md = redis_join_metadata(key);                    // GET, redis.clj:149
        if (!md) return DJO_JOIN_NOT_FOUND;                    // redis.clj:152
        if (pos >= md->count) return DJO_JOIN_BAD_POSITION;    // redis.clj:153
        // Lua add-join-and-count-them, redis.clj:133-140
        if (metadata.count(key) == 0) return DJO_JOIN_NOT_FOUND;  // redis.clj:139,158-159
        joins[key][pos] = record;                              // HSET: last writer wins
        join_ts[key] = now_ts;                                 // SET <id>:join-ts
        *hlen = (uint32_t)joins[key].size();                   // HLEN
        *md_out = *md;
        return 0;
    }
    // redis/complete with Lua complete-if-no-update, redis.clj:162-205
    CompleteResult redis_complete(const string& key) {
        auto jt = join_ts.find(key);
        auto jr = join_ts_read.find(key);
        bool jt_nil = jt == join_ts.end(), jr_nil = jr == join_ts_read.end();
        // redis.clj:179: two nils compare equal, nil never equals a value
        bool equal = (jt_nil && jr_nil) || (!jt_nil && !jr_nil && jt->second == jr->second);
        if (!equal) return CompleteResult::UpdatedJoinBeforeComplete;  // redis.clj:187,203-205
        metadata.erase(key);
        joins.erase(key);
        join_ts.erase(key);
        join_ts_read.erase(key);
        publish_queue.zrem(key);
        publish_queue_backup.zrem(key);
        return CompleteResult::Deleted;
    }
    // redis/all-joins and all-joins-status, redis.clj:210-246, with as-array redis.clj:76-80
    std::vector<optional<JoinRecord>> redis_all_joins(const string& key, uint32_t count,
                                                      bool update_timestamp) {
        if (update_timestamp) {                                // redis.clj:222-229
            auto jt = join_ts.find(key);
            if (jt != join_ts.end())
                join_ts_read[key] = jt->second;
            else
                join_ts_read.erase(key);
        }
        std::vector<optional<JoinRecord>> out(count);
        auto it = joins.find(key);                             // HGETALL
        if (it != joins.end())
            for (auto& kv : it->second)
                if (kv.first < count) out[kv.first] = kv.second;
        return out;
    }
    // redis/store-in-publish-queue, redis.clj:255-259
    void redis_store_in_publish_queue(const string& key, uint64_t publish_at) {
        publish_queue_backup.zadd(key, publish_at);
        publish_queue.zadd(key, publish_at);
    }
    // redis/store-nearest-in-publish-queue, redis.clj:261-281.  false = the script raised
    // (ZSCORE returned a nil bulk, i.e. Lua false, and `false > string` is an error).
    bool redis_store_nearest_in_publish_queue(const string& key, uint64_t publish_at) {
        auto cur = publish_queue.zscore(key);
        if (!cur) return false;
        if (decimal(*cur) > decimal(publish_at)) {            // string compare, redis.clj:269
            publish_queue_backup.zadd(key, publish_at);
            publish_queue.zadd(key, publish_at);
        }
        return true;
    }
    // redis/all-publish-messages, redis.clj:283-289
    std::vector<string> redis_all_publish_messages(uint64_t now) {
        auto r = publish_queue.zrangebyscore(now);
        publish_queue.zremrangebyscore(now);
        return r;
    }
    // redis/get-unprocessed-publish-messages, redis.clj:291-293
    std::vector<string> redis_get_unprocessed_publish_messages(uint64_t now) const {
        if (now < check_publish_delay_ms) return {};
        return publish_queue_backup.zrangebyscore(now - check_publish_delay_ms);
    }

    // ------------------------------------------------------------------- core.clj
    // core/fork, core.clj:24-33; the caller has already added the max-wait offset
    int core_fork(const string& key, uint32_t count, uint64_t timeout_secs) {
        if (count < 1 || count > fork_count_max) return DJO_FORK_BAD_COUNT;  // store.clj:32-34
        Metadata m;
        m.count = count;
        m.timeout_secs = timeout_secs;
        redis_fork(key, m);                                    // core.clj:29
        redis_store_in_publish_queue(key, timeout_secs);       // core.clj:30
        return DJO_FORK_CREATED;
    }
    // core/join, core.clj:44-58
    int core_join(const string& key, uint32_t pos, const JoinRecord& rec, uint64_t now) {
        if (rec.message.size() > join_message_len_max) return DJO_JOIN_TOO_LONG;  // store.clj:24-26
        Metadata md;
        uint32_t hlen = 0;
        int err = redis_join(key, pos, rec, now, &md, &hlen);
        if (err) return err;
        bool all_there = hlen == md.count;                     // core.clj:36-38
        if (all_there) {
            redis_store_in_publish_queue(key, now);            // core.clj:54
            return DJO_JOIN_COMPLETE;
        }
        if (rec.max_wait) {                                    // core.clj:40-42
            uint64_t date = now + 1000ull * *rec.max_wait;
            if (!redis_store_nearest_in_publish_queue(key, date)) return DJO_JOIN_NEAREST_NOT_QUEUED;
        }
        return DJO_JOIN_WAITING;
    }
    // is-timeout?, core.clj:86-89 (strictly greater)
    static bool is_timeout(const Metadata& m, uint64_t now) { return now > m.timeout_secs; }

    // update-to-retry-publish, core.clj:107-111
    void update_to_retry_publish(const string& key, Metadata m, bool timeout, uint32_t cnt, uint64_t now) {
        if (timeout) m.timeout_publish_retry_cnt = cnt; else m.publish_retry_cnt = cnt;
        redis_update_metadata(key, m);
        redis_store_in_publish_queue(key, now);
    }
    // reset-publish, core.clj:113-117
    void reset_publish(const string& key, uint64_t now) {
        auto md = redis_join_metadata(key);
        Metadata m = md ? *md : Metadata{};
        m.timeout_publish_retry_cnt.reset();
        m.publish_retry_cnt.reset();
        update_to_retry_publish(key, m, false, 0, now);
    }
    // complete, core.clj:119-130.  Returns true when the barrier was deleted.
    bool core_complete(const string& key, uint64_t now) {
        if (redis_complete(key) == CompleteResult::Deleted) return true;
        reset_publish(key, now);                               // core.clj:127-129
        return false;
    }
    // attempt-retry-publish 3-arity, core.clj:132-152
    int attempt_retry_publish(const string& key, const optional<Metadata>& md, uint64_t now) {
        if (!md)                                               // core.clj:134-136
            return core_complete(key, now) ? DJO_ACK_DONE_NO_METADATA : DJO_ACK_REQUEUED_DIRTY;
        bool timeout = is_timeout(*md, now);
        uint32_t cnt = timeout ? md->timeout_publish_retry_cnt.value_or(0)
                               : md->publish_retry_cnt.value_or(0);
        uint32_t max = timeout ? max_retry_post_timeout_publish : max_retry_post_publish;
        if (cnt >= max)                                        // core.clj:142-147
            return core_complete(key, now) ? DJO_ACK_DROPPED_EXHAUSTED : DJO_ACK_REQUEUED_DIRTY;
        update_to_retry_publish(key, *md, timeout, cnt + 1, now);  // core.clj:151
        return DJO_ACK_RETRY;
    }

    // ------------------------------------------------------------------ array glue
    const string& id_of(uint32_t idx) const { return ids.at(idx); }
    uint32_t idx_of(const string& id) const {
        auto it = id_index.find(id);
        return it == id_index.end() ? DJO_NIL : it->second;
    }
    void fill_cont(djo_cont* c, const string& id, const optional<Metadata>& md, uint64_t now,
                   uint64_t publish_at) const {
        memset(c, 0, sizeof *c);
        c->id_idx = idx_of(id);
        c->publish_at = publish_at;
        c->retry_publish = c->retry_timeout = DJO_NIL;
        if (!md) {
            c->flags = DJO_CONT_NO_METADATA;
            return;
        }
        c->count = md->count;
        c->deadline_ms = md->timeout_secs;
        if (md->publish_retry_cnt) c->retry_publish = *md->publish_retry_cnt;
        if (md->timeout_publish_retry_cnt) c->retry_timeout = *md->timeout_publish_retry_cnt;
        if (is_timeout(*md, now)) c->flags |= DJO_CONT_TIMEOUT;   // core.clj:91-95
    }
    static optional<Metadata> md_of_cont(const djo_cont& c) {
        if (c.flags & DJO_CONT_NO_METADATA) return std::nullopt;
        Metadata m;
        m.count = c.count;
        m.timeout_secs = c.deadline_ms;
        if (c.retry_publish != DJO_NIL) m.publish_retry_cnt = c.retry_publish;
        if (c.retry_timeout != DJO_NIL) m.timeout_publish_retry_cnt = c.retry_timeout;
        return m;
    }
    void write_joins(const std::vector<optional<JoinRecord>>& js, uint64_t pos_base, uint32_t* lens,
                     uint32_t* max_waits, uint8_t* payloads, uint64_t stride) const {
        for (size_t p = 0; p < js.size(); ++p) {
            uint64_t q = pos_base + p;
            if (!js[p]) {
                if (lens) lens[q] = DJO_NIL;
                if (max_waits) max_waits[q] = 0;
                continue;
            }
            if (lens) lens[q] = (uint32_t)js[p]->message.size();
            if (max_waits) max_waits[q] = js[p]->max_wait.value_or(0);
            if (payloads) memcpy(payloads + q * stride, js[p]->message.data(), js[p]->message.size());
        }
    }
};

extern "C" {

djo* djo_create(uint64_t check_publish_delay_ms, uint32_t max_retry_publish,
                uint32_t max_retry_timeout_publish, uint32_t max_count, uint32_t max_message_len) {
    djo* o = new djo();
    o->check_publish_delay_ms = check_publish_delay_ms;
    o->max_retry_post_publish = max_retry_publish;
    o->max_retry_post_timeout_publish = max_retry_timeout_publish;
    o->fork_count_max = max_count;
    o->join_message_len_max = max_message_len;
    return o;
}
void djo_destroy(djo* o) { delete o; }

void djo_set_ids(djo* o, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_ids) {
    o->ids.clear();
    o->id_index.clear();
    o->ids.reserve(n_ids);
    o->id_index.reserve(n_ids * 2);
    for (uint64_t i = 0; i < n_ids; ++i) {
        o->ids.emplace_back((const char*)bytes + offsets[i], offsets[i + 1] - offsets[i]);
        o->id_index.emplace(o->ids.back(), (uint32_t)i);
    }
}

void djo_fork_batch(djo* o, uint32_t n, const uint32_t* id_idx, const uint32_t* count,
                    const uint64_t* deadline_ms, int32_t* status) {
    for (uint32_t i = 0; i < n; ++i) status[i] = o->core_fork(o->id_of(id_idx[i]), count[i], deadline_ms[i]);
}

void djo_join_batch(djo* o, uint32_t n, const uint32_t* id_idx, const uint32_t* position,
                    const uint32_t* len, const uint32_t* max_wait_s, const uint8_t* payloads,
                    uint64_t pitch, uint64_t now_ms, int32_t* status) {
    for (uint32_t i = 0; i < n; ++i) {
        JoinRecord r;
        r.message.assign((const char*)payloads + (uint64_t)i * pitch, len[i]);
        if (max_wait_s[i]) r.max_wait = max_wait_s[i];
        status[i] = o->core_join(o->id_of(id_idx[i]), position[i], r, now_ms);
    }
}

int64_t djo_pop_due(djo* o, uint64_t now_ms, uint32_t cap, djo_cont* out) {
    // peek first so that a too-small cap leaves the state untouched
    auto due = o->publish_queue.zrangebyscore(now_ms);
    if (due.size() > cap) return -1;
    std::vector<uint64_t> scores;
    for (auto& id : due) scores.push_back(*o->publish_queue.zscore(id));
    auto popped = o->redis_all_publish_messages(now_ms);
    for (size_t k = 0; k < popped.size(); ++k)
        o->fill_cont(&out[k], popped[k], o->redis_join_metadata(popped[k]), now_ms, scores[k]);
    return (int64_t)popped.size();
}

// core/process-publish-messages up to the POST (core.clj:180-188, 173-178, 157-166)
int64_t djo_sweep(djo* o, uint64_t now_ms, uint32_t cap_conts, uint64_t cap_positions,
                  djo_cont* conts, uint32_t* lens, uint32_t* max_waits, uint8_t* payloads,
                  uint64_t stride, uint64_t* n_positions) {
    auto due = o->publish_queue.zrangebyscore(now_ms);
    if (due.size() > cap_conts) return -1;
    uint64_t need = 0;
    for (auto& id : due) {
        auto md = o->redis_join_metadata(id);
        if (md) need += md->count;
    }
    if (need > cap_positions) return -1;
    std::vector<uint64_t> scores;
    for (auto& id : due) scores.push_back(*o->publish_queue.zscore(id));
    auto popped = o->redis_all_publish_messages(now_ms);      // core.clj:184
    uint64_t pos = 0;
    for (size_t k = 0; k < popped.size(); ++k) {
        const string& id = popped[k];
        auto md = o->redis_join_metadata(id);                  // core.clj:175
        o->fill_cont(&conts[k], id, md, now_ms, scores[k]);    // core.clj:176
        conts[k].pos_base = pos;
        if (!md) continue;                                     // core.clj:159-161: completed by the ack
        auto js = o->redis_all_joins(id, md->count, true);     // core.clj:163
        o->write_joins(js, pos, lens, max_waits, payloads, stride);
        pos += md->count;
    }
    if (n_positions) *n_positions = pos;
    return (int64_t)popped.size();
}

void djo_ack_batch(djo* o, uint32_t n, const djo_cont* conts, const uint8_t* ok,
                   uint64_t now_ms, int32_t* status) {
    for (uint32_t i = 0; i < n; ++i) {
        const string& id = o->id_of(conts[i].id_idx);
        auto md = djo::md_of_cont(conts[i]);
        bool published = ok == nullptr || ok[i] != 0;
        if (!md) {
            // core.clj:159-161 (publish path) and core.clj:134-136 (retry path): complete as failed
            status[i] = o->core_complete(id, now_ms) ? DJO_ACK_DONE_NO_METADATA : DJO_ACK_REQUEUED_DIRTY;
        } else if (published) {
            // core.clj:168: (complete barrier-id true publish-resp)
            status[i] = o->core_complete(id, now_ms) ? DJO_ACK_DONE : DJO_ACK_REQUEUED_DIRTY;
        } else {
            // core.clj:167,170: attempt-retry-publish with the metadata held by the publisher
            status[i] = o->attempt_retry_publish(id, md, now_ms);
        }
    }
}

// core/recover-unprocessed-publish-messages, core.clj:193-201 with the 2-arity
// attempt-retry-publish, core.clj:153-155
int64_t djo_recover(djo* o, uint64_t now_ms, uint32_t cap, djo_cont* out, int32_t* status) {
    auto idsv = o->redis_get_unprocessed_publish_messages(now_ms);
    if (idsv.size() > cap) return -1;
    for (size_t k = 0; k < idsv.size(); ++k) {
        auto md = o->redis_join_metadata(idsv[k]);
        o->fill_cont(&out[k], idsv[k], md, now_ms, *o->publish_queue_backup.zscore(idsv[k]));
        status[k] = o->attempt_retry_publish(idsv[k], md, now_ms);
    }
    return (int64_t)idsv.size();
}

// core/status read side (core.clj:60-68) or redis/all-joins when update_ts is set
int djo_snapshot_one(djo* o, uint32_t id_idx, int update_ts, djo_cont* meta, uint32_t* lens,
                     uint32_t* max_waits, uint8_t* payloads, uint64_t stride) {
    const string& id = o->id_of(id_idx);
    auto md = o->redis_join_metadata(id);
    auto pq = o->publish_queue.zscore(id);
    o->fill_cont(meta, id, md, 0, pq ? *pq : DJO_NOT_QUEUED);
    meta->flags &= ~DJO_CONT_TIMEOUT;
    if (!md) return 0;
    auto js = o->redis_all_joins(id, md->count, update_ts != 0);
    o->write_joins(js, 0, lens, max_waits, payloads, stride);
    return 0;
}

void djo_store_in_publish_queue(djo* o, uint32_t id_idx, uint64_t at_ms) {
    o->redis_store_in_publish_queue(o->id_of(id_idx), at_ms);
}

void djo_queue_scores(djo* o, uint32_t id_idx, uint64_t* primary_at, uint64_t* backup_at) {
    auto p = o->publish_queue.zscore(o->id_of(id_idx));
    auto b = o->publish_queue_backup.zscore(o->id_of(id_idx));
    *primary_at = p ? *p : DJO_NOT_QUEUED;
    *backup_at = b ? *b : DJO_NOT_QUEUED;
}

uint64_t djo_live_barriers(djo* o) { return o->metadata.size(); }

// ---- the store namespace command by command (redis.clj's public functions, nothing of core.clj) ----
namespace {
Metadata md_from(uint32_t count, uint64_t timeout_secs, uint32_t retry_publish, uint32_t retry_timeout) {
    Metadata m;
    m.count = count;
    m.timeout_secs = timeout_secs;
    if (retry_publish != DJO_NIL) m.publish_retry_cnt = retry_publish;
    if (retry_timeout != DJO_NIL) m.timeout_publish_retry_cnt = retry_timeout;
    return m;
}
int64_t ids_out(djo* o, const std::vector<string>& v, uint32_t cap, uint32_t* id_idx) {
    if (v.size() > cap) return -1;
    for (size_t k = 0; k < v.size(); ++k) id_idx[k] = o->idx_of(v[k]);
    return (int64_t)v.size();
}
}  // namespace

void djo_redis_fork(djo* o, uint32_t id_idx, uint32_t count, uint64_t timeout_secs) {
    o->redis_fork(o->id_of(id_idx), md_from(count, timeout_secs, DJO_NIL, DJO_NIL));
}
void djo_redis_update_metadata(djo* o, uint32_t id_idx, uint32_t count, uint64_t timeout_secs, uint32_t retry_publish,
                               uint32_t retry_timeout) {
    o->redis_update_metadata(o->id_of(id_idx), md_from(count, timeout_secs, retry_publish, retry_timeout));
}
int djo_redis_join(djo* o, uint32_t id_idx, uint32_t position, const uint8_t* message, uint32_t len,
                   uint32_t max_wait_s, uint64_t now_ms, uint32_t* hlen) {
    JoinRecord r;
    r.message.assign((const char*)message, len);
    if (max_wait_s) r.max_wait = max_wait_s;
    Metadata md;
    return o->redis_join(o->id_of(id_idx), position, r, now_ms, &md, hlen);
}
int djo_redis_complete(djo* o, uint32_t id_idx) {
    return o->redis_complete(o->id_of(id_idx)) == CompleteResult::Deleted ? 1 : 0;
}
int djo_redis_store_nearest_in_publish_queue(djo* o, uint32_t id_idx, uint64_t publish_at) {
    return o->redis_store_nearest_in_publish_queue(o->id_of(id_idx), publish_at) ? 1 : 0;
}
int64_t djo_redis_all_publish_messages(djo* o, uint64_t now_ms, uint32_t cap, uint32_t* id_idx) {
    if (o->publish_queue.zrangebyscore(now_ms).size() > cap) return -1;
    return ids_out(o, o->redis_all_publish_messages(now_ms), cap, id_idx);
}
int64_t djo_redis_get_unprocessed_publish_messages(djo* o, uint64_t now_ms, uint32_t cap, uint32_t* id_idx) {
    return ids_out(o, o->redis_get_unprocessed_publish_messages(now_ms), cap, id_idx);
}

}  // extern "C"
