"""Parity harness: applies one op-log to the HIP engine and to the CPU oracle and compares
every observable result (per-event statuses, popped continuations with their position-
ordered messages and timeout classification, ack / recover outcomes, final snapshots).

Bit-exact bar: statuses, counters, scores and message bytes must be equal.  The engine hands a sweep's
continuations over in no particular order; the host that holds the id strings sorts them the way Redis
returns a ZRANGEBYSCORE (score, then member bytes: redis.clj:283-289), and that list must equal the oracle's.
"""
import numpy as np

from distributed_join_amd import capi

MASK64 = (1 << 64) - 1


def splitmix64(x):
    """Vectorised SplitMix64 finaliser on uint64 arrays."""
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def payload_bytes(barrier, pos, version, pitch):
    """Deterministic message bytes: byte i of (barrier, position, version) is a hash of all four,
    so a misplaced or stale byte is detectable.  -> uint8 array (n, pitch)."""
    barrier = np.asarray(barrier, dtype=np.uint64)
    pos = np.asarray(pos, dtype=np.uint64)
    version = np.asarray(version, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = splitmix64(barrier * np.uint64(1024) + pos + (version << np.uint64(40)))
        words = splitmix64(base[:, None] + np.arange(pitch // 8, dtype=np.uint64)[None, :] * np.uint64(0x100000001B3))
    return words.view(np.uint8).reshape(len(barrier), pitch)


class ParityDriver:
    def __init__(self, engine, oracle, ids, oracle_caps=None):
        self.e, self.o = engine, oracle
        self.oracle_caps = oracle_caps or {}      # cap_conts / cap_positions of the oracle's sweep buffers (rows of its whole-message width)
        self.ids = list(ids)
        self.fps = capi.fingerprint_many(self.ids)
        self.by_fp = {(int(a), int(b)): k for k, (a, b) in enumerate(self.fps)}
        assert len(self.by_fp) == len(self.ids), "fingerprint collision among test ids"
        self.stride = engine.stride

    # ---- ops -------------------------------------------------------------------------------
    def fork(self, bidx, count, deadline):
        bidx = np.asarray(bidx, dtype=np.int64)
        recs = np.zeros(len(bidx), dtype=capi.FORK_REC)
        recs["fp_lo"], recs["fp_hi"] = self.fps[bidx, 0], self.fps[bidx, 1]
        recs["count"], recs["deadline_ms"] = count, deadline
        got = self.e.fork_batch(recs)
        want = self.o.fork_batch(bidx, count, deadline)
        np.testing.assert_array_equal(got, want, err_msg="fork statuses")
        return got

    def join(self, bidx, pos, length, mw, payloads, now):
        bidx = np.asarray(bidx, dtype=np.int64)
        recs = np.zeros(len(bidx), dtype=capi.JOIN_REC)
        recs["fp_lo"], recs["fp_hi"] = self.fps[bidx, 0], self.fps[bidx, 1]
        recs["position"], recs["len"], recs["max_wait_s"] = pos, length, mw
        got = self.e.join_batch(recs, payloads, now)
        want = self.o.join_batch(bidx, pos, length, mw, payloads, now)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (f"join statuses differ at {bad[:10]}: engine {got[bad[:10]]} oracle {want[bad[:10]]} "
                               f"barrier {bidx[bad[:10]]} pos {np.asarray(pos)[bad[:10]]}")
        return got

    def _canon_engine(self, conts, lens=None, mws=None, pay=None):
        out = {}
        for c in conts:
            if int(c["flags"]) & capi.CONT_VOID:
                continue
            k = self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))]
            out[k] = self._canon(c, k, lens, mws, pay)
        assert len(out) == sum(1 for c in conts if not int(c["flags"]) & capi.CONT_VOID), "barrier popped twice in one sweep"
        return out

    def _canon_oracle(self, conts, lens=None, mws=None, pay=None):
        return {int(c["id_idx"]): self._canon(c, int(c["id_idx"]), lens, mws, pay) for c in conts}

    @staticmethod
    def _canon(c, k, lens, mws, pay):
        head = (k, int(c["count"]), int(c["flags"]), int(c["deadline_ms"]), int(c["publish_at"]),
                int(c["retry_publish"]), int(c["retry_timeout"]))
        if lens is None or int(c["flags"]) & capi.DJ_CONT_NO_METADATA:
            return head, None
        b = int(c["pos_base"])
        msgs = []
        for p in range(int(c["count"])):
            ln = int(lens[b + p])
            msgs.append(None if ln == capi.DJ_NIL else (bytes(pay[b + p, :ln]), int(mws[b + p])))
        return head, tuple(msgs)

    def redis_order(self, conts):
        """Engine continuations in the order Redis pops them: by score, then by member (id) bytes."""
        live = [c for c in conts if not int(c["flags"]) & capi.CONT_VOID]
        return sorted(live, key=lambda c: (int(c["publish_at"]), self.ids[self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))]].encode()))

    def sweep(self, now):
        conts, lens, mws, pay, more = self.e.sweep(now)
        assert not more, "engine sweep hit its host caps in a parity test"
        oc, ol, om, op_ = self.o.sweep(now, **self.oracle_caps)
        got, want = self._canon_engine(conts, lens, mws, pay), self._canon_oracle(oc, ol, om, op_)
        assert got.keys() == want.keys(), f"popped sets differ: only engine {sorted(got.keys() - want.keys())[:10]} only oracle {sorted(want.keys() - got.keys())[:10]}"
        for k in want:
            assert got[k] == want[k], f"continuation of {self.ids[k]} differs: {self._diff(got[k], want[k])}"
        order_e = [self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))] for c in self.redis_order(conts)]
        order_o = [int(c["id_idx"]) for c in oc]
        assert order_e == order_o, "pop order differs from the oracle's (score, member) order"
        return conts, oc

    @staticmethod
    def _diff(g, w):
        """Short description of where two canonical continuations differ (messages can be long)."""
        if g[0] != w[0]:
            return f"head engine {g[0]} oracle {w[0]}"
        if g[1] is None or w[1] is None:
            return f"messages engine {g[1] is not None} oracle {w[1] is not None}"
        out = []
        for p, (a, b) in enumerate(zip(g[1], w[1])):
            if a == b:
                continue
            if a is None or b is None:
                out.append(f"pos {p}: engine {'missing' if a is None else len(a[0])} oracle {'missing' if b is None else len(b[0])}")
                continue
            first = next((i for i, (x, y) in enumerate(zip(a[0], b[0])) if x != y), min(len(a[0]), len(b[0])))
            out.append(f"pos {p}: len {len(a[0])}/{len(b[0])} max-wait {a[1]}/{b[1]} first differing byte {first} "
                       f"engine {a[0][first:first + 8].hex()} oracle {b[0][first:first + 8].hex()}")
        return "; ".join(out[:6])

    def pop(self, now):
        conts, more = self.e.pop_due(now)
        assert not more
        oc = self.o.pop_due(now)
        got, want = self._canon_engine(conts), self._canon_oracle(oc)
        assert got == want, f"pop_due differs: engine {got} oracle {want}"
        assert [self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))] for c in self.redis_order(conts)] == [int(c["id_idx"]) for c in oc]
        return conts, oc

    def ack(self, conts, oconts, ok_of, now):
        """ok_of: barrier index -> bool.  Acks the same continuations on both sides in the same order (the
        oracle's: the order of the pop).  A barrier may appear more than once (continuations of two pops acked
        together): the oracle takes them one by one, the engine side splits the list into calls without
        repeats, as the host layers do (one barrier at most once per dj_ack_batch call)."""
        by_idx = {}
        for c in conts:
            if not int(c["flags"]) & capi.CONT_VOID:
                by_idx.setdefault(self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))], []).append(c)
        if not len(oconts):
            return {}
        ok_ = [int(c["id_idx"]) for c in oconts]
        ordered = [by_idx[k].pop(0) for k in ok_]          # the engine's records, in the oracle's order
        oks = np.array([ok_of(k) for k in ok_], dtype=np.uint8)
        want = self.o.ack_batch(oconts, oks, now)
        got = np.zeros(len(ordered), dtype=np.int32)
        pending = list(range(len(ordered)))
        while pending:
            seen, batch, rest = set(), [], []
            for i in pending:
                (rest if ok_[i] in seen else batch).append(i)
                seen.add(ok_[i])
            got[batch] = self.e.ack_batch(np.array([ordered[i] for i in batch], dtype=capi.CONT), oks[batch], now)
            pending = rest
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, f"ack outcomes differ: {[(self.ids[ok_[i]], int(got[i]), int(want[i])) for i in bad[:10]]}"
        return dict(zip(ok_, got.tolist()))

    def recover(self, now):
        conts, st, more = self.e.recover(now)
        assert not more
        oc, ost = self.o.recover(now)
        g = {self.by_fp[(int(c["fp_lo"]), int(c["fp_hi"]))]: (self._canon(c, 0, None, None, None)[0][1:], int(s))
             for c, s in zip(conts, st)}
        w = {int(c["id_idx"]): (self._canon(c, 0, None, None, None)[0][1:], int(s)) for c, s in zip(oc, ost)}
        assert g == w, f"recover differs:\n engine {g}\n oracle {w}"
        return g

    def check_all_snapshots(self, which=None):
        for k in (range(len(self.ids)) if which is None else which):
            em, el, emw, ep = self.e.snapshot_one(self.fps[k, 0], self.fps[k, 1])
            om, ol, omw, op_ = self.o.snapshot_one(k)
            em = em.copy()
            em["pos_base"] = 0
            g = self._canon(em, k, el, emw, ep)
            w = self._canon(om, k, ol, omw, op_)
            assert g == w, f"snapshot of {self.ids[k]} differs:\n engine {g}\n oracle {w}"
            assert self.e.queue_scores(self.fps[k, 0], self.fps[k, 1]) == self.o.queue_scores(k), f"queue scores of {self.ids[k]}"


def routed_order(owners_by_rank):
    """The order in which ONE store must apply the streams of several ranks so that every barrier sees its events as the
    router delivers them: barriers of owner r get r's own events first (they never cross the exchange), then the other
    ranks' in rank order, each rank's in its submission order.  owners_by_rank[s] = owner rank of every event of rank s.
    -> [(source rank, indices into that rank's stream), ...]; barriers of different owners do not interact, so the
    segments of one owner may follow those of another."""
    world = len(owners_by_rank)
    out = []
    for r in range(world):
        for s in [r] + [q for q in range(world) if q != r]:
            idx = np.nonzero(np.asarray(owners_by_rank[s]) == r)[0]
            if len(idx):
                out.append((s, idx))
    return out
