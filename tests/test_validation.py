"""The request rows of the reference's input-validation test (test/distributed_join/ring_test.clj:42-126),
driven through the host mirror (core.ForkJoinService) on the CPU oracle store."""
import pytest

from distributed_join_amd.core import ForkJoinService
from distributed_join_amd.store import DEFAULT_ENV

PORT = 3001
DEST = f"http://localhost:{PORT}/v1/publish"
TDEST = f"http://localhost:{PORT}/v1/publish/timeout"
CNT = 7
HUNDRED = "0123456789" * 10


def large(n):
    return (HUNDRED * (n // 100 + 1))[:n]


ENV = dict(DEFAULT_ENV)
JID, TID = "a01", "b01"
LARGE_TID = large(ENV["transaction-id-len-max"] + 1)
LARGE_JID = large(ENV["join-id-len-max"] + 1)
LARGE_DEST = "http://" + large(ENV["destination-url-len-max"])
FMAX = ENV["fork-count-max"]
ABSENT = object()

# ring_test.clj:44-94: [join-id transaction-id count destination timeout-destination] expect-success?
FORK_ROWS = [
    ((None, None, CNT, DEST, TDEST), False), ((None, TID, CNT, DEST, TDEST), False), ((JID, None, CNT, DEST, TDEST), False),
    (("", "", CNT, DEST, TDEST), False), (("", TID, CNT, DEST, TDEST), False), ((JID, "", CNT, DEST, TDEST), False),
    ((1, TID, CNT, DEST, TDEST), False), ((JID, 2, CNT, DEST, TDEST), False), ((1, 2, CNT, DEST, TDEST), False),
    ((JID, TID, None, DEST, TDEST), False), ((JID, TID, "", DEST, TDEST), False), ((JID, TID, -CNT, DEST, TDEST), False),
    ((JID, TID, CNT + 0.5, DEST, TDEST), False), ((JID, TID, 0, DEST, TDEST), False), ((JID, TID, "A", DEST, TDEST), False),
    ((JID, TID, FMAX + 1, DEST, TDEST), False),
    (("a~b", TID, CNT, DEST, TDEST), False), (("a:b", TID, CNT, DEST, TDEST), False), ((JID, "a~b", CNT, DEST, TDEST), False),
    ((JID, "a:b", CNT, DEST, TDEST), False), (("a~b", "a:b", CNT, DEST, TDEST), False), (("a:b", "a~b", CNT, DEST, TDEST), False),
    ((JID, TID, CNT, "", ""), False), ((JID, TID, CNT, DEST, ""), False), ((JID, TID, CNT, "", TDEST), False),
    ((JID, TID, CNT, None, None), False), ((JID, TID, CNT, DEST, None), False), ((JID, TID, CNT, None, TDEST), False),
    ((JID, TID, CNT, "http", "http"), False), ((JID, TID, CNT, DEST, "http"), False), ((JID, TID, CNT, "http", TDEST), False),
    ((JID, TID, CNT, "http://", "http://"), False), ((JID, TID, CNT, DEST, "http://"), False),
    ((JID, TID, CNT, "http://", TDEST), False),
    ((JID, TID, CNT, "file", "file"), False), ((JID, TID, CNT, DEST, "file"), False), ((JID, TID, CNT, "file", TDEST), False),
    ((JID, TID, CNT, "file://something", "file://something"), False), ((JID, TID, CNT, DEST, "file://something"), False),
    ((JID, TID, CNT, "file://something", TDEST), False),
    ((LARGE_JID, TID, CNT, DEST, TDEST), False), ((JID, LARGE_TID, CNT, DEST, TDEST), False),
    ((JID, TID, CNT, LARGE_DEST, TDEST), False), ((JID, TID, CNT, DEST, LARGE_DEST), False),
    (("&&", "**", CNT, DEST, TDEST), True), (("&&", TID, CNT, DEST, TDEST), True), ((JID, "**", CNT, DEST, TDEST), True),
    ((JID, TID, FMAX, DEST, TDEST), True),
    ((JID, TID, CNT, "http://somewhere", "http://somewhere"), True),
    ((JID, TID, CNT, "https://somewhere", "https://somewhere"), True),
    ((JID, TID, CNT, DEST, TDEST), True),
]


@pytest.fixture()
def svc(oracle_lib):
    o = oracle_lib.Oracle([], max_count=999, payload_stride=256)
    clock = [1_700_000_000_000]
    s = ForkJoinService(oracle_lib.OracleStore(o), lambda: clock[0], lambda url, body: {"status": 201}, env=ENV)
    yield s
    o.close()


def fork_request(row):
    jid, tid, cnt, dest, tdest = row
    return {"join-id": jid, "transaction-id": tid, "count": cnt, "destination": dest, "timeout-destination": tdest}


@pytest.mark.parametrize("row,ok", FORK_ROWS)
def test_fork_interface_rows(svc, row, ok):
    st, body = svc.fork(fork_request(row))
    if not ok:
        assert st == 400, (row, st, body)
        return
    assert st == 201
    st2, status = svc.status(body)                       # generic-fork-test, ring_test.clj:135-147
    assert st2 == 200
    assert status["joins"] == [None] * row[2]
    assert {k: v for k, v in status.items() if k != "joins"} == fork_request(row)


def test_join_interface_rows(svc):
    # ring_test.clj:99-117
    st, bid = svc.fork(fork_request(("a02", "b02", CNT, DEST, TDEST)))
    assert st == 201
    mid, msg = CNT // 2, "msg02"
    rows = [(("XXX-XXX", mid, msg), 404), ((123456789, mid, msg), 400), (("a:b", mid, msg), 400), ((None, mid, msg), 400),
            ((bid, None, msg), 400), ((large(ENV["barrier-id-len-max"] + 1), mid, msg), 400),
            ((bid, mid, large(ENV["join-message-len-max"] + 1)), 400), ((bid, -CNT, msg), 400), ((bid, 1.5, msg), 400),
            ((bid, CNT, msg), 400), ((bid, "A", msg), 400), ((bid, mid, None), 400)]
    for (b, p, m), want in rows:
        st, body = svc.join({"barrier-id": b, "position": p, "message": m})
        assert st == want, ((b if not isinstance(b, str) or len(b) < 40 else b[:10], p), st, body)
        st2, status = svc.status(bid)
        assert st2 == 200 and status["joins"] == [None] * CNT     # failed joins leave the barrier clean
    st, _ = svc.join({"barrier-id": bid, "position": mid, "message": ""})   # "Empty Message" is accepted
    assert st == 202
    st2, status = svc.status(bid)
    assert [None if j is None else j["message"] for j in status["joins"]] == [None] * mid + [""] + [None] * (CNT - mid - 1)


def test_status_interface_rows(svc):
    # ring_test.clj:119-126, 276-279
    st, bid = svc.fork(fork_request(("a03", "b03", CNT, DEST, TDEST)))
    assert svc.status("XXX-XXX")[0] == 404
    assert svc.status("123456789")[0] == 404
    assert svc.status(None)[0] == 400
    assert svc.status(bid)[0] == 200
    assert svc.status(large(ENV["barrier-id-len-max"] + 1))[0] == 400


def test_max_wait_limits(svc):
    # store.clj:27 (join: strictly below join-max-wait) and store.clj:68 (fork: up to fork-max-wait)
    base = fork_request(("a04", "b04", 2, DEST, TDEST))
    assert svc.fork({**base, "max-wait": ENV["fork-max-wait"]})[0] == 201
    assert svc.fork({**base, "max-wait": ENV["fork-max-wait"] + 1})[0] == 400
    assert svc.fork({**base, "max-wait": 0})[0] == 400
    j = {"barrier-id": "a04~b04", "position": 0, "message": "m"}
    assert svc.join({**j, "max-wait": ENV["join-max-wait"] - 1})[0] == 202
    assert svc.join({**j, "max-wait": ENV["join-max-wait"]})[0] == 400
    assert svc.join({**j, "max-wait": 0})[0] == 400


def test_message_limit_counts_characters_like_the_reference():
    """store.clj:24-26 limits (count message): UTF-16 code units of the String, not bytes.  A message of exactly
    the limit in three-byte characters (786432 UTF-8 bytes) is legal; one character more is not; a character
    beyond the BMP counts as two."""
    from distributed_join_amd.store import DEFAULT_ENV, clojure_count, engine_message_bytes, validate_join
    n = DEFAULT_ENV["join-message-len-max"]
    ok = {"barrier-id": "a~b", "position": 0, "message": "雪" * n}
    assert len(ok["message"].encode()) == 3 * n == engine_message_bytes()
    assert validate_join(ok) is None
    assert validate_join({**ok, "message": "雪" * (n + 1)}) is not None
    assert clojure_count("a😀") == 3
    assert validate_join({**ok, "message": "😀" * (n // 2)}) is None
    assert validate_join({**ok, "message": "😀" * (n // 2) + "a"}) is not None
