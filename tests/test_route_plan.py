"""dj_route_plan (the exchange plan both routers use): on every rank the same matrix of bucket sizes must give a
consistent set of rounds - what s sends to r in a round is what r expects from s, no receiver takes more than
its buffer holds in one round, and the rounds hand every receiver its incoming records exactly once: its own
rows first (they stay on the rank), then the other sources in rank order."""
import numpy as np

from distributed_join_amd import capi


def check(matrix, max_recv):
    w = matrix.shape[0]
    plans = {}
    rounds = None
    for rank in range(w):
        k, _, _, _ = capi.route_plan(matrix, rank, max_recv, 0)
        rounds = k if rounds is None else rounds
        assert k == rounds, "every rank must compute the same number of rounds"
        for rnd in range(rounds):
            plans[rank, rnd] = capi.route_plan(matrix, rank, max_recv, rnd)[1:]
    want_rounds = max(1, max(-(-int(matrix[:, r].sum()) // max_recv) for r in range(w)))
    assert rounds == want_rounds
    for r in range(w):
        taken = [0] * w                        # rows of each source's bucket already received by r
        for rnd in range(rounds):
            recv = plans[r, rnd][2]
            assert sum(recv) <= max_recv
            started = False
            for s in [r] + [q for q in range(w) if q != r]:      # the receiver's stream order
                so, sc, _ = plans[s, rnd]
                assert sc[r] == recv[s], "sender and receiver disagree"
                if sc[r]:
                    assert so[r] == taken[s], "pieces of a bucket must follow each other"
                    # source order inside the receiver's stream: its own rows first, then the other sources in rank order;
                    # every source ahead of s is complete before s starts
                    ahead = [] if s == r else [r] + [q for q in range(s) if q != r]
                    assert all(taken[q] == matrix[q, r] for q in ahead)
                    started = True
                taken[s] += sc[r]
            # a round is full unless it is the receiver's last
            if rnd < rounds - 1 and sum(taken) < matrix[:, r].sum():
                assert sum(recv) == max_recv or not started
        assert taken == [int(x) for x in matrix[:, r]]


def test_uniform_and_skewed_matrices():
    rng = np.random.default_rng(4)
    for w in (1, 2, 3, 8):
        for _ in range(6):
            m = rng.integers(0, 2000, (w, w)).astype(np.uint32)
            if rng.random() < 0.5:
                m[:, rng.integers(0, w)] *= 9          # a hot owner
            if rng.random() < 0.2:
                m[rng.integers(0, w), :] = 0           # a rank with nothing to send
            for cap in (700, 4096, 10**6):
                check(m, cap)
        check(rng.integers(0, 5, (w, w)).astype(np.uint32), 1)     # one record per round


def test_empty_and_bad_arguments():
    z = np.zeros((4, 4), dtype=np.uint32)
    assert capi.route_plan(z, 0, 100, 0) == (1, [0] * 4, [0] * 4, [0] * 4)
    assert capi.load().dj_route_plan(0, 0, None, 1, 0, None, None, None) == 0
