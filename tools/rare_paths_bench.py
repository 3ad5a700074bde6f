#!/usr/bin/env python3
"""Timing of the paths bench.py never takes, at 1 M barriers: a sweep where every publish FAILS
(ack with ok = 0: retry write-back), the re-sweep, and a mass recover of popped-but-unacked
barriers.  Prints HIP-event times of the kernels involved."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine
    nb, c, P = 1 << 20, 2, 64
    T0 = 1_700_000_000_000
    ids = [f"r{b:x}~t" for b in range(nb)]
    fps = capi.fingerprint_many(ids)
    dev = torch.device("cuda", 0)
    e = Engine(capacity_barriers=nb, arena_bytes=nb * 320 + (64 << 20), max_count=c, payload_stride=P, max_batch=1 << 21,
               check_publish_delay_ms=1000)
    frecs = np.zeros(nb, dtype=capi.FORK_REC)
    frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[:, 0], fps[:, 1], c, T0 + 10**8
    d_f = torch.from_numpy(frecs.view(np.uint8).reshape(-1).copy()).to(dev)
    jrecs = np.zeros(nb * c, dtype=capi.JOIN_REC)
    b_of = np.repeat(np.arange(nb), c)
    jrecs["fp_lo"], jrecs["fp_hi"], jrecs["position"], jrecs["len"] = fps[b_of, 0], fps[b_of, 1], np.tile(np.arange(c), nb), P
    d_j = torch.from_numpy(jrecs.view(np.uint8).reshape(-1).copy()).to(dev)
    d_pay = torch.randint(0, 256, (nb * c, P), dtype=torch.uint8, device=dev)
    d_st = e.empty(nb * c * 4)
    d_conts, d_lens, d_out, d_res, d_ack = e.empty(nb * 64), e.empty(nb * c * 4), e.empty(nb * c * P), e.empty(16), e.empty(nb * 4)
    d_fail = torch.zeros(nb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e.fork_batch_dev(nb, d_f, d_st)
    e.join_batch_dev(nb * c, d_j, d_pay, P, T0 + 1, d_st)
    e.profile_enable(True)
    res = np.zeros(1, dtype=capi.SWEEP_RESULT)
    out = []

    def note(tag):
        for name, ms in e.profile_read().items():
            out.append(f"  {tag:34s} {name:10s} {ms:.4f} ms")

    e.sweep_dev(T0 + 2, nb, nb * c, d_conts, d_lens, None, d_out, d_res, res)
    assert int(res[0]["n_conts"]) == nb
    note("sweep of 1 M complete barriers")
    for attempt in range(3):                                    # publish fails: retry counters 1, 2, 3
        e.ack_batch_dev(nb, None, d_conts, d_fail, T0 + 3 + attempt, d_ack)
        note(f"ack, all failed (retry {attempt + 1})")
        assert (d_ack[:nb * 4].view(torch.int32) == capi.DJ_ACK_RETRY).all()
        e.sweep_dev(T0 + 3 + attempt, nb, nb * c, d_conts, d_lens, None, d_out, d_res, res)
        assert int(res[0]["n_conts"]) == nb
    # popped and never acked: the recover pass finds all of them after the delay and retries once more -> exhausted
    d_rc, d_rs = e.empty(nb * 64), e.empty(nb * 4)
    e.recover_dev(T0 + 5000, nb, d_rc, d_rs, d_res, res)
    note("recover of 1 M unacked barriers")
    assert int(res[0]["n_conts"]) == nb
    assert (d_rs[:nb * 4].view(torch.int32) == capi.DJ_ACK_DROPPED_EXHAUSTED).all()
    assert e.stats()["live"] == 0
    print("\n".join(out))
    e.close()


if __name__ == "__main__":
    main()
