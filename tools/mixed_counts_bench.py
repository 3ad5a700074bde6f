#!/usr/bin/env python3
"""Fork / join / sweep / ack timing with MIXED barrier sizes (counts ~ Zipf(1) on [2, 64], BASELINE
config #3's shape): the slab allocator sees several size classes inside every wave and block, unlike
bench.py's uniform count.  Prints per-kernel HIP-event times of one job."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from distributed_join_amd import capi
    from distributed_join_amd.engine import Engine
    # usage: mixed_counts_bench.py [barriers [fixed-count [hot-path-mode [events-per-batch]]]]
    nb, P = (int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20), 64
    fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(3)
    k = np.arange(2, 65)
    p = 1.0 / k
    counts = rng.choice(k, size=nb, p=p / p.sum()).astype(np.int64) if not fixed else np.full(nb, fixed, dtype=np.int64)
    max_count = int(counts.max())
    total = int(counts.sum())
    ids = [f"m{b:x}~t" for b in range(nb)]
    fps = capi.fingerprint_many(ids)
    dev = torch.device("cuda", 0)
    slab = (1 << (max_count - 1).bit_length()) * (8 + P) + 256
    e = Engine(capacity_barriers=nb, arena_bytes=nb * slab + (256 << 20), max_count=max_count, payload_stride=P,
               max_batch=1 << 21)
    if len(sys.argv) > 3:
        e.set_hot_path(int(sys.argv[3]))
    frecs = np.zeros(nb, dtype=capi.FORK_REC)
    frecs["fp_lo"], frecs["fp_hi"], frecs["count"], frecs["deadline_ms"] = fps[:, 0], fps[:, 1], counts, 1_700_000_000_000 + 10**9
    d_f = torch.from_numpy(frecs.view(np.uint8).reshape(-1).copy()).to(dev)
    # all events of the job, shuffled, in batches of 2^20
    b_of = np.repeat(np.arange(nb), counts)
    p_of = np.concatenate([np.arange(c) for c in counts])
    perm = rng.permutation(total)
    b_of, p_of = b_of[perm], p_of[perm]
    jrecs = np.zeros(total, dtype=capi.JOIN_REC)
    jrecs["fp_lo"], jrecs["fp_hi"], jrecs["position"], jrecs["len"] = fps[b_of, 0], fps[b_of, 1], p_of, P
    d_j = torch.from_numpy(jrecs.view(np.uint8).reshape(-1).copy()).to(dev)
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
    d_pay = torch.randint(0, 256, (batch, P), dtype=torch.uint8, device=dev)
    d_st = e.empty((1 << 21) * 4)
    cap_c, cap_p = nb, total
    d_conts, d_lens, d_out, d_res, d_ack = e.empty(cap_c * 64), e.empty(cap_p * 4), e.empty(cap_p * P), e.empty(16), e.empty(cap_c * 4)
    torch.cuda.synchronize()
    for job in range(3):
        acc = {}
        if job == 2:
            e.profile_enable(True)
        e.fork_batch_dev(nb, d_f, d_st)
        now = 1_700_000_000_000 + job * 1000
        for lo in range(0, total, batch):
            n = min(batch, total - lo)
            now += 1
            e.join_batch_dev(n, d_j[lo * 32:], d_pay, P, now, d_st)
            e.sweep_dev(now, cap_c, cap_p, d_conts, d_lens, None, d_out, d_res)
            e.ack_batch_dev(cap_c, d_res, d_conts, None, now, d_ack)
            if job == 2:
                for name, ms in e.profile_read().items():
                    acc.setdefault(name, []).append(ms)
        e.sync()
        assert e.stats()["live"] == 0
    print(f"{nb} barriers, {total} events in batches of {batch}, counts {'Zipf(1) on [2, 64]' if not fixed else fixed}, {P} B messages; per launch (ms):")
    for name, v in acc.items():
        print(f"  {name:14s} avg {sum(v) / len(v):.4f}  x{len(v)}  total {sum(v):.3f}")
    e.close()


if __name__ == "__main__":
    main()
