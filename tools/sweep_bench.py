#!/usr/bin/env python3
"""What a sweep costs on a large, mostly quiet table: capacity 16 M barriers (32 M slots, 256 MB of queue words),
1 M live barriers, 64 K of them completed by each join batch.  Times the pop kernel (and gather, ack) with
sweeps that scan the queue words and with sweeps that drain the ready list; then the same table with barriers that
time out instead (65 536 and 1024 per sweep, spread over the whole table).
Usage: python tools/sweep_bench.py [log2 slots, default 25]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_join_amd import capi            # noqa: E402
from distributed_join_amd.engine import Engine   # noqa: E402

T0 = 1_700_000_000_000


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    cap = 1 << (lg - 1)
    nb, due, c = 1 << 20, 1 << 16, 2
    e = Engine(capacity_barriers=cap, arena_bytes=nb * 256 + (64 << 20), max_count=c, payload_stride=64, max_batch=nb)
    dev = e.device
    fps = torch.from_numpy(capi.fingerprint_many([f"q{b:x}~t" for b in range(nb)]).view(np.int64)).to(dev)
    forks = torch.empty((nb, 4), dtype=torch.int64, device=dev)
    forks[:, 0], forks[:, 1], forks[:, 2], forks[:, 3] = fps[:, 0], fps[:, 1], c, T0 + 30 * 86_400_000
    st = e.empty(nb * 4)
    pay = torch.zeros((due * c, 64), dtype=torch.uint8, device=dev)
    conts, lens, out = e.empty((due + 1024) * 64), e.empty((due + 1024) * c * 4), e.empty((due + 1024) * c * 64)
    res = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for mode, name in ((0, "scan"), (1, "ready list")):
        e.set_sweep_path(mode)
        e.fork_batch_dev(nb, forks, st)
        e.sync()
        acc = {}
        now = T0
        for r in range(12):
            lo = (r * due) % nb
            recs = torch.empty((due * c, 4), dtype=torch.int64, device=dev)
            idx = torch.arange(lo, lo + due, device=dev).repeat_interleave(c)
            recs[:, 0], recs[:, 1] = fps[idx, 0], fps[idx, 1]
            recs[:, 2] = torch.arange(due * c, device=dev) % c | (64 << 32)
            recs[:, 3] = 0
            torch.cuda.synchronize()
            now += 1
            if r >= 2:
                e.profile_enable(True)
            e.join_batch_dev(due * c, recs, pay, 64, now, st)
            e.sweep_dev(now, due + 1024, (due + 1024) * c, conts, lens, None, out, res)
            e.ack_batch_dev(due + 1024, res, conts, None, now, st)
            e.sync()
            assert int(res[0]) == due, int(res[0])
            if r >= 2:
                for k, v in e.profile_read().items():
                    acc.setdefault(k, []).append(v)
                e.profile_enable(False)
            # fork them again so the table keeps 1 M live barriers
            e.fork_batch_dev(due, forks[lo:lo + due], st)
        print(f"{1 << lg} slots, {nb} live barriers, {due} due per sweep, {name}: " +
              ", ".join(f"{k} {1e3 * sum(v) / len(v):.1f} us" for k, v in acc.items() if k in ("join_resolve", "pop", "gather", "ack")))
        # empty the table for the next mode
        e.sweep_dev(T0 + 40 * 86_400_000, nb + 1024, (nb + 1024) * c, e.empty((nb + 1024) * 64), e.empty((nb + 1024) * c * 4), None,
                    e.empty((nb + 1024) * c * 64), res)
        e.sync()
    # second scenario: nothing completes, barriers time out.  1 M live barriers whose deadlines are spread evenly over
    # SPAN sweeps, so every sweep pops nb / SPAN of them from all over the table (and acks: they are then forked again
    # with a deadline SPAN sweeps ahead, which keeps the table and the rate of expiry steady)
    for span in (16, 1024):
        for mode, name in ((0, "scan"), (1, "ready list + region bounds")):
            e.set_sweep_path(mode)
            now = T0 + 50 * 86_400_000 + (span * 1000 if mode else 0)
            per = nb // span
            order = torch.randperm(nb, device=dev)
            forks[:, 3] = now + 1 + (torch.arange(nb, device=dev) // per)[torch.argsort(order)]
            torch.cuda.synchronize()              # the engine's stream does not wait for torch's
            e.fork_batch_dev(nb, forks, st)
            e.sync()
            acc = {}
            cap = per + 1024
            for r in range(12):
                now += 1
                if r >= 2:
                    e.profile_enable(True)
                e.sweep_dev(now, cap, cap * c, conts, lens, None, out, res)
                e.ack_batch_dev(cap, res, conts, None, now, st)
                e.sync()
                if int(res[0]) != per:
                    cc = conts.view(torch.int64).reshape(-1, 8)[:int(res[0])]
                    print("  sweep", r, "popped", int(res[0]), "want", per, "deadline - now:", torch.unique(cc[:, 2] - now, return_counts=True))
                    raise SystemExit(1)
                if r >= 2:
                    for k, v in e.profile_read().items():
                        acc.setdefault(k, []).append(v)
                    e.profile_enable(False)
                # the same barriers again, due span sweeps from now
                idx = (forks[:, 3] == now).nonzero().flatten()
                again = forks[idx].clone()
                again[:, 3] = now + span
                forks[idx, 3] = now + span
                torch.cuda.synchronize()
                e.fork_batch_dev(len(idx), again, st)
            print(f"{1 << lg} slots, {nb} live barriers, {per} time out per sweep, {name}: " +
                  ", ".join(f"{k} {1e3 * sum(v) / len(v):.1f} us" for k, v in acc.items() if k in ("pop", "gather", "ack")))
            e.sweep_dev(now + 40 * 86_400_000, nb + 1024, (nb + 1024) * c, e.empty((nb + 1024) * 64), e.empty((nb + 1024) * c * 4), None,
                        e.empty((nb + 1024) * c * 64), res)
            e.ack_batch_dev(nb + 1024, res, e.empty((nb + 1024) * 64), None, now, st)
            e.sync()
    e.close()


if __name__ == "__main__":
    main()
