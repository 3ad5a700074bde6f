"""Longer randomized parity runs than the test suite holds: engine (through the C ABI) against the CPU oracle
on seeded op-logs with thousands of barriers and batches of tens of thousands of events, in every
configuration the suite covers (short and long chains, multi-word bitmaps, variable lengths, messages longer
than a row, both long-chain kernel modes, with and without the fused link pass, both sweep forms, re-forks to
other counts under failing publishes).  Any difference raises.  usage: soak_parity.py [seeds [scale]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py  # noqa: E402
from tests import test_engine_parity  # noqa: E402
from tests.test_engine_parity import make_pair, run_oplog  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle_py.build()
    cases = [
        # name, barriers, count range, stride, rounds, batch, extra run_oplog arguments, engine arguments
        ("small counts", 4000 * scale, (1, 9), 64, 20, 30000 * scale, dict(unknown=50), dict()),
        ("hot barrier", 2000 * scale, (2, 65), 64, 15, 40000 * scale, dict(hot=7), dict()),
        ("hot barrier, parallel kernels", 2000 * scale, (2, 65), 64, 15, 40000 * scale, dict(hot=7), dict(hot_mode=1)),
        ("many hot barriers", 100, (2, 65), 64, 8, 60000 * scale, dict(), dict()),
        ("many hot barriers, parallel kernels", 100, (2, 65), 64, 8, 60000 * scale, dict(), dict(hot_mode=1)),
        ("ten events per barrier and batch", 6000, (2, 17), 64, 6, 60000 * scale, dict(), dict()),
        ("ten events per barrier and batch, wide replay pinned", 3000, (2, 17), 64, 6, 30000 * scale, dict(), dict(hot_mode=3)),
        ("large counts", 300 * scale, (60, 1000), 64, 10, 60000 * scale, dict(hot=11), dict()),
        ("large counts, parallel kernels", 300 * scale, (60, 1000), 64, 10, 60000 * scale, dict(hot=11), dict(hot_mode=1)),
        ("variable lengths", 3000 * scale, (1, 20), 256, 12, 20000 * scale, dict(var_len=True), dict()),
        ("long messages", 1500 * scale, (1, 12), 64, 10, 8000 * scale, dict(long_to=4096), dict(max_message=4096)),
        ("small counts, separate copy pass", 4000 * scale, (1, 9), 64, 12, 30000 * scale, dict(unknown=50), dict(hot_mode=4)),
        ("re-forks to other counts, every second publish fails", 1500 * scale, (1, 13), 64, 25, 12000 * scale,
         dict(recount=(1, 12), fail_every=2), dict()),
        ("the same across the bitmap forms", 600 * scale, (40, 91), 64, 20, 20000 * scale,
         dict(recount=(40, 90), fail_every=2), dict()),
        ("the same with counts 2..300 and a hot barrier", 400 * scale, (2, 301), 64, 16, 30000 * scale,
         dict(recount=(2, 300), fail_every=2, hot=5), dict()),
        ("the same with long messages", 800 * scale, (1, 91), 64, 12, 8000 * scale,
         dict(long_to=4096, recount=(1, 90), fail_every=2), dict(max_message=4096)),
        # the bucketed join passes pinned (by themselves they start at 196608 events per batch)
        ("small counts, bucketed passes", 4000 * scale, (1, 9), 64, 20, 30000 * scale, dict(unknown=50), dict(hot_mode=16)),
        ("hot barrier, bucketed passes", 2000 * scale, (2, 65), 64, 15, 40000 * scale, dict(hot=7), dict(hot_mode=16)),
        ("ten events per barrier and batch, bucketed passes", 6000, (2, 17), 64, 6, 60000 * scale, dict(), dict(hot_mode=16)),
        ("large counts, bucketed passes", 300 * scale, (60, 1000), 64, 10, 60000 * scale, dict(hot=11), dict(hot_mode=16)),
        ("variable lengths, bucketed passes", 3000 * scale, (1, 20), 256, 12, 20000 * scale, dict(var_len=True), dict(hot_mode=16)),
        ("re-forks to other counts, every second publish fails, bucketed passes", 1500 * scale, (1, 13), 64, 25, 12000 * scale,
         dict(recount=(1, 12), fail_every=2), dict(hot_mode=16)),
        ("counts 2..300 with re-forks and a hot barrier, bucketed passes", 400 * scale, (2, 301), 64, 16, 30000 * scale,
         dict(recount=(2, 300), fail_every=2, hot=5), dict(hot_mode=16)),
    ]
    for seed in range(seeds):
        for name, nb, (lo, hi), stride, rounds, batch, kw, ekw in cases:
            rng = np.random.default_rng(1000 + seed)
            test_engine_parity.SWEEP_MODE[0] = seed % 2          # sweeps that scan / sweeps that drain the ready list
            ids = [f"s{seed}b{b:x}~t{b:x}" for b in range(nb)] + [f"ghost{k}~x" for k in range(kw.get("unknown", 0))]
            counts = rng.integers(lo, hi, nb)
            t0 = time.time()
            e, o, d = make_pair(oracle_py, ids, stride, capacity=nb + 64, arena=2 << 30, max_batch=1 << 18, **ekw)
            try:
                run_oplog(d, rng, nb, counts, stride, rounds=rounds, batch=batch, **kw)
            finally:
                e.close()
            print(f"seed {seed} {name:40s} ok  {time.time() - t0:6.1f} s", flush=True)
    print("soak ok")


if __name__ == "__main__":
    main()
