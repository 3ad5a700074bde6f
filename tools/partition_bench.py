"""Times dj_partition_joins (the stable partition by owner rank in front of the all-to-all) alone on one GPU.

    python tools/partition_bench.py [events log2 = 20] [ranks = 8] [pitch = 64]

Prints the average time of one call (its three kernels) and the bytes it has to move: every record (32 B) and
message row read and written once, the records' owner words read once more by the counting pass, 4 B of origin
index written per event.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_join_amd import capi  # noqa: E402
from distributed_join_amd.engine import Engine  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    pitch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    n = 1 << lg
    e = Engine(capacity_barriers=1024, arena_bytes=1 << 22, max_count=8, payload_stride=64, max_batch=n)
    dev = e.device
    rng = np.random.default_rng(1)
    recs = np.zeros(n, dtype=capi.JOIN_REC)
    recs["fp_lo"] = rng.integers(2, 1 << 62, n, dtype=np.uint64)
    recs["fp_hi"] = rng.integers(1, 1 << 62, n, dtype=np.uint64)
    d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
    d_pay = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, device=dev)
    d_ro, d_po = torch.empty_like(d_recs), torch.empty_like(d_pay)
    d_origin = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(64, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    reps = 50
    for with_pay in (True, False):
        def call():
            e.partition_joins_dev(n, ranks, d_recs, d_pay if with_pay else None, pitch, d_ro, d_po if with_pay else None,
                                  d_origin, d_counts, stream_ptr=st.cuda_stream)
        for _ in range(5):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(reps):
            call()
        b.record(st)
        st.synchronize()
        us = a.elapsed_time(b) * 1000 / reps
        moved = n * (2 * 32 + 8 + 4 + (2 * pitch if with_pay else 0))
        print(f"{n} events, {ranks} ranks, {'%d-byte rows' % pitch if with_pay else 'records only'}: {us:.1f} us per call, "
              f"{moved / 1e6:.0f} MB to move, {moved / us / 1e3:.0f} GB/s")
    e.close()


if __name__ == "__main__":
    main()
