"""What the join path delivers when the caller's records and messages live in HOST memory: pinned buffers,
copies to the device on a second stream (double-buffered), dj_join_batch + sweep + ack on the engine's
stream, statuses copied back.  The bench's `value` never includes this; it is the PCIe-inclusive rate.
usage: host_ingest_bench.py [barriers [count [payload]]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from distributed_join_amd import capi  # noqa: E402
from distributed_join_amd.engine import Engine  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    c = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda", 0)
    ids = [f"j{b:x}~t{b:x}" for b in range(nb)]
    fps = capi.fingerprint_many(ids).view(np.int64)
    rng = np.random.default_rng(1)
    perm = rng.permutation(nb * c)
    recs = np.empty((nb * c, 4), dtype=np.int64)
    recs[:, 0], recs[:, 1] = fps[perm // c, 0], fps[perm // c, 1]
    recs[:, 2] = (perm % c) | (P << 32)
    recs[:, 3] = 0
    h_recs = torch.from_numpy(recs).pin_memory()
    h_pay = torch.randint(0, 256, (nb * c, P), dtype=torch.uint8).pin_memory()
    h_status = torch.empty(nb * c, dtype=torch.int32).pin_memory()
    fork = torch.empty((nb, 4), dtype=torch.int64)
    fork[:, 0], fork[:, 1] = torch.from_numpy(fps[:, 0].copy()), torch.from_numpy(fps[:, 1].copy())
    fork[:, 2], fork[:, 3] = c, 1_700_000_000_000 + 3_600_000
    d_fork = fork.to(dev)
    slab = 64 * (1 + (8 * (1 << (c - 1).bit_length()) + 63) // 64) + (1 << (c - 1).bit_length()) * ((P + 63) // 64 * 64)
    e = Engine(capacity_barriers=nb, arena_bytes=nb * (slab + 64) + (64 << 20), max_count=c, payload_stride=(P + 63) // 64 * 64,
               max_batch=nb)
    es = e.torch_stream()
    copy_stream = torch.cuda.Stream(device=dev)
    d_recs = [torch.empty((nb, 4), dtype=torch.int64, device=dev) for _ in range(2)]
    d_pay = [torch.empty((nb, P), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_st = [torch.empty(nb, dtype=torch.int32, device=dev) for _ in range(2)]
    arrived = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    cap_c = nb
    d_conts, d_lens, d_out, d_res, d_ack = e.empty(cap_c * 64), e.empty(cap_c * c * 4), e.empty(cap_c * c * P), e.empty(16), e.empty(cap_c * 4)
    d_status0 = e.empty(nb * 4)
    now = 1_700_000_000_000

    def job(timed):
        nonlocal now
        e.fork_batch_dev(nb, d_fork, d_status0)
        for k in range(c):
            s = k & 1
            with torch.cuda.stream(copy_stream):
                if k >= 2:
                    copy_stream.wait_event(consumed[s])          # the engine has applied the batch this side held
                d_recs[s].copy_(h_recs[k * nb:(k + 1) * nb], non_blocking=True)
                d_pay[s].copy_(h_pay[k * nb:(k + 1) * nb], non_blocking=True)
                arrived[s].record(copy_stream)
            es.wait_event(arrived[s])
            now += 1
            e.join_batch_dev(nb, d_recs[s], d_pay[s], P, now, d_st[s])
            with torch.cuda.stream(es):
                h_status[k * nb:(k + 1) * nb].copy_(d_st[s], non_blocking=True)
                consumed[s].record(es)
            e.sweep_dev(now, cap_c, cap_c * c, d_conts, d_lens, None, d_out, d_res)
            e.ack_batch_dev(cap_c, d_res, d_conts, None, now, d_ack)
        e.sync()
        torch.cuda.synchronize()

    job(False)
    assert int((h_status == capi.DJ_JOIN_COMPLETE).sum()) == nb
    t0 = time.perf_counter()
    jobs = 3
    for _ in range(jobs):
        job(True)
    dt = time.perf_counter() - t0
    ev = jobs * nb * c
    gb = ev * (32 + P + 4) / 1e9
    print(f"{nb} barriers x count {c}, {P} B messages from pinned host memory: {ev / dt / 1e9:.3f} G join events/s, "
          f"{gb / dt:.1f} GB/s over PCIe (records + messages in, statuses out), {dt / (jobs * c) * 1e3:.3f} ms per batch")
    # torch's pinned-memory allocator notes the streams a block was used on and touches them again when the block
    # is freed: everything that was used on the engine's stream goes before the engine (and its stream) does
    del h_status, h_recs, h_pay, d_st, d_recs, d_pay, arrived, consumed
    import gc
    gc.collect()
    torch.cuda.synchronize()
    e.close()


if __name__ == "__main__":
    main()
