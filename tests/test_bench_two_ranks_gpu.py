"""bench.py's N>1 flow (routing two batches ahead, fork routing during a job's last step, the per-kernel
leg on every rank) rehearsed with two ranks that share the one GPU of the box; the collectives go over
gloo because RCCL refuses two ranks on one device.  Checks the bookkeeping, not the speed."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, STEPS = 16384, 16


def worker(rank, port, out_dir, extra=()):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--gpus", "2", "--steps", str(STEPS), "--warmup", "8", "--barriers", str(NB),
                "--rehearse-on-one-gpu", "--no-cpu-baseline", "--min-timed-ms", "1"] + list(extra)
    sys.stdout = open(os.path.join(out_dir, f"out_{rank}.txt"), "w")
    import bench
    bench.main()
    sys.stdout.flush()


def test_bench_with_two_ranks_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    port = 33500 + os.getpid() % 2000
    mp.spawn(worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    lines = [ln for ln in open(tmp_path / "out_0.txt").read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    assert not [ln for ln in open(tmp_path / "out_1.txt").read().splitlines() if ln.startswith("{")]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == STEPS and out["scaling"] == "weak"
    # every rank generates NB barriers per job; 16 steps = 2 jobs; every barrier fires exactly once
    assert out["timed_steps"] == out["repeats"] * STEPS
    assert out["continuations_fired"] == 2 * NB * (out["timed_steps"] // 8)
    assert out["value"] > 0 and out["roofline"]["kernel"] in out["kernels"]


def test_bench_with_a_hot_key_and_small_receive_buffers(tmp_path):
    """Half of every rank's joins on one barrier, receive buffers of 3/4 batch: the exchange of every batch takes
    several rounds on both ranks; the run must end (no rank waits in a collective the other never enters) and
    every other barrier must still fire exactly once."""
    import torch.multiprocessing as mp
    port = 35500 + os.getpid() % 2000
    mp.spawn(worker, args=(port, str(tmp_path), ("--hot-frac", "0.5", "--recv-cap", str(NB * 3 // 4), "--no-profile")),
             nprocs=2, join=True)
    lines = [ln for ln in open(tmp_path / "out_0.txt").read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    # barriers that lost a position to the hot key never complete; the others fire once per job
    assert 0 < out["continuations_fired"] < 2 * NB * (out["timed_steps"] // 8)


def one_rank_worker(rank, port, out_dir, extra):
    os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--route", "--steps", str(STEPS), "--warmup", "8", "--barriers", str(NB),
                "--no-cpu-baseline", "--min-timed-ms", "1", "--config3", "0", "--no-profile"] + list(extra)
    sys.stdout = open(os.path.join(out_dir, "out_0.txt"), "w")
    import bench
    bench.main()
    sys.stdout.flush()


@pytest.mark.parametrize("extra", [(), ("--recv-cap", str(NB * 3 // 8))], ids=["whole", "in rounds"])
def test_bench_routed_over_rccl_with_one_rank(tmp_path, extra):
    """The routed flow over the real RCCL backend (one rank: the only form a one-GPU box allows): partition, bucket
    sizes and rows on the router's streams, two batches ahead, also with receive buffers that make every batch take
    three rounds.  Every barrier must fire exactly once per job."""
    import torch.multiprocessing as mp
    port = 37500 + os.getpid() % 2000
    mp.spawn(one_rank_worker, args=(port, str(tmp_path), extra), nprocs=1, join=True)
    lines = [ln for ln in open(tmp_path / "out_0.txt").read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and "partition" in out["config"]["sharding"]
    assert out["continuations_fired"] == NB * (out["timed_steps"] // 8)
