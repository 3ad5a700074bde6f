"""The router behind the C ABI (dj_router_*, csrc/dj_router.cpp) with several ranks.  RCCL refuses two ranks of one
communicator on one device, so the ranks are threads of one program (tests/rccl_standin/router_ranks.cpp), each with an
engine of its own on the GPU, connected by a stand-in for librccl (tests/rccl_standin/rccl_standin.cpp) that only that
program's process gets to see (its directory is put on the dynamic linker's path for the child alone).  The router's
own code runs unchanged: partition, all-gather of the bucket sizes, the plan, grouped sends and receives per round,
buffer sets in flight.  A hot barrier makes its owner take every batch in several rounds."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "rccl_standin")


@pytest.mark.parametrize("ranks,n,max_recv,hot", [(2, 30000, 8192, 50), (4, 20000, 8192, 40), (8, 10000, 4096, 60), (3, 5000, 100000, 0)])
def test_native_router_with_several_ranks(ranks, n, max_recv, hot):
    exe, lib = os.path.join(DIR, "router_ranks"), os.path.join(DIR, "librccl.so.1")
    assert os.path.exists(exe) and os.path.exists(lib), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, str(ranks), str(n), str(max_recv), str(hot)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "every rank received its records in order and every record's answer came back" in p.stdout
    rounds = int(p.stdout.split(":")[-1].split("rounds")[0])
    assert (rounds > 1) == (hot > 0), p.stdout


@pytest.mark.parametrize("ranks,nb,n,max_recv,hot", [(2, 3000, 20000, 8192, 50), (3, 1500, 9000, 100000, 0), (4, 1500, 12000, 4096, 40),
                                                      (8, 800, 8000, 4096, 60)])
def test_native_router_brings_every_answer_home(ranks, nb, n, max_recv, hot):
    """Routed mode end to end behind the C ABI (tests/rccl_standin/router_answers.cpp): forks and joins go to their
    owners' engines, every record's answer returns to the rank that submitted it (dj_route_return / dj_route_statuses),
    and each rank's answers, in its own submission order, are one oracle store's on the concatenated stream — with a
    hot barrier that forces rounds, ids nobody forked (404) and positions beyond the count (400)."""
    exe, lib = os.path.join(DIR, "router_answers"), os.path.join(DIR, "librccl.so.1")
    assert os.path.exists(exe) and os.path.exists(lib), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, str(ranks), str(nb), str(n), str(max_recv), str(hot)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "every rank got one store's answers" in p.stdout
    rounds = int(p.stdout.split("due at the end,")[-1].split("rounds")[0])
    assert (rounds > 1) == (hot > 0), p.stdout


@pytest.mark.parametrize("ranks,threads,barriers", [(2, 4, 600), (8, 2, 800)])
def test_node_front_answers_every_blocking_caller(ranks, threads, barriers):
    """The per-request front of a sharded node (dj_batcher_create_routed; tests/rccl_standin/node_front.cpp): request
    threads on every rank call dj_batcher_fork_wait / dj_batcher_join_wait for barriers owned by any rank and each gets
    the answer one oracle store gives for its barrier's request sequence (404 before the fork, 201, WAITING / COMPLETE,
    400, replays, re-forks); a barrier joined by all threads at once completes exactly once; the ranks' publishers
    together pop what the one store pops."""
    exe, lib = os.path.join(DIR, "node_front"), os.path.join(DIR, "librccl.so.1")
    assert os.path.exists(exe) and os.path.exists(lib), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, str(ranks), str(threads), str(barriers)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "every caller got one store's answer" in p.stdout
