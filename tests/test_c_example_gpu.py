"""examples/fork_join_smoke.c: the reference's fork-join-test.sh (one barrier of count 3, three joins, the publisher's
poll, the ack) plus a barrier that times out, driven from plain C through include/dj_engine.h alone."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_example_runs_the_smoke_script():
    exe = os.path.join(ROOT, "examples", "fork_join_smoke")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l and "amdgpu.ids" not in l]
    assert lines == ['smoke~1 -> normal destination: "part zero" "part one" "part two"',
                     'late~1 -> timeout destination: nil "only part" nil', "ok"], lines
