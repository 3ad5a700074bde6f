"""Longer runs of the namespace-seam comparisons than the test suite holds: core.clj restated function by function
(core_unchanged.Core) over the engine's single-key entry points against the same over the oracle's command-level
functions, on seeded request streams and on direct command streams.  Any difference raises.
usage: soak_namespace.py [seeds [steps]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py  # noqa: E402
from distributed_join_amd.store import DEFAULT_ENV  # noqa: E402
from tests.test_namespace_seam import engine_view, oracle_view, run_commands, run_stream  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle_py.build()
    env = dict(DEFAULT_ENV)
    for seed in range(100, 100 + seeds):
        for name, run in (("request stream", run_stream), ("command stream", run_commands)):
            engines = []
            try:
                want = run(oracle_view(oracle_py, env, None), seed, steps)
                got = run(engine_view(env, None, engines), seed, steps)
                for k, (w, g) in enumerate(zip(want, got)):
                    if w != g:
                        raise SystemExit(f"seed {seed} {name} step {k}:\noracle {w}\nengine {g}")
            finally:
                for e in engines:
                    e.close()
            print(f"seed {seed} {name:15s} ok ({len(want)} steps)", flush=True)
    print("namespace soak ok")


if __name__ == "__main__":
    main()
