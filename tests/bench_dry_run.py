"""TEST WRAPPER around bench.py: runs its real step loop (public driver iteration, sharded engine, all-reduce, barrier +
max-over-ranks timing, self-launch of the ranks) on the CPU over gloo with an engine DOUBLE from tests/ — the line is marked
dry_run and carries no measurement.  bench.py itself has no test hook and imports nothing from tests/: this script hands it the
engine factory and its own path as the launcher target, so that the ranks `bench.py --gpus N` starts are this script again.

    BENCH_DRY_RUN_DOUBLE=module:Class python tests/bench_dry_run.py [bench.py arguments]"""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import bench  # noqa: E402


def main():
    mod, cls = os.environ.get("BENCH_DRY_RUN_DOUBLE", "oracle_engine:OracleEngine").split(":")

    def factory():
        return getattr(importlib.import_module(mod), cls)()
    bench.main(engine_factory=factory, launcher=os.path.abspath(__file__))


if __name__ == "__main__":
    main()
