"""CPU: `python bench.py --gpus 2` outside a distributed launcher starts its own ranks (torch.distributed.run on
127.0.0.1) and relays rank 0's JSON line as the last line of stdout — the form in which the driver runs the scaling
bench.  Dry run: tests/bench_dry_run.py imports bench.py and hands its main() the oracle-backed engine double — bench.py
itself carries no test hook, no environment switch and no import from tests/; the ranks sit on the CPU over gloo, the line is
marked dry_run and carries no measurement, but every rank walks the real step loop
(public driver iteration, sharded engine, all-reduce, barrier + max-over-ranks timing)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    script = "bench.py" if (env_extra or {}).get("REAL_BENCH") else os.path.join("tests", "bench_dry_run.py")
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, env=env,
                          timeout=600, cwd=ROOT)


@pytest.mark.parametrize("config", ["c2", "c3"])
def test_bench_self_launches_its_ranks(config):
    p = _run(["--gpus", "2", "--rows", "3000", "--steps", "2", "--warmup", "1", "--config", config, "--no-cpu-baseline"],
             {"BENCH_DRY_RUN_DOUBLE": "oracle_engine:OracleEngine"})
    assert p.returncode == 0, p.stderr[-2000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["dry_run"] is True and d["value"] is None
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["ms_per_step"] > 0
    K, D = d["config"]["K"], d["config"]["Dz"]
    assert d["allreduce_bytes_per_step"] == 8 * (K * (1 + D + D * D) + 4)
    pr = d["ms_per_step_per_rank"]
    assert len(pr["all"]) == 2 and pr["max"] == max(pr["all"]) and abs(pr["max"] - d["ms_per_step"]) < 1e-9
    assert "rank order" in d["allreduce_route"]
    if config == "c2":
        assert d["elbo_first_last"][1] >= d["elbo_first_last"][0]      # the bound rises over the sharded iterations


def test_bench_propagates_a_failing_rank():
    p = _run(["--gpus", "2", "--rows", "1000", "--steps", "1", "--warmup", "0"], {"BENCH_DRY_RUN_DOUBLE": "no_such_module:X"})
    assert p.returncode != 0


def test_bench_py_carries_no_test_hook():
    """The measurement script has no dry-run argument, no environment switch for one and no reference to tests/."""
    p = _run(["--gpus", "2", "--rows", "1000", "--dry-run-engine", "oracle_engine:OracleEngine"], {"REAL_BENCH": "1"})
    assert p.returncode != 0 and "unrecognized arguments" in p.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "DRY_RUN_ENGINE" not in src and "oracle_engine" not in src and '"tests"' not in src and "'tests'" not in src


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "2", "--rows", "1000"], {"RANK": "0", "WORLD_SIZE": "1", "REAL_BENCH": "1"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
