"""bench.py --gpus N must produce N ranks by itself: the parent starts torch.distributed.run
as a child before anything touches the GPU and relays rank 0's JSON line.  Rehearsed here on CPU: gloo backend, the CPU
oracle behind the engine protocol (tests/oracle_engine.py) instead of RolloutEngine -- the line is marked rehearsal.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    env["PYTHONPATH"] = HERE + os.pathsep + ROOT + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rehearsal-engine", "oracle_engine:bench_rehearsal_engine",
           "--batch", "48", "--horizon", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--global-batch", "37"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus2_spawns_two_ranks_and_reports_them():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rehearsal"] is True
    assert line["config"]["global_batch"] == 96 and line["config"]["batch_per_gpu"] == 48
    assert line["config"]["collective_backend"] == "gloo" and "gloo all-gather" in line["config"]["workload"]
    assert "RCCL" not in line["config"]["workload"]
    assert line["scaling"] == "weak" and line["value"] > 0
    assert line["metric"] == "pHNN-MPC rollouts+grads/sec, cartpole H=6 batch=48"
    assert line["strong_scaling"]["global_batch"] == 37  # ragged split over the two ranks


def test_gpus1_runs_in_process_with_unchanged_line_shape():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "strong_scaling" not in line and line["config"]["collective_backend"] is None
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line


def test_world_size_mismatch_is_an_error_not_a_warning():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
