#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (2, 3, 5) on one GPU; bench.py carries the headline
configuration (4, per GPU).  Prints one JSON line per configuration.  Run on the GPU box:
    python tools/bench_configs.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd import _capi  # noqa: E402
from phnn_mpc_amd.engine import RolloutEngine  # noqa: E402
from phnn_mpc_amd.solver import shooting_solve  # noqa: E402


def weights(name):
    with np.load(os.path.join(ROOT, "tests", "golden", f"weights_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


def inputs(n, B, H, amp):
    rx, ru = np.random.default_rng(1234), np.random.default_rng(5678)
    scale = np.array([1.0, 0.3, 0.5, 0.5]) if n == 4 else np.array([np.pi, 1.0])
    return ((rx.uniform(-1, 1, size=(B, n)) * scale).astype(np.float32), ru.uniform(-amp, amp, size=(B, H, 1)).astype(np.float32))


PEAK_F32_MFMA, PEAK_16BIT_MFMA, PEAK_HBM = 157.3e12, 2.5e15, 8.0e12  # MI355X_MICROARCH.md
SPLIT_PRODUCTS = {"f16x2": 3, "bf16x3": 6}


def roofline(eng, flop_per_rollout, B, t, stash_bytes=0):
    """Roofline object of one row: algorithmic FLOPs / wall time against the matrix peak of the
    engine's product mode; `hbm_stash_frac` = the K1 -> K2 stash written and read once / time against the HBM peak
    (the bound of the RK4 stash path)."""
    mm = eng.matmul_mode
    peak = PEAK_F32_MFMA if mm == "f32" else PEAK_16BIT_MFMA / SPLIT_PRODUCTS[mm]
    ach = flop_per_rollout * B / t
    r = {"bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
         "frac": round(ach / peak, 4), "matmul": mm, "flop_per_rollout": flop_per_rollout}
    if stash_bytes:
        r["stash_gb"] = round(stash_bytes / 1e9, 2)
        r["hbm_stash_frac"] = round(2 * stash_bytes / t / PEAK_HBM, 3)
    return r


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    dev = "cuda:0"
    cart = _capi.make_cost(4, 1, [10.0, 200.0, 1.0, 10.0], [0.01], None, -15.0, 15.0)
    pend = _capi.make_cost(2, 1, [10.0, 1.0], [0.01], None, -2.0, 2.0)
    out = []
    # config 1 (plumbing): MPCController.compute_control, pHNN, H=20, 30 Adam iterations, ONE plant, and 4096 plants
    import yaml
    from phnn_mpc_amd.models import pHNN
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    cfgp = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
    m = pHNN(cfgp)
    m.load_state_dict({k: torch.tensor(v) for k, v in weights("phnn_cartpole").items()})
    ctl = create_mpc_from_config(m, yaml.safe_load(open(cfgp)))
    st = np.array([0.0, 0.1, 0.0, 0.0], np.float32)
    t = timed(lambda: ctl.compute_control(st), reps=10, warm=2)
    out.append({"config": "1: MPCController.compute_control, pHNN, H=20, 30 Adam iterations, B=1", "ms": round(t * 1e3, 3)})
    S = np.tile(st, (4096, 1)) + np.random.default_rng(0).normal(size=(4096, 4)).astype(np.float32) * 0.05
    t = timed(lambda: ctl.compute_control_batch(S), reps=5, warm=1)
    out.append({"config": "1b: compute_control_batch, same settings, 4096 plants at once", "ms": round(t * 1e3, 3),
                "controls_per_s": round(4096 / t, 1)})
    # config 2: canonical, H=50, B=4096, forward kernel only
    eng = RolloutEngine(weights("canonical_cartpole"), dev)
    x0, U = [torch.tensor(a, device=dev) for a in inputs(4, 4096, 50, 5.0)]
    for integ in ("euler", "rk4"):
        t = timed(lambda: eng.rollout_cost(x0, U, cart, integ, 0.02), reps=20)
        out.append({"config": f"2: canonical cart-pole, {integ}, H=50, B=4096, K1 only", "ms": round(t * 1e3, 4),
                    "rollouts_per_s": round(4096 / t, 1),
                    "roofline": roofline(eng, 50 * (1 if integ == "euler" else 4) * 67.9e3, 4096, t)})
    # config 3: pHNN, H=100, B=65536, 20 Adam iterations (K1+K2+K3 each)
    eng = RolloutEngine(weights("phnn_cartpole"), dev)
    x0, U = [torch.tensor(a, device=dev) for a in inputs(4, 65536, 100, 5.0)]
    U0 = torch.zeros_like(U)
    t = timed(lambda: shooting_solve(eng, x0, U0, cart, "euler", 0.02, 0.015, 20, track_best=True, u_min=-15.0,
                                     u_max=15.0, record_costs=False), reps=2, warm=1)
    out.append({"config": "3: pHNN cart-pole, euler, H=100, B=65536, 20 Adam iterations (K1+K2+K3)",
                "ms": round(t * 1e3, 3), "rollouts_grads_per_s": round(65536 * 20 / t, 1),
                "roofline": roofline(eng, 100 * (73.0e3 + 72.7e3) * 20, 65536, t)})
    # config 5: ODEFunc(2,1), classic RK4, H=200, B=65536
    eng = RolloutEngine(weights("odefunc_pendulum"), dev)
    x0, U = [torch.tensor(a, device=dev) for a in inputs(2, 65536, 200, 2.0)]
    t = timed(lambda: eng.rollout_cost(x0, U, pend, "rk4", 0.05), reps=3, warm=1)
    out.append({"config": "5: ODEFunc(2,1), rk4, H=200, B=65536, K1 only", "ms": round(t * 1e3, 3),
                "rollouts_per_s": round(65536 / t, 1), "roofline": roofline(eng, 200 * 4 * 66.8e3, 65536, t)})
    for stash in (True, False):  # K1 + K2 on the stage-tape stash (default) and with K2 recomputing
        eng.use_stash = stash
        ws = {}
        t = timed(lambda: eng.rollout_cost_grad(x0, U, pend, "rk4", 0.05, workspace=ws), reps=2, warm=1)
        nst = eng.workspace_bytes(65536, 200, "rk4") if stash else 0
        out.append({"config": "5: ODEFunc(2,1), rk4, H=200, B=65536, K1+K2" + ("" if stash else " (no stash: K2 recomputes)"),
                    "ms": round(t * 1e3, 3), "rollouts_grads_per_s": round(65536 / t, 1),
                    "roofline": roofline(eng, 106.9e6, 65536, t, nst)})
        del ws
        torch.cuda.empty_cache()
    eng.use_stash = True
    # cart-pole pHNN with RK4 at the headline batch (not a BASELINE row; the RK4 stash path on the flagship model)
    eng = RolloutEngine(weights("phnn_cartpole"), dev)
    x0, U = [torch.tensor(a, device=dev) for a in inputs(4, 65536, 50, 5.0)]
    for stash in (True, False):
        eng.use_stash = stash
        ws = {}
        t = timed(lambda: eng.rollout_cost_grad(x0, U, cart, "rk4", 0.02, workspace=ws), reps=3, warm=1)
        nst = eng.workspace_bytes(65536, 50, "rk4") if stash else 0
        out.append({"config": "4': pHNN cart-pole, rk4, H=50, B=65536, K1+K2" + ("" if stash else " (no stash: K2 recomputes)"),
                    "ms": round(t * 1e3, 3), "rollouts_grads_per_s": round(65536 / t, 1),
                    "roofline": roofline(eng, 50 * 4 * (73.0e3 + 72.7e3), 65536, t, nst)})
        del ws
        torch.cuda.empty_cache()
    eng.use_stash = True
    # training side: forward rollout + adjoint with records + record reduction -> d loss / d theta
    for name, n, H, B, dt in (("phnn_cartpole", 4, 20, 4096, 0.02), ("phnn_cartpole", 4, 50, 65536, 0.02),
                              ("canonical_cartpole", 4, 20, 4096, 0.02)):
        eng = RolloutEngine(weights(name), dev)
        x0, U = [torch.tensor(a, device=dev) for a in inputs(n, B, H, 5.0)]
        tb = torch.randn(B, H + 1, n, device=dev)

        def train_pass(tapes=True):  # what _RolloutFn does: K1 keeps its tapes for the adjoint + reduction
            traj = eng.rollout_trajectory(x0, U, "euler", dt, tapes=tapes)
            eng.rollout_wgrad(x0, U, traj, "euler", dt, traj_bar=tb, tape_token=eng.tape_token if tapes else None)

        t = timed(train_pass, reps=5, warm=2)
        t_rc = timed(lambda: train_pass(False), reps=5, warm=2)
        # algorithmic FLOPs per rollout-step: forward 73.0 k + VJP 74.2 k + the parameter outer products:
        # 2 x 2 x 128 x 128 (W2) + 2 x 16 x 128 (V2) + small = 69.9 k
        out.append({"config": f"f4: {name} training pass (rollout + parameter gradient), euler, H={H}, B={B}",
                    "ms": round(t * 1e3, 3), "rollouts_wgrads_per_s": round(B / t, 1),
                    "ms_adjoint_recomputes": round(t_rc * 1e3, 3),
                    "workspace_bytes_per_rollout_step": eng.lib.phnn_wgrad_workspace_bytes(eng.h, 16, 1, 0) // 16})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
