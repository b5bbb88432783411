#!/usr/bin/env python3
"""Parity margins on the TRAINED cart-pole weights, per matmul mode (GPU box; test infrastructure: CPU oracle + goldens).
(1) the reference's golden rollouts: error vs the reference's float64, in units of the reference's own float32-vs-float64
    deviation on the same rollout (tests/test_trained_weights.py explains why that is the yardstick there);
(2) MPC-like rollouts (x0 near upright, |u| <= 1, H = 20 / 50): well-conditioned, so the stated tolerances apply:
    fraction of tolerance used, against the float64 oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from phnn_mpc_amd.engine import RolloutEngine  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


g, w = ol.load_golden(ol.TRAINED), ol.load_weights(ol.TRAINED)
m64 = ol.OracleModel(w, "f64")
cost = ol.cost_from_golden(g)
for mode in ("f16x2", "bf16x3", "f32"):
    eng = RolloutEngine(w, matmul=mode)
    worst_c = worst_g = 0.0
    for integ in ("euler", "rk4"):
        for B, H in ol.ROLL_CASES:
            key = f"roll_{integ}_B{B}_H{H}"
            fc, fg, gmax = ol.trained_rollout_floor(g, key)
            c, gu = eng.rollout_cost_grad(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
            ec = np.abs(npy(c) / g[key + "_cost_f64"] - 1)
            eg = np.abs(npy(gu) - g[key + "_gu_f64"]).max(axis=(1, 2)) / gmax
            rc, rg = (ec / (fc + 2.5e-6)).max(), (eg / (fg + 2.5e-5)).max()
            worst_c, worst_g = max(worst_c, rc), max(worst_g, rg)
            print(f"  {mode:7s} {key:22s} cost err {ec.max():.2e} (reference f32: {fc.max():.2e})  grad err {eg.max():.2e} "
                  f"(reference f32: {fg.max():.2e})  ratio to floor+tol/4: cost {rc:5.2f} grad {rg:5.2f}")
    # MPC-like rollouts
    rng = np.random.default_rng(5)
    res = []
    for H in (20, 50):
        B = 512
        x0 = (rng.uniform(-1, 1, size=(B, 4)) * np.array([0.3, 0.1, 0.2, 0.2])).astype(np.float32)
        U = rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32)
        ref = m64.rollout(x0, U, cost, "euler", 0.02, nthreads=8)
        c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02)
        ce = np.abs(npy(c) / ref["cost"] - 1).max() / 1e-5
        gmax = np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True)
        ge = (np.abs(npy(gu) - ref["grad_u"]) / gmax).max() / 1e-4
        res.append(f"H={H}: cost {ce:.2f} x tol, grad {ge:.2f} x tol (max |grad| {gmax.max():.3g})")
    print(f"{mode:7s} golden rollouts: worst ratio cost {worst_c:.2f} grad {worst_g:.2f} | MPC-like Euler rollouts: " + "; ".join(res), flush=True)
