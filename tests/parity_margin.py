#!/usr/bin/env python3
"""Print how far the GPU results sit from the stated tolerances (fraction of tolerance used), per model/integrator.
Run on the GPU box: python tests/parity_margin.py  (test infrastructure: uses the CPU oracle)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # oracle: test infrastructure
import oracle_lib as ol  # noqa: E402
from phnn_mpc_amd.engine import RolloutEngine  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


for name in ol.MODELS:
    g, w = ol.load_golden(name), ol.load_weights(name)
    eng, m64 = RolloutEngine(w), ol.OracleModel(w, "f64")
    rng = np.random.default_rng(99)
    n = eng.n
    B, H = 512, 100
    x0 = (rng.uniform(-1, 1, size=(B, n)) * np.array([1.0, 0.3, 0.5, 0.5][:n])).astype(np.float32)
    U = rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * float(g["u_max"]) * 0.5
    cost = ol.cost_from_golden(g)
    for integ in ("euler", "rk4"):
        ref = m64.rollout(x0, U, cost, integ, float(g["dt"]), nthreads=8)
        c, gu = eng.rollout_cost_grad(x0, U, cost, integ, float(g["dt"]))
        _, tr = eng.rollout_cost(x0, U, cost, integ, float(g["dt"]), want_traj=True)
        ce = np.abs(npy(c) / ref["cost"] - 1).max() / 1e-5
        te = np.abs(npy(tr) - ref["traj"]).max()
        gmax = np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True)
        ge = (np.abs(npy(gu) - ref["grad_u"]) / gmax).max() / 1e-4
        print(f"{name:20s} {integ:5s} B={B} H={H}: cost {ce:5.2f} x tol(1e-5)   traj abs {te:.2e}   grad {ge:5.2f} x tol(1e-4)")
