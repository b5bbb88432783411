#!/usr/bin/env python3
"""Write cost and grad_u of a fixed seeded batch to an .npz (to compare two builds of the library bit for bit)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phnn_mpc_amd.engine import RolloutEngine
from phnn_mpc_amd import _capi
w = dict(np.load(os.path.join(ROOT, "tests/golden/weights_phnn_cartpole.npz")))
eng = RolloutEngine(w)
rng = np.random.default_rng(1234)
B, H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 50
x0 = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
U = rng.uniform(-8, 8, size=(B, H, 1)).astype(np.float32)
cost = _capi.make_cost(4, 1, np.diag([10.0, 100.0, 1.0, 10.0]), 0.01, np.zeros(4), -10.0, 10.0, None, None, 1000.0)
c, g = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02)
c2, g2 = eng.rollout_cost_grad(x0[8:8 + 1000], U[8:8 + 1000], cost, "euler", 0.02)
np.savez(sys.argv[1], c=c.cpu().numpy(), g=g.cpu().numpy(), c2=c2.cpu().numpy(), g2=g2.cpu().numpy())
g, g2 = g.cpu().numpy(), g2.cpu().numpy()
d = np.abs(g[8:1008] - g2)
print("shifted-chunk: max abs diff", d.max(), "rel to max|g|", d.max() / np.abs(g).max(), "n differing", (d > 0).sum())
