#!/usr/bin/env python3
"""Per-step latency and fixed cost of the split-tile kernels for ONE tile (B <= 16): K1 + K2 at several horizons.
Run under `rocprofv3 --kernel-trace --stats`: the kernel durations against H give slope (per step) and intercept
(launch + weight-image staging)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd import _capi
from phnn_mpc_amd.engine import RolloutEngine
with np.load(os.path.join(ROOT, "tests", "golden", "weights_phnn_cartpole.npz")) as z:
    w = {k: z[k] for k in z.files}
cost = _capi.make_cost(4, 1, [10.0, 200.0, 1.0, 10.0], [0.01], None, -15.0, 15.0)
eng = RolloutEngine(w, split=os.environ.get("SPLIT", "always"))
x0 = torch.tensor([[0.0, 0.1, 0.0, 0.0]], device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for H in (1, 5, 10, 20, 40, 80):
    u = torch.zeros(1, H, 1, device="cuda")
    ws = {}
    for _ in range(5):
        eng.rollout_cost_grad(x0, u, cost, "euler", 0.02, workspace=ws)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(100):
        eng.rollout_cost_grad(x0, u, cost, "euler", 0.02, workspace=ws)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"H={H:3d}: K1+K2 {ev[0].elapsed_time(ev[1]) * 10:.1f} us per call", flush=True)
