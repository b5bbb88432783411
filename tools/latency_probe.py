#!/usr/bin/env python3
"""Single-plant latency of one MPC solve (BASELINE config 1: pHNN, H=20, 30 Adam iterations): eager launches vs one HIP
graph, whole-tile vs split-tile kernels.  Under `rocprofv3 --kernel-trace --stats` the kernel durations show how much
of the wall time is kernels and how much is launch gaps."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd import _capi
from phnn_mpc_amd.engine import RolloutEngine
from phnn_mpc_amd.solver import GraphedSolve, shooting_solve

with np.load(os.path.join(ROOT, "tests", "golden", "weights_phnn_cartpole.npz")) as z:
    w = {k: z[k] for k in z.files}
cost = _capi.make_cost(4, 1, [10.0, 200.0, 1.0, 10.0], [0.01], None, -15.0, 15.0)
x0 = torch.tensor([[0.0, 0.1, 0.0, 0.0]], device="cuda")
u0 = torch.zeros(1, 20, 1, device="cuda")
for mode in ("never", "always"):
    eng = RolloutEngine(w, split=mode)
    for label, solve in (("eager", shooting_solve), ("graph", GraphedSolve(eng))):
        args = (eng, x0, u0, cost, "euler", 0.02, 0.015, 30)
        kw = dict(u_min=-15.0, u_max=15.0, record_costs=False)
        for _ in range(3):
            solve(*args, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            solve(*args, **kw)
        torch.cuda.synchronize()
        print(f"split={mode:6s} {label}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per solve", flush=True)
# the same solve through phnn_solve (one C call enqueues the 90 launches)
eng = RolloutEngine(w)
ws = {}
for _ in range(3):
    eng.solve(x0, u0, cost, "euler", 0.02, lr=0.015, iters=30, record_costs=False, workspace=ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.solve(x0, u0, cost, "euler", 0.02, lr=0.015, iters=30, record_costs=False, workspace=ws)
torch.cuda.synchronize()
print(f"phnn_solve            : {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per solve", flush=True)
