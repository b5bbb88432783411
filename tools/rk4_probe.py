#!/usr/bin/env python3
"""BASELINE config 5 (ODEFunc(2,1), classic RK4, H=200, B=65536): K1 + K2 on the stage-tape stash.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (K1 writes the stash, K2 reads it)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd import _capi
from phnn_mpc_amd.engine import RolloutEngine
name = os.environ.get("MODEL", "odefunc_pendulum")
with np.load(os.path.join(ROOT, "tests", "golden", f"weights_{name}.npz")) as z:
    w = {k: z[k] for k in z.files}
eng = RolloutEngine(w)
n = eng.n
B, H = int(os.environ.get("B", 65536)), int(os.environ.get("H", 200 if n == 2 else 50))
dt = 0.05 if n == 2 else 0.02
cost = (_capi.make_cost(2, 1, [10.0, 1.0], [0.01], None, -2.0, 2.0) if n == 2 else
        _capi.make_cost(4, 1, [10.0, 200.0, 1.0, 10.0], [0.01], None, -15.0, 15.0))
rng = np.random.default_rng(0)
scale = np.array([np.pi, 1.0]) if n == 2 else np.array([1.0, 0.3, 0.5, 0.5])
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, n)) * scale).astype(np.float32), device="cuda")
U = torch.tensor(rng.uniform(-2, 2, size=(B, H, 1)).astype(np.float32), device="cuda")
eng.use_stash = os.environ.get("STASH", "1") == "1"
ws = {}
for _ in range(2):
    eng.rollout_cost_grad(x0, U, cost, "rk4", dt, workspace=ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.rollout_cost_grad(x0, U, cost, "rk4", dt, workspace=ws)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 5
print(f"{name} rk4 B={B} H={H} stash={eng.use_stash}: {t*1e3:.2f} ms per K1+K2, {B/t/1e6:.2f} M rollouts+grads/s, "
      f"stash {eng.workspace_bytes(B, H, 'rk4')/1e9 if eng.use_stash else 0:.1f} GB")
