#!/usr/bin/env python3
"""Training-side pass at scale (pHNN cart-pole, Euler, H=50, B=65536): forward rollout, adjoint + records, record
reduction.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd.engine import RolloutEngine
with np.load(os.path.join(ROOT, "tests", "golden", "weights_phnn_cartpole.npz")) as z:
    w = {k: z[k] for k in z.files}
eng = RolloutEngine(w)
rng = np.random.default_rng(0)
B, H = int(os.environ.get("B", 65536)), int(os.environ.get("H", 50))
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
U = torch.tensor(rng.uniform(-5, 5, size=(B, H, 1)).astype(np.float32), device="cuda")
tb = torch.randn(B, H + 1, 4, device="cuda")
TAPES = os.environ.get("TAPES", "1") == "1"  # K1 keeps its tapes for the adjoint + reduction (default) / the adjoint recomputes
for _ in range(2):
    traj = eng.rollout_trajectory(x0, U, "euler", 0.02, tapes=TAPES)
    eng.rollout_wgrad(x0, U, traj, "euler", 0.02, traj_bar=tb, tape_token=eng.tape_token)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    traj = eng.rollout_trajectory(x0, U, "euler", 0.02, tapes=TAPES)
    eng.rollout_wgrad(x0, U, traj, "euler", 0.02, traj_bar=tb, tape_token=eng.tape_token)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"B={B} H={H}: {dt*1e3:.2f} ms per training pass, {B/dt/1e6:.2f} M rollouts+wgrads/s, workspace {eng._wg_ws.numel()/2**30:.2f} GiB, tapes={TAPES}")
