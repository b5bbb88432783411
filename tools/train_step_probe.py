#!/usr/bin/env python3
"""End-to-end time of ONE optimisation step of the reference's cart-pole training loop
(scripts/train_cartpole_phnn.py:112-178: Euler rollout from x_batch[:,0], position MSE + (1 - cos) angle loss +
velocity MSE + 0.01 * H(0)^2, Adam on every parameter) through this package's drop-in modules on the GPU:
fused forward launch, adjoint + record reduction in backward, torch.optim.Adam, weights re-packed before the next forward
(on the device when the parameters live there).  Two shapes: the reference's own (16 windows x 16 steps) and the bench shape (65536 x 50).
Prints where the wall time of a step goes (GPU-side kernels vs host: autograd glue, optimizer, re-pack + upload).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnn_mpc_amd.integrators import rollout_trajectory_differentiable  # noqa: E402
from phnn_mpc_amd.models import pHNN  # noqa: E402

CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
dev = torch.device("cuda:0")
with np.load(os.path.join(ROOT, "tests", "golden", "weights_phnn_cartpole.npz")) as z:
    sd = {k: torch.tensor(z[k]) for k in z.files}


PARAMS_ON = os.environ.get("PARAMS", "cuda")  # where the nn.Parameters (and so the optimizer state) live


def make():
    m = pHNN(CFG)
    m.load_state_dict(sd)
    if PARAMS_ON == "cuda":
        m = m.to(dev)  # gradients and Adam stay on the device; only the re-pack reads the parameters back
    return m.use_device(dev)


def step(model, opt, xb, ub, dt=0.02):
    opt.zero_grad()
    X, dX = rollout_trajectory_differentiable(model, xb[:, 0, :], ub[:, :-1, :], dt, "euler", return_derivatives=True)
    l_pos = torch.mean((X[:, :, 0] - xb[:, :, 0]) ** 2)
    l_theta = torch.mean(1 - torch.cos(X[:, :, 1] - xb[:, :, 1]))
    l_vel = torch.mean((X[:, :, 2:] - xb[:, :, 2:]) ** 2)
    _, H0 = model(torch.zeros(1, 4, device=dev, requires_grad=True), torch.zeros(1, 1, device=dev))
    loss = l_pos + l_theta + l_vel + 0.01 * torch.mean(H0 ** 2)
    loss.backward()
    opt.step()
    return loss


def run(B, T, reps):
    rng = np.random.default_rng(0)
    xb = torch.tensor((rng.uniform(-1, 1, size=(B, T, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device=dev)
    ub = torch.tensor(rng.uniform(-5, 5, size=(B, T, 1)).astype(np.float32), device=dev)
    model = make()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    for _ in range(3):
        step(model, opt, xb, ub)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step(model, opt, xb, ub)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / reps
    # the weight refresh alone (what an optimizer step triggers before the next forward): the path this module takes
    # (parameters on the GPU: torch.cat on the device + the one-workgroup packing kernel, phnn_update_weights_dev;
    # parameters on the CPU: host packing + upload), and the host path for comparison
    eng = model.engine
    t0 = time.perf_counter()
    for _ in range(20):
        model._repack()
    torch.cuda.synchronize()
    pack = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        eng.update_weights(model.state_dict())
    torch.cuda.synchronize()
    pack_host = (time.perf_counter() - t0) / 20
    print(f"[parameters on {PARAMS_ON}, {torch.get_num_threads()} CPU threads] training step B={B} T={T}: {per * 1e3:.3f} ms end to end "
          f"({B / per / 1e3:.1f} k windows/s), of which the weight refresh {pack * 1e3:.3f} ms (host packing + upload would be "
          f"{pack_host * 1e3:.3f} ms); loss {float(loss.detach()):.5f}", flush=True)


if __name__ == "__main__":
    if "THREADS" in os.environ:
        torch.set_num_threads(int(os.environ["THREADS"]))
    run(16, 16, 50)       # the reference's batch (cartpole_mpc_config.yaml: batch_size 16, seq_len 16)
    run(4096, 16, 20)
    run(65536, 51, 5)     # the bench shape: 65536 rollouts x 50 steps

if os.environ.get("PROFILE") == "1":  # where the host time of the small-batch step goes
    import cProfile
    import pstats
    rng = np.random.default_rng(0)
    xb = torch.tensor((rng.uniform(-1, 1, size=(16, 16, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device=dev)
    ub = torch.tensor(rng.uniform(-5, 5, size=(16, 16, 1)).astype(np.float32), device=dev)
    model = make()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    for _ in range(5):
        step(model, opt, xb, ub)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step(model, opt, xb, ub)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
