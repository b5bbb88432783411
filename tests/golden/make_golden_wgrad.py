#!/usr/bin/env python3
"""Golden vectors for the training-side path: parameter gradients, produced by running
the REFERENCE itself.  A separate script from make_golden.py so that the round-1 fixtures stay bit-identical.

Runs only in the build container (reference mounted read-only at /root/reference).  Loads the committed weight
fixtures into the reference's own nn.Modules (src/pHNN.py, src/pHNN_canonical.py, src/baseline_node.py), evaluates
them and torch.autograd on seeded inputs, and stores inputs + expected outputs in golden_wgrad.npz.  Nothing of the
reference's source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_wgrad.py

Sets (keys '<model>/<set>_<name>'; parameter gradients as 'g.<state_dict key>'):
  G14 pt_*     point-wise: loss = sum(lam * model(x,u)[0]) + sum(Hbar * model(x,u)[1]); autograd.grad w.r.t. every
               parameter (48 random points), float64 (reference cast to double).
  G15 tr_*     one optimisation step's loss and gradients of each training loop, Euler rollouts from x_batch[:,0]:
                 phnn_cartpole       scripts/train_cartpole_phnn.py:112-178  (MSE(x) + mean(1-cos dtheta) + MSE(vel)
                                     + 0.01 * H(0)^2), windows cut from data/cartpole_training_data.pt
                 phnn_pendulum       main.py:93-148  (MSE(X_pred, x) + MSE(dX_pred, dx)), the shipped trained weights
                 canonical_cartpole  scripts/train_cartpole_phnn_canonical.py:83-196 compute_integrated_loss (called
                                     as is, integrator='euler'; its inner loop steps with manual Euler either way)
               stored: inputs, loss value, gradient per parameter, and the predicted trajectory.
  G16 rw_*     generic reverse pass: random cotangents on the trajectory and on the per-step derivatives dX, Euler and
               RK4 (src/integrators.py:39-84) -> gradients w.r.t. parameters, controls and x0, float64.
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.append(os.path.join(REF, "src"))
sys.path.append(os.path.join(REF, "scripts"))
os.chdir(REF)

from pHNN import pHNN  # noqa: E402
from pHNN_canonical import pHNN_Canonical  # noqa: E402
import integrators  # noqa: E402
from baseline_node import ODEFunc  # noqa: E402

torch.set_num_threads(1)


class OdeAdapter(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, y, u):
        self.f.current_action = u
        return self.f(0.0, y), torch.zeros(y.shape[0], dtype=y.dtype)

    def named_parameters(self, *a, **k):
        return self.f.named_parameters(*a, **k)


def load_weights(name):
    with np.load(os.path.join(OUT, f"weights_{name}.npz")) as z:
        return {k: torch.tensor(z[k]) for k in z.files}


def named_grads(model, loss, out, prefix, tag):
    names, params = zip(*[(n, p) for n, p in model.named_parameters()])
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    for n, p, g in zip(names, params, grads):
        out[f"{prefix}g.{n}_{tag}"] = (torch.zeros_like(p) if g is None else g).detach().numpy()


def point_set(name, model, n, m, xlo, xhi, uamp, seed, out):
    rng = np.random.default_rng(seed)
    N = 48
    x = rng.uniform(xlo, xhi, size=(N, n)).astype(np.float32)
    u = rng.uniform(-uamp, uamp, size=(N, m)).astype(np.float32)
    lam = rng.normal(size=(N, n)).astype(np.float32)
    Hbar = rng.normal(size=(N,)).astype(np.float32)
    pre = f"{name}/pt_"
    out[pre + "x"], out[pre + "u"], out[pre + "lam"], out[pre + "Hbar"] = x, u, lam, Hbar
    for tag, dtype in (("f64", torch.float64),):
        mm = copy.deepcopy(model).to(dtype)
        res = mm(torch.tensor(x, dtype=dtype, requires_grad=True), torch.tensor(u, dtype=dtype))
        loss = (res[0] * torch.tensor(lam, dtype=dtype)).sum() + (res[1] * torch.tensor(Hbar, dtype=dtype)).sum()
        named_grads(mm, loss, out, pre, tag)


def reverse_set(name, model, n, m, dt, xlo, xhi, uamp, seed, out, B=6, H=12):
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(xlo, xhi, size=(B, n)).astype(np.float32)
    U = rng.uniform(-uamp, uamp, size=(B, H, m)).astype(np.float32)
    Wt = rng.normal(size=(B, H + 1, n)).astype(np.float32)
    Wd = rng.normal(size=(B, H, n)).astype(np.float32)
    pre = f"{name}/rw_"
    out[pre + "x0"], out[pre + "U"], out[pre + "traj_bar"], out[pre + "dx_bar"], out[pre + "dt"] = x0, U, Wt, Wd, np.float64(dt)
    for integ in ("euler", "rk4"):
        mm = copy.deepcopy(model).double()
        y = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
        Ut = torch.tensor(U, dtype=torch.float64, requires_grad=True)
        y0 = y
        traj, dxs = [y], []
        for t in range(H):
            u_t = Ut[:, t, :]
            dxs.append(mm(y, u_t)[0])  # derivative at the step's first stage, as the training loops collect dX_pred
            y = integrators.euler_step(mm, y, u_t, dt) if integ == "euler" else integrators.rk4_step(mm, y, u_t, dt)
            traj.append(y)
        traj, dX = torch.stack(traj, dim=1), torch.stack(dxs, dim=1)
        loss = (traj * torch.tensor(Wt, dtype=torch.float64)).sum() + (dX * torch.tensor(Wd, dtype=torch.float64)).sum()
        gu, gx = torch.autograd.grad(loss, [Ut, y0], retain_graph=True)
        named_grads(mm, loss, out, f"{pre}{integ}_", "f64")
        out[f"{pre}{integ}_gu_f64"], out[f"{pre}{integ}_gx0_f64"] = gu.numpy(), gx.numpy()
        out[f"{pre}{integ}_traj_f64"], out[f"{pre}{integ}_dX_f64"] = traj.detach().numpy(), dX.detach().numpy()


def main():
    out = {}
    xlo_c = np.array([-1.0, -0.3, -0.5, -0.5])
    xlo_p = np.array([-np.pi, -1.0])

    phnn = pHNN("cartpole_mpc_config.yaml")
    phnn.load_state_dict(load_weights("phnn_cartpole"))
    can = pHNN_Canonical("cartpole_mpc_config.yaml")
    can.load_state_dict(load_weights("canonical_cartpole"))
    pend = pHNN("pendulum_config.yaml")
    pend.load_state_dict(load_weights("phnn_pendulum"))
    ode = ODEFunc(2, 1)
    ode.load_state_dict(load_weights("odefunc_pendulum"))
    ode_model = OdeAdapter(ode)

    for name, model, n, dt, xlo, uamp, seed in (("phnn_cartpole", phnn, 4, 0.02, xlo_c, 5.0, 1401),
                                                ("canonical_cartpole", can, 4, 0.02, xlo_c, 5.0, 1402),
                                                ("phnn_pendulum", pend, 2, 0.05, xlo_p, 2.5, 1403),
                                                ("odefunc_pendulum", ode_model, 2, 0.05, xlo_p, 2.5, 1404)):
        point_set(name, model, n, 1, xlo, -xlo, uamp, seed, out)
        if name != "odefunc_pendulum":  # ODEFunc: point set only (its training loop is not on SURVEY 8 f4's list)
            reverse_set(name, model, n, 1, dt, 0.5 * xlo, -0.5 * xlo, uamp, seed + 100, out)

    # ------------------------------------------------------------------ G15: the training loops' own losses
    d = torch.load("data/cartpole_training_data.pt", weights_only=True)
    S, C = d["states"], d["controls"]
    rng = np.random.default_rng(15)
    B, L = 8, 12
    idx = rng.choice(S.shape[0], size=B, replace=False)
    start = rng.integers(0, S.shape[1] - L, size=B)
    xb = torch.stack([S[i, s:s + L] for i, s in zip(idx, start)]).float()
    ub = torch.stack([C[i, s:s + L] for i, s in zip(idx, start)]).float()
    out["tr_cart_x"], out["tr_cart_u"] = xb.numpy(), ub.numpy()

    # (a) scripts/train_cartpole_phnn.py:112-178 on the cart-pole pHNN
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        model = copy.deepcopy(phnn).to(dtype)
        x_batch, u_batch = xb.to(dtype), ub.to(dtype)
        dt = 0.02
        loss_fn = nn.MSELoss()
        x0_batch = x_batch[:, 0, :].requires_grad_(True)
        X_pred = [x0_batch]
        for t in range(x_batch.shape[1] - 1):
            dx, _ = model(X_pred[-1], u_batch[:, t, :])
            X_pred.append(X_pred[-1] + dt * dx)
        X_pred = torch.stack(X_pred, dim=1)
        l_pos = loss_fn(X_pred[:, :, 0], x_batch[:, :, 0])
        l_theta = torch.mean(1 - torch.cos(X_pred[:, :, 1] - x_batch[:, :, 1]))
        l_vel = loss_fn(X_pred[:, :, 2:], x_batch[:, :, 2:])
        zero_state = torch.zeros(1, 4, dtype=dtype, requires_grad=True)
        _, H_zero = model(zero_state, torch.zeros(1, 1, dtype=dtype))
        loss = 1.0 * l_pos + 1.0 * l_theta + 1.0 * l_vel + 0.01 * torch.mean(H_zero ** 2)
        out[f"phnn_cartpole/tr_loss_{tag}"] = np.float64(loss.item())
        out[f"phnn_cartpole/tr_X_{tag}"] = X_pred.detach().numpy()
        named_grads(model, loss, out, "phnn_cartpole/tr_", tag)

    # (b) scripts/train_cartpole_phnn_canonical.py:83-196, the function itself
    from train_cartpole_phnn_canonical import compute_integrated_loss
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        model = copy.deepcopy(can).to(dtype)
        loss, parts = compute_integrated_loss(model, xb.to(dtype).clone(), ub.to(dtype), 0.02,
                                              {"position": 1.0, "velocity": 0.5}, torch.device("cpu"), integrator="euler")
        out[f"canonical_cartpole/tr_loss_{tag}"] = np.float64(loss.item())
        out[f"canonical_cartpole/tr_loss_position_{tag}"] = np.float64(parts["position"])
        out[f"canonical_cartpole/tr_loss_velocity_{tag}"] = np.float64(parts["velocity_reconstruction"])
        named_grads(model, loss, out, "canonical_cartpole/tr_", tag)

    # (c) main.py:93-148 on the trained pendulum pHNN (learned G): synthetic windows of the pendulum's ranges
    rngp = np.random.default_rng(16)
    Bp, Lp = 8, 10
    xp = (rngp.uniform(-1, 1, size=(Bp, Lp, 2)) * np.array([1.5, 1.0])).astype(np.float32)
    xp = np.cumsum(0.05 * xp, axis=1).astype(np.float32) + rngp.uniform(-1, 1, size=(Bp, 1, 2)).astype(np.float32)
    up = rngp.uniform(-2, 2, size=(Bp, Lp, 1)).astype(np.float32)
    dxp = rngp.normal(size=(Bp, Lp, 2)).astype(np.float32)
    out["tr_pend_x"], out["tr_pend_u"], out["tr_pend_dx"] = xp, up, dxp
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        model = copy.deepcopy(pend).to(dtype)
        x_batch, u_batch, dx_batch = torch.tensor(xp).to(dtype), torch.tensor(up).to(dtype), torch.tensor(dxp).to(dtype)
        dt = 0.05
        loss_fn = nn.MSELoss()
        x0_batch = x_batch[:, 0, :].requires_grad_(True)
        X_pred, dX_pred = [x0_batch], []
        for t in range(x_batch.shape[1] - 1):
            dx, _ = model(X_pred[-1], u_batch[:, t, :])
            dX_pred.append(dx)
            X_pred.append(X_pred[-1] + dt * dx)
        X_pred, dX_pred = torch.stack(X_pred, dim=1), torch.stack(dX_pred, dim=1)
        loss = loss_fn(X_pred, x_batch) + loss_fn(dX_pred, dx_batch[:, 0:-1, :])
        out[f"phnn_pendulum/tr_loss_{tag}"] = np.float64(loss.item())
        named_grads(model, loss, out, "phnn_pendulum/tr_", tag)

    np.savez(os.path.join(OUT, "golden_wgrad.npz"), **out)
    print("wrote golden_wgrad.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
