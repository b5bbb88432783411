#!/usr/bin/env python3
"""Golden vectors for models with TWO (golden_m2.npz) and THREE / FOUR (golden_m34.npz) control inputs,
produced by running the REFERENCE.

Build container only (reference mounted read-only at /root/reference).  Constructs the reference's own pHNN (fixed G and
learned G_net, src/pHNN.py:31-38,86-92) and pHNN_Canonical with input_dim = 2 from temporary YAML files (the shipped
cart-pole configuration with input_dim / G_value / G_mlp changed), seeded, and stores weights, inputs and outputs in
golden_m2.npz.  Nothing of the reference's source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_m2.py

Sets per model <name> in {phnn_m2_fix, phnn_m2_gnet, canonical_m2}: 'w/<name>/<state_dict key>' weights,
'<name>/fwd_*' model(x,u), '<name>/vjp_*' VJPs (ubar is (N,2)), '<name>/roll_<integ>_*' rollouts with cost
x^T Q x + u^T R u (full 2x2 R), gradients w.r.t. controls (B,H,2) and x0, '<name>/pt_g.*' parameter gradients.
"""
import copy
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.append(os.path.join(REF, "src"))
os.chdir(REF)

from pHNN import pHNN  # noqa: E402
from pHNN_canonical import pHNN_Canonical  # noqa: E402
import integrators  # noqa: E402

torch.set_num_threads(1)


G4 = [[0.0, 0.3, -0.2, 0.1], [0.2, 0.0, 0.4, -0.3], [1.0, -0.5, 0.25, 0.6], [0.0, 0.8, -0.7, 0.35]]
R4 = np.array([[0.02, 0.004, 0.0, -0.001], [-0.002, 0.05, 0.003, 0.0], [0.001, 0.0, 0.03, 0.002], [0.0, -0.004, 0.001, 0.04]], np.float32)


def build(cls, seed, fixed_G, m=2):
    cfg = yaml.safe_load(open("cartpole_mpc_config.yaml"))
    cfg["model"]["input_dim"] = m
    cfg["model"]["fixed_G"] = fixed_G
    cfg["model"]["G_value"] = [row[:m] for row in G4]
    cfg["model"]["G_mlp"] = {"activation": "nn.Tanh", "bias": True, "dropout": 0.0, "hidden_sizes": [128], "layer_norm": False}
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as tf:
        yaml.safe_dump(cfg, tf)
    torch.manual_seed(seed)
    m = cls(tf.name)
    os.unlink(tf.name)
    return m


def block(name, model, out, seed, m=2):
    rng = np.random.default_rng(seed)
    n, dt = 4, 0.02
    for k, v in model.state_dict().items():
        out[f"w/{name}/{k}"] = v.detach().numpy().copy()
    xlo = np.array([-1.0, -0.3, -0.5, -0.5])
    N = 64
    x = rng.uniform(xlo, -xlo, size=(N, n)).astype(np.float32)
    u = rng.uniform(-4, 4, size=(N, m)).astype(np.float32)
    lam = rng.normal(size=(N, n)).astype(np.float32)
    Hbar = rng.normal(size=(N,)).astype(np.float32)
    out[f"{name}/x"], out[f"{name}/u"], out[f"{name}/lam"], out[f"{name}/Hbar"] = x, u, lam, Hbar
    mm = copy.deepcopy(model).double()
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    res = mm(xt, ut)
    out[f"{name}/fwd_dx"], out[f"{name}/fwd_H"] = res[0].detach().numpy(), res[1].detach().numpy()
    xb, ub = torch.autograd.grad((res[0] * torch.tensor(lam, dtype=torch.float64)).sum(), [xt, ut], retain_graph=True)
    out[f"{name}/vjp_xbar"], out[f"{name}/vjp_ubar"] = xb.numpy(), ub.numpy()
    loss = (res[0] * torch.tensor(lam, dtype=torch.float64)).sum() + (res[1] * torch.tensor(Hbar, dtype=torch.float64)).sum()
    names, params = zip(*mm.named_parameters())
    for nm, p, g in zip(names, params, torch.autograd.grad(loss, params, allow_unused=True)):
        out[f"{name}/pt_g.{nm}"] = (torch.zeros_like(p) if g is None else g).numpy()
    # rollouts with a full 2x2 control weight
    B, H = 6, 30
    x0 = rng.uniform(xlo, -xlo, size=(B, n)).astype(np.float32)
    U = rng.uniform(-12, 12, size=(B, H, m)).astype(np.float32)
    U[0, 1, 0], U[0, 1, 1], U[1, 4, 1], U[2, 0, 0] = 13.0, -11.0, 10.0, -10.0  # outside / on the clamp bounds +-10
    if m > 2:
        U[3, 2, m - 1] = -14.0
    Q = np.diag([10.0, 200.0, 1.0, 10.0]).astype(np.float32)
    R = np.ascontiguousarray(R4[:m, :m])
    out[f"{name}/roll_x0"], out[f"{name}/roll_U"], out[f"{name}/Q"], out[f"{name}/R"] = x0, U, Q, R
    for integ in ("euler", "rk4"):
        y0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
        Ur = torch.tensor(U, dtype=torch.float64, requires_grad=True)
        Uc = torch.clamp(Ur, -10.0, 10.0)
        traj = integrators.rollout_trajectory_differentiable(mm, y0, Uc, dt, integ)
        Qt, Rt = torch.tensor(Q).double(), torch.tensor(R).double()
        cost = torch.einsum("bti,ij,btj->b", traj, Qt, traj) + torch.einsum("bti,ij,btj->b", Uc, Rt, Uc)
        gu, gx = torch.autograd.grad(cost.sum(), [Ur, y0])
        out[f"{name}/roll_{integ}_traj"], out[f"{name}/roll_{integ}_cost"] = traj.detach().numpy(), cost.detach().numpy()
        out[f"{name}/roll_{integ}_gu"], out[f"{name}/roll_{integ}_gx0"] = gu.numpy(), gx.numpy()


def main():
    out = {}
    block("phnn_m2_fix", build(pHNN, 21, True), out, 2101)
    block("phnn_m2_gnet", build(pHNN, 22, False), out, 2102)
    can = build(pHNN_Canonical, 23, True)
    with torch.no_grad():
        can.M_net.log_a.fill_(0.25)
        can.M_net.b.fill_(0.30)
        can.M_net.log_c.fill_(-0.15)
        can.R_diag_raw.copy_(torch.tensor([0.2, -0.3, 0.6, 0.15]))
    block("canonical_m2", can, out, 2103)
    np.savez(os.path.join(OUT, "golden_m2.npz"), **out)
    print("wrote golden_m2.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB")
    # three and four control inputs (golden_m34.npz): one model of each family
    out = {}
    block("phnn_m3_fix", build(pHNN, 31, True, 3), out, 3101, 3)
    block("phnn_m4_gnet", build(pHNN, 32, False, 4), out, 3102, 4)
    can = build(pHNN_Canonical, 33, True, 3)
    with torch.no_grad():
        can.M_net.log_a.fill_(0.25)
        can.M_net.b.fill_(0.30)
        can.M_net.log_c.fill_(-0.15)
        can.R_diag_raw.copy_(torch.tensor([0.2, -0.3, 0.6, 0.15]))
    block("canonical_m3", can, out, 3103, 3)
    np.savez(os.path.join(OUT, "golden_m34.npz"), **out)
    print("wrote golden_m34.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
