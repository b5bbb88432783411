#!/usr/bin/env python3
"""Golden vectors for pHNN_Canonical with the general MassMatrixNetwork, produced by running the
REFERENCE (src/mass_matrix.py:15-216 through src/pHNN_canonical.py:67-86, config model.mass_matrix.type in
{'constant', 'diagonal', 'full'}).  Build container only; nothing of the reference's source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mass.py

Per type <t>: 'w/<t>/<key>' seeded weights (the M_net output layer, which the reference initialises to zero, is given
seeded non-zero values so that M depends on q), '<t>/fwd_*', '<t>/vjp_*', '<t>/roll_<integ>_*' (cost, gradients w.r.t.
controls and x0) and '<t>/pt_g.*' parameter gradients (float64).
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

from pHNN_canonical import pHNN_Canonical  # noqa: E402
import integrators  # noqa: E402

torch.set_num_threads(1)


def build(mass_type, seed):
    cfg = yaml.safe_load(open("cartpole_mpc_config.yaml"))
    cfg["model"]["mass_matrix"] = {"type": mass_type, "hidden_sizes": [64, 64], "activation": "nn.Tanh", "init_scale": 1.0}
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as tf:
        yaml.safe_dump(cfg, tf)
    torch.manual_seed(seed)
    m = pHNN_Canonical(tf.name)
    os.unlink(tf.name)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        m.R_diag_raw.copy_(torch.tensor([0.10, -0.40, 0.70, 0.25]))
        if mass_type == "constant":
            m.M_net.L_tril.copy_(torch.tensor([[0.9, 0.7], [0.35, 1.2]]))  # upper entry is ignored by torch.tril
        else:
            last = m.M_net.mlp[-1]
            last.weight.copy_(0.3 * torch.randn(last.weight.shape, generator=g))
            last.bias.add_(0.2 * torch.randn(last.bias.shape, generator=g))
    return m


def block(name, model, out, seed):
    rng = np.random.default_rng(seed)
    n, dt = 4, 0.02
    for k, v in model.state_dict().items():
        out[f"w/{name}/{k}"] = v.detach().numpy().copy()
    xlo = np.array([-1.0, -0.3, -0.5, -0.5])
    N = 64
    x = rng.uniform(xlo, -xlo, size=(N, n)).astype(np.float32)
    u = rng.uniform(-5, 5, size=(N, 1)).astype(np.float32)
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
    B, H = 6, 40
    x0 = rng.uniform(xlo, -xlo, size=(B, n)).astype(np.float32)
    U = rng.uniform(-16, 16, size=(B, H, 1)).astype(np.float32)
    out[f"{name}/roll_x0"], out[f"{name}/roll_U"] = x0, U
    Q = torch.diag(torch.tensor([10.0, 200.0, 1.0, 10.0], dtype=torch.float64))
    for integ in ("euler", "rk4"):
        y0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
        Ur = torch.tensor(U, dtype=torch.float64, requires_grad=True)
        Uc = torch.clamp(Ur, -15.0, 15.0)
        traj = integrators.rollout_trajectory_differentiable(mm, y0, Uc, dt, integ)
        cost = torch.einsum("bti,ij,btj->b", traj, Q, traj) + float(np.float32(0.01)) * (Uc ** 2).sum(dim=(1, 2))
        gu, gx = torch.autograd.grad(cost.sum(), [Ur, y0])
        out[f"{name}/roll_{integ}_traj"], out[f"{name}/roll_{integ}_cost"] = traj.detach().numpy(), cost.detach().numpy()
        out[f"{name}/roll_{integ}_gu"], out[f"{name}/roll_{integ}_gx0"] = gu.numpy(), gx.numpy()


def main():
    out = {}
    for i, t in enumerate(("constant", "diagonal", "full")):
        block(t, build(t, 31 + i), out, 3101 + i)
    np.savez(os.path.join(OUT, "golden_mass.npz"), **out)
    print("wrote golden_mass.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
