"""Stock-PyTorch CPU restatement of the batched oracle (second CPU baseline).

TEST INFRASTRUCTURE, like phnn_oracle.c: only tests/ and bench.py's cpu_baseline leg import this file; nothing under
phnn_mpc_amd/ does.  It restates, with plain torch ops and torch.autograd, what the reference computes for the
batched path (SURVEY.md 8c "batched oracle composition"):

    traj = rollout_trajectory_differentiable(model, y0, U, dt, 'euler')        src/integrators.py:192-258
    cost_b = sum_t (x_t - x*)^T Q (x_t - x*) + R sum_t u_t^2                   src/mpc_controller.py:75-114
    grad   = autograd.grad(cost.sum(), U)                                      src/mpc_controller.py:192

with pHNN.forward as in src/pHNN.py:52-100 (dH/dx by autograd.grad(create_graph=True) inside forward, R = S S^T with
S = sym(R_net(x).view(n,n)), J - J^T without the 1/2, fixed G).  Pinned to the golden vectors in
tests/test_oracle_golden.py.  The reference's own files cannot travel to the GPU box; this file is what is timed
there as "a stock PyTorch-CPU restatement".
"""
import torch


def _mlp(x, layers):
    for i, (W, b) in enumerate(layers):
        x = torch.addmm(b, x, W.t())
        if i + 1 < len(layers):
            x = torch.tanh(x)
    return x


def _layers(w, prefix, dtype):
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in w if k.startswith(prefix) and k.endswith(".weight")})
    return [(torch.as_tensor(w[f"{prefix}{i}.weight"], dtype=dtype), torch.as_tensor(w[f"{prefix}{i}.bias"], dtype=dtype))
            for i in idx]


class TorchPhnn:
    """pHNN with a fixed G (the cart-pole configuration), weights from a reference state_dict (numpy arrays)."""

    def __init__(self, w, dtype=torch.float32):
        self.dtype = dtype
        self.J = torch.as_tensor(w["J"], dtype=dtype)
        self.G = torch.as_tensor(w["G_fixed"], dtype=dtype)
        self.H_layers = _layers(w, "H_net.net.", dtype)
        self.R_layers = _layers(w, "R_net.net.", dtype)
        self.n = self.J.shape[0]

    def forward(self, x, u):
        B, n = x.shape[0], self.n
        if not x.requires_grad:
            x = x.requires_grad_(True)
        H = _mlp(x, self.H_layers).squeeze(-1)
        dH = torch.autograd.grad(H.sum(), x, create_graph=True)[0]
        R_raw = _mlp(x, self.R_layers).view(B, n, n)
        S = (R_raw + R_raw.transpose(1, 2)) / 2
        R = torch.bmm(S, S.transpose(1, 2))
        A = (self.J - self.J.t()).unsqueeze(0) - R
        dx = torch.bmm(A, dH.unsqueeze(-1)).squeeze(-1) + torch.bmm(self.G.unsqueeze(0).expand(B, -1, -1), u.unsqueeze(-1)).squeeze(-1)
        return dx, H

    def rollout_cost_grad(self, x0, U, Q_diag, R, x_target, u_min, u_max, dt):
        """x0 (B,n), U (B,H,1) -> (cost (B,), grad_u (B,H,1)); Euler; controls clamped inside the graph."""
        x0 = torch.as_tensor(x0, dtype=self.dtype)
        U = torch.as_tensor(U, dtype=self.dtype).clone().requires_grad_(True)
        import numpy as np
        Q = torch.tensor(np.array(Q_diag, dtype=np.float64), dtype=self.dtype)
        xt = torch.tensor(np.array(x_target, dtype=np.float64), dtype=self.dtype)
        Uc = torch.clamp(U, u_min, u_max)
        x = x0.clone().requires_grad_(True)
        cost = (((x - xt) ** 2) * Q).sum(dim=1)
        for t in range(U.shape[1]):
            dx, _ = self.forward(x, Uc[:, t, :])
            x = x + dt * dx
            cost = cost + (((x - xt) ** 2) * Q).sum(dim=1)
        cost = cost + R * (Uc ** 2).sum(dim=(1, 2))
        g = torch.autograd.grad(cost.sum(), U)[0]
        return cost.detach(), g
