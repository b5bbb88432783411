"""Drop-in for src/mpc_controller_canonical.py: MPCControllerCanonical with the same constructor, methods, return
values and info dict, solved by the fused kernels, plus batched entry points.

Reference behaviour kept: full Q (n,n) and R (m,m) matrices; Euler rollout; the iterate returned is the BEST
clamped one (cost measured before the Adam step of the same iteration, strict '<', src/mpc_controller_canonical.py:
208-214); control() warm-starts by shifting the previous sequence by one and zero-filling the tail (:252-255) and
returns (u (m,), info{'u_sequence','solve_time','optimization'{'costs','final_cost','num_steps'}}).
"""
import os
import time

import numpy as np
import torch

from . import _capi
from .solver import solver_for


class MPCControllerCanonical:
    def __init__(self, model, horizon=20, dt=0.02, Q=None, R=None, x_target=None, u_min=-10.0, u_max=10.0,
                 optimizer_steps=50, learning_rate=0.1, verbose=False):
        self.model = model
        self.model.eval()
        self.horizon, self.dt = horizon, dt
        self.optimizer_steps, self.learning_rate, self.verbose = optimizer_steps, learning_rate, verbose
        self.state_dim, self.input_dim = model.state_dim, model.input_dim
        if Q is None:
            Q = np.diag([10.0, 100.0, 1.0, 10.0])  # src/mpc_controller_canonical.py:68-74
        self.Q = torch.tensor(Q, dtype=torch.float32)
        if R is None:
            R = 0.01 * np.eye(self.input_dim)
        self.R = torch.tensor(R, dtype=torch.float32)
        if x_target is None:
            x_target = np.zeros(self.state_dim)
        self.x_target = torch.tensor(x_target, dtype=torch.float32)
        self.u_min, self.u_max = u_min, u_max
        self.integrator = "euler"
        # True (or PHNN_GRAPH=1): replay the whole solve as one HIP graph instead of 3 x iterations launches
        self.use_graph = os.environ.get("PHNN_GRAPH", "0") == "1"
        self._graphed = None

    def _solver(self, eng):
        self._graphed = solver_for(eng, self.use_graph, self._graphed)
        return self._graphed

    def _cost(self, clamp=True):
        c = _capi.make_cost(self.state_dim, self.input_dim, self.Q.numpy(), self.R.numpy(), self.x_target.numpy(),
                            self.u_min, self.u_max)
        if not clamp:
            c.has_u_bounds = 0
        return c

    @property
    def engine(self):
        return self.model.engine

    # ------------------------------------------------------------------ reference methods
    def compute_cost(self, x_pred, u_seq):
        """(H+1,n), (H,m) -> scalar cost (src/mpc_controller_canonical.py:91-120), host arithmetic."""
        x_pred = torch.as_tensor(x_pred, dtype=torch.float32)
        u_seq = torch.as_tensor(u_seq, dtype=torch.float32)
        e = x_pred[: self.horizon + 1] - self.x_target
        u = u_seq[: self.horizon]
        return ((e @ self.Q) * e).sum() + ((u @ self.R) * u).sum()

    def rollout(self, x0, u_seq):
        """x0 (n,), u_seq (H,m) -> (H+1,n), controls used as given (src/mpc_controller_canonical.py:122-161)."""
        eng = self.engine
        x0d = torch.as_tensor(x0, dtype=torch.float32).reshape(1, -1).to(eng.device)
        ud = torch.as_tensor(u_seq, dtype=torch.float32).reshape(1, -1, self.input_dim).to(eng.device)
        _, traj = eng.rollout_cost(x0d, ud, self._cost(clamp=False), self.integrator, self.dt, want_traj=True)
        return traj[0].cpu()

    def optimize_control(self, x0, u_init=None):
        """x0 (n,), u_init (H,m) or None -> (u_opt (H,m) tensor, info)   (src/mpc_controller_canonical.py:163-228)"""
        x0 = torch.as_tensor(x0, dtype=torch.float32).reshape(1, -1)
        ui = None if u_init is None else torch.as_tensor(u_init, dtype=torch.float32).reshape(1, self.horizon, -1)
        out = self.optimize_control_batch(x0, ui)
        costs = [float(v) for v in out["costs"][:, 0].cpu()]
        if self.verbose:
            for step, c in enumerate(costs):
                if step % 10 == 0 or step == self.optimizer_steps - 1:
                    print(f"  Step {step:3d}: cost = {c:.4f}")
        info = {"costs": costs, "final_cost": float(out["best_cost"][0].cpu()), "num_steps": self.optimizer_steps}
        return out["best_u"][0].cpu(), info

    def control(self, x_current, u_prev=None):
        """x_current (n,), u_prev (H,m) or None -> (u (m,) np.ndarray, info)  (src/mpc_controller_canonical.py:230-273)"""
        start = time.time()
        x0 = torch.tensor(np.asarray(x_current), dtype=torch.float32)
        u_init = None
        if u_prev is not None:
            up = torch.tensor(np.asarray(u_prev), dtype=torch.float32)
            u_init = torch.cat([up[1:], torch.zeros(1, self.input_dim)], dim=0)
        u_seq_opt, opt_info = self.optimize_control(x0, u_init)
        u = u_seq_opt[0].detach().numpy()
        info = {"u_sequence": u_seq_opt.detach().numpy(), "solve_time": time.time() - start, "optimization": opt_info}
        return u, info

    # ------------------------------------------------------------------ batched (new)
    def optimize_control_batch(self, x0, u_init=None, record_costs=True):
        """x0 (B,n), u_init (B,H,m) or None -> dict(best_u (B,H,m) clamped, best_cost (B), costs (steps,B), u_last)."""
        eng = self.engine
        x0 = torch.as_tensor(x0, dtype=torch.float32).reshape(-1, self.state_dim).to(eng.device)
        B = x0.shape[0]
        if u_init is None:
            u0 = torch.zeros(B, self.horizon, self.input_dim, dtype=torch.float32, device=eng.device)
        else:
            u0 = torch.as_tensor(u_init, dtype=torch.float32).reshape(B, self.horizon, self.input_dim).to(eng.device)
        return self._solver(eng)(eng, x0, u0, self._cost(), self.integrator, self.dt, self.learning_rate,
                                 self.optimizer_steps, track_best=True, u_min=self.u_min, u_max=self.u_max,
                                 record_costs=record_costs)

    def control_batch(self, x_current, u_prev=None):
        """x_current (B,n), u_prev (B,H,m) or None -> (u (B,m), u_sequence (B,H,m), best_cost (B)) numpy arrays."""
        u_init = None
        if u_prev is not None:
            up = torch.as_tensor(u_prev, dtype=torch.float32)
            u_init = torch.cat([up[:, 1:], torch.zeros(up.shape[0], 1, self.input_dim)], dim=1)
        out = self.optimize_control_batch(x_current, u_init, record_costs=False)
        seq = out["best_u"].cpu().numpy()
        return seq[:, 0, :], seq, out["best_cost"].cpu().numpy()


def create_mpc_controller(model, config):
    """src/mpc_controller_canonical.py:276-316"""
    mpc = config.get("mpc", {})
    return MPCControllerCanonical(
        model=model, horizon=mpc.get("horizon", 20), dt=config["cartpole"]["dt"],
        Q=np.diag(mpc.get("Q_diag", [10.0, 100.0, 1.0, 10.0])), R=np.diag(mpc.get("R_diag", [0.01])),
        x_target=np.array(mpc.get("x_target", [0.0, 0.0, 0.0, 0.0])), u_min=mpc.get("u_min", -10.0),
        u_max=mpc.get("u_max", 10.0), optimizer_steps=mpc.get("optimizer_steps", 50),
        learning_rate=mpc.get("learning_rate", 0.1), verbose=mpc.get("verbose", False))
