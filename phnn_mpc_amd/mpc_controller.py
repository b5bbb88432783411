"""Drop-in for src/mpc_controller.py: MPCController with the same constructor and methods, solved by the fused
rollout / adjoint / Adam kernels, plus a batched entry point (many plants at once).

Reference behaviour kept:
  - cost sums the state error over t = 0..H inclusive, control effort over t = 0..H-1, optional soft barrier;
  - controls are clamped inside the differentiated closure only when BOTH u_min and u_max are given;
  - every call cold-starts from zeros, runs max_iterations Adam steps and returns clamp(u_0) of the LAST iterate
    as a numpy array of shape (1,).
"""
import os

import numpy as np
import torch

from . import _capi
from .solver import solver_for


class MPCController:
    def __init__(self, phnn_model, horizon, dt, Q, R, target_state=None, u_min=None, u_max=None, x_min=None, x_max=None,
                 optimizer_type="Adam", lr=0.1, max_iterations=50):
        self.model = phnn_model
        self.model.eval()
        self.horizon, self.dt = horizon, dt
        self.Q = torch.diag(torch.tensor(Q, dtype=torch.float32)) if isinstance(Q, list) else torch.diag(Q)
        self.R = R
        self.state_dim = phnn_model.J.shape[0]
        if target_state is None:
            self.target_state = torch.zeros(self.state_dim)
        else:
            self.target_state = torch.tensor(target_state, dtype=torch.float32)
        self.u_min, self.u_max = u_min, u_max
        self.x_min = torch.tensor(x_min, dtype=torch.float32) if x_min is not None else None
        self.x_max = torch.tensor(x_max, dtype=torch.float32) if x_max is not None else None
        self.optimizer_type, self.lr, self.max_iterations = optimizer_type, lr, max_iterations
        self.integrator = "euler"  # src/mpc_controller.py:137-138
        # True (or PHNN_GRAPH=1): replay the whole solve as one HIP graph instead of 3 x iterations launches
        self.use_graph = os.environ.get("PHNN_GRAPH", "0") == "1"
        self._graphed = None

    # ------------------------------------------------------------------ kernel parameters
    def _solver(self, eng):
        self._graphed = solver_for(eng, self.use_graph, self._graphed)
        return self._graphed

    def _cost(self):
        return _capi.make_cost(self.state_dim, 1, self.Q.numpy(), float(self.R), self.target_state.numpy(),
                               self.u_min, self.u_max,
                               None if self.x_min is None else self.x_min.numpy(),
                               None if self.x_max is None else self.x_max.numpy(), 1000.0)

    def _cost_noclamp(self):
        c = self._cost()
        c.has_u_bounds = 0
        return c

    @property
    def engine(self):
        return self.model.engine

    # ------------------------------------------------------------------ reference methods (B = 1)
    def rollout_dynamics(self, x0, controls):
        """x0 (n,), controls (H,1) -> states (H+1,n); controls are used as given (src/mpc_controller.py:116-141)."""
        eng = self.engine
        x0d = torch.as_tensor(x0, dtype=torch.float32).reshape(1, -1).to(eng.device)
        ud = torch.as_tensor(controls, dtype=torch.float32).reshape(1, -1, 1).to(eng.device)
        _, traj = eng.rollout_cost(x0d, ud, self._cost_noclamp(), self.integrator, self.dt, want_traj=True)
        return traj[0].cpu()

    def compute_cost(self, states, controls):
        """Quadratic cost (+ barrier) of a given trajectory (src/mpc_controller.py:75-114); host arithmetic on
        (H+1,n)/(H,1) tensors -- the kernels fuse the same sum into the march."""
        states = torch.as_tensor(states, dtype=torch.float32)
        controls = torch.as_tensor(controls, dtype=torch.float32)
        e = states[: self.horizon + 1] - self.target_state
        cost = ((e @ self.Q) * e).sum()
        if self.x_min is not None:
            cost = cost + 1000.0 * (torch.relu(self.x_min - states[: self.horizon + 1]) ** 2).sum()
        if self.x_max is not None:
            cost = cost + 1000.0 * (torch.relu(states[: self.horizon + 1] - self.x_max) ** 2).sum()
        return cost + self.R * (controls[: self.horizon] ** 2).sum()

    def compute_control(self, current_state):
        """current_state (n,) -> optimal first control, np.ndarray (1,)   (src/mpc_controller.py:143-209)"""
        if isinstance(current_state, np.ndarray):
            current_state = torch.tensor(current_state, dtype=torch.float32)
        if self.optimizer_type == "LBFGS":
            return self._compute_control_lbfgs(current_state)
        u = self.compute_control_batch(current_state.reshape(1, -1))
        return u[0]

    def _compute_control_lbfgs(self, current_state):
        """The reference's L-BFGS branch (src/mpc_controller.py:169-170,196-197): torch.optim.LBFGS(lr, max_iter=20)
        stepped max_iterations times; the closure's cost and gradient come from K1/K2 instead of autograd.  One
        plant at a time (L-BFGS keeps a curvature history per problem)."""
        eng = self.engine
        x0 = current_state.reshape(1, -1).to(eng.device, torch.float32)
        control_sequence = torch.zeros(self.horizon, 1, requires_grad=True)
        optimizer = torch.optim.LBFGS([control_sequence], lr=self.lr, max_iter=20)
        cost_struct, ws = self._cost(), {}

        def closure():
            optimizer.zero_grad()
            c, g = eng.rollout_cost_grad(x0, control_sequence.detach().reshape(1, self.horizon, 1).to(eng.device), cost_struct,
                                         self.integrator, self.dt, workspace=ws)
            control_sequence.grad = g.reshape(self.horizon, 1).to(control_sequence.device).clone()
            return c.reshape(()).to(control_sequence.device).clone()

        for _ in range(self.max_iterations):
            optimizer.step(closure)
        with torch.no_grad():
            u0 = control_sequence[0]
            if self.u_min is not None and self.u_max is not None:
                u0 = torch.clamp(u0, self.u_min, self.u_max)
        return u0.detach().numpy()

    # ------------------------------------------------------------------ batched (new)
    def solve_batch(self, states, record_costs=False):
        """states (B,n) -> dict with the last iterate of B independent problems (all on the engine's device)."""
        if self.optimizer_type == "LBFGS":
            raise NotImplementedError("L-BFGS keeps a curvature history per problem: use compute_control (one plant "
                                      "at a time); the batched solve is Adam only")
        if self.optimizer_type != "Adam":
            raise ValueError(f"Unknown optimizer type: {self.optimizer_type}")
        eng = self.engine
        x0 = torch.as_tensor(states, dtype=torch.float32).reshape(-1, self.state_dim).to(eng.device)
        u0 = torch.zeros(x0.shape[0], self.horizon, 1, dtype=torch.float32, device=eng.device)
        return self._solver(eng)(eng, x0, u0, self._cost(), self.integrator, self.dt, self.lr, self.max_iterations,
                                 track_best=False, u_min=self.u_min, u_max=self.u_max, record_costs=record_costs)

    def compute_control_batch(self, states):
        """states (B,n) -> np.ndarray (B,1): first control of each plant's optimised sequence."""
        out = self.solve_batch(states)
        u0 = out["u_last"][:, 0, :]
        if self.u_min is not None and self.u_max is not None:
            u0 = torch.clamp(u0, self.u_min, self.u_max)
        return u0.cpu().numpy()


def create_mpc_from_config(phnn_model, config):
    """Accepts both key schemas the reference uses: src/mpc_controller.py:212-241 (Q/R/dt/lr/max_iterations) and
    scripts/run_cartpole_mpc.py:57-88 (Q_diag/R_diag/learning_rate/optimizer_steps, dt from config['cartpole'])."""
    mpc = config["mpc"]
    if "Q_diag" in mpc:
        return MPCController(phnn_model=phnn_model, horizon=mpc.get("horizon", 20), dt=config["cartpole"]["dt"],
                             Q=mpc.get("Q_diag", [10.0, 100.0, 1.0, 10.0]), R=mpc.get("R_diag", [0.01])[0],
                             target_state=mpc.get("x_target", [0.0, 0.0, 0.0, 0.0]), u_min=mpc.get("u_min", -10.0),
                             u_max=mpc.get("u_max", 10.0), optimizer_type="Adam", lr=mpc.get("learning_rate", 0.1),
                             max_iterations=mpc.get("optimizer_steps", 50))
    return MPCController(phnn_model=phnn_model, horizon=mpc["horizon"], dt=mpc["dt"], Q=mpc["Q"], R=mpc["R"],
                         target_state=mpc.get("target_state", None), u_min=mpc.get("u_min", None),
                         u_max=mpc.get("u_max", None), x_min=mpc.get("x_min", None), x_max=mpc.get("x_max", None),
                         optimizer_type=mpc.get("optimizer", "Adam"), lr=mpc.get("lr", 0.1),
                         max_iterations=mpc.get("max_iterations", 50))
