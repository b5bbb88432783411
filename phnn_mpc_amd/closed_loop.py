"""Closed-loop receding-horizon drivers for MANY plants at once.

  BatchedCartPole        the reference's ground-truth plant (src/cartpole_simulator.py:63-112: float64, explicit
                         Euler, standard cart-pole equations, termination |x| > 10 or |theta| > 0.5) vectorised over
                         B plants with numpy -- it runs once per control step on the host, like the reference's.
  run_mpc_batch          the loop of scripts/run_cartpole_mpc.py:91-182 / scripts/run_mpc_canonical.py:26-110 for B
                         plants: every control step is ONE batched solve on the GPU (cold start for MPCController,
                         warm start by shift for MPCControllerCanonical), then one plant step.
  run_mpc_batch_device   the same loop with NOTHING on the host: the plant (k_plant_step, float64), the warm-start shift,
                         the logs and the done mask are device kernels, one control step = solve + plant step is
                         captured as a HIP graph and replayed num_steps times; the host synchronises once, at the end.
  stability_report       the "stability achieved" criterion of scripts/run_cartpole_mpc.py:117-159 with the
                         tolerances of the `stability` config section, per plant.
"""
import numpy as np


class BatchedCartPole:
    def __init__(self, dt=0.02):
        self.dt = dt
        self.gravity, self.masscart, self.masspole, self.length = 9.8, 1.0, 0.1, 0.5
        self.polemass_length = self.masspole * self.length
        self.total_mass = self.masspole + self.masscart
        self.state = None

    def reset(self, initial_states):
        self.state = np.array(initial_states, dtype=np.float64).reshape(-1, 4)
        return self.state.copy()

    def step(self, action):
        """action (B,) or (B,1) forces -> (states (B,4), done (B,) bool)"""
        force = np.asarray(action, dtype=np.float64).reshape(-1)
        x, theta, x_dot, theta_dot = self.state.T
        costheta, sintheta = np.cos(theta), np.sin(theta)
        temp = (force + self.polemass_length * theta_dot ** 2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        self.state = np.stack([x + self.dt * x_dot, theta + self.dt * theta_dot, x_dot + self.dt * xacc,
                               theta_dot + self.dt * thetaacc], axis=1)
        done = (np.abs(self.state[:, 0]) > 10.0) | (np.abs(self.state[:, 1]) > 0.5)
        return self.state.copy(), done

    def get_state(self):
        return self.state.copy()


def run_mpc_batch(simulator, controller, initial_states, num_steps):
    """-> dict(states (T+1,B,4), controls (T,B,1), done_step (B,) first step a plant terminated at, or -1)."""
    x = simulator.reset(initial_states)
    B = x.shape[0]
    states, controls = [x.copy()], []
    done_step = np.full(B, -1, dtype=np.int64)
    canonical = hasattr(controller, "control_batch")
    u_prev = None
    for step in range(num_steps):
        if canonical:
            u, u_prev, _ = controller.control_batch(x.astype(np.float32), u_prev)
        else:
            u = controller.compute_control_batch(x.astype(np.float32))
        x, done = simulator.step(u)
        newly = done & (done_step < 0)
        done_step[newly] = step
        states.append(x.copy())
        controls.append(np.asarray(u, dtype=np.float64).reshape(B, -1))
    return {"states": np.stack(states), "controls": np.stack(controls), "done_step": done_step}


class DeviceClosedLoop:
    """Closed loop of B cart-poles resident on the engine's GPU (SURVEY.md 8 rows f1 + f3).

    controller: MPCController (cold start, last iterate, clamp(u_0)) or MPCControllerCanonical (warm start by shift,
    best clamped iterate).  One control step enqueues: iters x (K1, K2, K3), k_plant_step, k_shift_controls; with
    use_graph the step is captured once and replayed.  Per-plant arithmetic is identical to run_mpc_batch (same
    kernels, same order); the plant differs from the numpy one only by the device's double-precision sin/cos.
    """

    def __init__(self, controller, initial_states, num_steps, use_graph=True, dt=None):
        import torch
        from . import _capi
        self.torch, self.ctl = torch, controller
        eng = self.eng = controller.engine
        dev = eng.device
        self.canonical = hasattr(controller, "control_batch")
        x = np.asarray(initial_states, dtype=np.float64).reshape(-1, 4)
        B, H, m = x.shape[0], controller.horizon, 1
        self.B, self.T = B, int(num_steps)
        self.plant = _capi.Plant.default(controller.dt if dt is None else dt)
        self.state = torch.tensor(x, dtype=torch.float64, device=dev)
        self.x32 = self.state.to(torch.float32)  # what the reference hands the controller (float32 of the state)
        self.log_states = torch.empty(self.T + 1, B, 4, dtype=torch.float64, device=dev)
        self.log_states[0].copy_(self.state)
        self.log_controls = torch.empty(self.T, B, dtype=torch.float32, device=dev)
        self.done_step = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.u_init = torch.zeros(B, H, m, dtype=torch.float32, device=dev)
        self.u = torch.empty_like(self.u_init)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.u), torch.zeros_like(self.u)
        self.best_cost = torch.empty(B, dtype=torch.float32, device=dev) if self.canonical else None
        self.best_u = torch.empty_like(self.u) if self.canonical else None
        self.ws = {}
        self.cost = controller._cost()
        self.iters = controller.optimizer_steps if self.canonical else controller.max_iterations
        self.lr = controller.learning_rate if self.canonical else controller.lr
        if not self.canonical and controller.optimizer_type != "Adam":
            raise NotImplementedError("the device closed loop runs the batched Adam solve")
        self.graph = None
        self.use_graph = use_graph and dev.type == "cuda"

    def _control_step(self):
        eng, c = self.eng, self.ctl
        self.u.copy_(self.u_init)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        if self.canonical:
            self.best_cost.fill_(float("inf"))
            self.best_u.zero_()
        for k in range(self.iters):
            cost, g = eng.rollout_cost_grad(self.x32, self.u, self.cost, c.integrator, c.dt, workspace=self.ws)
            eng.adam_step(self.u, g, self.exp_avg, self.exp_avg_sq, self.lr, k + 1,
                          cost=cost if self.canonical else None, best_cost=self.best_cost, best_u=self.best_u,
                          u_min=c.u_min, u_max=c.u_max)
        H = self.u.shape[1] * self.u.shape[2]
        if self.canonical:  # best clamped iterate; next call warm-starts from its shift
            eng.plant_step(self.plant, self.state, self.best_u, H, state_f32=self.x32, done_step=self.done_step,
                           step_dev=self.step_dev, log_states=self.log_states, log_controls=self.log_controls)
            eng.shift_controls(self.best_u, self.u_init, step_dev=self.step_dev)
        else:  # last iterate, clamp(u_0); every call cold-starts from zeros (u_init stays zero)
            eng.plant_step(self.plant, self.state, self.u, H, u_min=c.u_min, u_max=c.u_max, state_f32=self.x32,
                           done_step=self.done_step, step_dev=self.step_dev, log_states=self.log_states,
                           log_controls=self.log_controls)
            eng.advance_step(self.step_dev)

    def run(self):
        torch = self.torch
        dev = self.eng.device
        start = 0
        if self.use_graph and self.T > 1:
            # first step eagerly on a side stream (allocates the workspace outside the capture), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._control_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            start = 1
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._control_step()
            # the capture itself does not execute: replay for steps 1 .. T-1
            for _ in range(start, self.T):
                self.graph.replay()
        else:
            for _ in range(self.T):
                self._control_step()
        torch.cuda.synchronize(dev)
        return {"states": self.log_states.cpu().numpy(), "controls": self.log_controls.cpu().numpy()[:, :, None].astype(np.float64),
                "done_step": self.done_step.cpu().numpy().astype(np.int64)}


def run_mpc_batch_device(controller, initial_states, num_steps, use_graph=True):
    """Device-resident version of run_mpc_batch: same return dict, one host synchronisation at the end."""
    return DeviceClosedLoop(controller, initial_states, num_steps, use_graph=use_graph).run()


def stability_report(states, target, tolerance, min_duration, dt):
    """Per plant: was |x_t - target| <= tolerance held for at least min_duration seconds?  states (T+1,B,n)."""
    ok = np.all(np.abs(states - np.asarray(target)) <= np.asarray(tolerance), axis=2)  # (T+1,B)
    need = max(int(np.ceil(min_duration / dt)), 1)
    run = np.zeros(ok.shape[1], dtype=np.int64)
    best = np.zeros(ok.shape[1], dtype=np.int64)
    for t in range(ok.shape[0]):
        run = np.where(ok[t], run + 1, 0)
        best = np.maximum(best, run)
    return {"stable": best >= need, "longest_run_s": best * dt}
