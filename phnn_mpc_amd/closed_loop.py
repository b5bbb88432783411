"""Closed-loop receding-horizon drivers for MANY plants at once.

  BatchedCartPole        the reference's ground-truth plant (src/cartpole_simulator.py:63-112: float64, explicit
                         Euler, standard cart-pole equations, termination |x| > 10 or |theta| > 0.5) vectorised over
                         B plants with numpy -- it runs once per control step on the host, like the reference's.
  run_mpc_batch          the loop of scripts/run_cartpole_mpc.py:91-182 / scripts/run_mpc_canonical.py:26-110 for B
                         plants: every control step is ONE batched solve on the GPU (cold start for MPCController,
                         warm start by shift for MPCControllerCanonical), then one plant step.
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
