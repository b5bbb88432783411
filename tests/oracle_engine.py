"""OracleEngine: the engine protocol (what controllers / solver / distributed code call) served by the CPU
oracle on CPU tensors.  TEST INFRASTRUCTURE: lets the host logic be exercised without a GPU.  The product's
engine is phnn_mpc_amd.engine.RolloutEngine; nothing in the package imports this file.
"""
import numpy as np
import torch

import oracle_lib as ol
from phnn_mpc_amd import _capi


class OracleEngine:
    def __init__(self, state_dict, precision="f32"):
        self.m_ = ol.OracleModel(state_dict, precision)
        self.n, self.m = self.m_.n, self.m_.m
        self.device = torch.device("cpu")
        self.np_dtype = self.m_.dtype
        from phnn_mpc_amd import weights
        self.layout = weights.blob_layout(state_dict, kind=self.m_.desc.kind)
        self.has_wgrad = self.m_.desc.kind != _capi.MODEL_ODEFUNC  # mirrors the product: pHNN and canonical only

    tape_token = None  # the CPU stand-in keeps no tapes (the product's K1 -> K2w shortcut)

    def rollout_trajectory(self, x0, u, integrator="euler", dt=0.02, want_dx=False, tapes=False):
        r = self.m_.rollout_wgrad(np.asarray(x0), np.asarray(u), integrator, dt)
        return (self._out(r["traj"]), self._out(r["dX"])) if want_dx else self._out(r["traj"])

    def rollout_wgrad(self, x0, u, traj, integrator="euler", dt=0.02, traj_bar=None, dx_bar=None, grad_theta=None,
                      accumulate=False, tape_token=None):
        r = self.m_.rollout_wgrad(np.asarray(x0), np.asarray(u), integrator, dt,
                                  None if traj_bar is None else np.asarray(traj_bar),
                                  None if dx_bar is None else np.asarray(dx_bar))
        return self._out(r["grad_theta"]), self._out(r["grad_u"]), self._out(r["grad_x0"])

    def model_wgrad(self, x, u, lam, Hbar=None, grad_theta=None, accumulate=False):
        g = self.m_.wgrad(np.asarray(x), np.asarray(u), np.asarray(lam), None if Hbar is None else np.asarray(Hbar))
        xb, ub = self.m_.vjp(np.asarray(x), np.asarray(u), np.asarray(lam))
        if Hbar is not None:  # cotangent on H adds Hbar dH/dx: dH/dx = VJP of H, taken by finite structure of the oracle
            xb = xb + np.asarray(Hbar, self.np_dtype)[:, None] * self._dH(np.asarray(x), np.asarray(u))
        return self._out(g), self._out(xb), self._out(ub)

    def _dH(self, x, u):
        """dH/dx by central differences of the oracle's H in float64 (test infrastructure; only the energy-anchor
        path needs it and there the input gradient is not consumed)."""
        eps = 1e-6
        out = np.zeros((x.shape[0], self.n))
        for i in range(self.n):
            xp, xm = np.array(x, np.float64), np.array(x, np.float64)
            xp[:, i] += eps
            xm[:, i] -= eps
            out[:, i] = (self.m_.forward(xp, u)[1] - self.m_.forward(xm, u)[1]) / (2 * eps)
        return out.astype(self.np_dtype)

    def named_grads(self, grad_theta):
        from phnn_mpc_amd import weights
        return weights.unpack_grad_blob(None, grad_theta, layout=self.layout)

    def _out(self, a):
        return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))

    def forward(self, x, u):
        dx, H = self.m_.forward(x.numpy(), u.numpy())
        return self._out(dx), self._out(H)

    def vjp(self, x, u, lam):
        xb, ub = self.m_.vjp(x.numpy(), u.numpy(), lam.numpy())
        return self._out(xb), self._out(ub)

    def rollout_cost(self, x0, u, cost, integrator="euler", dt=0.02, want_traj=False, traj_out=None):
        r = self.m_.rollout(np.asarray(x0), np.asarray(u), cost, integrator, dt, grad=False, traj=True)
        c, t = self._out(r["cost"]), self._out(r["traj"])
        return (c, t) if want_traj else c

    def rollout_cost_grad(self, x0, u, cost, integrator="euler", dt=0.02, want_grad_x0=False, workspace=None,
                          after_forward=None):
        r = self.m_.rollout(np.asarray(x0), np.asarray(u), cost, integrator, dt, grad=True, traj=False)
        out = (self._out(r["cost"]), self._out(r["grad_u"]))
        if after_forward is not None:
            after_forward(out[0])
        return out + (self._out(r["grad_x0"]),) if want_grad_x0 else out

    def adam_step(self, u, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, cost=None,
                  best_cost=None, best_u=None, u_min=None, u_max=None):
        if best_cost is not None:
            better = cost < best_cost
            uc = torch.clamp(u, u_min, u_max) if (u_min is not None and u_max is not None) else u
            best_u[better] = uc[better]
            best_cost[better] = cost[better]
        f32 = ol.OracleModel  # noqa: F841  (Adam restatement lives in the f32 oracle build)
        lib = ol.lib()
        import ctypes as C
        for t in (u, grad, exp_avg, exp_avg_sq):
            assert t.dtype == torch.float32 and t.is_contiguous()
        lib.oracle_adam_f32(C.c_void_p(u.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(exp_avg.data_ptr()),
                            C.c_void_p(exp_avg_sq.data_ptr()), u.numel(), lr, beta1, beta2, eps, step)


def bench_rehearsal_engine(state_dict):
    """Factory for `bench.py --rehearsal-engine oracle_engine:bench_rehearsal_engine` (tests/test_bench_spawn.py)."""
    return OracleEngine(state_dict)
