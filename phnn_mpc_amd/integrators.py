"""Drop-in for src/integrators.py: same function names, arguments and return shapes.

Single steps (euler_step, rk4_step, rk4_step_with_energy) compose model calls exactly as the reference does
(each model call is one kernel launch).  The rollout functions send the whole horizon to the fused forward
kernel K1 in ONE launch when the model is engine-backed, and rollout_trajectory_differentiable is
differentiable w.r.t. y0 and controls through the adjoint kernel (phnn_rollout_vjp).
"""
import torch

from . import _capi
from .models import _EngineBacked, ODEFunc, _no_wgrad_warning, grad_params, live_engine, packed_engine, split_param_grads


def _call(model, y, u):
    if isinstance(model, ODEFunc):
        return model.dynamics(y, u)
    return model(y, u)


def euler_step(model, y, u, dt):
    """y_{t+1} = y_t + dt f(y_t,u_t)   (src/integrators.py:13-36)"""
    return y + dt * _call(model, y, u)[0]


def rk4_step(model, y, u, dt):
    """classic RK4 with u held over the step (src/integrators.py:39-84)"""
    k1 = _call(model, y, u)[0]
    k2 = _call(model, y + (dt / 2) * k1, u)[0]
    k3 = _call(model, y + (dt / 2) * k2, u)[0]
    k4 = _call(model, y + dt * k3, u)[0]
    return y + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)


def rk4_step_with_energy(model, y, u, dt):
    """RK4 step + H at the CURRENT state (stage k1), src/integrators.py:87-125"""
    res = _call(model, y, u)
    k1, H = res[0], res[1]
    k2 = _call(model, y + (dt / 2) * k1, u)[0]
    k3 = _call(model, y + (dt / 2) * k2, u)[0]
    k4 = _call(model, y + dt * k3, u)[0]
    return y + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4), H


def _zero_cost(n, m):
    import numpy as np
    return _capi.make_cost(n, m, np.zeros((n, n)), np.zeros((m, m)))


class _RolloutFn(torch.autograd.Function):
    """(trajectory, dX) = rollout(y0, controls) on K1.  backward: the adjoint march on K2 -- w.r.t. y0 and controls,
    and, when the model's parameters require grad or the loss uses dX, the weight-gradient path (phnn_rollout_wgrad):
    what loss.backward() does in the reference's training loops (scripts/train_cartpole_phnn.py:112-178, main.py:93-148).
    """

    @staticmethod
    def forward(ctx, y0, controls, model, dt, integrator, keys, *params):
        eng = ctx.eng = packed_engine(model)  # _rollout has just checked it against the parameters
        y0d = y0.detach().to(eng.device, torch.float32).contiguous()
        ud = controls.detach().to(eng.device, torch.float32).contiguous()
        # parameters that need gradients: K1 keeps its tapes for the weight-gradient pass of backward()
        traj, dX = eng.rollout_trajectory(y0d, ud, integrator, dt, want_dx=True, tapes=bool(keys))
        ctx.tape_token = eng.tape_token if keys else None
        ctx.model, ctx.dt, ctx.integ, ctx.keys = model, dt, integrator, keys
        ctx.devs = (y0.device, controls.device, params[0].device if params else None)
        ctx.save_for_backward(y0d, ud, traj)
        ctx.set_materialize_grads(False)  # an unused output (dX, mostly) arrives as None, not as zeros
        return traj.to(y0.device), dX.to(y0.device)

    @staticmethod
    def backward(ctx, gtraj, gdx):
        y0d, ud, traj = ctx.saved_tensors
        eng = live_engine(ctx.eng)
        want_params = bool(ctx.keys) and any(ctx.needs_input_grad[6:])
        nk = len(ctx.keys)
        tb = None if gtraj is None else gtraj.to(eng.device, torch.float32).contiguous()
        db = None if gdx is None else gdx.to(eng.device, torch.float32).contiguous()
        if want_params or (db is not None and eng.has_wgrad):
            gth, gu, gx = eng.rollout_wgrad(y0d, ud, traj, ctx.integ, ctx.dt, traj_bar=tb, dx_bar=db,
                                            tape_token=ctx.tape_token)
            pg = (split_param_grads(eng, gth, ctx.keys, ctx.devs[2], ctx.model, (y0d.shape[0], ud.shape[1], ctx.integ))
                  if want_params else (None,) * nk)
            return (gx.to(ctx.devs[0]), gu.to(ctx.devs[1]), None, None, None, None) + pg
        if db is not None and bool((db != 0).any()):
            raise NotImplementedError("a loss on the per-step derivatives needs the weight-gradient kernels (pHNN / canonical)")
        zero = torch.zeros(y0d.shape[0], dtype=torch.float32, device=eng.device)
        if tb is None:
            tb = torch.zeros_like(traj)
        gu, gx = eng.rollout_vjp(y0d, ud, traj, _zero_cost(eng.n, eng.m), ctx.integ, ctx.dt, traj_bar=tb, cost_bar=zero)
        return (gx.to(ctx.devs[0]), gu.to(ctx.devs[1]), None, None, None, None) + (None,) * nk


def _rollout(model, y0, controls, dt, integrator):
    keys, params = [], []
    eng = model.engine  # re-packs the weights if a parameter has changed
    if torch.is_grad_enabled():
        if eng.has_wgrad:
            keys, params = grad_params(model)
        elif any(p.requires_grad for p in model.parameters()):
            _no_wgrad_warning(model)
    return _RolloutFn.apply(y0, controls, model, dt, integrator, keys, *params)


def _check_integrator(integrator):
    if integrator not in ("rk4", "euler"):
        raise ValueError(f"Unknown integrator: {integrator}")


def _energies(model, traj, controls):
    """H along a trajectory with the reference's indexing quirks handled by the callers."""
    B, T1, n = traj.shape
    m = controls.shape[-1]
    u_pad = torch.cat([controls, controls[:, -1:]], dim=1)  # u used only as a formal argument of model(y,u)
    H = _call(model, traj.reshape(B * T1, n), u_pad.reshape(B * T1, m))[1]
    return H.reshape(B, T1)


def rollout_trajectory(model, y0, controls, dt, integrator="rk4"):
    """-> trajectory (B,T+1,n), energies (B,T+1)   (src/integrators.py:128-189; H at every stored state)."""
    _check_integrator(integrator)
    if not isinstance(model, _EngineBacked):
        raise TypeError("rollout_trajectory needs an engine-backed model (phnn_mpc_amd.models)")
    with torch.no_grad():
        traj = _rollout(model, y0, controls, dt, integrator)[0]
        energies = _energies(model, traj, controls)
    return traj, energies


def rollout_trajectory_differentiable(model, y0, controls, dt, integrator="rk4", return_energies=False,
                                      return_derivatives=False):
    """-> trajectory (B,T+1,n) [, energies (B,T+1)]   (src/integrators.py:192-258).

    Differentiable w.r.t. y0, the controls AND the model's parameters (pHNN / canonical pHNN): one fused forward
    launch, and one adjoint + one reduction launch in backward, replace the reference's Python time loop and its
    autograd graph.  return_derivatives=True (not in the reference's signature) additionally returns
    dX (B,T,n) = f(x_t,u_t) of every step, differentiable too -- the dX_pred the training loops collect
    (main.py:116-122).  With return_energies the reference returns H at the CURRENT state of each step, so
    energies[:,1] duplicates energies[:,0]; reproduced here.
    """
    _check_integrator(integrator)
    if not isinstance(model, _EngineBacked):
        raise TypeError("rollout_trajectory_differentiable needs an engine-backed model (phnn_mpc_amd.models)")
    traj, dX = _rollout(model, y0, controls, dt, integrator)
    if return_derivatives:
        if return_energies:
            raise ValueError("return_derivatives and return_energies are exclusive")
        return traj, dX
    if not return_energies:
        return traj
    with torch.no_grad():
        Hall = _energies(model, traj, controls)
    energies = torch.cat([Hall[:, :1], Hall[:, :-1]], dim=1)
    return traj, energies


def compare_integrators(model, y0, controls, dt):
    """Euler against RK4 on the same controls (src/integrators.py:261-308): both trajectories and energy histories, the
    per-step distance between the two trajectories and each integrator's energy drift |H_T - H_0|."""
    with torch.no_grad():
        et, ee = rollout_trajectory(model, y0, controls, dt, integrator="euler")
        rt, re = rollout_trajectory(model, y0, controls, dt, integrator="rk4")
        return {"euler_trajectory": et, "rk4_trajectory": rt, "trajectory_difference": torch.norm(et - rt, dim=-1),
                "euler_energies": ee, "rk4_energies": re, "euler_energy_drift": torch.abs(ee[:, -1] - ee[:, 0]),
                "rk4_energy_drift": torch.abs(re[:, -1] - re[:, 0])}
