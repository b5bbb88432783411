"""Batched single-shooting solve: Adam on the control sequences of B independent MPC problems.

Host logic shared by both controller classes.  It restates the optimisation loops of the reference
  src/mpc_controller.py:164-209            (cold start zeros, Adam, returns the LAST iterate, clamped)
  src/mpc_controller_canonical.py:163-228  (optional warm start, Adam, returns the BEST clamped iterate; the cost
                                            of iterate k is measured before step k is applied, strict '<')
for B problems at once: Adam is element-wise, so B stacked problems of shape (H,m) behave exactly like B
separate torch.optim.Adam instances.  All arithmetic is delegated to an engine object (RolloutEngine on the
GPU): rollout_cost_grad (K1+K2) and adam_step (K3).
"""
import torch


def shooting_solve(engine, x0, u_init, cost, integrator, dt, lr, iters, track_best=False, u_min=None, u_max=None,
                   record_costs=True):
    """x0 (B,n), u_init (B,H,m) on engine.device -> dict(u_last, costs[, best_u, best_cost]).

    u_last   : unclamped last iterate (B,H,m)
    costs    : (iters,B) cost of each iterate (measured before its Adam step), if record_costs
    best_u   : clamped best iterate (track_best)
    """
    dev = x0.device
    u = u_init.detach().clone().contiguous()
    exp_avg = torch.zeros_like(u)
    exp_avg_sq = torch.zeros_like(u)
    B = u.shape[0]
    costs = torch.empty(iters, B, dtype=torch.float32, device=dev) if record_costs else None
    best_cost = best_u = None
    if track_best:
        best_cost = torch.full((B,), float("inf"), dtype=torch.float32, device=dev)
        best_u = torch.zeros_like(u)
    ws = {}
    for k in range(iters):
        c, g = engine.rollout_cost_grad(x0, u, cost, integrator, dt, workspace=ws)
        if record_costs:
            costs[k].copy_(c)
        engine.adam_step(u, g, exp_avg, exp_avg_sq, lr, k + 1, cost=c if track_best else None, best_cost=best_cost,
                         best_u=best_u, u_min=u_min, u_max=u_max)
    out = {"u_last": u, "costs": costs}
    if track_best:
        out["best_u"], out["best_cost"] = best_u, best_cost
    return out
