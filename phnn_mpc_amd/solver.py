"""Batched single-shooting solve: Adam on the control sequences of B independent MPC problems.

Host logic shared by both controller classes.  It restates the optimisation loops of the reference
  src/mpc_controller.py:164-209            (cold start zeros, Adam, returns the LAST iterate, clamped)
  src/mpc_controller_canonical.py:163-228  (optional warm start, Adam, returns the BEST clamped iterate; the cost
                                            of iterate k is measured before step k is applied, strict '<')
for B problems at once: Adam is element-wise, so B stacked problems of shape (H,m) behave exactly like B
separate torch.optim.Adam instances.  All arithmetic is delegated to an engine object (RolloutEngine on the
GPU): rollout_cost_grad (K1+K2) and adam_step (K3).
"""
import ctypes

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


def _eager(engine, x0, u_init, cost, integrator, dt, lr, iters, track_best=False, u_min=None, u_max=None, record_costs=True):
    """The solve without a captured graph: the library's own loop (phnn_solve: one call enqueues every launch) when the
    engine has one and the Adam-side bounds are the cost's (they are for both controller classes); else the Python loop.
    Same launches, same order: identical results."""
    has_b = u_min is not None and u_max is not None
    same = bool(cost.has_u_bounds) == has_b and (not has_b or (float(cost.u_min) == float(ctypes.c_float(u_min).value)
                                                                and float(cost.u_max) == float(ctypes.c_float(u_max).value)))
    if hasattr(engine, "solve") and same:
        return engine.solve(x0, u_init, cost, integrator, dt, lr=lr, iters=iters, track_best=track_best, record_costs=record_costs)
    return shooting_solve(engine, x0, u_init, cost, integrator, dt, lr, iters, track_best=track_best, u_min=u_min, u_max=u_max,
                          record_costs=record_costs)


def solver_for(engine, use_graph, previous=None):
    """-> callable(engine, x0, u_init, cost, ...) : shooting_solve, or a GraphedSolve bound to `engine` (reused from
    `previous` when it already is one for this engine)."""
    if not use_graph or engine.device.type != "cuda":
        return _eager
    if isinstance(previous, GraphedSolve) and previous.engine is engine:
        return previous
    return GraphedSolve(engine)


class GraphedSolve:
    """shooting_solve captured once as a HIP graph (all `iters` x (K1, K2, K3) launches plus the state resets) and
    replayed per call: one graph launch per MPC solve instead of 3 x iters kernel launches through Python.  Worth it
    where the solve is launch-bound -- the reference's own use, one plant (or a few) per call in a closed loop.

    The graph is tied to (B, H, m, iters, cost struct, integrator, dt, lr, flags); a call with another signature
    re-captures.  Inputs are copied into the graph's static buffers, results are returned as fresh tensors.
    Same kernels, same order, same arithmetic as shooting_solve: results are bit-identical.
    """

    def __init__(self, engine):
        self.engine = engine
        self.key = None
        self.graph = None

    def _signature(self, x0, u_init, cost, integrator, dt, lr, iters, track_best, u_min, u_max, record_costs):
        return (tuple(x0.shape), tuple(u_init.shape), bytes(ctypes.string_at(ctypes.addressof(cost), ctypes.sizeof(cost))),
                integrator, float(dt), float(lr), int(iters), bool(track_best), u_min, u_max, bool(record_costs))

    def _capture(self, x0, u_init, cost, integrator, dt, lr, iters, track_best, u_min, u_max, record_costs):
        eng, dev = self.engine, x0.device
        self.x0 = x0.detach().clone().contiguous()
        self.u_init = u_init.detach().clone().contiguous()
        self.u = torch.empty_like(self.u_init)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.u), torch.zeros_like(self.u)
        B = self.u.shape[0]
        self.costs = torch.empty(iters, B, dtype=torch.float32, device=dev) if record_costs else None
        self.best_cost = torch.empty(B, dtype=torch.float32, device=dev) if track_best else None
        self.best_u = torch.empty_like(self.u) if track_best else None
        self.ws = {}

        def body():
            self.u.copy_(self.u_init)
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            if track_best:
                self.best_cost.fill_(float("inf"))
                self.best_u.zero_()
            for k in range(iters):
                c, g = eng.rollout_cost_grad(self.x0, self.u, cost, integrator, dt, workspace=self.ws)
                if record_costs:
                    self.costs[k].copy_(c)
                eng.adam_step(self.u, g, self.exp_avg, self.exp_avg_sq, lr, k + 1, cost=c if track_best else None,
                              best_cost=self.best_cost, best_u=self.best_u, u_min=u_min, u_max=u_max)

        # one eager pass on a side stream first: allocates the workspace outside the capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()

    def __call__(self, engine, x0, u_init, cost, integrator, dt, lr, iters, track_best=False, u_min=None, u_max=None,
                 record_costs=True):
        assert engine is self.engine
        key = self._signature(x0, u_init, cost, integrator, dt, lr, iters, track_best, u_min, u_max, record_costs)
        if key != self.key:
            self._capture(x0, u_init, cost, integrator, dt, lr, iters, track_best, u_min, u_max, record_costs)
            self.key = key
        self.x0.copy_(x0)
        self.u_init.copy_(u_init)
        self.graph.replay()
        out = {"u_last": self.u.clone(), "costs": self.costs.clone() if record_costs else None}
        if track_best:
            out["best_u"], out["best_cost"] = self.best_u.clone(), self.best_cost.clone()
        return out
