"""Debug: is the big-batch solve bitwise equal to a slice solved alone, and to itself run twice?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd.engine import RolloutEngine
from phnn_mpc_amd.solver import shooting_solve
g, w = ol.load_golden("phnn_cartpole"), ol.load_weights("phnn_cartpole")
eng = RolloutEngine(w)
junk = torch.randn(1 << 28, device="cuda")  # dirty 1 GiB so that fresh allocations are not zero
del junk
rng = np.random.default_rng(1234)
B, H, iters, lr = 65536, 100, 20, 0.015
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
cost = ol.cost_from_golden(g)
u0 = torch.zeros(B, H, 1, device="cuda")
def solve(xs, us):
    return shooting_solve(eng, xs, us, cost, "euler", 0.02, lr, iters, track_best=True, u_min=-15.0, u_max=15.0, record_costs=True)
a = solve(x0, u0); b = solve(x0, u0)
print("big vs big: costs equal", torch.equal(a["costs"], b["costs"]), "u", torch.equal(a["u_last"], b["u_last"]))
for lo in (0, 40000, 65536 - 300):
    s = solve(x0[lo:lo + 300], u0[lo:lo + 300])
    d = (s["costs"] - a["costs"][:, lo:lo + 300]).abs()
    it = (d.amax(dim=1) > 0).nonzero().flatten().tolist()
    print("lo", lo, "costs equal", torch.equal(s["costs"], a["costs"][:, lo:lo + 300]), "first differing iters", it[:5],
          "n differing rollouts", int((d.amax(dim=0) > 0).sum()), "max abs", float(d.max()))
    s2 = solve(x0[lo:lo + 300], u0[lo:lo + 300])
    print("   small vs small equal", torch.equal(s["costs"], s2["costs"]))
# one-shot K1+K2 consistency
U = torch.tensor(rng.uniform(-5, 5, size=(B, H, 1)).astype(np.float32), device="cuda")
c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02); c, gu = c.clone(), gu.clone()
for lo in (0, 40000):
    c2, g2 = eng.rollout_cost_grad(x0[lo:lo + 300], U[lo:lo + 300], cost, "euler", 0.02)
    print("one-shot lo", lo, torch.equal(c2, c[lo:lo + 300]), torch.equal(g2, gu[lo:lo + 300]),
          "n diff grads", int((g2 != gu[lo:lo + 300]).sum()), float((g2 - gu[lo:lo + 300]).abs().max()))
