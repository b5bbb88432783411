"""Debug: parity margin of the library in PHNN_LIB_PATH -- worst gradient / cost error against the float64 oracle as a
fraction of the stated tolerances (grad 1e-4 max|grad| per rollout, cost rtol 1e-5), cart-pole pHNN, seeded batch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd.engine import RolloutEngine
name = sys.argv[1] if len(sys.argv) > 1 else "phnn_cartpole"
g, w = ol.load_golden(name if name != "phnn_cartpole_odd" else "phnn_cartpole"), ol.load_weights(name)
m64 = ol.OracleModel(w, "f64")
eng = RolloutEngine(w, split="never")  # whole-tile kernels (the ones a variant build of the adjoint unit changes)
cost = ol.cost_from_golden(g)
rng = np.random.default_rng(7)
for B, H, amp in ((256, 50, 5.0), (256, 100, 15.0), (64, 200, 10.0)):
    x0 = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
    U = rng.uniform(-amp, amp, size=(B, H, 1)).astype(np.float32)
    ref = m64.rollout(x0, U, cost, "euler", 0.02, nthreads=8)
    c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02)
    c, gu = c.cpu().numpy().astype(np.float64), gu.cpu().numpy().astype(np.float64)
    ge = (np.abs(gu - ref["grad_u"]).max(axis=(1, 2)) / np.abs(ref["grad_u"]).max(axis=(1, 2))).max() / 1e-4
    ce = np.abs(c / ref["cost"] - 1).max() / 1e-5
    print(f"{os.path.basename(os.environ.get('PHNN_LIB_PATH', 'product'))} B={B} H={H} amp={amp}: grad error {ge:.3f} of tol, cost error {ce:.3f} of tol", flush=True)
