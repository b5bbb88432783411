"""Debug: count K2 (stash) non-repeatable rollouts for the library in PHNN_LIB_PATH."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd.engine import RolloutEngine
name = sys.argv[1] if len(sys.argv) > 1 else "phnn_cartpole"
g, w = ol.load_golden("phnn_cartpole"), ol.load_weights(name)
rng = np.random.default_rng(1234)
cost = ol.cost_from_golden(g)
B, H = 65536, 50
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
U = torch.tensor(rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * 0.3, device="cuda")
eng = RolloutEngine(w, **({"matmul": os.environ["MM"]} if "MM" in os.environ else {}))
traj = torch.empty(B, H + 1, 4, device="cuda"); cst = torch.empty(B, device="cuda"); gu = torch.empty(B, H, 1, device="cuda")
st = torch.empty(eng.workspace_bytes(B, H, 0), dtype=torch.uint8, device="cuda")
eng.lib.phnn_rollout_fwd(eng.h, eng._p(x0), eng._p(U), B, H, C.byref(cost), 0, 0.02, eng._p(cst), eng._p(traj), eng._p(st), eng._stream())
out = []
for mode, sp in (("stash", eng._p(st)), ("recompute", None)):
    runs = []
    for _ in range(8):
        eng.lib.phnn_rollout_grad(eng.h, eng._p(x0), eng._p(U), B, H, C.byref(cost), 0, 0.02, eng._p(traj), sp, eng._p(gu), None, eng._stream())
        torch.cuda.synchronize(); runs.append(gu.clone())
    ref = torch.stack(runs).median(dim=0).values
    out.append(mode + " " + str([int(((r - ref).abs().squeeze(-1).amax(dim=1) > 0).sum()) for r in runs]))
print(os.path.basename(os.environ.get("PHNN_LIB_PATH", "product")), eng.variant, "bad rollouts per run:", " | ".join(out), flush=True)
