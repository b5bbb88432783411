"""Debug: which kernel is not bitwise repeatable at B=65536, H=100?"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd import _capi
from phnn_mpc_amd.engine import RolloutEngine
g, w = ol.load_golden("phnn_cartpole"), ol.load_weights("phnn_cartpole")
rng = np.random.default_rng(1234)
B, H = 65536, 100
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
U = torch.tensor(rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * 0.3, device="cuda")
cost = ol.cost_from_golden(g)
for label, kw, stash in (("f16x2 8 waves stash", {}, True), ("f16x2 8 waves recompute", {}, False), ("f16x2 4 waves stash", {"max_waves": 4}, True),
                         ("f32 8 waves stash", {"matmul": "f32"}, True), ("bf16x3 8 waves stash", {"matmul": "bf16x3"}, True)):
    eng = RolloutEngine(w, **kw)
    eng.use_stash = stash
    integ = 0
    traj = torch.empty(B, H + 1, 4, device="cuda"); cst = torch.empty(B, device="cuda"); gu = torch.empty(B, H, 1, device="cuda")
    nst = eng.workspace_bytes(B, H, integ) if stash else 0
    st = torch.empty(max(nst, 1), dtype=torch.uint8, device="cuda")
    sp = eng._p(st) if stash else None
    def k1():
        eng.lib.phnn_rollout_fwd(eng.h, eng._p(x0), eng._p(U), B, H, C.byref(cost), integ, 0.02, eng._p(cst), eng._p(traj), sp, eng._stream())
    def k2():
        eng.lib.phnn_rollout_grad(eng.h, eng._p(x0), eng._p(U), B, H, C.byref(cost), integ, 0.02, eng._p(traj), sp, eng._p(gu), None, eng._stream())
    k1(); torch.cuda.synchronize()
    c0, t0, s0 = cst.clone(), traj.clone(), st.clone()
    k2(); torch.cuda.synchronize()
    g0 = gu.clone()
    bad1 = bad1s = bad2 = 0
    for rep in range(15):
        k1(); torch.cuda.synchronize()
        bad1 += int((traj != t0).any(dim=2).any(dim=1).sum()); bad1s += int((st != s0).sum())
        traj.copy_(t0); st.copy_(s0)
        k2(); torch.cuda.synchronize()
        bad2 += int((gu != g0).any(dim=2).any(dim=1).sum())
    print(f"{label}: over 15 repeats: K1 rollouts with differing traj {bad1}, differing stash bytes {bad1s}, K2 rollouts with differing grad {bad2}", flush=True)
