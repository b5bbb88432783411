"""Debug: which intermediate of the adjoint step first differs between two runs?  Uses the record-writing K2."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd.engine import RolloutEngine
w = ol.load_weights("phnn_cartpole")
rng = np.random.default_rng(1234)
B, H = 65536, 6
x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
U = torch.tensor(rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * 0.3, device="cuda")
tb = torch.tensor(rng.normal(size=(B, H + 1, 4)).astype(np.float32), device="cuda")
eng = RolloutEngine(w)
traj = eng.rollout_trajectory(x0, U, "euler", 0.02)
T, NB = 8, 5
VEC, REC = T * 256, 5 * T * 256 + 16 * 36
tiles = B // 16
recs = []
for r in range(4):
    eng._wg_ws = None
    g, gu, gx = eng.rollout_wgrad(x0, U, traj, "euler", 0.02, traj_bar=tb)
    torch.cuda.synchronize()
    ws = eng._wg_ws.view(torch.float32)[: tiles * H * REC].view(tiles, H, REC).clone()
    recs.append((ws, gu.clone(), g.clone()))
names = ["a2", "q1", "ad2", "qd", "hbR"]
a = recs[0][0]
for r in range(1, 4):
    b = recs[r][0]
    print("run", r, "grad_u equal", torch.equal(recs[0][1], recs[r][1]), "grad_theta equal", torch.equal(recs[0][2], recs[r][2]))
    for t in range(H - 1, -1, -1):
        line = []
        for v, nm in enumerate(names):
            d = (a[:, t, v * VEC:(v + 1) * VEC] != b[:, t, v * VEC:(v + 1) * VEC]).view(tiles, T, 64, 4)
            line.append(f"{nm}: tiles {int(d.any(dim=3).any(dim=2).any(dim=1).sum())}")
        sm = (a[:, t, NB * VEC:] != b[:, t, NB * VEC:]).view(tiles, 16, 36)
        fields = {"x": (0, 4), "v": (4, 8), "lam": (8, 12), "dH": (12, 16), "rbar": (16, 32)}
        line.append("small: " + ", ".join(f"{k} {int(sm[:, :, lo:hi].any(dim=2).any(dim=1).sum())}" for k, (lo, hi) in fields.items()))
        print("   t", t, " | ".join(line))
    # detail at the first processed step (t = H-1): which unit tiles / registers of the first differing vector
    t = H - 1
    for v, nm in enumerate(names):
        d = (a[:, t, v * VEC:(v + 1) * VEC] != b[:, t, v * VEC:(v + 1) * VEC]).view(tiles, T, 64, 4)
        if d.any():
            bt = d.any(dim=3).any(dim=2).any(dim=1).nonzero().flatten()
            print("   first step, vector", nm, ": unit-tile hist", d.any(dim=3).any(dim=2).sum(dim=0).tolist(),
                  "lanes-per-bad-tile", float(d.any(dim=3).sum(dim=2).sum(dim=1)[bt].float().mean()),
                  "wave-in-WG hist", torch.bincount(bt % 8, minlength=8).tolist())
            x, y = a[bt[0], t, v * VEC:(v + 1) * VEC].view(T, 64, 4), b[bt[0], t, v * VEC:(v + 1) * VEC].view(T, 64, 4)
            idx = (x != y).nonzero()[:6]
            for i in idx.tolist():
                print("      tile", int(bt[0]), "unit-tile/lane/reg", i, float(x[tuple(i)]), float(y[tuple(i)]))
            break
