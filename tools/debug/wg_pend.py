import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from phnn_mpc_amd import weights
from phnn_mpc_amd.engine import RolloutEngine
wg = ol.load_wgrad_golden()
for name in ("phnn_pendulum", "phnn_cartpole"):
    w = ol.load_weights(name)
    eng = RolloutEngine(w)
    pre = f"{name}/pt_"
    m64 = ol.OracleModel(w, "f64")
    for N in (48, 16, 5):
        ref = weights.unpack_grad_blob(w, m64.wgrad(wg[pre + "x"][:N], wg[pre + "u"][:N], wg[pre + "lam"][:N], wg[pre + "Hbar"][:N]))
        outs = []
        for rep in range(3):
            g, _, _ = eng.model_wgrad(wg[pre + "x"][:N], wg[pre + "u"][:N], wg[pre + "lam"][:N], wg[pre + "Hbar"][:N])
            outs.append(g.clone())
        named = {k: v.cpu().numpy() for k, v in eng.named_grads(outs[0]).items()}
        errs = {k: float(np.abs(named[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30)) for k in ref if np.abs(ref[k]).max() > 0}
        bad = {k: f"{v:.1e}" for k, v in errs.items() if v > 1e-4}
        print(name, "N", N, "repeatable", all(torch.equal(outs[0], o) for o in outs), "bad tensors", bad)
