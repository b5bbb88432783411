#!/usr/bin/env python3
"""Golden vectors for models whose MLPs use another activation than Tanh (src/NN.py:13 defaults to
nn.SiLU, src/pHNN.py:41 resolves any nn.* by name, src/baseline_node.py:49-58 offers relu / elu / gelu), produced by running the
REFERENCE.  Build container only (reference mounted read-only at /root/reference).

Models (seeded construction of the reference's own classes from the shipped cart-pole YAML with the activation changed):
    phnn_silu       pHNN, H_mlp / R_mlp activation nn.SiLU          phnn_relu      the same with nn.ReLU
    canonical_silu  pHNN_Canonical, H_mlp activation nn.SiLU        odefunc_relu   ODEFunc(2, 1, activation='relu')
    phnn_elu / phnn_gelu / canonical_elu / canonical_gelu: the same with nn.ELU / nn.GELU; odefunc_elu ODEFunc(2, 1, 'elu');
    odefunc_gelu ODEFunc(4, 1, 'gelu')
Per model: weights_<name>.npz (state_dict) and golden_<name>.npz with the sets of make_golden.py (G2 f(x,u), H; G3 VJPs;
G4 Euler / RK4 rollouts with cost, grad_u, grad_x0; G10 reverse pass with cotangents), float64 and float32.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_act.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(OUT, "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)  # chdir's to the reference and imports its modules
from pHNN import pHNN  # noqa: E402
from pHNN_canonical import pHNN_Canonical  # noqa: E402
from baseline_node import ODEFunc  # noqa: E402

torch.set_num_threads(1)
SCRATCH = os.path.join(ROOT, "build", "golden_scratch")
os.makedirs(SCRATCH, exist_ok=True)


def build(cls, act, seed):
    cfg = yaml.safe_load(open("cartpole_mpc_config.yaml"))
    for k in ("H_mlp", "R_mlp"):
        cfg["model"][k]["activation"] = act
    path = os.path.join(SCRATCH, f"cfg_{cls.__name__}_{act.split('.')[-1]}.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    torch.manual_seed(seed)
    return cls(path)


def main():
    CASES = [(1, 20), (8, 50), (4, 100)]
    Qc, Rc = np.diag([10.0, 200.0, 1.0, 10.0]), np.diag([0.01])
    xlo = np.array([-1.0, -0.3, -0.5, -0.5])
    only = set(sys.argv[1:])  # optional: regenerate only the named models
    for name, cls, act, seed in (("phnn_silu", pHNN, "nn.SiLU", 3), ("phnn_relu", pHNN, "nn.ReLU", 4),
                                 ("canonical_silu", pHNN_Canonical, "nn.SiLU", 5), ("phnn_elu", pHNN, "nn.ELU", 7),
                                 ("phnn_gelu", pHNN, "nn.GELU", 8), ("canonical_elu", pHNN_Canonical, "nn.ELU", 9),
                                 ("canonical_gelu", pHNN_Canonical, "nn.GELU", 10)):
        if only and name not in only:
            continue
        m = build(cls, act, seed)
        if cls is pHNN_Canonical:  # as make_golden.py: non-trivial mass-matrix / dissipation parameters
            with torch.no_grad():
                m.M_net.log_a.fill_(0.30)
                m.M_net.b.fill_(0.35)
                m.M_net.log_c.fill_(-0.20)
                m.R_diag_raw.copy_(torch.tensor([0.10, -0.40, 0.70, 0.25]))
        np.savez(os.path.join(OUT, f"weights_{name}.npz"), **mg.sd_numpy(m))
        blk = mg.model_block(name, m, 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0, xlo, -xlo, 5.0, 300 + seed, CASES)
        np.savez(os.path.join(OUT, f"golden_{name}.npz"), **blk)
        print(name, "done")
    # ODEFunc: (2,1) pendulum-sized with relu / elu, and the reference's default (4,1) with gelu (src/baseline_node.py:49-58)
    for name, n, act, seed in (("odefunc_relu", 2, "relu", 6), ("odefunc_elu", 2, "elu", 11), ("odefunc_gelu", 4, "gelu", 12)):
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        ode = ODEFunc(n, 1, activation=act)
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for mod in ode.modules():
                if isinstance(mod, nn.Linear):
                    mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
        np.savez(os.path.join(OUT, f"weights_{name}.npz"), **mg.sd_numpy(ode))
        if n == 2:
            blk = mg.model_block(name, mg.OdeAdapter(ode), 2, 1, 0.05, np.diag([10.0, 1.0]), Rc, np.zeros(2), -2.0, 2.0,
                                 np.array([-np.pi, -1.0]), np.array([np.pi, 1.0]), 2.0, 300 + seed, CASES)
        else:
            blk = mg.model_block(name, mg.OdeAdapter(ode), 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0, xlo, -xlo, 5.0,
                                 300 + seed, CASES)
        np.savez(os.path.join(OUT, f"golden_{name}.npz"), **blk)
        print(name, "done")


if __name__ == "__main__":
    main()
