"""HIP kernels for MLPs with SiLU / ReLU / ELU / GELU activations (src/NN.py:13 defaults to nn.SiLU; src/pHNN.py:41 resolves any nn.* by
name; src/baseline_node.py:49-58: relu, elu, gelu) against the reference's own outputs (tests/golden/make_golden_act.py) and the
float64 oracle: model(x,u), VJP, Euler / RK4 rollouts with cost and gradients, tape and recompute adjoints, and the
drop-in module classes built from a YAML that selects the activation.  Tolerances as for the Tanh models (stated in
tests/test_gpu_parity.py); ReLU gradients allow one unit's mask flipping where float32 rounds a pre-activation across 0."""
import os

import numpy as np
import pytest
import yaml

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
CASES = [(1, 20), (8, 50), (4, 100)]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


@pytest.fixture(scope="module", params=list(ol.ACT_MODELS))
def bundle(request, torch):
    from phnn_mpc_amd.engine import RolloutEngine
    name = request.param
    act = ol.ACT_MODELS[name]
    g, w = ol.load_golden(name), ol.load_weights(name)
    return name, act, g, ol.OracleModel(w, "f64", activation=act), RolloutEngine(w, activation=act)


def test_variant_and_point_sets(bundle):
    name, act, g, m64, eng = bundle
    assert act in eng.variant and "f16x2" not in eng.variant  # all-f32 kernels
    dx, H = eng.forward(g["fwd_x"], g["fwd_u"])
    assert rel(npy(dx), g["fwd_dx_f64"]) < 2e-5 and rel(npy(H), g["fwd_H_f64"]) < 2e-5
    xb, ub = eng.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    gtol = 2e-3 if act == "relu" else 1e-4
    assert rel(npy(xb), g["vjp_xbar_f64"]) < gtol and rel(npy(ub), g["vjp_ubar_f64"]) < gtol


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_rollouts_tape_and_recompute(bundle, integ):
    name, act, g, m64, eng = bundle
    cost = ol.cost_from_golden(g)
    gtol = 2e-3 if act == "relu" else 1e-4
    atol = 5e-5 if name.startswith("odefunc") else 1e-5
    for B, H in CASES:
        key = f"roll_{integ}_B{B}_H{H}"
        x0, U = g[key + "_x0"], g[key + "_U"]
        outs = []
        for stash in (True, False):
            eng.use_stash = stash
            try:
                c, gu, gx = eng.rollout_cost_grad(x0, U, cost, integ, float(g["dt"]), want_grad_x0=True)
                outs.append((npy(c), npy(gu), npy(gx)))
            finally:
                eng.use_stash = True
        _, tr = eng.rollout_cost(x0, U, cost, integ, float(g["dt"]), want_traj=True)
        assert np.allclose(npy(tr), g[key + "_traj_f64"], rtol=1e-5, atol=atol)
        gmax = np.abs(g[key + "_gu_f64"]).max(axis=(1, 2), keepdims=True)
        for c, gu, gx in outs:
            assert np.allclose(c, g[key + "_cost_f64"], rtol=1e-5)
            assert np.all(np.abs(gu - g[key + "_gu_f64"]) <= gtol * gmax)
            assert rel(gx, g[key + "_gx0_f64"]) < max(gtol, 2e-4)
        outside = (U > float(g["u_max"])) | (U < float(g["u_min"]))
        assert outside.any() and np.all(outs[0][1][outside] == 0.0)


def test_seeded_batch_vs_oracle_and_repeatable(bundle, torch):
    name, act, g, m64, eng = bundle
    rng = np.random.default_rng(17)
    n = eng.n
    B, H = 300, 30
    x0 = (rng.uniform(-1, 1, size=(B, n)) * ([1.0, 0.3, 0.5, 0.5] if n == 4 else [1.5, 0.8])).astype(np.float32)
    U = rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * float(g["u_max"]) * 0.5
    cost = ol.cost_from_golden(g)
    ref = m64.rollout(x0, U, cost, "euler", float(g["dt"]), nthreads=8)
    c1, g1 = [t.clone() for t in eng.rollout_cost_grad(x0, U, cost, "euler", float(g["dt"]))]
    c2, g2 = eng.rollout_cost_grad(x0, U, cost, "euler", float(g["dt"]))
    assert torch.equal(c1, c2) and torch.equal(g1, g2)
    assert np.allclose(npy(c1), ref["cost"], rtol=1e-5)
    gmax = np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True)
    frac_bad = np.mean(np.abs(npy(g1) - ref["grad_u"]) > 1e-4 * gmax)
    assert frac_bad <= (2e-3 if act == "relu" else 0.0)  # ReLU: a mask flip in a few rollouts at most


def test_drop_in_modules_select_the_activation_from_the_yaml(torch, tmp_path):
    """models.pHNN built from a YAML whose H_mlp / R_mlp say nn.SiLU runs the SiLU kernels; mixed activations and
    activations without kernels are refused with a clear error (the reference would accept any nn.Module)."""
    from phnn_mpc_amd.models import pHNN
    cfg = yaml.safe_load(open(CFG))
    for k in ("H_mlp", "R_mlp"):
        cfg["model"][k]["activation"] = "nn.SiLU"
    p = tmp_path / "silu.yaml"
    p.write_text(yaml.safe_dump(cfg))
    g, w = ol.load_golden("phnn_silu"), ol.load_weights("phnn_silu")
    m = pHNN(str(p))
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    x = torch.tensor(g["fwd_x"][:32], requires_grad=True)
    dx, H = m(x, torch.tensor(g["fwd_u"][:32]))
    assert "silu" in m.engine.variant
    assert rel(dx.detach().cpu().numpy(), g["fwd_dx_f64"][:32]) < 2e-5
    (dx * torch.tensor(g["vjp_lam"][:32])).sum().backward()  # input gradients through the plain VJP kernel
    assert x.grad is not None and torch.isfinite(x.grad).all()
    cfg["model"]["R_mlp"]["activation"] = "nn.Tanh"  # mixed
    p2 = tmp_path / "mixed.yaml"
    p2.write_text(yaml.safe_dump(cfg))
    with pytest.raises(NotImplementedError, match="ONE activation"):
        pHNN(str(p2)).engine
    for k in ("H_mlp", "R_mlp"):
        cfg["model"][k]["activation"] = "nn.Softplus"
    p3 = tmp_path / "softplus.yaml"
    p3.write_text(yaml.safe_dump(cfg))
    with pytest.raises(NotImplementedError):
        pHNN(str(p3)).engine


def test_odefunc_module_takes_the_reference_activation_names(torch):
    """models.ODEFunc(activation=...) (src/baseline_node.py:49-58: relu / tanh / elu / gelu) runs the matching kernels and
    reproduces the reference's own ODEFunc outputs; LayerNorm stays refused."""
    from phnn_mpc_amd.models import ODEFunc
    for name, n in (("odefunc_elu", 2), ("odefunc_gelu", 4)):
        act = ol.ACT_MODELS[name]
        g, w = ol.load_golden(name), ol.load_weights(name)
        f = ODEFunc(n, 1, activation=act)
        f.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
        f.current_action = torch.tensor(g["fwd_u"][:16])
        dx = f(0.0, torch.tensor(g["fwd_x"][:16]))
        assert act in f.engine.variant
        assert rel(dx.detach().cpu().numpy(), g["fwd_dx_f64"][:16]) < 2e-5
    with pytest.raises(NotImplementedError, match="LayerNorm"):
        ODEFunc(2, 1, activation="gelu", layer_norm=True).engine
