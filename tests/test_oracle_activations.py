"""The CPU oracle with SiLU / ReLU MLPs against the reference's own outputs (tests/golden/make_golden_act.py): the
activation is part of the model description (src/NN.py:13 defaults to nn.SiLU; src/pHNN.py:41; src/baseline_node.py:49-58).
float64: the restatement itself; float32: the stated tolerances.  ReLU: a pre-activation that float32 rounds across 0
flips one unit's mask, so the float32 gradient checks allow the size of one unit's contribution."""
import numpy as np
import pytest

import oracle_lib as ol

CASES = [(1, 20), (8, 50), (4, 100)]


@pytest.fixture(scope="module", params=list(ol.ACT_MODELS))
def bundle(request):
    name = request.param
    act = ol.ACT_MODELS[name]
    g, w = ol.load_golden(name), ol.load_weights(name)
    return name, act, g, ol.OracleModel(w, "f64", activation=act), ol.OracleModel(w, "f32", activation=act)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def test_forward_and_vjp(bundle):
    name, act, g, m64, m32 = bundle
    dx, H = m64.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f64"]) < 1e-12 and rel(H, g["fwd_H_f64"]) < 1e-12
    xb, ub = m64.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f64"]) < 1e-11 and rel(ub, g["vjp_ubar_f64"]) < 1e-11
    dx, H = m32.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f32"]) < 2e-5 and rel(H, g["fwd_H_f32"]) < 2e-5
    xb, ub = m32.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f32"]) < (2e-3 if act == "relu" else 2e-5)
    assert rel(ub, g["vjp_ubar_f32"]) < (2e-3 if act == "relu" else 2e-5)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("case", CASES)
def test_rollouts(bundle, integ, case):
    name, act, g, m64, m32 = bundle
    B, H = case
    key = f"roll_{integ}_B{B}_H{H}"
    cost = ol.cost_from_golden(g)
    r = m64.rollout(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
    assert np.abs(r["traj"] - g[key + "_traj_f64"]).max() < 1e-10
    assert rel(r["cost"], g[key + "_cost_f64"]) < 1e-11
    assert rel(r["grad_u"], g[key + "_gu_f64"]) < 1e-9 and rel(r["grad_x0"], g[key + "_gx0_f64"]) < 1e-9
    r = m32.rollout(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
    assert np.allclose(r["cost"], g[key + "_cost_f64"], rtol=1e-5)
    gmax = np.abs(g[key + "_gu_f64"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(r["grad_u"] - g[key + "_gu_f64"]) <= (2e-3 if act == "relu" else 1e-4) * gmax)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_reverse_pass_with_cotangents(bundle, integ):
    name, act, g, m64, _ = bundle
    gu, gx = m64.rollout_vjp(g["tvjp_x0"], g["tvjp_U"], ol.cost_from_golden(g), integ, float(g["dt"]),
                             traj_bar=g["tvjp_traj_bar"], cost_bar=g["tvjp_cost_bar"])
    assert rel(gu, g[f"tvjp_{integ}_gu_f64"]) < 1e-9 and rel(gx, g[f"tvjp_{integ}_gx0_f64"]) < 1e-9
