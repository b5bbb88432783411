"""Pin the CPU oracle (oracle/phnn_oracle.c) against golden vectors produced by the reference itself
(tests/golden/make_golden.py, sets G2-G4, G7, G8 of SURVEY.md 8c).

The float64 oracle must reproduce the reference-in-double to round-off; the float32 oracle must agree with
the reference-in-float32 within float32 noise (the reference's own f32-vs-f64 floor is ~1e-6 relative).
"""
import numpy as np
import pytest

import oracle_lib as ol

CASES = [(1, 20), (8, 50), (4, 100), (2, 200)]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module", params=ol.MODELS)
def bundle(request):
    name = request.param
    w = ol.load_weights(name)
    return name, ol.load_golden(name), ol.OracleModel(w, "f64"), ol.OracleModel(w, "f32")


def test_forward_f64(bundle):
    name, g, m64, _ = bundle
    dx, H = m64.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f64"]) < 1e-12
    assert np.abs(H - g["fwd_H_f64"]).max() < 1e-12


def test_forward_f32(bundle):
    name, g, _, m32 = bundle
    dx, H = m32.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f32"]) < 5e-6
    assert np.abs(H - g["fwd_H_f32"]).max() < 5e-6 * max(1.0, np.abs(g["fwd_H_f32"]).max())


def test_vjp_f64(bundle):
    name, g, m64, _ = bundle
    xb, ub = m64.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f64"]) < 1e-11
    assert rel(ub, g["vjp_ubar_f64"]) < 1e-11


def test_vjp_f32(bundle):
    name, g, _, m32 = bundle
    xb, ub = m32.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f32"]) < 2e-5
    assert rel(ub, g["vjp_ubar_f32"]) < 2e-5


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("case", CASES)
def test_rollout_f64(bundle, integ, case):
    name, g, m64, _ = bundle
    B, H = case
    key = f"roll_{integ}_B{B}_H{H}"
    r = m64.rollout(g[key + "_x0"], g[key + "_U"], ol.cost_from_golden(g), integ, float(g["dt"]))
    assert np.abs(r["traj"] - g[key + "_traj_f64"]).max() < 1e-10
    assert rel(r["cost"], g[key + "_cost_f64"]) < 1e-11
    assert rel(r["grad_u"], g[key + "_gu_f64"]) < 1e-9
    assert rel(r["grad_x0"], g[key + "_gx0_f64"]) < 1e-9
    # clamp mask: entries pushed outside the bounds have exactly zero gradient, the one ON the bound does not
    U = g[key + "_U"]
    outside = (U > float(g["u_max"])) | (U < float(g["u_min"]))
    assert outside.any() and np.all(r["grad_u"][outside] == 0.0)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("case", CASES)
def test_rollout_f32(bundle, integ, case):
    name, g, _, m32 = bundle
    B, H = case
    key = f"roll_{integ}_B{B}_H{H}"
    r = m32.rollout(g[key + "_x0"], g[key + "_U"], ol.cost_from_golden(g), integ, float(g["dt"]))
    # tolerance stated in BASELINE.md: cost rtol 1e-5, trajectory atol 1e-5 (+rtol 1e-5), grad 1e-4*max|grad|
    assert np.allclose(r["cost"], g[key + "_cost_f32"], rtol=1e-5, atol=0)
    assert np.allclose(r["traj"][:, -1], g[key + "_xH_f32"], rtol=1e-5, atol=1e-5)
    gmax = np.abs(g[key + "_gu_f32"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(r["grad_u"] - g[key + "_gu_f32"]) <= 1e-4 * gmax)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_full_nonsymmetric_Q_and_target(integ):
    g = ol.load_golden("phnn_cartpole")
    m64 = ol.OracleModel(ol.load_weights("phnn_cartpole"), "f64")
    cost = ol.cost_from_golden(g, Q=g["fullq_Q"], x_target=g["fullq_xt"])
    r = m64.rollout(g["fullq_x0"], g["fullq_U"], cost, integ, 0.02)
    assert rel(r["cost"], g[f"fullq_{integ}_cost_f64"]) < 1e-11
    assert rel(r["grad_u"], g[f"fullq_{integ}_gu_f64"]) < 1e-9
    assert rel(r["grad_x0"], g[f"fullq_{integ}_gx0_f64"]) < 1e-9


def test_pendulum_anchor_g7():
    g = ol.load_golden("phnn_pendulum")
    w = ol.load_weights("phnn_pendulum")
    cost = ol.cost_from_golden(g)
    for integ in ("euler", "rk4"):
        for prec, tol in (("f64", 1e-12), ("f32", 2e-6)):
            m = ol.OracleModel(w, prec)
            r = m.rollout(np.array([[0.5, 0.1]], np.float32), np.zeros((1, 10, 1)), cost, integ, 0.05)
            assert np.abs(r["traj"] - g[f"g7_{integ}_traj_{prec}"]).max() < tol


def test_dataset_windows_g8():
    """Realistic magnitudes cut from data/cartpole_training_data.pt (states up to +-11, controls +-14.5)."""
    with np.load(ol.GOLDEN + "/golden_dataset_windows.npz") as z:
        win = {k: z[k] for k in z.files}
    gc = ol.load_golden("phnn_cartpole")
    for nm, wn in (("phnn", "phnn_cartpole"), ("canonical", "canonical_cartpole")):
        m64 = ol.OracleModel(ol.load_weights(wn), "f64")
        r = m64.rollout(win["x0"], win["U"], ol.cost_from_golden(gc), "euler", 0.02)
        ok = win[f"{nm}_finite_f64"]
        assert ok.all()
        assert rel(r["cost"], win[f"{nm}_cost_f64"]) < 1e-10
        assert rel(r["grad_u"], win[f"{nm}_gu_f64"]) < 1e-8


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_rollout_vjp_with_trajectory_and_cost_cotangents_g10(bundle, integ):
    """G10: loss = <W, traj> + <w, cost> differentiated by the reference's autograd (float64)."""
    name, g, m64, _ = bundle
    gu, gx = m64.rollout_vjp(g["tvjp_x0"], g["tvjp_U"], ol.cost_from_golden(g), integ, float(g["dt"]),
                             traj_bar=g["tvjp_traj_bar"], cost_bar=g["tvjp_cost_bar"])
    assert rel(gu, g[f"tvjp_{integ}_gu_f64"]) < 1e-9
    assert rel(gx, g[f"tvjp_{integ}_gx0_f64"]) < 1e-9


def test_torch_restatement_matches_golden():
    """oracle/torch_oracle.py (the stock PyTorch-CPU restatement timed as bench.py's second CPU baseline) against the
    reference's own float64 rollouts (G4, Euler cases)."""
    import sys
    import torch
    sys.path.insert(0, ol.ORACLE_DIR)
    from torch_oracle import TorchPhnn
    g, w = ol.load_golden("phnn_cartpole"), ol.load_weights("phnn_cartpole")
    m = TorchPhnn(w, torch.float64)
    for B, H in ((8, 50), (4, 100)):
        key = f"roll_euler_B{B}_H{H}"
        c, gu = m.rollout_cost_grad(g[key + "_x0"], g[key + "_U"], np.diag(g["Q"]) if g["Q"].ndim == 2 else g["Q"],
                                    float(np.asarray(g["R"]).reshape(-1)[0]), g["x_target"], float(g["u_min"]),
                                    float(g["u_max"]), float(g["dt"]))
        # summation order of the cost differs from the reference's Python loops: 1e-9, not bitwise
        assert np.allclose(c.numpy(), g[key + "_cost_f64"], rtol=1e-8)
        assert np.allclose(gu.numpy(), g[key + "_gu_f64"], rtol=1e-6, atol=1e-8 * np.abs(g[key + "_gu_f64"]).max())


# ----------------------------------------------------------------------------- training side: G14-G16
WG_MODELS = ["phnn_cartpole", "canonical_cartpole", "phnn_pendulum", "odefunc_pendulum"]


@pytest.fixture(scope="module")
def wg():
    return ol.load_wgrad_golden()


def assert_param_grads(named, g, prefix, tag, rtol, what):
    """every parameter of the reference's named_parameters(): |ours - ref| <= rtol * max|ref| of that tensor
    (an all-zero reference gradient -- buffers, autograd constants, unused rows -- must be exactly zero)."""
    keys = [k for k in g if k.startswith(prefix + "g.") and k.endswith("_" + tag)]
    assert keys, prefix
    for k in keys:
        name = k[len(prefix) + 2:-len(tag) - 1]
        ref = np.asarray(g[k], np.float64)
        ours = np.asarray(named[name], np.float64).reshape(ref.shape)
        mx = np.abs(ref).max()
        if mx == 0:
            assert np.all(ours == 0), (what, name)
        else:
            assert np.abs(ours - ref).max() <= rtol * mx, (what, name, np.abs(ours - ref).max() / mx)


@pytest.mark.parametrize("name", WG_MODELS)
def test_oracle_point_wgrad_g14(wg, name):
    from phnn_mpc_amd import weights
    w = ol.load_weights(name)
    m = ol.OracleModel(w, "f64")
    pre = f"{name}/pt_"
    g = m.wgrad(wg[pre + "x"], wg[pre + "u"], wg[pre + "lam"], wg[pre + "Hbar"])
    assert_param_grads(weights.unpack_grad_blob(w, g), wg, pre, "f64", 1e-10, name)
    # buffers / autograd constants keep zero gradient in the blob
    named = weights.unpack_grad_blob(w, g)
    for k in ("G_fixed", "G", "M_net.log_a", "M_net.b", "M_net.log_c"):
        if k in named:
            assert np.all(named[k] == 0)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("name", WG_MODELS[:3])
def test_oracle_rollout_wgrad_g16(wg, name, integ):
    from phnn_mpc_amd import weights
    w = ol.load_weights(name)
    m = ol.OracleModel(w, "f64")
    pre = f"{name}/rw_"
    r = m.rollout_wgrad(wg[pre + "x0"], wg[pre + "U"], integ, float(wg[pre + "dt"]), wg[pre + "traj_bar"], wg[pre + "dx_bar"])
    assert np.allclose(r["traj"], wg[f"{pre}{integ}_traj_f64"], rtol=1e-10, atol=1e-12)
    assert np.allclose(r["dX"], wg[f"{pre}{integ}_dX_f64"], rtol=1e-10, atol=1e-12)
    assert np.abs(r["grad_u"] - wg[f"{pre}{integ}_gu_f64"]).max() <= 1e-10 * np.abs(wg[f"{pre}{integ}_gu_f64"]).max()
    assert np.abs(r["grad_x0"] - wg[f"{pre}{integ}_gx0_f64"]).max() <= 1e-10 * np.abs(wg[f"{pre}{integ}_gx0_f64"]).max()
    assert_param_grads(weights.unpack_grad_blob(w, r["grad_theta"]), wg, f"{pre}{integ}_", "f64", 1e-9, (name, integ))


# ----------------------------------------------------------------------------- two control inputs (m = 2), row f2
@pytest.mark.parametrize("fname,name,nu", ol.MULTI_INPUT_MODELS)
def test_oracle_several_inputs(fname, name, nu):
    from phnn_mpc_amd import _capi, weights
    g, ws = ol.load_named_golden(fname)
    w = ws[name]
    m = ol.OracleModel(w, "f64")
    assert (m.n, m.m) == (4, nu)
    dx, H = m.forward(g[f"{name}/x"], g[f"{name}/u"])
    assert np.allclose(dx, g[f"{name}/fwd_dx"], rtol=1e-10, atol=1e-12) and np.allclose(H, g[f"{name}/fwd_H"], rtol=1e-10, atol=1e-12)
    xb, ub = m.vjp(g[f"{name}/x"], g[f"{name}/u"], g[f"{name}/lam"])
    assert ub.shape == (64, nu)
    assert np.allclose(xb, g[f"{name}/vjp_xbar"], rtol=1e-9, atol=1e-11) and np.allclose(ub, g[f"{name}/vjp_ubar"], rtol=1e-9, atol=1e-11)
    cost = _capi.make_cost(4, nu, g[f"{name}/Q"], g[f"{name}/R"], None, -10.0, 10.0)
    for integ in ("euler", "rk4"):
        r = m.rollout(g[f"{name}/roll_x0"], g[f"{name}/roll_U"], cost, integ, 0.02)
        assert np.allclose(r["traj"], g[f"{name}/roll_{integ}_traj"], rtol=1e-9, atol=1e-11)
        assert np.allclose(r["cost"], g[f"{name}/roll_{integ}_cost"], rtol=1e-9)
        ref = g[f"{name}/roll_{integ}_gu"]
        assert r["grad_u"].shape == ref.shape == (6, 30, nu)
        assert np.abs(r["grad_u"] - ref).max() <= 1e-9 * np.abs(ref).max()
        assert np.abs(r["grad_x0"] - g[f"{name}/roll_{integ}_gx0"]).max() <= 1e-9 * np.abs(g[f"{name}/roll_{integ}_gx0"]).max()
        U = g[f"{name}/roll_U"]
        assert np.all(r["grad_u"][(U > 10.0) | (U < -10.0)] == 0.0)
    gt = m.wgrad(g[f"{name}/x"], g[f"{name}/u"], g[f"{name}/lam"], g[f"{name}/Hbar"])
    named = weights.unpack_grad_blob(w, gt)
    for k in [k for k in g if k.startswith(f"{name}/pt_g.")]:
        ref = g[k]
        ours = named[k.split("pt_g.", 1)[1]].reshape(ref.shape)
        assert np.abs(ours - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30), k


# ----------------------------------------------------------------------------- MassMatrixNetwork (row f2)
@pytest.mark.parametrize("name", ol.MASS_TYPES)
def test_oracle_mass_matrix_network(name):
    """pHNN_Canonical with MassMatrixNetwork constant / diagonal / full (src/mass_matrix.py:15-216)."""
    from phnn_mpc_amd import _capi, weights
    g, ws = ol.load_named_golden("golden_mass.npz")
    w = ws[name]
    d, _ = weights.pack_state_dict(w)
    assert d.mass_type == {"constant": 1, "diagonal": 2, "full": 3}[name]
    m = ol.OracleModel(w, "f64")
    dx, H = m.forward(g[f"{name}/x"], g[f"{name}/u"])
    assert np.allclose(dx, g[f"{name}/fwd_dx"], rtol=1e-9, atol=1e-11) and np.allclose(H, g[f"{name}/fwd_H"], rtol=1e-10, atol=1e-12)
    xb, ub = m.vjp(g[f"{name}/x"], g[f"{name}/u"], g[f"{name}/lam"])
    assert np.abs(xb - g[f"{name}/vjp_xbar"]).max() <= 1e-8 * np.abs(g[f"{name}/vjp_xbar"]).max()
    assert np.abs(ub - g[f"{name}/vjp_ubar"]).max() <= 1e-8 * np.abs(g[f"{name}/vjp_ubar"]).max()
    cost = _capi.make_cost(4, 1, [10.0, 200.0, 1.0, 10.0], [0.01], None, -15.0, 15.0)
    for integ in ("euler", "rk4"):
        r = m.rollout(g[f"{name}/roll_x0"], g[f"{name}/roll_U"], cost, integ, 0.02)
        assert np.allclose(r["traj"], g[f"{name}/roll_{integ}_traj"], rtol=1e-8, atol=1e-10)
        assert np.allclose(r["cost"], g[f"{name}/roll_{integ}_cost"], rtol=1e-8)
        for a, b in ((r["grad_u"], g[f"{name}/roll_{integ}_gu"]), (r["grad_x0"], g[f"{name}/roll_{integ}_gx0"])):
            assert np.abs(a - b).max() <= 1e-7 * np.abs(b).max()
    gt = m.wgrad(g[f"{name}/x"], g[f"{name}/u"], g[f"{name}/lam"], g[f"{name}/Hbar"])
    named = weights.unpack_grad_blob(w, gt)
    for k in [k for k in g if k.startswith(f"{name}/pt_g.")]:
        ref = g[k]
        ours = named[k.split("pt_g.", 1)[1]].reshape(ref.shape)
        assert np.abs(ours - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-30), (k, np.abs(ours - ref).max(), np.abs(ref).max())
