"""GPU parity: the gfx950 kernels, called through the C-ABI, against (a) the float64 CPU oracle on the same
seeded inputs and (b) the golden vectors the reference itself produced.

Stated float32 tolerances (BASELINE.md section 3, 5-10x the reference's own f32-vs-f64 floor):
  cost        rtol 1e-5
  trajectory  atol 1e-5 + rtol 1e-5 for the cart-pole models; atol 5e-5 for the pendulum models, whose
              trajectories swing to |x| ~ 11 and where the reference's OWN float32 run already differs from
              its float64 run by 1.5e-5 (golden roll_euler_B2_H200: xH_f32 vs traj_f64)
  grad_u      <= 1e-4 * max|grad| per rollout
  f(x,u), VJP <= 2e-5 * max|.| over the batch
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

CASES = [(1, 20), (8, 50), (4, 100), (2, 200)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module", params=ol.MODELS)
def bundle(request, torch_cuda):
    from phnn_mpc_amd.engine import RolloutEngine
    name = request.param
    w = ol.load_weights(name)
    return name, ol.load_golden(name), ol.OracleModel(w, "f64"), RolloutEngine(w, "cuda:0")


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


TRAJ_ATOL = {"phnn_cartpole": 1e-5, "canonical_cartpole": 1e-5, "phnn_pendulum": 5e-5, "odefunc_pendulum": 5e-5,
             "odefunc_cartpole": 1e-5, "phnn_cartpole_odd": 1e-5, "phnn_cartpole_trained": 1e-5}


def assert_rollout_close(cost, traj, gu, gx0, ref_cost, ref_traj, ref_gu, ref_gx0, traj_atol=1e-5):
    assert np.allclose(cost, ref_cost, rtol=1e-5, atol=0), np.abs(cost / ref_cost - 1).max()
    if traj is not None:
        assert np.allclose(traj, ref_traj, rtol=1e-5, atol=traj_atol), np.abs(traj - ref_traj).max()
    gmax = np.abs(ref_gu).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(gu - ref_gu) <= 1e-4 * gmax), (np.abs(gu - ref_gu) / gmax).max()
    if gx0 is not None:
        xmax = np.abs(ref_gx0).max(axis=1, keepdims=True)
        assert np.all(np.abs(gx0 - ref_gx0) <= 1e-4 * xmax), (np.abs(gx0 - ref_gx0) / xmax).max()


def test_model_forward(bundle):
    name, g, m64, eng = bundle
    dx, H = eng.forward(g["fwd_x"], g["fwd_u"])
    rdx, rH = m64.forward(g["fwd_x"], g["fwd_u"])
    assert np.abs(npy(dx) - rdx).max() <= 2e-5 * np.abs(rdx).max()
    assert np.abs(npy(H) - rH).max() <= 2e-5 * max(1.0, np.abs(rH).max())
    # and against the reference's own outputs
    assert np.abs(npy(dx) - g["fwd_dx_f64"]).max() <= 2e-5 * np.abs(rdx).max()


def test_model_vjp(bundle):
    name, g, m64, eng = bundle
    xb, ub = eng.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert np.abs(npy(xb) - g["vjp_xbar_f64"]).max() <= 2e-5 * np.abs(g["vjp_xbar_f64"]).max()
    assert np.abs(npy(ub) - g["vjp_ubar_f64"]).max() <= 2e-5 * max(1e-30, np.abs(g["vjp_ubar_f64"]).max())


def test_ragged_batches(bundle):
    """Batch sizes that do not fill a 16-rollout wave tile or a workgroup (1, 15, 17, 130)."""
    name, g, m64, eng = bundle
    for B in (1, 15, 17, 130):
        x, u = g["fwd_x"][:B], g["fwd_u"][:B]
        dx, H = eng.forward(x, u)
        rdx, rH = m64.forward(x, u)
        assert dx.shape == (B, eng.n)
        assert np.abs(npy(dx) - rdx).max() <= 2e-5 * np.abs(rdx).max()


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("case", CASES)
def test_rollout_vs_golden(bundle, integ, case):
    name, g, m64, eng = bundle
    B, H = case
    key = f"roll_{integ}_B{B}_H{H}"
    cost = ol.cost_from_golden(g)
    c, gu, gx0 = eng.rollout_cost_grad(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]), want_grad_x0=True)
    c2, traj = eng.rollout_cost(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]), want_traj=True)
    assert np.array_equal(npy(c), npy(c2))
    assert_rollout_close(npy(c), npy(traj), npy(gu), npy(gx0), g[key + "_cost_f64"], g[key + "_traj_f64"],
                         g[key + "_gu_f64"], g[key + "_gx0_f64"], TRAJ_ATOL[name])
    U = g[key + "_U"]
    outside = (U > float(g["u_max"])) | (U < float(g["u_min"]))
    assert np.all(npy(gu)[outside] == 0.0)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_rollout_vs_oracle_seeded(bundle, integ):
    """A larger seeded batch (B=300, ragged against 16 and 128) checked against the float64 oracle."""
    name, g, m64, eng = bundle
    rng = np.random.default_rng(4242)
    n = eng.n
    B, H = 300, 30
    x0 = (rng.uniform(-1, 1, size=(B, n)) * np.array([1.0, 0.3, 0.5, 0.5][:n])).astype(np.float32)
    amp = 1.3 * float(g["u_max"])
    U = rng.uniform(-amp, amp, size=(B, H, 1)).astype(np.float32)
    cost = ol.cost_from_golden(g)
    ref = m64.rollout(x0, U, cost, integ, float(g["dt"]), nthreads=8)
    c, gu, gx0 = eng.rollout_cost_grad(x0, U, cost, integ, float(g["dt"]), want_grad_x0=True)
    _, traj = eng.rollout_cost(x0, U, cost, integ, float(g["dt"]), want_traj=True)
    assert_rollout_close(npy(c), npy(traj), npy(gu), npy(gx0), ref["cost"], ref["traj"], ref["grad_u"], ref["grad_x0"],
                         TRAJ_ATOL[name])


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_full_Q_target_and_barrier(torch_cuda, integ):
    from phnn_mpc_amd import _capi
    from phnn_mpc_amd.engine import RolloutEngine
    g = ol.load_golden("phnn_cartpole")
    w = ol.load_weights("phnn_cartpole")
    eng, m64 = RolloutEngine(w), ol.OracleModel(w, "f64")
    cost = ol.cost_from_golden(g, Q=g["fullq_Q"], x_target=g["fullq_xt"])
    c, gu, gx0 = eng.rollout_cost_grad(g["fullq_x0"], g["fullq_U"], cost, integ, 0.02, want_grad_x0=True)
    assert_rollout_close(npy(c), None, npy(gu), npy(gx0), g[f"fullq_{integ}_cost_f64"], None,
                         g[f"fullq_{integ}_gu_f64"], g[f"fullq_{integ}_gx0_f64"])
    # soft state barrier (src/mpc_controller.py:96-107) against the oracle
    costb = _capi.make_cost(4, 1, g["Q"], g["R"], None, -15.0, 15.0, x_min=[-0.3, -0.1, -0.3, -0.2],
                            x_max=[0.3, 0.1, 0.25, 0.15])
    ref = m64.rollout(g["fullq_x0"], g["fullq_U"], costb, integ, 0.02)
    c, gu, gx0 = eng.rollout_cost_grad(g["fullq_x0"], g["fullq_U"], costb, integ, 0.02, want_grad_x0=True)
    assert_rollout_close(npy(c), None, npy(gu), npy(gx0), ref["cost"], None, ref["grad_u"], ref["grad_x0"])


def test_dataset_windows(torch_cuda):
    """Realistic magnitudes (G8): windows cut from the reference's own training data."""
    from phnn_mpc_amd.engine import RolloutEngine
    with np.load(ol.GOLDEN + "/golden_dataset_windows.npz") as z:
        win = {k: z[k] for k in z.files}
    gc = ol.load_golden("phnn_cartpole")
    for nm, wn in (("phnn", "phnn_cartpole"), ("canonical", "canonical_cartpole")):
        eng = RolloutEngine(ol.load_weights(wn))
        c, gu = eng.rollout_cost_grad(win["x0"], win["U"], ol.cost_from_golden(gc), "euler", 0.02)
        assert np.allclose(npy(c), win[f"{nm}_cost_f64"], rtol=1e-5), np.abs(npy(c) / win[f"{nm}_cost_f64"] - 1).max()
        gmax = np.abs(win[f"{nm}_gu_f64"]).max(axis=(1, 2), keepdims=True)
        assert np.all(np.abs(npy(gu) - win[f"{nm}_gu_f64"]) <= 1e-4 * gmax), (np.abs(npy(gu) - win[f"{nm}_gu_f64"]) / gmax).max()


def test_repeatable_and_shard_equivalent(bundle):
    """Bitwise run-to-run repeatability, and per-rollout results independent of how the batch is split
    (what makes the multi-GPU sharding exact)."""
    name, g, m64, eng = bundle
    rng = np.random.default_rng(7)
    B, H, n = 257, 25, eng.n
    x0 = (rng.uniform(-1, 1, size=(B, n)) * 0.3).astype(np.float32)
    U = rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32)
    cost = ol.cost_from_golden(g)
    c1, g1 = [npy(t).copy() for t in eng.rollout_cost_grad(x0, U, cost, "euler", float(g["dt"]))]
    c2, g2 = [npy(t).copy() for t in eng.rollout_cost_grad(x0, U, cost, "euler", float(g["dt"]))]
    assert np.array_equal(c1, c2) and np.array_equal(g1, g2)
    ca, ga = [npy(t).copy() for t in eng.rollout_cost_grad(x0[:100], U[:100], cost, "euler", float(g["dt"]))]
    cb, gb = [npy(t).copy() for t in eng.rollout_cost_grad(x0[100:], U[100:], cost, "euler", float(g["dt"]))]
    assert np.array_equal(np.concatenate([ca, cb]), c1) and np.array_equal(np.concatenate([ga, gb]), g1)


def test_errors(torch_cuda):
    from phnn_mpc_amd.engine import PhnnError, RolloutEngine
    w = ol.load_weights("phnn_cartpole")
    eng = RolloutEngine(w)
    g = ol.load_golden("phnn_cartpole")
    with pytest.raises(ValueError):
        eng.rollout_cost(g["fwd_x"][:4], np.zeros((4, 5, 1)), ol.cost_from_golden(g), "leapfrog", 0.02)
    bad = dict(w)  # narrower layers are zero-padded to a kernel width; wider than 128 has no kernel
    bad["H_net.net.2.weight"] = np.zeros((160, 128), np.float32)
    bad["H_net.net.2.bias"] = np.zeros(160, np.float32)
    bad["H_net.net.4.weight"] = np.zeros((1, 160), np.float32)
    with pytest.raises(PhnnError, match="exceeds the widest kernel"):
        RolloutEngine(bad)
    deep = dict(w)  # a third hidden layer in R_net: no kernel
    deep["R_net.net.2.weight"] = np.zeros((128, 128), np.float32)
    deep["R_net.net.2.bias"] = np.zeros(128, np.float32)
    deep["R_net.net.4.weight"] = np.zeros((16, 128), np.float32)
    deep["R_net.net.4.bias"] = np.zeros(16, np.float32)
    with pytest.raises(PhnnError):
        RolloutEngine(deep)


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_stash_and_recompute_modes_agree(bundle, integ):
    """K2 fed by K1's activation stash (default; RK4: four stage tapes + stage states per step) vs K2 recomputing the
    forward tape(s): same costs, same gradients."""
    name, g, m64, eng = bundle
    rng = np.random.default_rng(21)
    B, H, n = 333, 40, eng.n
    x0 = (rng.uniform(-1, 1, size=(B, n)) * 0.4).astype(np.float32)
    U = rng.uniform(-1, 1, size=(B, H, 1)).astype(np.float32) * float(g["u_max"])
    cost = ol.cost_from_golden(g)
    assert eng.use_stash and 0 < eng.workspace_bytes(B, H, "euler") < eng.workspace_bytes(B, H, "rk4")
    c1, g1, x1 = [npy(t).copy() for t in eng.rollout_cost_grad(x0, U, cost, integ, float(g["dt"]), want_grad_x0=True)]
    eng.use_stash = False
    try:
        c2, g2, x2 = [npy(t).copy() for t in eng.rollout_cost_grad(x0, U, cost, integ, float(g["dt"]), want_grad_x0=True)]
    finally:
        eng.use_stash = True
    assert np.array_equal(c1, c2)
    gmax = np.abs(g2).max(axis=(1, 2), keepdims=True)
    # ODEFunc keeps its tape as 24-bit fixed point (absolute error 6e-8 per tanh output): agreement to 5e-6 instead of 1e-6
    tol = 5e-6 if name.startswith("odefunc") else 1e-6
    assert np.all(np.abs(g1 - g2) <= tol * gmax) and np.allclose(x1, x2, rtol=1e-5, atol=tol * np.abs(x2).max())
    ref = m64.rollout(x0, U, cost, integ, float(g["dt"]), nthreads=8)
    assert np.all(np.abs(g2 - ref["grad_u"]) <= 1e-4 * np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True))
    assert np.all(np.abs(g1 - ref["grad_u"]) <= 1e-4 * np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True))


@pytest.mark.parametrize("name", ["phnn_cartpole", "canonical_cartpole"])
def test_matmul_variants(torch_cuda, name):
    """The 128x128 products run as an exact 2-way f16 split on the matrix pipe by default (matmul="f16x2");
    bf16x3 is the 3-way bf16 split, f32 the all-f32-MFMA kernels.  Each must sit within the stated tolerances of
    the float64 oracle.  The mode is an explicit phnn_options field (phnn_create_ex), not hidden global state."""
    from phnn_mpc_amd.engine import RolloutEngine
    g, w = ol.load_golden(name), ol.load_weights(name)
    m64 = ol.OracleModel(w, "f64")
    rng = np.random.default_rng(31)
    B, H = 200, 60
    x0 = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
    U = rng.uniform(-17, 17, size=(B, H, 1)).astype(np.float32)
    cost = ol.cost_from_golden(g)
    ref = m64.rollout(x0, U, cost, "euler", 0.02, nthreads=8)
    res = {}
    for mode in ("f16x2", "bf16x3", "f32"):
        eng = RolloutEngine(w, matmul=mode)
        assert eng.matmul_mode == mode
        c, gu, gx = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02, want_grad_x0=True)
        _, tr = eng.rollout_cost(x0, U, cost, "euler", 0.02, want_traj=True)
        assert_rollout_close(npy(c), npy(tr), npy(gu), npy(gx), ref["cost"], ref["traj"], ref["grad_u"], ref["grad_x0"])
        res[mode] = (npy(c), npy(gu))
    assert not np.array_equal(res["f32"][1], res["bf16x3"][1])  # they really are different kernels
    assert not np.array_equal(res["f16x2"][1], res["bf16x3"][1])
    assert np.allclose(res["f32"][0], res["bf16x3"][0], rtol=2e-6) and np.allclose(res["f32"][0], res["f16x2"][0], rtol=4e-6)


def test_f16x2_scaling_with_extreme_weights(torch_cuda):
    """Weights far outside f16's comfortable range (output layer x 2e4, hidden layer x 6, tiny first layer): the
    power-of-two scales of the f16x2 path (S on W2, Sb on the backward-type operands, per-rollout normalisation of
    the Hessian-vector input) must keep model(x,u) and its VJP at f32 accuracy."""
    from phnn_mpc_amd.engine import RolloutEngine
    g = ol.load_golden("phnn_cartpole")
    w = dict(ol.load_weights("phnn_cartpole"))
    w["H_net.net.4.weight"] = w["H_net.net.4.weight"] * np.float32(2.0e4)
    w["H_net.net.2.weight"] = w["H_net.net.2.weight"] * np.float32(6.0)
    w["H_net.net.0.weight"] = w["H_net.net.0.weight"] * np.float32(0.05)
    eng, m64 = RolloutEngine(w), ol.OracleModel(w, "f64")
    assert eng.matmul_mode == "f16x2"
    x, u = g["vjp_x"], g["vjp_u"]
    lam = g["vjp_lam"] * np.float32(1.0e-6)  # tiny and, below, huge cotangents
    for scale in (1.0, 1.0e9):
        l2 = (lam * np.float32(scale)).astype(np.float32)
        xb, ub = eng.vjp(x, u, l2)
        rxb, rub = m64.vjp(x, u, l2)
        assert np.abs(npy(xb) - rxb).max() <= 3e-5 * np.abs(rxb).max()
        assert np.abs(npy(ub) - rub).max() <= 3e-5 * max(np.abs(rub).max(), 1e-30)
    dx, H = eng.forward(g["fwd_x"], g["fwd_u"])
    rdx, rH = m64.forward(g["fwd_x"], g["fwd_u"])
    assert np.abs(npy(dx) - rdx).max() <= 3e-5 * np.abs(rdx).max()
    assert np.abs(npy(H) - rH).max() <= 3e-5 * np.abs(rH).max()


def _random_phnn_weights(rng, n, m, hH, hR, hG):
    """state_dict-shaped random weights for a pHNN with the given hidden sizes (hG None -> fixed G)."""
    def lin(o, i):
        b = 1.0 / np.sqrt(i)
        return rng.uniform(-b, b, size=(o, i)).astype(np.float32), rng.uniform(-b, b, size=(o,)).astype(np.float32)
    w = {"J": rng.normal(size=(n, n)).astype(np.float32)}
    dims = [n] + list(hH) + [1]
    for k in range(len(dims) - 1):
        w[f"H_net.net.{2 * k}.weight"], w[f"H_net.net.{2 * k}.bias"] = lin(dims[k + 1], dims[k])
    dims = [n] + list(hR) + [n * n]
    for k in range(len(dims) - 1):
        w[f"R_net.net.{2 * k}.weight"], w[f"R_net.net.{2 * k}.bias"] = lin(dims[k + 1], dims[k])
    if hG is None:
        w["G_fixed"] = rng.normal(size=(n, m)).astype(np.float32)
    else:
        dims = [n] + list(hG) + [n * m]
        for k in range(len(dims) - 1):
            w[f"G_net.net.{2 * k}.weight"], w[f"G_net.net.{2 * k}.bias"] = lin(dims[k + 1], dims[k])
    return w


@pytest.mark.parametrize("shape", [(4, (100, 128), (72,), (90,)), (2, (128, 96), (128,), (80,)), (2, (70, 65), (33,), None),
                                   (4, (40, 64), (64,), None), (2, (24, 48), (16,), (20,))])
def test_other_phnn_shapes_vs_oracle(torch_cuda, shape):
    """(n, G fixed|learned, hidden widths) combinations beyond the shipped configs -- zero-padded to a kernel width --
    against the float64 oracle (which is pinned to the reference on the shipped shapes and generic in the sizes)."""
    from phnn_mpc_amd import _capi
    from phnn_mpc_amd.engine import RolloutEngine
    n, hH, hR, hG = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    w = _random_phnn_weights(rng, n, 1, hH, hR, hG)
    eng, m64 = RolloutEngine(w), ol.OracleModel(w, "f64")
    x = (rng.uniform(-1, 1, size=(40, n)) * 0.8).astype(np.float32)
    u = rng.uniform(-2, 2, size=(40, 1)).astype(np.float32)
    lam = rng.normal(size=(40, n)).astype(np.float32)
    dx, H = eng.forward(x, u)
    rdx, rH = m64.forward(x, u)
    assert np.abs(npy(dx) - rdx).max() <= 2e-5 * np.abs(rdx).max() and np.abs(npy(H) - rH).max() <= 2e-5 * max(1.0, np.abs(rH).max())
    xb, ub = eng.vjp(x, u, lam)
    rxb, rub = m64.vjp(x, u, lam)
    assert np.abs(npy(xb) - rxb).max() <= 3e-5 * np.abs(rxb).max() and np.abs(npy(ub) - rub).max() <= 3e-5 * np.abs(rub).max()
    cost = _capi.make_cost(n, 1, np.linspace(1.0, 5.0, n), [0.02], None, -3.0, 3.0)
    x0, U = x[:33] * 0.5, rng.uniform(-4, 4, size=(33, 25, 1)).astype(np.float32)
    for integ in ("euler", "rk4"):
        ref = m64.rollout(x0, U, cost, integ, 0.01, nthreads=8)
        c, gu, gx = eng.rollout_cost_grad(x0, U, cost, integ, 0.01, want_grad_x0=True)
        assert_rollout_close(npy(c), None, npy(gu), npy(gx), ref["cost"], None, ref["grad_u"], ref["grad_x0"])


def test_default_matmul_mode_by_width(torch_cuda):
    """128-wide models default to the f16x2 products, 64-wide ones to all-f32.  f16x2 on a 64-wide model is known to
    exceed the stated tolerance in long rollouts of the trained pendulum model: phnn_create_ex refuses it unless the
    caller forces it."""
    import os
    from phnn_mpc_amd.engine import PhnnError, RolloutEngine
    assert "PHNN_MATMUL" not in os.environ
    assert RolloutEngine(ol.load_weights("phnn_cartpole")).matmul_mode == "f16x2"
    assert RolloutEngine(ol.load_weights("phnn_pendulum")).matmul_mode == "f32"
    with pytest.raises(PhnnError, match="force_matmul"):
        RolloutEngine(ol.load_weights("phnn_pendulum"), matmul="f16x2")
    eng = RolloutEngine(ol.load_weights("phnn_pendulum"), matmul="f16x2", force_matmul=True)
    assert eng.matmul_mode == "f16x2"
    g = ol.load_golden("phnn_pendulum")
    dx, H = eng.forward(g["fwd_x"], g["fwd_u"])
    assert np.abs(npy(dx) - g["fwd_dx_f64"]).max() <= 2e-5 * np.abs(g["fwd_dx_f64"]).max()
