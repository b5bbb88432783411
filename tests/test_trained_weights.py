"""Parity on TRAINED weights: weights_phnn_cartpole_trained.npz is the state_dict after 860
epochs of the reference's own training loop on its own dataset (tests/golden/make_trained_cartpole.py; the checkpoint
the reference's config names, models/checkpoint_epoch_860.pth, is not shipped).

What the goldens show about these weights: single evaluations (f, H, VJP) behave like the seed-0 ones, but rollouts under
random controls are ill-conditioned -- |d cost / d u| reaches 3e8 and the REFERENCE's float32 run differs from its
float64 run by up to 8e-2 in cost and several times the largest gradient entry.  A fixed tolerance is therefore
meaningless there; the yardstick is the reference's own float32-vs-float64 deviation, rollout by rollout:

    err(implementation vs reference float64)  <=  K * err(reference float32 vs reference float64) + stated tolerance

K = 4 for float32 arithmetic (CPU oracle here; K per matmul mode for the GPU in test_gpu_trained.py).
The closed-loop golden (cl_*) is the reference's own driver (scripts/run_cartpole_mpc.py:57-182) on these weights: with this
training run the reference does NOT stabilise the plant (the pole falls after 30 control steps), so the behavioural check
is that the drop-in controller reproduces that outcome, not that it balances.
"""
import numpy as np
import pytest

import oracle_lib as ol

K_F32 = 4.0


@pytest.fixture(scope="module")
def tb():
    g, w = ol.load_golden(ol.TRAINED), ol.load_weights(ol.TRAINED)
    return g, w, ol.OracleModel(w, "f64"), ol.OracleModel(w, "f32")


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


def test_point_sets_at_the_standard_tolerances(tb):
    g, w, m64, m32 = tb
    dx, H = m64.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f64"]) < 1e-11 and rel(H, g["fwd_H_f64"]) < 1e-11
    xb, ub = m64.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f64"]) < 1e-10 and rel(ub, g["vjp_ubar_f64"]) < 1e-10
    dx, H = m32.forward(g["fwd_x"], g["fwd_u"])
    assert rel(dx, g["fwd_dx_f64"]) < 2e-5 and rel(H, g["fwd_H_f64"]) < 2e-5
    xb, ub = m32.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(xb, g["vjp_xbar_f64"]) < 5e-5 and rel(ub, g["vjp_ubar_f64"]) < 5e-5


@pytest.mark.parametrize("integ", ["euler", "rk4"])
@pytest.mark.parametrize("case", ol.ROLL_CASES)
def test_rollouts_against_the_reference_noise_floor(tb, integ, case):
    g, w, m64, m32 = tb
    B, H = case
    key = f"roll_{integ}_B{B}_H{H}"
    cost = ol.cost_from_golden(g)
    fc, fg, gmax = ol.trained_rollout_floor(g, key)
    # float64 oracle: the restatement itself -- float64 rounding amplified by the rollout's own sensitivity
    r = m64.rollout(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
    assert np.all(np.abs(r["cost"] / g[key + "_cost_f64"] - 1) <= 1e-7 * np.maximum(fc / 1e-7, 1.0) * 1e-2 + 1e-11)
    assert np.all(np.abs(r["grad_u"] - g[key + "_gu_f64"]).max(axis=(1, 2)) / gmax <= 1e-2 * np.maximum(fg, 1e-9) + 1e-10)
    # float32 oracle against the reference's float64, measured in units of the reference's own float32 deviation
    r = m32.rollout(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
    ec = np.abs(r["cost"] / g[key + "_cost_f64"] - 1)
    eg = np.abs(r["grad_u"] - g[key + "_gu_f64"]).max(axis=(1, 2)) / gmax
    assert np.all(ec <= K_F32 * fc + 1e-5), (ec, fc)
    assert np.all(eg <= K_F32 * fg + 1e-4), (eg, fg)


def test_reference_closed_loop_outcome_is_recorded(tb):
    g = tb[0]
    assert int(g["train_epochs"]) == 860
    assert not bool(g["cl_stability_achieved"])  # this training run does not yield a stabilising model (see the module docstring)
    assert 5 < g["cl_controls"].shape[0] < 300 and abs(g["cl_states"][-1][1]) > 0.5  # the reference's loop ended on |theta| > 0.5
