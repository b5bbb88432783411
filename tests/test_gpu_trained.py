"""HIP path on the TRAINED cart-pole weights (tests/golden/make_trained_cartpole.py).

On these weights rollouts under random controls are ill-conditioned and the reference's own float32 results sit far from
its float64 ones (tests/test_trained_weights.py); the bar for every matmul mode of the kernels is therefore

    err(kernel vs reference float64)  <=  K_GPU * err(reference float32 vs reference float64) + stated tolerance

per rollout, with ONE K for the three modes (measured worst ratios on the GPU, tests/parity_margin_trained.py: f16x2 3.9 /
3.8, bf16x3 4.8 / 1.9, all-f32 2.0 / 4.2 for cost / gradient: the modes are indistinguishable from the reference's own
float32 noise, which is what decides that f16x2 stays the default on trained weights).  Single evaluations keep the
standard tolerances.  The controller (G5) and the closed loop are checked against the reference's own run on these
weights: same control, same cost history, same (non-stabilising) closed-loop outcome.
"""
import os

import numpy as np
import pytest
import yaml

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
K_GPU = 8.0
MODES = ["f16x2", "bf16x3", "f32"]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("mode", MODES)
def test_point_sets_standard_tolerances(torch, mode):
    from phnn_mpc_amd.engine import RolloutEngine
    g, w = ol.load_golden(ol.TRAINED), ol.load_weights(ol.TRAINED)
    eng = RolloutEngine(w, matmul=mode)
    dx, H = eng.forward(g["fwd_x"], g["fwd_u"])
    assert rel(npy(dx), g["fwd_dx_f64"]) < 2e-5 and rel(npy(H), g["fwd_H_f64"]) < 2e-5
    xb, ub = eng.vjp(g["vjp_x"], g["vjp_u"], g["vjp_lam"])
    assert rel(npy(xb), g["vjp_xbar_f64"]) < 1e-4 and rel(npy(ub), g["vjp_ubar_f64"]) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_golden_rollouts_against_the_reference_noise_floor(torch, mode):
    from phnn_mpc_amd.engine import RolloutEngine
    g, w = ol.load_golden(ol.TRAINED), ol.load_weights(ol.TRAINED)
    eng = RolloutEngine(w, matmul=mode)
    cost = ol.cost_from_golden(g)
    worst = [0.0, 0.0]
    for integ in ("euler", "rk4"):
        for B, H in ol.ROLL_CASES:
            key = f"roll_{integ}_B{B}_H{H}"
            fc, fg, gmax = ol.trained_rollout_floor(g, key)
            c, gu = eng.rollout_cost_grad(g[key + "_x0"], g[key + "_U"], cost, integ, float(g["dt"]))
            ec = np.abs(npy(c) / g[key + "_cost_f64"] - 1)
            eg = np.abs(npy(gu) - g[key + "_gu_f64"]).max(axis=(1, 2)) / gmax
            assert np.all(ec <= K_GPU * fc + 1e-5), (mode, key, ec, fc)
            assert np.all(eg <= K_GPU * fg + 1e-4), (mode, key, eg, fg)
            U = g[key + "_U"]
            outside = (U > float(g["u_max"])) | (U < float(g["u_min"]))
            assert np.all(npy(gu)[outside] == 0.0)  # clamp mask, exact
            worst = [max(worst[0], float((ec / (fc + 2.5e-6)).max())), max(worst[1], float((eg / (fg + 2.5e-5)).max()))]
    print(f"trained weights, {mode}: worst error / (reference float32 deviation + tol/4): cost {worst[0]:.2f}, grad {worst[1]:.2f}")


def test_controller_and_closed_loop_reproduce_the_reference_run(torch):
    """G5 on the trained weights and the reference's closed loop (scripts/run_cartpole_mpc.py:57-182 on CartPoleSimulator):
    same first control, same cost history, the same states until the reference's episode ends, the same ending."""
    from phnn_mpc_amd.closed_loop import BatchedCartPole, run_mpc_batch
    from phnn_mpc_amd.models import pHNN
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    g, w = ol.load_golden(ol.TRAINED), ol.load_weights(ol.TRAINED)
    cfg = yaml.safe_load(open(CFG))
    model = pHNN(CFG)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    c = create_mpc_from_config(model, cfg)
    u0 = c.compute_control(g["mpc_x0"].copy())
    costs = np.asarray(c.last_costs) if hasattr(c, "last_costs") else None
    print("compute_control on trained weights: ours %.6f, reference %.6f" % (float(u0[0]), float(g["mpc_u0"][0])))
    assert abs(float(u0[0]) - float(g["mpc_u0"][0])) <= 2e-4 * max(1.0, abs(float(g["mpc_u0"][0])))
    if costs is not None:
        assert np.allclose(costs, g["mpc_costs"], rtol=1e-4)
    T = g["cl_controls"].shape[0]
    out = run_mpc_batch(BatchedCartPole(cfg["cartpole"]["dt"]), c, g["cl_x0"][None, :], T + 5)
    st, ct = out["states"][:, 0], out["controls"][:, 0, 0]
    end = int(out["done_step"][0])
    head = 10
    dev_c = np.abs(ct[:4] - g["cl_controls"][:4]).max()
    dev_s = np.abs(st[: head + 1] - g["cl_states"][: head + 1]).max()
    print("closed loop on trained weights: first 4 controls within %.2e, first %d states within %.2e, ended at step %d "
          "(reference: %d)" % (dev_c, head, dev_s, end, T - 1))
    # The loop is unstable and its solves ill-conditioned (|d cost / d u| ~ 1e9 over the horizon): float32 differences --
    # between this build and the reference, or between two builds of these kernels that differ in one rounding -- are
    # amplified step by step.  Checked tightly where they have not grown yet (first controls, first ten states), and by
    # outcome afterwards: the pole falls (|theta| > 0.5 ends the episode) within a few steps of where the reference's did.
    assert dev_c < 2e-4 and dev_s < 2e-3
    assert end >= 0 and abs(end - (T - 1)) <= 3 and abs(st[end + 1][1]) > 0.5 and not bool(g["cl_stability_achieved"])
