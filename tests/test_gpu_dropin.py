"""GPU tests of the drop-in Python surface (model classes, integrators, controllers) and of the remaining C-ABI
entry points (general reverse pass, Adam kernel), against the reference's golden outputs and the CPU oracle.
"""
import os

import numpy as np
import pytest
import yaml

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
CFG_PEND = os.path.join(ROOT, "configs", "pendulum.yaml")


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def ctl():
    with np.load(os.path.join(ol.GOLDEN, "golden_controllers.npz")) as z:
        return {k: z[k] for k in z.files}


def _load(cls, cfg, name, torch):
    m = cls(cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in ol.load_weights(name).items()})
    return m


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


def test_model_classes_forward_backward(torch):
    from phnn_mpc_amd.models import ODEFunc, pHNN, pHNN_Canonical
    for cls, cfg, name in ((pHNN, CFG, "phnn_cartpole"), (pHNN_Canonical, CFG, "canonical_cartpole"),
                           (pHNN, CFG_PEND, "phnn_pendulum")):
        g = ol.load_golden(name)
        m = _load(cls, cfg, name, torch)
        x = torch.tensor(g["vjp_x"], requires_grad=True)  # CPU tensors in, CPU tensors out (as the reference)
        u = torch.tensor(g["vjp_u"], requires_grad=True)
        res = m(x, u)
        assert res[0].device.type == "cpu" and (len(res) == 3) == (cls is pHNN_Canonical)
        (res[0] * torch.tensor(g["vjp_lam"])).sum().backward()
        assert np.abs(npy(x.grad) - g["vjp_xbar_f64"]).max() <= 2e-5 * np.abs(g["vjp_xbar_f64"]).max()
        assert np.abs(npy(u.grad) - g["vjp_ubar_f64"]).max() <= 2e-5 * np.abs(g["vjp_ubar_f64"]).max()
        with torch.no_grad():  # unlike the reference, no requires_grad / grad mode is needed
            dx, H = m(torch.tensor(g["fwd_x"]), torch.tensor(g["fwd_u"]))[:2]
        assert np.abs(npy(dx) - g["fwd_dx_f64"]).max() <= 2e-5 * np.abs(g["fwd_dx_f64"]).max()
        assert np.abs(npy(H) - g["fwd_H_f64"]).max() <= 2e-5 * max(1.0, np.abs(g["fwd_H_f64"]).max())
    g = ol.load_golden("odefunc_pendulum")
    o = _load(lambda _: ODEFunc(2, 1), None, "odefunc_pendulum", torch)
    o.current_action = torch.tensor(g["fwd_u"])
    dx = o(0.0, torch.tensor(g["fwd_x"]))
    assert np.abs(npy(dx) - g["fwd_dx_f64"]).max() <= 2e-5 * np.abs(g["fwd_dx_f64"]).max()


def test_integrator_functions(torch):
    from phnn_mpc_amd import integrators as I
    from phnn_mpc_amd.models import pHNN
    name = "phnn_cartpole"
    g = ol.load_golden(name)
    m = _load(pHNN, CFG, name, torch)
    m64 = ol.OracleModel(ol.load_weights(name), "f64")
    x, u = torch.tensor(g["fwd_x"][:32]), torch.tensor(g["fwd_u"][:32])
    k1 = m64.forward(g["fwd_x"][:32], g["fwd_u"][:32])[0]
    assert np.abs(npy(I.euler_step(m, x, u, 0.02)) - (g["fwd_x"][:32] + 0.02 * k1)).max() < 2e-6
    # rk4_step against a one-step oracle rollout (which also returns the trajectory)
    zero = ol.cost_from_golden(g)
    zero.has_u_bounds = 0
    r = m64.rollout(g["fwd_x"][:32], g["fwd_u"][:32, None, :], zero, "rk4", 0.02)
    assert np.abs(npy(I.rk4_step(m, x, u, 0.02)) - r["traj"][:, 1]).max() < 2e-6
    y, H = I.rk4_step_with_energy(m, x, u, 0.02)
    assert np.abs(npy(H) - g["fwd_H_f64"][:32]).max() < 2e-5 and np.abs(npy(y) - r["traj"][:, 1]).max() < 2e-6
    # compare_integrators (src/integrators.py:261-308): both rollouts, their distance and the energy drifts
    U3 = torch.tensor(g["fwd_u"][:32, None, :]).repeat(1, 3, 1)
    cmp_ = I.compare_integrators(m, x, U3, 0.02)
    et, ee = I.rollout_trajectory(m, x, U3, 0.02, "euler")
    rt, re = I.rollout_trajectory(m, x, U3, 0.02, "rk4")
    assert torch.equal(cmp_["euler_trajectory"], et) and torch.equal(cmp_["rk4_trajectory"], rt)
    assert torch.allclose(cmp_["trajectory_difference"], torch.norm(et - rt, dim=-1)) and cmp_["trajectory_difference"].shape == (32, 4)
    assert torch.allclose(cmp_["euler_energy_drift"], (ee[:, -1] - ee[:, 0]).abs()) and torch.allclose(cmp_["rk4_energy_drift"], (re[:, -1] - re[:, 0]).abs())
    with pytest.raises(ValueError):
        I.rollout_trajectory(m, x, torch.zeros(32, 3, 1), 0.02, "leapfrog")
    with pytest.raises(ValueError):
        I.rollout_trajectory_differentiable(m, x, torch.zeros(32, 3, 1), 0.02, "verlet")


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_rollout_trajectory_differentiable_and_g10(torch, integ):
    """Trajectory vs the reference's own (G4), backward through it vs the reference's autograd (G10)."""
    from phnn_mpc_amd import integrators as I
    from phnn_mpc_amd.models import ODEFunc, pHNN, pHNN_Canonical
    for cls, cfg, name in ((pHNN, CFG, "phnn_cartpole"), (pHNN_Canonical, CFG, "canonical_cartpole"),
                           (pHNN, CFG_PEND, "phnn_pendulum"), (lambda _: ODEFunc(2, 1), None, "odefunc_pendulum")):
        g = ol.load_golden(name)
        m = _load(cls, cfg, name, torch)
        dt, umin, umax = float(g["dt"]), float(g["u_min"]), float(g["u_max"])
        key = f"roll_{integ}_B8_H50"
        U = torch.clamp(torch.tensor(g[key + "_U"]), umin, umax)
        traj = I.rollout_trajectory_differentiable(m, torch.tensor(g[key + "_x0"]), U, dt, integ)
        assert traj.shape == (8, 51, m.engine.n)
        assert np.allclose(npy(traj), g[key + "_traj_f64"], rtol=1e-5, atol=5e-5 if "pendulum" in name else 1e-5)
        # backward: loss = <W, traj> (+ nothing on the cost) -> compare with G10 restricted to cost_bar = 0 via oracle
        y0 = torch.tensor(g["tvjp_x0"], requires_grad=True)
        Ur = torch.tensor(g["tvjp_U"], requires_grad=True)
        traj = I.rollout_trajectory_differentiable(m, y0, torch.clamp(Ur, umin, umax), dt, integ)
        (traj * torch.tensor(g["tvjp_traj_bar"])).sum().backward()
        m64 = ol.OracleModel(ol.load_weights(name), "f64")
        gu, gx = m64.rollout_vjp(g["tvjp_x0"], g["tvjp_U"], ol.cost_from_golden(g), integ, dt,
                                 traj_bar=g["tvjp_traj_bar"], cost_bar=np.zeros(4))
        assert np.abs(npy(Ur.grad) - gu).max() <= 1e-4 * np.abs(gu).max()
        assert np.abs(npy(y0.grad) - gx).max() <= 1e-4 * np.abs(gx).max()
        # full G10 (trajectory AND cost cotangents) through the C-ABI entry point
        eng = m.engine
        cost = ol.cost_from_golden(g)
        _, tr = eng.rollout_cost(g["tvjp_x0"], g["tvjp_U"], cost, integ, dt, want_traj=True)
        gu2, gx2 = eng.rollout_vjp(g["tvjp_x0"], g["tvjp_U"], tr, cost, integ, dt, traj_bar=g["tvjp_traj_bar"],
                                   cost_bar=g["tvjp_cost_bar"])
        ref_gu, ref_gx = g[f"tvjp_{integ}_gu_f64"], g[f"tvjp_{integ}_gx0_f64"]
        assert np.abs(npy(gu2) - ref_gu).max() <= 1e-4 * np.abs(ref_gu).max()
        assert np.abs(npy(gx2) - ref_gx).max() <= 1e-4 * np.abs(ref_gx).max()


def test_energy_outputs_and_quirk9(torch):
    from phnn_mpc_amd import integrators as I
    from phnn_mpc_amd.models import pHNN
    g = ol.load_golden("phnn_pendulum")
    m = _load(pHNN, CFG_PEND, "phnn_pendulum", torch)
    x0, U = torch.tensor(g["roll_rk4_B8_H50_x0"]), torch.clamp(torch.tensor(g["roll_rk4_B8_H50_U"]), -2.0, 2.0)
    traj, en = I.rollout_trajectory(m, x0, U, 0.05, "rk4")
    m64 = ol.OracleModel(ol.load_weights("phnn_pendulum"), "f64")
    Href = m64.forward(npy(traj).reshape(-1, 2), np.zeros((8 * 51, 1)))[1].reshape(8, 51)
    assert np.abs(npy(en) - Href).max() < 5e-5 * max(1.0, np.abs(Href).max())
    traj2, en2 = I.rollout_trajectory_differentiable(m, x0, U, 0.05, "rk4", return_energies=True)
    assert np.array_equal(npy(en2[:, 0]), npy(en2[:, 1]))  # energies[1] duplicates energies[0]
    assert np.allclose(npy(en2[:, 2:]), npy(en[:, 1:-1]))


def test_controllers_on_gpu_match_reference(torch, ctl):
    from phnn_mpc_amd.models import pHNN, pHNN_Canonical
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    from phnn_mpc_amd.mpc_controller_canonical import create_mpc_controller
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_from_config(_load(pHNN, CFG, "phnn_cartpole", torch), cfg)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert u0.shape == (1,) and abs(u0[0] - ctl["mpc_u0"][0]) < 1e-4, (u0, ctl["mpc_u0"])
    out = c.solve_batch(ctl["mpc_x0"][None], record_costs=True)
    assert np.allclose(npy(out["costs"][:, 0]), ctl["mpc_costs"], rtol=1e-5)
    st = c.rollout_dynamics(ctl["mpc_x0"], np.zeros((20, 1), np.float32))
    assert np.allclose(npy(st), ctl["mpc_zero_states"], atol=1e-5)
    cc = create_mpc_controller(_load(pHNN_Canonical, CFG, "canonical_cartpole", torch), cfg)
    u_a, info_a = cc.control(ctl["mpc_x0"].copy(), None)
    assert abs(u_a[0] - ctl["can_u_a"][0]) < 2e-4
    assert np.allclose(info_a["optimization"]["costs"], ctl["can_costs_a"], rtol=1e-5)
    assert np.allclose(info_a["u_sequence"], ctl["can_useq_a"], atol=5e-4)
    u_b, info_b = cc.control(ctl["can_x_b"].copy(), ctl["can_useq_a"])
    assert abs(u_b[0] - ctl["can_u_b"][0]) < 5e-4
    assert np.allclose(info_b["optimization"]["costs"], ctl["can_costs_b"], rtol=1e-5)
    # many plants at once == one at a time
    rng = np.random.default_rng(5)
    X = (rng.uniform(-1, 1, size=(40, 4)) * [0.5, 0.2, 0.3, 0.3]).astype(np.float32)
    ub, seq, best = cc.control_batch(X)
    for b in (0, 17, 39):
        u1, info = cc.control(X[b])
        assert np.array_equal(u1, ub[b]) and np.array_equal(info["u_sequence"], seq[b])


def test_graphed_solve_is_bit_identical(torch, ctl):
    """use_graph: the whole solve replayed as one HIP graph == the eager launch sequence, bit for bit, also when the
    graph is replayed with new inputs (closed loop) and re-captured for another batch size."""
    import time
    from phnn_mpc_amd.models import pHNN, pHNN_Canonical
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    from phnn_mpc_amd.mpc_controller_canonical import create_mpc_controller
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_from_config(_load(pHNN, CFG, "phnn_cartpole", torch), cfg)
    cc = create_mpc_controller(_load(pHNN_Canonical, CFG, "canonical_cartpole", torch), cfg)
    rng = np.random.default_rng(11)
    X = (rng.uniform(-1, 1, size=(3, 40, 4)) * [0.5, 0.2, 0.3, 0.3]).astype(np.float32)
    eager = [c.compute_control_batch(X[k]) for k in range(3)] + [c.compute_control(X[0, 0])]
    eager_c = [cc.control_batch(X[k]) for k in range(2)]
    c.use_graph = cc.use_graph = True
    graphed = [c.compute_control_batch(X[k]) for k in range(3)] + [c.compute_control(X[0, 0])]
    graphed_c = [cc.control_batch(X[k]) for k in range(2)]
    for a, b in zip(eager, graphed):
        assert np.array_equal(a, b)
    for a, b in zip(eager_c, graphed_c):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    u_a, info_a = cc.control(ctl["mpc_x0"].copy(), None)  # warm-start path through the graph, against the golden run
    assert abs(u_a[0] - ctl["can_u_a"][0]) < 2e-4 and np.allclose(info_a["optimization"]["costs"], ctl["can_costs_a"], rtol=1e-5)
    # latency of the reference's own use (one plant per call), eager vs graph -- printed, not asserted
    for flag in (False, True):
        c.use_graph = flag
        c.compute_control(X[0, 0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10):
            c.compute_control(X[0, k])
        torch.cuda.synchronize()
        print("compute_control, one plant, H=%d, %d iterations, graph=%s: %.2f ms" % (
            c.horizon, c.max_iterations, flag, (time.perf_counter() - t0) * 100))


def test_adam_kernel_matches_torch_adam_order(torch):
    """K3 against the float32 Adam restatement of the oracle AND against torch.optim.Adam itself on the CPU."""
    from phnn_mpc_amd.engine import RolloutEngine
    eng = RolloutEngine(ol.load_weights("phnn_cartpole"))
    rng = np.random.default_rng(2)
    p0 = rng.normal(size=(64, 20, 1)).astype(np.float32)
    grads = [rng.normal(size=p0.shape).astype(np.float32) * s for s in (1.0, 0.1, 3.0, 1e-3, 2.0)]
    u = torch.tensor(p0, device="cuda")
    m, v = torch.zeros_like(u), torch.zeros_like(u)
    pt = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([pt], lr=0.015)
    for k, g in enumerate(grads):
        eng.adam_step(u, torch.tensor(g, device="cuda"), m, v, 0.015, k + 1)
        pt.grad = torch.tensor(g)
        opt.step()
    assert np.abs(npy(u) - pt.detach().numpy()).max() < 2e-6


def test_full_size_properties(torch):
    """BASELINE config sizes (B=65536, H=50): finite everywhere, chunk-consistent (a slice of the big batch equals
    that slice run alone, bitwise), clamp mask exact, and a 192-rollout sample checked against the f64 oracle."""
    from phnn_mpc_amd.engine import RolloutEngine
    g = ol.load_golden("phnn_cartpole")
    w = ol.load_weights("phnn_cartpole")
    eng = RolloutEngine(w)
    rng = np.random.default_rng(1234)
    B, H = 65536, 50
    x0 = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
    U = rng.uniform(-18, 18, size=(B, H, 1)).astype(np.float32)
    cost = ol.cost_from_golden(g)
    c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02)
    c, gu = npy(c), npy(gu)
    assert np.isfinite(c).all() and np.isfinite(gu).all()
    assert np.all(gu[np.abs(U) > 15.0] == 0.0)
    lo, hi = 31000, 31700
    c2, g2 = eng.rollout_cost_grad(x0[lo:hi], U[lo:hi], cost, "euler", 0.02)
    assert np.array_equal(npy(c2), c[lo:hi]) and np.array_equal(npy(g2), gu[lo:hi])
    idx = rng.choice(B, size=192, replace=False)
    ref = ol.OracleModel(w, "f64").rollout(x0[idx], U[idx], cost, "euler", 0.02, nthreads=8)
    assert np.allclose(c[idx], ref["cost"], rtol=1e-5)
    gmax = np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(gu[idx] - ref["grad_u"]) <= 1e-4 * gmax)


def test_edge_sizes(torch):
    """Empty batch, one-step horizon, a batch of one: shapes and values."""
    from phnn_mpc_amd.engine import RolloutEngine
    g = ol.load_golden("phnn_cartpole")
    w = ol.load_weights("phnn_cartpole")
    eng, m64 = RolloutEngine(w), ol.OracleModel(w, "f64")
    cost = ol.cost_from_golden(g)
    c, gu = eng.rollout_cost_grad(np.zeros((0, 4), np.float32), np.zeros((0, 7, 1), np.float32), cost, "euler", 0.02)
    assert c.shape == (0,) and gu.shape == (0, 7, 1)
    dx, H = eng.forward(np.zeros((0, 4), np.float32), np.zeros((0, 1), np.float32))
    assert dx.shape == (0, 4) and H.shape == (0,)
    x0, U = g["fwd_x"][:3], g["fwd_u"][:3, None, :]
    for integ in ("euler", "rk4"):
        ref = m64.rollout(x0, U, cost, integ, 0.02)
        c, gu, gx = eng.rollout_cost_grad(x0, U, cost, integ, 0.02, want_grad_x0=True)
        assert np.allclose(npy(c), ref["cost"], rtol=1e-5)
        assert np.allclose(npy(gu), ref["grad_u"], rtol=1e-4, atol=1e-5 * np.abs(ref["grad_u"]).max())
        assert np.allclose(npy(gx), ref["grad_x0"], rtol=1e-4, atol=1e-5 * np.abs(ref["grad_x0"]).max())


def test_million_rollouts_properties(torch):
    """BASELINE config 4's global batch (B = 2^20, H = 50) on one GPU: finite, slice-consistent with small runs."""
    from phnn_mpc_amd.engine import RolloutEngine
    g = ol.load_golden("phnn_cartpole")
    eng = RolloutEngine(ol.load_weights("phnn_cartpole"))
    gen = torch.Generator(device="cuda").manual_seed(5)
    B, H = 1 << 20, 50
    x0 = (torch.rand(B, 4, device="cuda", generator=gen) * 2 - 1) * torch.tensor([1.0, 0.3, 0.5, 0.5], device="cuda")
    U = (torch.rand(B, H, 1, device="cuda", generator=gen) * 2 - 1) * 5.0
    cost = ol.cost_from_golden(g)
    ws = {}
    c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02, workspace=ws)
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(gu).all())
    for lo in (0, 524288 - 100, B - 300):
        c2, g2 = eng.rollout_cost_grad(x0[lo:lo + 300], U[lo:lo + 300], cost, "euler", 0.02)
        assert torch.equal(c2, c[lo:lo + 300]) and torch.equal(g2, gu[lo:lo + 300])
    idx = torch.tensor([3, 77777, 555555, B - 1], device="cuda")
    ref = ol.OracleModel(ol.load_weights("phnn_cartpole"), "f64").rollout(npy(x0[idx]), npy(U[idx]), cost, "euler", 0.02)
    assert np.allclose(npy(c[idx]), ref["cost"], rtol=1e-5)


def test_batched_closed_loop_on_gpu(torch):
    """Rows f1/f3: 64 plants driven at once (one batched solve per control step) == plants driven one at a time."""
    from phnn_mpc_amd.closed_loop import BatchedCartPole, run_mpc_batch
    from phnn_mpc_amd.models import pHNN_Canonical
    from phnn_mpc_amd.mpc_controller_canonical import create_mpc_controller
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_controller(_load(pHNN_Canonical, CFG, "canonical_cartpole", torch), cfg)
    c.optimizer_steps = 8
    rng = np.random.default_rng(6)
    X0 = rng.uniform(-1, 1, size=(64, 4)) * [0.2, 0.08, 0.1, 0.1]
    out = run_mpc_batch(BatchedCartPole(0.02), c, X0, 6)
    assert np.isfinite(out["states"]).all() and out["controls"].shape == (6, 64, 1)
    assert np.all(np.abs(out["controls"]) <= 15.0)
    for b in (0, 63):
        one = run_mpc_batch(BatchedCartPole(0.02), c, X0[b:b + 1], 6)
        assert np.array_equal(one["controls"][:, 0], out["controls"][:, b])
        assert np.array_equal(one["states"][:, 0], out["states"][:, b])
