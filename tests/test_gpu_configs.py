"""GPU tests at the BASELINE.json configuration sizes that round 1 left untested, and of the golden sets that were
only checked through the CPU oracle (G9 soft barrier, G13 L-BFGS branch), all through the C-ABI.

  config 3   pHNN cart-pole, Euler, H=100, B=65536, 20 Adam iterations on the controls from the zero start
             (src/mpc_controller.py:164-209): finite, slice-consistent, and a 64-rollout sample of every iterate's
             cost and of the final controls against the float64 oracle driven through the same 20 Adam steps.
  config 5   ODEFunc(2,1), classic RK4, H=200, B=65536, K1 + K2: same properties, 64-rollout oracle sample.
  G9 / G13   MPCController with the soft state barrier / with optimizer_type='LBFGS' on the HIP path.
"""
import os

import numpy as np
import pytest
import yaml

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def ctl():
    with np.load(os.path.join(ol.GOLDEN, "golden_controllers.npz")) as z:
        return {k: z[k] for k in z.files}


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _load(cls, cfg, name, torch):
    m = cls(cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in ol.load_weights(name).items()})
    return m


def oracle_adam_solve(m64, x0, cost, integ, dt, H, lr, iters):
    """The reference's optimisation loop (src/mpc_controller.py:164-209) in float64 on the CPU oracle: zero start,
    `iters` x (rollout + cost + gradient, Adam step).  -> (costs (iters,B), u_last (B,H,1))"""
    B = x0.shape[0]
    u = np.zeros((B, H, 1))
    mom, vel = np.zeros_like(u), np.zeros_like(u)
    costs = np.empty((iters, B))
    for k in range(iters):
        r = m64.rollout(x0, u, cost, integ, dt, traj=False, nthreads=8)
        costs[k] = r["cost"]
        g = np.ascontiguousarray(r["grad_u"])
        m64.adam(u, g, mom, vel, lr, k + 1)
    return costs, u


def test_config3_adam_solve_full_size(torch):
    """BASELINE config 3 at full size."""
    from phnn_mpc_amd.engine import RolloutEngine
    from phnn_mpc_amd.solver import shooting_solve
    g, w = ol.load_golden("phnn_cartpole"), ol.load_weights("phnn_cartpole")
    eng = RolloutEngine(w)
    rng = np.random.default_rng(1234)
    B, H, iters, lr = 65536, 100, 20, 0.015
    x0h = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
    x0 = torch.tensor(x0h, device="cuda")
    cost = ol.cost_from_golden(g)
    u0 = torch.zeros(B, H, 1, device="cuda")
    out = shooting_solve(eng, x0, u0, cost, "euler", 0.02, lr, iters, track_best=True, u_min=-15.0, u_max=15.0,
                         record_costs=True)
    costs, u_last, best_cost = out["costs"], out["u_last"], out["best_cost"]
    assert costs.shape == (iters, B) and bool(torch.isfinite(costs).all()) and bool(torch.isfinite(u_last).all())
    assert bool((best_cost <= costs[0]).all()) and torch.equal(best_cost, costs.min(dim=0).values)
    assert float(costs[-1].mean()) < float(costs[0].mean())  # Adam does descend on average
    # slice consistency: a 300-rollout slice solved alone == the same rows of the big solve, bitwise
    lo = 40000
    sub = shooting_solve(eng, x0[lo:lo + 300], u0[lo:lo + 300], cost, "euler", 0.02, lr, iters, track_best=True,
                         u_min=-15.0, u_max=15.0, record_costs=True)
    assert torch.equal(sub["costs"], costs[:, lo:lo + 300]) and torch.equal(sub["u_last"], u_last[lo:lo + 300])
    assert torch.equal(sub["best_u"], out["best_u"][lo:lo + 300])
    # 64-rollout sample against the float64 oracle driven through the same 20 Adam steps
    idx = rng.choice(B, size=64, replace=False)
    rc, ru = oracle_adam_solve(ol.OracleModel(w, "f64"), x0h[idx].astype(np.float64), cost, "euler", 0.02, H, lr, iters)
    gc = npy(costs)[:, idx]
    assert np.allclose(gc, rc, rtol=1e-5), np.abs(gc / rc - 1).max()
    # controls after 20 steps: each Adam step moves an entry by <= lr, f32-vs-f64 drift stays far below one step
    du = np.abs(npy(u_last)[idx] - ru).max()
    assert du <= 0.05 * lr, du
    print("config 3: max cost rel err %.2e over 20 iterates, max |u_last - oracle| %.2e" % (np.abs(gc / rc - 1).max(), du))


def test_config5_odefunc_rk4_full_size(torch):
    """BASELINE config 5 at full size: ODEFunc(2,1), classic RK4 of src/integrators.py:39-84, H=200, B=65536."""
    from phnn_mpc_amd import _capi
    from phnn_mpc_amd.engine import RolloutEngine
    w = ol.load_weights("odefunc_pendulum")
    eng = RolloutEngine(w)
    rng = np.random.default_rng(5678)
    B, H, dt = 65536, 200, 0.05
    x0h = (rng.uniform(-1, 1, size=(B, 2)) * [np.pi, 1.0]).astype(np.float32)
    Uh = rng.uniform(-2.4, 2.4, size=(B, H, 1)).astype(np.float32)  # some entries outside the clamp [-2, 2]
    cost = _capi.make_cost(2, 1, [10.0, 1.0], [0.01], None, -2.0, 2.0)
    x0, U = torch.tensor(x0h, device="cuda"), torch.tensor(Uh, device="cuda")
    ws = {}
    c, gu, gx = eng.rollout_cost_grad(x0, U, cost, "rk4", dt, want_grad_x0=True, workspace=ws)
    c, gu, gx = c.clone(), gu.clone(), gx.clone()
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(gu).all()) and bool(torch.isfinite(gx).all())
    assert bool((gu[(U > 2.0) | (U < -2.0)] == 0).all())  # clamp mask exact
    # the run above used the RK4 stage-tape stash (54.5 GB at this size); a second pass over the same workspace repeats it bit for bit
    assert eng.use_stash and ws["stash"] is not None and ws["stash"].numel() == eng.workspace_bytes(B, H, "rk4")
    c_r, gu_r = eng.rollout_cost_grad(x0, U, cost, "rk4", dt, workspace=ws)
    assert torch.equal(c_r, c) and torch.equal(gu_r, gu)
    lo = 12345
    c2, g2 = eng.rollout_cost_grad(x0[lo:lo + 300], U[lo:lo + 300], cost, "rk4", dt)
    assert torch.equal(c2, c[lo:lo + 300]) and torch.equal(g2, gu[lo:lo + 300])
    eng.use_stash = False  # the recomputing adjoint (three forward evaluations + four recomputing VJPs per step): same gradients to rounding
    try:
        c3, g3 = eng.rollout_cost_grad(x0[lo:lo + 300], U[lo:lo + 300], cost, "rk4", dt)
    finally:
        eng.use_stash = True
    assert torch.equal(c3, c2)
    assert float((g3 - g2).abs().max()) <= 2e-6 * float(g2.abs().max())
    idx = rng.choice(B, size=64, replace=False)
    ref = ol.OracleModel(w, "f64").rollout(x0h[idx], Uh[idx], cost, "rk4", dt, nthreads=8)
    cg, gg = npy(c)[idx], npy(gu)[idx]
    assert np.allclose(cg, ref["cost"], rtol=1e-5), np.abs(cg / ref["cost"] - 1).max()
    gmax = np.abs(ref["grad_u"]).max(axis=(1, 2), keepdims=True)
    assert np.all(np.abs(gg - ref["grad_u"]) <= 1e-4 * gmax), (np.abs(gg - ref["grad_u"]) / gmax).max()
    print("config 5: max cost rel err %.2e, max grad err / max|grad| %.2e" % (
        np.abs(cg / ref["cost"] - 1).max(), (np.abs(gg - ref["grad_u"]) / gmax).max()))


def test_barrier_golden_g9_on_gpu(torch, ctl):
    """G9: MPCController with soft state bounds (src/mpc_controller.py:96-107), the reference's own outputs."""
    from phnn_mpc_amd.models import pHNN
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    model = _load(pHNN, CFG, "phnn_cartpole", torch)
    cfg = {"mpc": {"horizon": 20, "dt": 0.02, "Q": [10.0, 200.0, 1.0, 10.0], "R": 0.01, "u_min": -15.0, "u_max": 15.0,
                   "x_min": list(ctl["bar_xmin"]), "x_max": list(ctl["bar_xmax"]), "lr": 0.015, "max_iterations": 5}}
    c = create_mpc_from_config(model, cfg)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert abs(u0[0] - ctl["bar_u0_after5"][0]) < 1e-4, (u0, ctl["bar_u0_after5"])
    st = c.rollout_dynamics(torch.tensor(ctl["mpc_x0"]), torch.tensor(ctl["bar_u"]))
    assert np.allclose(npy(st), ctl["bar_states"], atol=1e-5)
    r = model.engine.rollout_cost_grad(torch.tensor(ctl["mpc_x0"][None]), torch.tensor(ctl["bar_u"][None]), c._cost(),
                                       "euler", 0.02)
    assert abs(float(r[0][0]) / float(ctl["bar_cost"]) - 1) < 1e-5
    assert np.all(np.abs(npy(r[1][0]) - ctl["bar_grad"]) <= 1e-4 * np.abs(ctl["bar_grad"]).max())


def test_lbfgs_golden_g13_on_gpu(torch, ctl):
    """G13: optimizer_type='LBFGS' (src/mpc_controller.py:169-170,196-197) with cost and gradient from K1/K2."""
    from phnn_mpc_amd.models import pHNN
    from phnn_mpc_amd.mpc_controller import MPCController
    c = MPCController(phnn_model=_load(pHNN, CFG, "phnn_cartpole", torch), horizon=20, dt=0.02,
                      Q=[10.0, 200.0, 1.0, 10.0], R=0.01, target_state=[0.0, 0.0, 0.0, 0.0], u_min=-15.0, u_max=15.0,
                      optimizer_type="LBFGS", lr=0.5, max_iterations=3)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert u0.shape == (1,) and abs(u0[0] - ctl["lbfgs_u0"][0]) < 5e-4 * max(1.0, abs(ctl["lbfgs_u0"][0])), (u0, ctl["lbfgs_u0"])


def test_options_struct_and_weight_updates(torch):
    """phnn_create_ex options (no environment variables inside the library); the nn.Module wrappers re-pack their
    weights when a parameter changes (optimizer step / in-place edit), so forward never serves stale weights."""
    from phnn_mpc_amd.engine import PhnnError, RolloutEngine
    from phnn_mpc_amd.models import pHNN
    w = ol.load_weights("phnn_cartpole")
    g = ol.load_golden("phnn_cartpole")
    e4 = RolloutEngine(w, max_waves=4)
    e8 = RolloutEngine(w)
    assert e4.kernel_info(65536)["rollouts_per_workgroup"] == 64 and e8.kernel_info(65536)["rollouts_per_workgroup"] == 128
    a, b = e4.forward(g["fwd_x"], g["fwd_u"]), e8.forward(g["fwd_x"], g["fwd_u"])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(PhnnError, match="Tanh"):
        RolloutEngine(w, activation="Softplus")  # Tanh / SiLU / ReLU / ELU / GELU have kernels (tests/test_gpu_activations.py), nothing else
    with pytest.raises(ValueError):
        RolloutEngine(w, matmul="fp8")
    m = _load(pHNN, CFG, "phnn_cartpole", torch)
    x, u = torch.tensor(g["fwd_x"]), torch.tensor(g["fwd_u"])
    d0 = m(x, u)[0].clone()
    with torch.no_grad():
        m.H_net.net[4].weight.mul_(1.5)  # in-place edit, as an optimizer step would do
    d1 = m(x, u)[0].clone()
    w2 = dict(w)
    w2["H_net.net.4.weight"] = w["H_net.net.4.weight"] * np.float32(1.5)
    ref = ol.OracleModel(w2, "f64").forward(g["fwd_x"], g["fwd_u"])[0]
    assert not torch.equal(d0, d1)
    assert np.abs(npy(d1) - ref).max() <= 2e-5 * np.abs(ref).max()


def test_device_restored_after_calls(torch):
    """A C-ABI call leaves the calling thread's current device as it found it (single-GPU box: device 0 stays 0)."""
    from phnn_mpc_amd.engine import RolloutEngine
    before = torch.cuda.current_device()
    eng = RolloutEngine(ol.load_weights("phnn_cartpole"))
    g = ol.load_golden("phnn_cartpole")
    eng.forward(g["fwd_x"], g["fwd_u"])
    assert torch.cuda.current_device() == before


# ----------------------------------------------------------------------------- the plant on the device (row f3)
def test_plant_kernel_matches_reference_simulator_g11(torch, ctl):
    """k_plant_step against the reference simulator (float64).  The C-ABI takes the force as float32 -- what the
    controllers return -- so (a) plant 2 of G11, whose forces are float32-exact (14.0), is compared with the
    reference's own trajectory directly, and (b) all three plants are compared with the host restatement
    BatchedCartPole (pinned to G11 in tests/test_host_logic.py) fed the same float32-rounded forces.  Every
    operation is IEEE double in the reference's order; only the last bit of the device's double-precision sin/cos
    can differ from numpy's, hence 1e-13 absolute instead of bitwise."""
    from phnn_mpc_amd import _capi
    from phnn_mpc_amd.closed_loop import BatchedCartPole
    from phnn_mpc_amd.engine import RolloutEngine
    eng = RolloutEngine(ol.load_weights("phnn_cartpole"))
    init = ctl["plant_init"]
    forces = ctl["plant_forces"].astype(np.float32)
    assert np.array_equal(forces[:, 2].astype(np.float64), ctl["plant_forces"][:, 2])
    B, T = init.shape[0], forces.shape[0]
    state = torch.tensor(init, dtype=torch.float64, device="cuda")
    x32 = torch.empty(B, 4, dtype=torch.float32, device="cuda")
    done_step = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    logs = torch.zeros(T + 1, B, 4, dtype=torch.float64, device="cuda")
    logc = torch.zeros(T, B, dtype=torch.float32, device="cuda")
    plant = _capi.Plant.default(0.02)
    host = BatchedCartPole(0.02)
    host.reset(init)
    host_done = np.full(B, -1)
    for t in range(T):
        a = torch.tensor(forces[t].reshape(B), device="cuda")
        eng.plant_step(plant, state, a, 1, state_f32=x32, done_step=done_step, step=t, log_states=logs, log_controls=logc)
        hs, hd = host.step(forces[t].astype(np.float64))
        host_done[(host_done < 0) & hd] = t
        assert np.allclose(npy(state), hs, rtol=0, atol=1e-13)
        assert np.allclose(npy(state)[2], ctl["plant_states"][t + 1, 2], rtol=0, atol=1e-13)
        assert np.allclose(npy(x32), hs.astype(np.float32).astype(np.float64), rtol=2e-7, atol=1e-30)
    assert np.allclose(npy(logs)[1:, 2], ctl["plant_states"][1:, 2], rtol=0, atol=1e-13)
    assert np.array_equal(logc.cpu().numpy(), forces)
    assert np.array_equal(done_step.cpu().numpy(), host_done)
    assert done_step[2].item() == int(np.argmax(ctl["plant_done"][:, 2]))


def test_device_closed_loop_equals_host_loop(torch):
    """run_mpc_batch_device (plant, warm-start shift, logs on the device; one HIP graph per control step; one host
    sync at the end) == run_mpc_batch (numpy plant on the host, one sync per step): same controls bit for bit, same
    plant states to the sin/cos bound, for both controller classes."""
    from phnn_mpc_amd.closed_loop import BatchedCartPole, run_mpc_batch, run_mpc_batch_device
    from phnn_mpc_amd.models import pHNN, pHNN_Canonical
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    from phnn_mpc_amd.mpc_controller_canonical import create_mpc_controller
    cfg = yaml.safe_load(open(CFG))
    rng = np.random.default_rng(6)
    X0 = rng.uniform(-1, 1, size=(200, 4)) * [0.2, 0.08, 0.1, 0.1]
    cc = create_mpc_controller(_load(pHNN_Canonical, CFG, "canonical_cartpole", torch), cfg)
    cc.optimizer_steps = 6
    cp = create_mpc_from_config(_load(pHNN, CFG, "phnn_cartpole", torch), cfg)
    cp.max_iterations = 6
    for c in (cc, cp):
        host = run_mpc_batch(BatchedCartPole(0.02), c, X0, 7)
        for use_graph in (False, True):
            dev = run_mpc_batch_device(c, X0, 7, use_graph=use_graph)
            assert dev["controls"].shape == host["controls"].shape and dev["states"].shape == host["states"].shape
            assert np.array_equal(dev["controls"], host["controls"])
            assert np.allclose(dev["states"], host["states"], rtol=0, atol=1e-12)
            assert np.array_equal(dev["done_step"], host["done_step"])


def test_device_closed_loop_300_steps_4096_plants(torch):
    """The closed loop of scripts/run_cartpole_mpc.py:91-182 (300 control steps, config settings H=20, 30 Adam
    iterations) for 4096 plants with nothing on the host.  No trained cart-pole checkpoint ships with the reference
    (SURVEY.md section 0), so with the seed-0 fixture weights the behavioural 'stability achieved' outcome is not
    asserted; what is: the loop runs to the end, logs are finite up to each plant's termination, the termination
    bookkeeping is the simulator's (|x| > 10 or |theta| > 0.5), the stability criterion of
    cartpole_mpc_config.yaml:69-75 is evaluated per plant, and 8 plants re-run on the host loop agree."""
    import time
    from phnn_mpc_amd.closed_loop import BatchedCartPole, run_mpc_batch, run_mpc_batch_device, stability_report
    from phnn_mpc_amd.models import pHNN
    from phnn_mpc_amd.mpc_controller import create_mpc_from_config
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_from_config(_load(pHNN, CFG, "phnn_cartpole", torch), cfg)
    rng = np.random.default_rng(9)
    B, T = 4096, 300
    X0 = rng.uniform(-1, 1, size=(B, 4)) * [0.5, 0.1, 0.2, 0.2]
    t0 = time.perf_counter()
    out = run_mpc_batch_device(c, X0, T, use_graph=True)
    el = time.perf_counter() - t0
    st, ds = out["states"], out["done_step"]
    assert st.shape == (T + 1, B, 4) and out["controls"].shape == (T, B, 1)
    assert np.all(np.abs(out["controls"]) <= 15.0)
    for b in range(0, B, 97):
        last = T if ds[b] < 0 else ds[b] + 1
        assert np.isfinite(st[: last + 1, b]).all()
        viol = (np.abs(st[1:, b, 0]) > 10.0) | (np.abs(st[1:, b, 1]) > 0.5)
        assert (ds[b] < 0 and not viol.any()) or (ds[b] >= 0 and ds[b] == np.argmax(viol))
    stab = cfg.get("stability", {})
    rep = stability_report(st, cfg["mpc"].get("x_target", [0, 0, 0, 0]), stab.get("tolerance", [0.5, 0.1, 0.5, 0.5]),
                           stab.get("min_duration", 1.0), 0.02)
    assert rep["stable"].shape == (B,)
    host = run_mpc_batch(BatchedCartPole(0.02), c, X0[:8], 12)
    assert np.array_equal(host["controls"], out["controls"][:12, :8])
    print("device closed loop: %d plants x %d control steps (H=20, 30 Adam iterations) in %.2f s = %.0f controls/s; "
          "%d plants terminated, %d meet the stability criterion (seed-0 weights, untrained)" % (
              B, T, el, B * T / el, int((ds >= 0).sum()), int(rep["stable"].sum())))


@pytest.mark.parametrize("name", ["phnn_cartpole", "canonical_cartpole"])
def test_full_size_bitwise_repeatable(torch, name):
    """The bench workload (B=65536, H=50, two waves per SIMD) run repeatedly on the same inputs: K1 and K2, stash and
    recompute mode, must be bitwise repeatable.  Regression test: the adjoint kernels built with LLVM's max-ILP
    scheduling strategy (round 1's last +2 %) were NOT -- a few hundred rollouts per launch, mostly in the second wave
    of a SIMD, differed by 2e-7 ... 4e-2 of their largest gradient entry from run to run, which small batches (one
    wave per SIMD) never show."""
    from phnn_mpc_amd.engine import RolloutEngine
    g, w = ol.load_golden("phnn_cartpole"), ol.load_weights(name)
    eng = RolloutEngine(w)
    rng = np.random.default_rng(1234)
    B, H = 65536, 50
    x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
    U = torch.tensor(rng.uniform(-5, 5, size=(B, H, 1)).astype(np.float32), device="cuda")
    cost = ol.cost_from_golden(g)
    assert eng.kernel_info(B)["rollouts_per_workgroup"] == 128
    for stash in (True, False):
        eng.use_stash = stash
        ws = {}
        c0, g0 = [t.clone() for t in eng.rollout_cost_grad(x0, U, cost, "euler", 0.02, workspace=ws)]
        for _ in range(6):
            c, gu = eng.rollout_cost_grad(x0, U, cost, "euler", 0.02, workspace=ws)
            assert torch.equal(c, c0)
            nbad = int((gu != g0).any(dim=2).any(dim=1).sum())
            assert nbad == 0, (name, "stash" if stash else "recompute", nbad)
    eng.use_stash = True


@pytest.mark.parametrize("name", ["phnn_cartpole", "canonical_cartpole"])
def test_split_tile_kernels_bitwise_equal_whole_tile(torch, name):
    """Small batches run on the split-tile kernels (four waves per 16-rollout tile).  They keep every
    summation order of the whole-tile kernels, so costs, trajectories and gradients must be BITWISE identical --
    Euler with and without the activation stash, RK4, ragged batch sizes; and the automatic choice must pick them
    for small batches only."""
    from phnn_mpc_amd.engine import RolloutEngine
    g, w = ol.load_golden("phnn_cartpole"), ol.load_weights(name)
    whole, split, auto = RolloutEngine(w, split="never"), RolloutEngine(w, split="always"), RolloutEngine(w)
    assert auto.kernel_info(4096)["rollouts_per_workgroup"] == 16 and auto.kernel_info(65536)["rollouts_per_workgroup"] == 128
    assert whole.kernel_info(4096)["rollouts_per_workgroup"] > 0 and split.kernel_info(64)["rollouts_per_workgroup"] == 16
    rng = np.random.default_rng(33)
    cost = ol.cost_from_golden(g)
    for B, H in ((1, 20), (17, 25), (300, 31)):
        x0 = (rng.uniform(-1, 1, size=(B, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32)
        U = rng.uniform(-17, 17, size=(B, H, 1)).astype(np.float32)
        for integ in ("euler", "rk4"):
            for stash in (True, False):
                res = []
                for eng in (whole, split):
                    eng.use_stash = stash
                    c, gu, gx = eng.rollout_cost_grad(x0, U, cost, integ, 0.02, want_grad_x0=True)
                    _, tr = eng.rollout_cost(x0, U, cost, integ, 0.02, want_traj=True)
                    traj, dX = eng.rollout_trajectory(x0, U, integ, 0.02, want_dx=True)
                    res.append([t.clone() for t in (c, gu, gx, tr, traj, dX)])
                    eng.use_stash = True
                for a, b, what in zip(res[0], res[1], ("cost", "grad_u", "grad_x0", "traj", "train traj", "dX")):
                    assert torch.equal(a, b), (name, B, H, integ, stash, what, float((a - b).abs().max()))
    # mixed: K1 by one kernel family, K2 by the other, through the shared stash format
    import ctypes as C
    B, H = 300, 31
    x0t, Ut = torch.tensor(x0, device="cuda"), torch.tensor(U, device="cuda")
    out = {}
    for k1, k2 in ((whole, split), (split, whole), (whole, whole)):
        traj = torch.empty(B, H + 1, 4, device="cuda"); cst = torch.empty(B, device="cuda"); gu = torch.empty(B, H, 1, device="cuda")
        st = torch.empty(k1.workspace_bytes(B, H, 0), dtype=torch.uint8, device="cuda")
        k1.lib.phnn_rollout_fwd(k1.h, k1._p(x0t), k1._p(Ut), B, H, C.byref(cost), 0, 0.02, k1._p(cst), k1._p(traj), k1._p(st), k1._stream())
        k2.lib.phnn_rollout_grad(k2.h, k2._p(x0t), k2._p(Ut), B, H, C.byref(cost), 0, 0.02, k2._p(traj), k2._p(st), k2._p(gu), None, k2._stream())
        out[(k1 is split, k2 is split)] = gu.clone()
    assert torch.equal(out[(False, True)], out[(False, False)]) and torch.equal(out[(True, False)], out[(False, False)])


def test_small_batch_latency_report(torch):
    """Config 1 (one plant, H=20, 30 Adam iterations) and config 2 (canonical, H=50, B=4096, forward) with and without
    the split-tile kernels -- printed, and the split kernels must not be slower."""
    import time
    from phnn_mpc_amd.engine import RolloutEngine
    from phnn_mpc_amd.solver import shooting_solve
    g = ol.load_golden("phnn_cartpole")
    cost = ol.cost_from_golden(g)
    rep = {}
    for mode in ("never", "always"):
        eng = RolloutEngine(ol.load_weights("phnn_cartpole"), split=mode)
        x0 = torch.tensor([[0.0, 0.1, 0.0, 0.0]], device="cuda")
        u0 = torch.zeros(1, 20, 1, device="cuda")
        for _ in range(3):
            shooting_solve(eng, x0, u0, cost, "euler", 0.02, 0.015, 30, u_min=-15.0, u_max=15.0, record_costs=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            shooting_solve(eng, x0, u0, cost, "euler", 0.02, 0.015, 30, u_min=-15.0, u_max=15.0, record_costs=False)
        torch.cuda.synchronize()
        rep[("c1", mode)] = (time.perf_counter() - t0) / 10 * 1e3
        engc = RolloutEngine(ol.load_weights("canonical_cartpole"), split=mode)
        rng = np.random.default_rng(1)
        X = torch.tensor((rng.uniform(-1, 1, size=(4096, 4)) * [1.0, 0.3, 0.5, 0.5]).astype(np.float32), device="cuda")
        U = torch.tensor(rng.uniform(-5, 5, size=(4096, 50, 1)).astype(np.float32), device="cuda")
        for integ in ("euler", "rk4"):
            for _ in range(3):
                engc.rollout_cost(X, U, cost, integ, 0.02)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                engc.rollout_cost(X, U, cost, integ, 0.02)
            torch.cuda.synchronize()
            rep[("c2" + integ, mode)] = (time.perf_counter() - t0) / 20 * 1e3
    print("config 1 (one plant, H=20, 30 Adam iterations): whole-tile %.2f ms, split-tile %.2f ms | config 2 (canonical H=50 "
          "B=4096 K1): euler %.3f -> %.3f ms, rk4 %.3f -> %.3f ms" % (
              rep[("c1", "never")], rep[("c1", "always")], rep[("c2euler", "never")], rep[("c2euler", "always")],
              rep[("c2rk4", "never")], rep[("c2rk4", "always")]))
    assert rep[("c1", "always")] < rep[("c1", "never")] and rep[("c2euler", "always")] < rep[("c2euler", "never")]


def test_bench_two_ranks_on_one_gpu_rehearsal(torch):
    """The N > 1 flow of bench.py on real kernels: `python bench.py --gpus 2` launches its own two ranks, both on this
    one GPU with the gloo rehearsal backend (RCCL refuses two ranks per device; the driver's runs use nccl, one GPU per
    rank).  The line must report two ranks, weak scaling over 2 x batch, the real backend in its labels, a
    strong-scaling object for the ragged split, and a roofline from the HIP events of rank 0."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PHNN_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8192", "--steps", "5", "--warmup", "2",
                        "--global-batch", "9001", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16384 and line["config"]["collective_backend"] == "gloo"
    assert "RCCL" not in line["config"]["workload"] and line["metric"].endswith("batch=8192")
    assert line["value"] > 1e6 and line["roofline"]["frac"] > 0 and line["strong_scaling"]["global_batch"] == 9001
    assert "rehearsal" not in line  # real kernels: only the collective backend is the rehearsal one
