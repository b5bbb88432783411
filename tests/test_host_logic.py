"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the weight packer, the model
classes' state_dict layout, and the controller host logic -- driven by the CPU oracle through the engine
protocol (tests/oracle_engine.py) and pinned to the reference's own controller outputs (golden sets G5, G6, G9).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import yaml

import oracle_lib as ol
from oracle_engine import OracleEngine
from phnn_mpc_amd import _capi, weights
from phnn_mpc_amd.models import ODEFunc, pHNN, pHNN_Canonical
from phnn_mpc_amd.mpc_controller import MPCController, create_mpc_from_config
from phnn_mpc_amd.mpc_controller_canonical import MPCControllerCanonical, create_mpc_controller

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "cartpole_mpc.yaml")
CFG_PEND = os.path.join(ROOT, "configs", "pendulum.yaml")


@pytest.fixture(scope="module")
def ctl():
    with np.load(os.path.join(ol.GOLDEN, "golden_controllers.npz")) as z:
        return {k: z[k] for k in z.files}


# ----------------------------------------------------------------------------- C-ABI surface
def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    header = open(os.path.join(ROOT, "include", "phnn_mpc.h")).read()
    declared = set(re.findall(r"\b(phnn_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTED), declared ^ set(_capi.EXPORTED)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.phnn_version() >= 100


def test_struct_sizes_match_header():
    # phnn_desc: 4 ints + 3 * (1 + 4) ints + activation + mass_type + m_net (1 + 4) ; phnn_cost: 64+16+8 floats, 2 floats, int, 16 floats, 2 ints,
    # float ; phnn_options: 3 ints + 5 reserved ; phnn_plant: 7 doubles
    assert C.sizeof(_capi.Desc) == 4 * (4 + 3 * 5 + 1 + 1 + 5)
    assert C.sizeof(_capi.Cost) == 4 * (64 + 16 + 8 + 2 + 1 + 16 + 2 + 1)
    assert C.sizeof(_capi.Options) == 4 * 8
    assert C.sizeof(_capi.Plant) == 8 * 7


def test_activation_is_part_of_the_description():
    """A checkpoint trained with another activation has the same keys and shapes: the description carries the
    activation explicitly; Tanh, SiLU, ReLU, ELU and GELU have kernels, everything else is refused (before any device is touched),
    and so are SiLU / ReLU in the split-precision product modes (their activations are unbounded)."""
    lib = _capi.load_library()
    d, blob = weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="nn.Softplus")
    assert d.activation == _capi.ACT_OTHER
    h = C.c_void_p()
    rc = lib.phnn_create_ex(C.byref(d), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, 0, None, C.byref(h))
    assert rc == -2 and b"Tanh" in lib.phnn_last_error(None)
    ds, blobs = weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="nn.SiLU")
    assert ds.activation == _capi.ACT_SILU and weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="relu")[0].activation == _capi.ACT_RELU
    assert weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="nn.GELU")[0].activation == _capi.ACT_GELU
    assert weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="elu")[0].activation == _capi.ACT_ELU
    o16 = _capi.Options()
    o16.matmul_mode = _capi.MATMUL_MODES["f16x2"]
    rc = lib.phnn_create_ex(C.byref(ds), blobs.ctypes.data_as(C.POINTER(C.c_float)), blobs.size, 0, C.byref(o16), C.byref(h))
    assert rc == -2 and b"all-f32" in lib.phnn_last_error(None)
    d2, _ = weights.pack_state_dict(ol.load_weights("phnn_cartpole"), activation="nn.Tanh")
    assert d2.activation == _capi.ACT_TANH
    # f16x2 on a 64-wide model is refused unless forced (known to miss the tolerance on the trained pendulum model)
    d3, blob3 = weights.pack_state_dict(ol.load_weights("phnn_pendulum"))
    opt = _capi.Options()
    opt.matmul_mode = _capi.MATMUL_MODES["f16x2"]
    rc = lib.phnn_create_ex(C.byref(d3), blob3.ctypes.data_as(C.POINTER(C.c_float)), blob3.size, 0, C.byref(opt), C.byref(h))
    assert rc == -2 and b"force_matmul" in lib.phnn_last_error(None)
    opt.max_waves = 99
    rc = lib.phnn_create_ex(C.byref(d3), blob3.ctypes.data_as(C.POINTER(C.c_float)), blob3.size, 0, C.byref(opt), C.byref(h))
    assert rc == -1


@pytest.mark.parametrize("name", ol.MODELS)
def test_weight_count_and_packer(name):
    lib = _capi.load_library()
    d, blob = weights.pack_state_dict(ol.load_weights(name))
    assert lib.phnn_weight_count(C.byref(d)) == blob.size
    assert blob.dtype == np.float32 and blob.flags.c_contiguous


def test_create_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load_library()
    d, blob = weights.pack_state_dict(ol.load_weights("phnn_cartpole"))
    h = C.c_void_p()
    rc = lib.phnn_create(C.byref(d), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, 0, C.byref(h))
    assert rc < 0 and not h.value
    assert b"no HIP device" in lib.phnn_last_error(None) or b"hip" in lib.phnn_last_error(None).lower()
    from phnn_mpc_amd.engine import PhnnError, RolloutEngine
    with pytest.raises(PhnnError):
        RolloutEngine(ol.load_weights("phnn_cartpole"))


def test_unsupported_descriptions_are_refused():
    lib = _capi.load_library()
    w = ol.load_weights("phnn_cartpole")
    d, blob = weights.pack_state_dict(w)
    h = C.c_void_p()
    assert lib.phnn_create(C.byref(d), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size - 1, 0, C.byref(h)) == -1
    d.h_net.hidden[1] = 96  # H_net widths differ: no kernel
    assert lib.phnn_weight_count(C.byref(d)) != blob.size


def test_packer_errors():
    with pytest.raises(ValueError):
        weights.pack_state_dict({"foo": np.zeros(3)})
    w = ol.load_weights("canonical_cartpole")
    del w["M_net.b"]
    with pytest.raises(ValueError):
        weights.pack_state_dict(w)
    wrapped = {"model_state_dict": ol.load_weights("phnn_pendulum")}
    d, _ = weights.pack_state_dict(wrapped)
    assert (d.kind, d.n, d.m, d.fixed_G) == (_capi.MODEL_PHNN, 2, 1, 0)


def test_make_cost_rules():
    c = _capi.make_cost(4, 1, [1, 2, 3, 4], 0.5, None, -1.0, None)  # one-sided bound: the reference does not clamp
    assert c.has_u_bounds == 0 and c.Q[5] == 2.0 and c.R[0] == 0.5
    with pytest.raises(ValueError):
        _capi.make_cost(4, 1, np.eye(3), 0.1)


# ----------------------------------------------------------------------------- model classes
def test_model_state_dict_layout_matches_reference_checkpoints():
    for cls, cfg, name in ((pHNN, CFG, "phnn_cartpole"), (pHNN_Canonical, CFG, "canonical_cartpole"),
                           (pHNN, CFG_PEND, "phnn_pendulum")):
        m = cls(cfg)
        w = ol.load_weights(name)
        assert set(m.state_dict().keys()) == set(w.keys())
        m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})  # strict
        d, blob = weights.pack_state_dict(m.state_dict())
        d2, blob2 = weights.pack_state_dict(w)
        assert np.array_equal(blob, blob2)
    o = ODEFunc(2, 1)
    w = ol.load_weights("odefunc_pendulum")
    assert set(o.state_dict().keys()) == set(w.keys())
    o.load_state_dict({k: torch.tensor(v) for k, v in w.items()})


def test_seeded_construction_reproduces_reference_init():
    """Same parameter-creation order and initialisers as the reference => same seed-0 weights (torch 2.10)."""
    torch.manual_seed(0)
    m = pHNN(CFG)
    w = ol.load_weights("phnn_cartpole")
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), w[k]), k


def test_model_forward_through_engine_protocol():
    w = ol.load_weights("phnn_cartpole")
    g = ol.load_golden("phnn_cartpole")
    m = pHNN(CFG)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    m.set_engine(OracleEngine(w))
    x = torch.tensor(g["fwd_x"][:8], requires_grad=True)
    u = torch.tensor(g["fwd_u"][:8], requires_grad=True)
    dx, H = m(x, u)
    assert np.allclose(dx.detach().numpy(), g["fwd_dx_f32"][:8], rtol=1e-5, atol=1e-6)
    lam = torch.tensor(g["vjp_lam"][:8])
    (dx * lam).sum().backward()
    ref = ol.OracleModel(w, "f64").vjp(g["fwd_x"][:8], g["fwd_u"][:8], g["vjp_lam"][:8])
    assert np.allclose(x.grad.numpy(), ref[0], rtol=1e-4, atol=1e-5)
    assert np.allclose(u.grad.numpy(), ref[1], rtol=1e-4, atol=1e-5)
    # 1-D input is promoted to a batch of one, as the reference does
    dx1, H1 = m(torch.tensor(g["fwd_x"][0]), torch.tensor(g["fwd_u"][0]))
    assert dx1.shape == (1, 4) and H1.shape == (1,)


def test_backward_through_H_without_wgrad_kernels_raises():
    """A variant without weight-gradient kernels (bf16x3, 64-wide f16x2) has only the plain VJP, which takes no cotangent
    on H: a loss through H must raise instead of silently dropping that term."""
    w = ol.load_weights("phnn_cartpole")
    g = ol.load_golden("phnn_cartpole")
    m = pHNN(CFG)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    eng = OracleEngine(w)
    eng.has_wgrad = False
    m.set_engine(eng)
    for p in m.parameters():
        p.requires_grad_(False)
    x = torch.tensor(g["fwd_x"][:4], requires_grad=True)
    u = torch.tensor(g["fwd_u"][:4])
    dx, H = m(x, u)
    with pytest.raises(NotImplementedError):
        (dx.sum() + H.sum()).backward()
    dx, H = m(x, u)
    dx.sum().backward()  # the dx output alone is fine
    assert x.grad is not None


def test_unsupported_model_options_raise():
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["H_mlp"]["activation"] = "nn.ReLU"
    p = os.path.join(ROOT, "gpurun_out", "_tmp_relu.yaml")
    os.makedirs(os.path.dirname(p), exist_ok=True)
    yaml.safe_dump(cfg, open(p, "w"))
    m = pHNN(p)
    with pytest.raises(NotImplementedError):
        m.engine
    with pytest.raises(ValueError):
        ODEFunc(2, 1, activation="swish")
    with pytest.raises(RuntimeError):
        ODEFunc(2, 1)(0.0, torch.zeros(1, 2))


# ----------------------------------------------------------------------------- controllers vs the reference
def _phnn_with_oracle():
    w = ol.load_weights("phnn_cartpole")
    m = pHNN(CFG)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return m.set_engine(OracleEngine(w))


def _canon_with_oracle():
    w = ol.load_weights("canonical_cartpole")
    m = pHNN_Canonical(CFG)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return m.set_engine(OracleEngine(w))


def test_mpc_controller_matches_reference_g5(ctl):
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_from_config(_phnn_with_oracle(), cfg)
    assert (c.horizon, c.lr, c.max_iterations) == (20, 0.015, 30)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert isinstance(u0, np.ndarray) and u0.shape == (1,)
    assert abs(u0[0] - ctl["mpc_u0"][0]) < 2e-5, (u0, ctl["mpc_u0"])
    out = c.solve_batch(ctl["mpc_x0"][None], record_costs=True)
    assert np.allclose(out["costs"][:, 0].numpy(), ctl["mpc_costs"], rtol=2e-6)
    # the controller's own B=1 rollout / cost methods
    st = c.rollout_dynamics(torch.tensor(ctl["mpc_x0"]), torch.zeros(20, 1))
    assert np.allclose(st.numpy(), ctl["mpc_zero_states"], atol=2e-6)
    assert abs(float(c.compute_cost(st, torch.zeros(20, 1))) - float(ctl["mpc_zero_cost"])) < 1e-4


def test_mpc_controller_other_schema_barrier_and_errors(ctl):
    model = _phnn_with_oracle()
    cfg = {"mpc": {"horizon": 20, "dt": 0.02, "Q": [10.0, 200.0, 1.0, 10.0], "R": 0.01, "u_min": -15.0, "u_max": 15.0,
                   "x_min": list(ctl["bar_xmin"]), "x_max": list(ctl["bar_xmax"]), "lr": 0.015, "max_iterations": 5}}
    c = create_mpc_from_config(model, cfg)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert abs(u0[0] - ctl["bar_u0_after5"][0]) < 2e-5
    st = c.rollout_dynamics(torch.tensor(ctl["mpc_x0"]), torch.tensor(ctl["bar_u"]))
    assert np.allclose(st.numpy(), ctl["bar_states"], atol=2e-6)
    assert abs(float(c.compute_cost(st, torch.tensor(ctl["bar_u"]))) / float(ctl["bar_cost"]) - 1) < 1e-5
    r = model.engine.rollout_cost_grad(torch.tensor(ctl["mpc_x0"][None]), torch.tensor(ctl["bar_u"][None]), c._cost(),
                                       "euler", 0.02)
    assert abs(float(r[0][0]) / float(ctl["bar_cost"]) - 1) < 1e-5
    assert np.allclose(r[1][0].numpy(), ctl["bar_grad"], rtol=1e-4, atol=1e-4 * np.abs(ctl["bar_grad"]).max())
    c.optimizer_type = "SGD"
    with pytest.raises(ValueError):
        c.compute_control(ctl["mpc_x0"].copy())


def test_canonical_controller_matches_reference_g6(ctl):
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_controller(_canon_with_oracle(), cfg)
    u_a, info_a = c.control(ctl["mpc_x0"].copy(), None)
    assert u_a.shape == (1,) and info_a["u_sequence"].shape == (20, 1)
    assert set(info_a) == {"u_sequence", "solve_time", "optimization"}
    assert set(info_a["optimization"]) == {"costs", "final_cost", "num_steps"}
    assert abs(u_a[0] - ctl["can_u_a"][0]) < 5e-5
    assert np.allclose(info_a["optimization"]["costs"], ctl["can_costs_a"], rtol=5e-6)
    assert abs(info_a["optimization"]["final_cost"] - float(ctl["can_final_a"])) < 1e-3
    assert np.allclose(info_a["u_sequence"], ctl["can_useq_a"], atol=1e-4)
    # warm start: shift by one, zero tail -- fed with the REFERENCE's previous sequence
    u_b, info_b = c.control(ctl["can_x_b"].copy(), ctl["can_useq_a"])
    assert abs(u_b[0] - ctl["can_u_b"][0]) < 1e-4
    assert np.allclose(info_b["optimization"]["costs"], ctl["can_costs_b"], rtol=5e-6)
    assert np.allclose(info_b["u_sequence"], ctl["can_useq_b"], atol=2e-4)


def test_batched_solve_equals_per_sample_solves():
    """B stacked problems behave exactly like B separate solves (Adam is element-wise, rollouts independent)."""
    cfg = yaml.safe_load(open(CFG))
    c = create_mpc_controller(_canon_with_oracle(), cfg)
    c.optimizer_steps = 6
    rng = np.random.default_rng(3)
    X = (rng.uniform(-1, 1, size=(5, 4)) * [0.5, 0.2, 0.3, 0.3]).astype(np.float32)
    ub, seq, best = c.control_batch(X)
    for b in range(5):
        u1, info = c.control(X[b])
        assert np.array_equal(u1, ub[b]) and np.array_equal(info["u_sequence"], seq[b])


# ----------------------------------------------------------------------------- closed loop (rows f1 / f3)
def test_batched_plant_matches_reference_simulator_g11(ctl):
    from phnn_mpc_amd.closed_loop import BatchedCartPole
    sim = BatchedCartPole(dt=0.02)
    x = sim.reset(ctl["plant_init"])
    assert np.array_equal(x, ctl["plant_init"])
    for t in range(60):
        x, done = sim.step(ctl["plant_forces"][t])
        assert np.allclose(x, ctl["plant_states"][t + 1], rtol=0, atol=1e-13)
        assert np.array_equal(done, ctl["plant_done"][t])
    assert ctl["plant_done"][-1, 2] and not ctl["plant_done"][0].any()


def test_batched_closed_loop_equals_per_plant_loops():
    """run_mpc_batch (one batched solve per control step) == B separate reference-style loops."""
    from phnn_mpc_amd.closed_loop import BatchedCartPole, run_mpc_batch, stability_report
    cfg = yaml.safe_load(open(CFG))
    rng = np.random.default_rng(4)
    X0 = rng.uniform(-1, 1, size=(3, 4)) * [0.2, 0.08, 0.1, 0.1]
    for make in (lambda: create_mpc_controller(_canon_with_oracle(), cfg), lambda: create_mpc_from_config(_phnn_with_oracle(), cfg)):
        c = make()
        if hasattr(c, "optimizer_steps"):
            c.optimizer_steps = 4
        else:
            c.max_iterations = 4
        out = run_mpc_batch(BatchedCartPole(0.02), c, X0, 5)
        assert out["states"].shape == (6, 3, 4) and out["controls"].shape == (5, 3, 1)
        for b in range(3):
            sim = BatchedCartPole(0.02)
            x = sim.reset(X0[b])[0]
            u_prev = None
            for t in range(5):
                if hasattr(c, "control"):
                    u, info = c.control(x.astype(np.float32), u_prev)
                    u_prev = info["u_sequence"]
                else:
                    u = c.compute_control(x.astype(np.float32))
                assert np.array_equal(np.asarray(u, np.float64).reshape(-1), out["controls"][t, b])
                x = sim.step(u)[0][0]
                assert np.array_equal(x, out["states"][t + 1, b])
    rep = stability_report(out["states"], [0, 0, 0, 0], cfg["stability"]["tolerance"], cfg["stability"]["min_duration"], 0.02)
    assert rep["stable"].shape == (3,) and rep["longest_run_s"].shape == (3,)


def test_coordinate_transforms_match_reference_g12(ctl):
    from phnn_mpc_amd import coordinate_transforms as CT
    m = pHNN_Canonical(CFG)
    m.load_state_dict({k: torch.tensor(v) for k, v in ol.load_weights("canonical_cartpole").items()})
    y = torch.tensor(ctl["ct_y"])
    z = CT.kinematic_to_canonical(y, m.M_net)
    assert np.allclose(z.numpy(), ctl["ct_z"], rtol=1e-6, atol=1e-6)
    assert np.allclose(CT.canonical_to_kinematic(z, m.M_net).numpy(), ctl["ct_y_back"], rtol=1e-6, atol=1e-6)
    assert np.allclose(CT.velocity_to_momentum(y[:, :2], y[:, 2:], m.M_net).numpy(), ctl["ct_p"], rtol=1e-6, atol=1e-6)
    assert np.allclose(m.get_velocity_reconstruction(y).numpy(), ctl["ct_vrec"], rtol=1e-6, atol=1e-6)
    q, v = CT.split_state(y)
    assert q.shape == (32, 2) and v.shape == (32, 2)
    # the remaining helpers of src/coordinate_transforms.py:133-237, against the golden transforms
    p = torch.tensor(ctl["ct_p"])
    assert torch.equal(CT.combine_state(q, v), y)
    assert np.allclose(CT.batch_matrix_vector_product(m.M_net(q), v).numpy(), ctl["ct_p"], rtol=1e-6, atol=1e-6)
    T = CT.compute_kinetic_energy(q, p, m.M_net)
    assert T.shape == (32,) and np.allclose(T.numpy(), 0.5 * (ctl["ct_p"] * ctl["ct_y_back"][:, 2:]).sum(1), rtol=1e-5, atol=1e-6)
    ok, worst = CT.verify_coordinate_transform(y, m.M_net)
    assert ok and worst < 1e-5
    err = CT.compute_velocity_reconstruction_error(q, v, p, m.M_net)
    assert err.shape == (32,) and float(err.max()) < 1e-9
    assert m.M_net.get_parameters_dict().keys() == {"a", "b", "c"}


def test_lbfgs_branch_matches_reference_g13(ctl):
    """optimizer_type='LBFGS' (src/mpc_controller.py:169-170): torch's L-BFGS driven by the engine's cost/gradient."""
    c = MPCController(phnn_model=_phnn_with_oracle(), horizon=20, dt=0.02, Q=[10.0, 200.0, 1.0, 10.0], R=0.01,
                      target_state=[0.0, 0.0, 0.0, 0.0], u_min=-15.0, u_max=15.0, optimizer_type="LBFGS", lr=0.5,
                      max_iterations=3)
    u0 = c.compute_control(ctl["mpc_x0"].copy())
    assert u0.shape == (1,) and abs(u0[0] - ctl["lbfgs_u0"][0]) < 5e-4 * max(1.0, abs(ctl["lbfgs_u0"][0])), (u0, ctl["lbfgs_u0"])
    with pytest.raises(NotImplementedError):
        c.compute_control_batch(ctl["mpc_x0"][None])


# ----------------------------------------------------------------------------- training side (row f4), host logic on the oracle
def _named_param_grads(model):
    return {k: (p.grad.detach().numpy().astype(np.float64) if p.grad is not None else np.zeros(tuple(p.shape)))
            for k, p in model.named_parameters()}


def _check_named(named, wg, prefix, tag, rtol):
    keys = [k for k in wg if k.startswith(prefix + "g.") and k.endswith("_" + tag)]
    assert keys
    for k in keys:
        name = k[len(prefix) + 2:-len(tag) - 1]
        ref = np.asarray(wg[k], np.float64)
        mx = np.abs(ref).max()
        ours = named[name].reshape(ref.shape)
        assert np.abs(ours - ref).max() <= rtol * max(mx, 1e-30), (name, np.abs(ours - ref).max(), mx)


def test_training_step_phnn_cartpole_g15():
    """One optimisation step's loss and parameter gradients of scripts/train_cartpole_phnn.py:112-178, written against
    the drop-in API: the fused differentiable rollout for X_pred, a model call for the energy anchor H(0)^2.  The
    engine is the float64 oracle here (host logic + autograd plumbing); the GPU twin is tests/test_gpu_wgrad.py."""
    from phnn_mpc_amd.integrators import rollout_trajectory_differentiable
    wg = ol.load_wgrad_golden()
    w = ol.load_weights("phnn_cartpole")
    model = pHNN(CFG)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model.set_engine(OracleEngine(w, "f64"))
    x_batch, u_batch = torch.tensor(wg["tr_cart_x"]), torch.tensor(wg["tr_cart_u"])
    loss_fn = torch.nn.MSELoss()
    X_pred = rollout_trajectory_differentiable(model, x_batch[:, 0, :], u_batch[:, :-1, :], 0.02, "euler")
    l_pos = loss_fn(X_pred[:, :, 0], x_batch[:, :, 0])
    l_theta = torch.mean(1 - torch.cos(X_pred[:, :, 1] - x_batch[:, :, 1]))
    l_vel = loss_fn(X_pred[:, :, 2:], x_batch[:, :, 2:])
    _, H_zero = model(torch.zeros(1, 4), torch.zeros(1, 1))
    loss = 1.0 * l_pos + 1.0 * l_theta + 1.0 * l_vel + 0.01 * torch.mean(H_zero ** 2)
    loss.backward()
    assert abs(loss.item() / float(wg["phnn_cartpole/tr_loss_f64"]) - 1) < 1e-6  # float32 tensors around a float64 engine
    assert np.allclose(X_pred.detach().numpy(), wg["phnn_cartpole/tr_X_f64"], atol=2e-6)
    _check_named(_named_param_grads(model), wg, "phnn_cartpole/tr_", "f64", 2e-6)


def test_training_step_pendulum_and_canonical_g15():
    """main.py:93-148 (loss on X_pred and on dX_pred; pendulum pHNN with a learned G) and
    scripts/train_cartpole_phnn_canonical.py:83-196 (per-step model calls with return_intermediate=True, manual
    Euler -- the reference's loop as is, every model call one engine call) on the oracle engine."""
    from phnn_mpc_amd.integrators import rollout_trajectory_differentiable
    from phnn_mpc_amd.coordinate_transforms import split_state
    wg = ol.load_wgrad_golden()
    w = ol.load_weights("phnn_pendulum")
    model = pHNN(CFG_PEND)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model.set_engine(OracleEngine(w, "f64"))
    x_batch, u_batch, dx_batch = (torch.tensor(wg[k]) for k in ("tr_pend_x", "tr_pend_u", "tr_pend_dx"))
    loss_fn = torch.nn.MSELoss()
    X_pred, dX_pred = rollout_trajectory_differentiable(model, x_batch[:, 0, :], u_batch[:, :-1, :], 0.05, "euler",
                                                        return_derivatives=True)
    loss = loss_fn(X_pred, x_batch) + loss_fn(dX_pred, dx_batch[:, 0:-1, :])
    loss.backward()
    assert abs(loss.item() / float(wg["phnn_pendulum/tr_loss_f64"]) - 1) < 1e-6
    _check_named(_named_param_grads(model), wg, "phnn_pendulum/tr_", "f64", 2e-6)
    # canonical: the reference's own loop shape (compute_integrated_loss), model call per step
    w = ol.load_weights("canonical_cartpole")
    can = pHNN_Canonical(CFG)
    can.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    can.set_engine(OracleEngine(w, "f64"))
    x_batch, u_batch = torch.tensor(wg["tr_cart_x"]), torch.tensor(wg["tr_cart_u"])
    y_pred, vel_err = [x_batch[:, 0, :]], []
    for t in range(x_batch.shape[1] - 1):
        dy, _, inter = can(y_pred[-1], u_batch[:, t, :], return_intermediate=True)
        y_pred.append(y_pred[-1] + 0.02 * dy)
        _, qd_true = split_state(x_batch[:, t, :])
        vel_err.append(torch.sum((inter["q_dot_reconstructed"] - qd_true) ** 2, dim=1).mean())
    y_pred = torch.stack(y_pred, dim=1)
    l_pos = torch.mean((y_pred[:, :, 0] - x_batch[:, :, 0]) ** 2) + torch.mean(1 - torch.cos(y_pred[:, :, 1] - x_batch[:, :, 1]))
    l_vel = torch.mean(torch.stack(vel_err))
    loss = 1.0 * l_pos + 0.5 * l_vel
    loss.backward()
    assert abs(l_pos.item() / float(wg["canonical_cartpole/tr_loss_position_f64"]) - 1) < 1e-5
    assert abs(l_vel.item() / float(wg["canonical_cartpole/tr_loss_velocity_f64"]) - 1) < 1e-5
    _check_named(_named_param_grads(can), wg, "canonical_cartpole/tr_", "f64", 5e-6)
