#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference.  It imports the reference's Python modules (src/pHNN.py, src/pHNN_canonical.py,
src/integrators.py, src/mpc_controller.py, src/mpc_controller_canonical.py, src/baseline_node.py),
evaluates them on seeded inputs and stores inputs + outputs as .npz data files.  Nothing of the
reference's source text is stored; the fixtures are weights, inputs and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Vector sets:
  G1 weights          weights_*.npz          state_dict arrays
  G2 f(x,u), H        golden_<model>.npz     fwd_*
  G3 VJP              golden_<model>.npz     vjp_*
  G4 rollouts         golden_<model>.npz     roll_<integ>_<case>_*
  G5 MPCController.compute_control           golden_controllers.npz  mpc_*
  G6 MPCControllerCanonical.control (+warm)  golden_controllers.npz  can_*
  G7 pendulum 10-step rollouts               golden_phnn_pendulum.npz  g7_*
  G8 dataset windows                         golden_dataset_windows.npz
  G9 soft state barrier (MPCController x_min/x_max) golden_controllers.npz  bar_*
  G10 reverse pass with trajectory + cost cotangents golden_<model>.npz  tvjp_*
  G11 the NumPy plant CartPoleSimulator.step         golden_controllers.npz  plant_*
  G12 coordinate transforms with the cart-pole M_net golden_controllers.npz  ct_*
  G13 MPCController with optimizer_type="LBFGS"      golden_controllers.npz  lbfgs_*
Each quantity is stored twice: *_f64 from the reference cast to double, *_f32 from the
reference as shipped (float32, torch CPU).
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.append(os.path.join(REF, "src"))
os.chdir(REF)  # the reference opens its YAML files by relative path

from pHNN import pHNN  # noqa: E402
from pHNN_canonical import pHNN_Canonical  # noqa: E402
import integrators  # noqa: E402
from mpc_controller import MPCController  # noqa: E402
from mpc_controller_canonical import MPCControllerCanonical, create_mpc_controller  # noqa: E402
from baseline_node import ODEFunc  # noqa: E402

torch.set_num_threads(1)


class OdeAdapter(nn.Module):
    """model(y,u) -> (dy, H) view of ODEFunc (src/baseline_node.py:88-116); H is not defined -> 0."""

    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, y, u):
        self.f.current_action = u
        return self.f(0.0, y), torch.zeros(y.shape[0], dtype=y.dtype)


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def batched_cost(traj, Uc, Q, R, xt):
    e = traj - xt
    sc = torch.einsum("bti,ij,btj->b", e, Q, e)
    cc = torch.einsum("bti,ij,btj->b", Uc, R, Uc)
    return sc + cc


def rollout_case(model, x0, U, dt, integ, Q, R, xt, umin, umax, dtype):
    m = copy.deepcopy(model).to(dtype)
    y0 = torch.tensor(x0, dtype=dtype, requires_grad=True)
    Ur = torch.tensor(U, dtype=dtype, requires_grad=True)
    Uc = torch.clamp(Ur, umin, umax)
    traj = integrators.rollout_trajectory_differentiable(m, y0, Uc, dt, integ)
    # cost weights are float32 values in the reference (torch.tensor(Q, dtype=torch.float32),
    # src/mpc_controller_canonical.py:75,80,87); the double run uses the same float32-representable numbers
    f32 = lambda a: torch.tensor(np.asarray(a, np.float32)).to(dtype)
    cost = batched_cost(traj, Uc, f32(Q), f32(R), f32(xt))
    gu, gx = torch.autograd.grad(cost.sum(), [Ur, y0])
    return (traj.detach().numpy(), cost.detach().numpy(), gu.numpy(), gx.numpy())


def model_block(name, model, n, m, dt, Q, R, xt, umin, umax, xlo, xhi, uamp, seed, cases):
    rng = np.random.default_rng(seed)
    out = {"n": np.int32(n), "m": np.int32(m), "dt": np.float64(dt), "Q": np.asarray(Q, np.float64),
           "R": np.asarray(R, np.float64), "x_target": np.asarray(xt, np.float64),
           "u_min": np.float64(umin), "u_max": np.float64(umax)}
    # ---- G2: forward
    NB = 256
    x = rng.uniform(xlo, xhi, size=(NB, n)).astype(np.float32)
    u = rng.uniform(-uamp, uamp, size=(NB, m)).astype(np.float32)
    out["fwd_x"], out["fwd_u"] = x, u
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        mm = copy.deepcopy(model).to(dtype)
        xt_ = torch.tensor(x, dtype=dtype, requires_grad=True)
        ut_ = torch.tensor(u, dtype=dtype)
        res = mm(xt_, ut_)
        out[f"fwd_dx_{tag}"] = res[0].detach().numpy()
        out[f"fwd_H_{tag}"] = res[1].detach().numpy()
    # ---- G3: VJP
    NV = 64
    x = rng.uniform(xlo, xhi, size=(NV, n)).astype(np.float32)
    u = rng.uniform(-uamp, uamp, size=(NV, m)).astype(np.float32)
    lam = rng.normal(size=(NV, n)).astype(np.float32)
    out["vjp_x"], out["vjp_u"], out["vjp_lam"] = x, u, lam
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        mm = copy.deepcopy(model).to(dtype)
        xt_ = torch.tensor(x, dtype=dtype, requires_grad=True)
        ut_ = torch.tensor(u, dtype=dtype, requires_grad=True)
        dx = mm(xt_, ut_)[0]
        xb, ub = torch.autograd.grad((dx * torch.tensor(lam, dtype=dtype)).sum(), [xt_, ut_])
        out[f"vjp_xbar_{tag}"] = xb.numpy()
        out[f"vjp_ubar_{tag}"] = ub.numpy()
    # ---- G4: rollouts
    for (B, H) in cases:
        x0 = rng.uniform(xlo, xhi, size=(B, n)).astype(np.float32)
        U = rng.uniform(-uamp, uamp, size=(B, H, m)).astype(np.float32)
        # exercise the clamp mask: outside, and exactly on, the bounds (torch.clamp backward is inclusive)
        U[0, 1 % H, 0] = np.float32(umax * 1.2)
        U[0, 3 % H, 0] = np.float32(umin * 1.1)
        U[0, 5 % H, 0] = np.float32(umax)
        U[-1, H - 1, 0] = np.float32(umin)
        for integ in ("euler", "rk4"):
            key = f"roll_{integ}_B{B}_H{H}"
            out[f"{key}_x0"], out[f"{key}_U"] = x0, U
            for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
                traj, cost, gu, gx = rollout_case(model, x0, U, dt, integ, Q, R, xt, umin, umax, dtype)
                assert np.isfinite(traj).all() and np.isfinite(gu).all(), (name, key, tag)
                if tag == "f64":
                    out[f"{key}_traj_{tag}"] = traj
                else:
                    out[f"{key}_xH_{tag}"] = traj[:, -1]
                out[f"{key}_cost_{tag}"] = cost
                out[f"{key}_gu_{tag}"] = gu
                out[f"{key}_gx0_{tag}"] = gx
    # ---- G10: reverse pass with a random cotangent on the trajectory and on the cost
    B, H = 4, 30
    x0 = rng.uniform(xlo, xhi, size=(B, n)).astype(np.float32)
    U = rng.uniform(-uamp, uamp, size=(B, H, m)).astype(np.float32)
    U[1, 2, 0] = np.float32(umax * 1.5)
    Wt = rng.normal(size=(B, H + 1, n)).astype(np.float32)
    wc = rng.normal(size=(B,)).astype(np.float32)
    out["tvjp_x0"], out["tvjp_U"], out["tvjp_traj_bar"], out["tvjp_cost_bar"] = x0, U, Wt, wc
    for integ in ("euler", "rk4"):
        mm = copy.deepcopy(model).double()
        y0 = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
        Ur = torch.tensor(U, dtype=torch.float64, requires_grad=True)
        Uc = torch.clamp(Ur, umin, umax)
        traj = integrators.rollout_trajectory_differentiable(mm, y0, Uc, dt, integ)
        f32 = lambda a: torch.tensor(np.asarray(a, np.float32)).double()
        cost = batched_cost(traj, Uc, f32(Q), f32(R), f32(xt))
        loss = (traj * torch.tensor(Wt, dtype=torch.float64)).sum() + (cost * torch.tensor(wc, dtype=torch.float64)).sum()
        gu, gx = torch.autograd.grad(loss, [Ur, y0])
        out[f"tvjp_{integ}_gu_f64"], out[f"tvjp_{integ}_gx0_f64"] = gu.numpy(), gx.numpy()
    return out


def main():
    cfg = yaml.safe_load(open("cartpole_mpc_config.yaml"))
    CASES = [(1, 20), (8, 50), (4, 100), (2, 200)]
    Qc = np.diag([10.0, 200.0, 1.0, 10.0])
    Rc = np.diag([0.01])
    xlo_c = np.array([-1.0, -0.3, -0.5, -0.5])
    xhi_c = -xlo_c

    # ------------------------------------------------------------------ G1 weights
    torch.manual_seed(0)
    phnn = pHNN("cartpole_mpc_config.yaml")
    np.savez(os.path.join(OUT, "weights_phnn_cartpole.npz"), **sd_numpy(phnn))

    torch.manual_seed(0)
    can = pHNN_Canonical("cartpole_mpc_config.yaml")
    # seed-0 construction leaves M_net / R_diag_raw at their constants; perturb them so the mass-matrix
    # and dissipation paths are exercised with non-trivial values (still "weights", i.e. data).
    with torch.no_grad():
        can.M_net.log_a.fill_(0.30)
        can.M_net.b.fill_(0.35)
        can.M_net.log_c.fill_(-0.20)
        can.R_diag_raw.copy_(torch.tensor([0.10, -0.40, 0.70, 0.25]))
    np.savez(os.path.join(OUT, "weights_canonical_cartpole.npz"), **sd_numpy(can))

    pend = pHNN("pendulum_config.yaml")
    pend.load_state_dict(torch.load("pendulum_pHNN_weights.pth", weights_only=True))
    np.savez(os.path.join(OUT, "weights_phnn_pendulum.npz"), **sd_numpy(pend))

    torch.manual_seed(0)
    ode = ODEFunc(2, 1)
    # xavier init leaves all biases zero; give them seeded values so the bias path is covered
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for mod in ode.modules():
            if isinstance(mod, nn.Linear):
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    np.savez(os.path.join(OUT, "weights_odefunc_pendulum.npz"), **sd_numpy(ode))
    ode_model = OdeAdapter(ode)

    # ------------------------------------------------------------------ G2-G4 per model
    blk = model_block("phnn_cartpole", phnn, 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0,
                      xlo_c, xhi_c, 5.0, 101, CASES)
    # one extra case: non-symmetric full Q, non-zero target, 16 rollouts
    rng = np.random.default_rng(77)
    Qf = Qc + rng.normal(size=(4, 4)) * 0.5
    xt = np.array([0.2, -0.05, 0.1, 0.0])
    x0 = rng.uniform(xlo_c, xhi_c, size=(16, 4)).astype(np.float32)
    U = rng.uniform(-20, 20, size=(16, 30, 1)).astype(np.float32)
    blk["fullq_Q"], blk["fullq_xt"], blk["fullq_x0"], blk["fullq_U"] = Qf, xt, x0, U
    for integ in ("euler", "rk4"):
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            traj, cost, gu, gx = rollout_case(phnn, x0, U, 0.02, integ, Qf, Rc, xt, -15.0, 15.0, dtype)
            blk[f"fullq_{integ}_cost_{tag}"], blk[f"fullq_{integ}_gu_{tag}"] = cost, gu
            blk[f"fullq_{integ}_gx0_{tag}"] = gx
    np.savez(os.path.join(OUT, "golden_phnn_cartpole.npz"), **blk)

    blk = model_block("canonical_cartpole", can, 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0,
                      xlo_c, xhi_c, 5.0, 202, CASES)
    np.savez(os.path.join(OUT, "golden_canonical_cartpole.npz"), **blk)

    Qp, Rp = np.diag([10.0, 1.0]), np.diag([0.01])
    xlo_p = np.array([-np.pi, -1.0])
    blk = model_block("phnn_pendulum", pend, 2, 1, 0.05, Qp, Rp, np.zeros(2), -2.0, 2.0,
                      xlo_p, -xlo_p, 2.5, 303, CASES)
    # ---- G7: 10-step rollouts from [0.5, 0.1], u = 0 (SURVEY 8c sanity anchor)
    x0 = np.array([[0.5, 0.1]], np.float32)
    U = np.zeros((1, 10, 1), np.float32)
    for integ in ("euler", "rk4"):
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            traj, cost, gu, gx = rollout_case(pend, x0, U, 0.05, integ, Qp, Rp, np.zeros(2), -2.0, 2.0, dtype)
            blk[f"g7_{integ}_traj_{tag}"] = traj
    np.savez(os.path.join(OUT, "golden_phnn_pendulum.npz"), **blk)

    blk = model_block("odefunc_pendulum", ode_model, 2, 1, 0.05, Qp, Rp, np.zeros(2), -2.0, 2.0,
                      xlo_p, -xlo_p, 2.5, 404, CASES)
    np.savez(os.path.join(OUT, "golden_odefunc_pendulum.npz"), **blk)

    # a pHNN with hidden widths no kernel is instantiated for (H_mlp [96, 80], R_mlp [48]): the engine embeds it
    # exactly in the 128-wide kernels by zero padding
    import tempfile
    cfg_odd = yaml.safe_load(open("cartpole_mpc_config.yaml"))
    cfg_odd["model"]["H_mlp"]["hidden_sizes"] = [96, 80]
    cfg_odd["model"]["R_mlp"]["hidden_sizes"] = [48]
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as tf:
        yaml.safe_dump(cfg_odd, tf)
    torch.manual_seed(3)
    phnn_odd = pHNN(tf.name)
    os.unlink(tf.name)
    np.savez(os.path.join(OUT, "weights_phnn_cartpole_odd.npz"), **sd_numpy(phnn_odd))
    blk = model_block("phnn_cartpole_odd", phnn_odd, 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0,
                      xlo_c, xhi_c, 5.0, 606, CASES)
    np.savez(os.path.join(OUT, "golden_phnn_cartpole_odd.npz"), **blk)

    # the reference's DEFAULT ODEFunc: state_dim=4, action_dim=1 (cart-pole), same treatment
    torch.manual_seed(0)
    ode4 = ODEFunc(4, 1)
    with torch.no_grad():
        g4 = torch.Generator().manual_seed(2)
        for mod in ode4.modules():
            if isinstance(mod, nn.Linear):
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g4))
    np.savez(os.path.join(OUT, "weights_odefunc_cartpole.npz"), **sd_numpy(ode4))
    blk = model_block("odefunc_cartpole", OdeAdapter(ode4), 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0,
                      xlo_c, xhi_c, 5.0, 505, CASES)
    np.savez(os.path.join(OUT, "golden_odefunc_cartpole.npz"), **blk)

    # ------------------------------------------------------------------ G5 / G6 / G9 controllers
    ctl = {}
    mpc = cfg["mpc"]
    x_init = np.array([0.0, 0.1, 0.0, 0.0], np.float32)
    # G5: MPCController (src/mpc_controller.py:143-209) with the mapping of scripts/run_cartpole_mpc.py:57-88
    c = MPCController(phnn_model=phnn, horizon=mpc["horizon"], dt=cfg["cartpole"]["dt"], Q=mpc["Q_diag"],
                      R=mpc["R_diag"][0], target_state=mpc["x_target"], u_min=mpc["u_min"], u_max=mpc["u_max"],
                      optimizer_type="Adam", lr=mpc["learning_rate"], max_iterations=mpc["optimizer_steps"])
    costs = []
    orig = c.compute_cost

    def logged(states, controls):
        v = orig(states, controls)
        costs.append(float(v.item()))
        return v

    c.compute_cost = logged
    u0 = c.compute_control(x_init.copy())
    ctl["mpc_x0"], ctl["mpc_u0"], ctl["mpc_costs"] = x_init, np.asarray(u0), np.asarray(costs)
    ctl["mpc_horizon"], ctl["mpc_lr"], ctl["mpc_iters"] = (np.int32(mpc["horizon"]), np.float64(mpc["learning_rate"]),
                                                             np.int32(mpc["optimizer_steps"]))
    # single fwd+bwd through the controller's own B=1 path, zero controls (SURVEY 8c anchor)
    cs = torch.zeros(mpc["horizon"], 1, requires_grad=True)
    st = c.rollout_dynamics(torch.tensor(x_init), torch.clamp(cs, mpc["u_min"], mpc["u_max"]))
    cost = orig(st, torch.clamp(cs, mpc["u_min"], mpc["u_max"]))
    cost.backward()
    ctl["mpc_zero_cost"], ctl["mpc_zero_grad"], ctl["mpc_zero_states"] = (np.float32(cost.item()), cs.grad.numpy().copy(),
                                                                          st.detach().numpy())
    # G6: MPCControllerCanonical.control, cold then warm-started (src/mpc_controller_canonical.py:230-273)
    cc = create_mpc_controller(can, cfg)
    u_a, info_a = cc.control(x_init.copy(), None)
    u_b, info_b = cc.control(np.array([0.001, 0.098, 0.02, -0.05], np.float32), info_a["u_sequence"])
    ctl["can_u_a"], ctl["can_useq_a"] = u_a, info_a["u_sequence"]
    ctl["can_costs_a"], ctl["can_final_a"] = np.asarray(info_a["optimization"]["costs"]), np.float64(info_a["optimization"]["final_cost"])
    ctl["can_x_b"] = np.array([0.001, 0.098, 0.02, -0.05], np.float32)
    ctl["can_u_b"], ctl["can_useq_b"] = u_b, info_b["u_sequence"]
    ctl["can_costs_b"], ctl["can_final_b"] = np.asarray(info_b["optimization"]["costs"]), np.float64(info_b["optimization"]["final_cost"])
    # G9: soft state barrier (src/mpc_controller.py:96-107), B=1, H=20
    xmin, xmax = [-0.02, -0.09, -0.3, -0.2], [0.03, 0.095, 0.25, 0.15]
    cb = MPCController(phnn_model=phnn, horizon=20, dt=0.02, Q=mpc["Q_diag"], R=0.01, target_state=mpc["x_target"],
                       u_min=-15.0, u_max=15.0, x_min=xmin, x_max=xmax, lr=0.015, max_iterations=5)
    rngb = np.random.default_rng(9)
    ub = rngb.uniform(-6, 6, size=(20, 1)).astype(np.float32)
    ubt = torch.tensor(ub, requires_grad=True)
    st = cb.rollout_dynamics(torch.tensor(x_init), torch.clamp(ubt, -15.0, 15.0))
    cost = cb.compute_cost(st, torch.clamp(ubt, -15.0, 15.0))
    cost.backward()
    ctl["bar_xmin"], ctl["bar_xmax"], ctl["bar_u"] = np.asarray(xmin), np.asarray(xmax), ub
    ctl["bar_cost"], ctl["bar_grad"], ctl["bar_states"] = np.float32(cost.item()), ubt.grad.numpy().copy(), st.detach().numpy()
    ctl["bar_u0_after5"] = np.asarray(cb.compute_control(x_init.copy()))
    # G13: the L-BFGS branch of MPCController (src/mpc_controller.py:169-170,196-197), 3 outer iterations
    cl = MPCController(phnn_model=phnn, horizon=20, dt=0.02, Q=mpc["Q_diag"], R=0.01, target_state=mpc["x_target"],
                       u_min=-15.0, u_max=15.0, optimizer_type="LBFGS", lr=0.5, max_iterations=3)
    ctl["lbfgs_u0"] = np.asarray(cl.compute_control(x_init.copy()))
    # G11: the reference plant (src/cartpole_simulator.py:63-112), 3 plants x 60 steps of seeded forces
    from cartpole_simulator import CartPoleSimulator
    rngp = np.random.default_rng(12)
    init = rngp.uniform(-1, 1, size=(3, 4)) * np.array([0.5, 0.1, 0.3, 0.3])
    forces = rngp.uniform(-15, 15, size=(60, 3))
    forces[:, 2] = 14.0  # drives plant 2 over the termination limits
    traj, dones = [], []
    for b in range(3):
        sim = CartPoleSimulator(dt=0.02)
        sim.reset(init[b])
        tb, db = [init[b].copy()], []
        for t in range(60):
            s_, d_ = sim.step(np.array([forces[t, b]]))
            tb.append(s_)
            db.append(d_)
        traj.append(np.stack(tb))
        dones.append(np.array(db))
    ctl["plant_init"], ctl["plant_forces"] = init, forces
    ctl["plant_states"], ctl["plant_done"] = np.stack(traj, axis=1), np.stack(dones, axis=1)
    # G12: coordinate transforms with the canonical model's mass matrix (src/coordinate_transforms.py:20-130)
    import coordinate_transforms as CT
    rngc = np.random.default_rng(13)
    yk = (rngc.uniform(-1, 1, size=(32, 4)) * np.array([1.0, 3.0, 2.0, 2.0])).astype(np.float32)
    yt = torch.tensor(yk)
    zc = CT.kinematic_to_canonical(yt, can.M_net)
    ctl["ct_y"], ctl["ct_z"] = yk, zc.detach().numpy()
    ctl["ct_y_back"] = CT.canonical_to_kinematic(zc, can.M_net).detach().numpy()
    ctl["ct_p"] = CT.velocity_to_momentum(yt[:, :2], yt[:, 2:], can.M_net).detach().numpy()
    ctl["ct_vrec"] = can.get_velocity_reconstruction(yt).detach().numpy()
    np.savez(os.path.join(OUT, "golden_controllers.npz"), **ctl)

    # ------------------------------------------------------------------ G8 dataset windows
    d = torch.load("data/cartpole_training_data.pt", weights_only=True)
    S, C = d["states"].numpy(), d["controls"].numpy()
    rng = np.random.default_rng(8)
    idx = rng.choice(S.shape[0], size=16, replace=False)
    start = rng.integers(0, S.shape[1] - 41, size=16)
    x0 = np.stack([S[i, s] for i, s in zip(idx, start)]).astype(np.float32)
    U = np.stack([C[i, s:s + 40] for i, s in zip(idx, start)]).astype(np.float32)
    win = {"x0": x0, "U": U}
    for nm, mdl in (("phnn", phnn), ("canonical", can)):
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            traj, cost, gu, gx = rollout_case(mdl, x0, U, 0.02, "euler", Qc, Rc, np.zeros(4), -15.0, 15.0, dtype)
            ok = np.isfinite(cost) & np.isfinite(gu).all(axis=(1, 2))
            win[f"{nm}_cost_{tag}"], win[f"{nm}_gu_{tag}"], win[f"{nm}_finite_{tag}"] = cost, gu, ok
    np.savez(os.path.join(OUT, "golden_dataset_windows.npz"), **win)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
