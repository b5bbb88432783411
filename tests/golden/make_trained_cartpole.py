#!/usr/bin/env python3
"""A TRAINED cart-pole pHNN as a fixture, produced by the reference's own training loop.

The reference ships the dataset (data/cartpole_training_data.pt) and the training script
(scripts/train_cartpole_phnn.py) but no trained cart-pole checkpoint (cartpole_mpc_config.yaml:83 names
models/checkpoint_epoch_860.pth; there is no models/ directory).  This script -- build container only, reference
mounted read-only at /root/reference -- imports that training script and calls ITS functions unchanged:

    load_config / load_training_data / create_dataloader / train_phnn   (scripts/train_cartpole_phnn.py:25-211)

with torch.manual_seed(0) before the model is constructed, the config's own schedule (Adam lr 1e-4, batch 16, windows of
16 steps; `--epochs` bounds the epoch count, default 860 = the checkpoint the config names), on CPU.  train_phnn writes
its checkpoints / loss plot relative to the working directory, which is a scratch directory under build/ (git-ignored).

Two phases (the training takes ~8 s per epoch on this container's CPUs):

    python tests/golden/make_trained_cartpole.py train  --epochs 860
    python tests/golden/make_trained_cartpole.py export --checkpoint build/trained_scratch/models/checkpoint_epoch_860.pth

`export` stores data only (numeric arrays):
    weights_phnn_cartpole_trained.npz   the state_dict
    golden_phnn_cartpole_trained.npz    G2 f(x,u), H; G3 VJPs; G4 Euler / RK4 rollouts with cost, grad_u, grad_x0; G10
                                        reverse pass with cotangents -- the same sets make_golden.py stores for the seed-0
                                        weights, on the SAME seeded inputs -- plus
                                        mpc_*   G5  MPCController.compute_control([0, .1, 0, 0]) with per-iteration costs
                                        cl_*    the reference's closed loop, run by its own functions
                                                (scripts/run_cartpole_mpc.py:57-182: create_mpc_from_config,
                                                run_mpc_control on CartPoleSimulator, the config's 300 steps) from
                                                [0, .1, 0, 0]: states, controls, H log, and the "stability achieved"
                                                outcome of cartpole_mpc_config.yaml:69-75 as run_mpc_control evaluates it
                                        train_* epochs run, final state loss
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
SCRATCH = os.path.join(ROOT, "build", "trained_scratch")
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.append(os.path.join(REF, "src"))
sys.path.append(os.path.join(REF, "scripts"))
CFG = os.path.join(REF, "cartpole_mpc_config.yaml")


def train(epochs, threads):
    import train_cartpole_phnn as T  # the reference's training script, imported as a module
    from pHNN import pHNN
    os.makedirs(os.path.join(SCRATCH, "results"), exist_ok=True)
    os.makedirs(os.path.join(SCRATCH, "models"), exist_ok=True)
    os.chdir(SCRATCH)  # train_phnn writes models/checkpoint_epoch_N.pth and results/training_loss.png relative to cwd
    torch.set_num_threads(threads)
    cfg = T.load_config(CFG)
    states, controls, derivs = T.load_training_data(os.path.join(REF, cfg["data"]["save_path"]))
    torch.manual_seed(0)
    model = pHNN(CFG)
    loader = T.create_dataloader(states, controls, derivs, cfg)
    cfg["training"]["epochs"] = int(epochs)
    t0 = time.time()
    T.train_phnn(model, loader, cfg, torch.device("cpu"))
    torch.save(model.state_dict(), os.path.join(SCRATCH, "models", f"final_epoch_{epochs}.pth"))
    print(f"trained {epochs} epochs in {time.time() - t0:.0f} s")


def export(checkpoint, epochs_label):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(OUT, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)  # chdir's to the reference, imports its modules; gives model_block / sd_numpy
    import yaml
    from pHNN import pHNN
    from mpc_controller import MPCController
    from cartpole_simulator import CartPoleSimulator
    torch.set_num_threads(1)
    cfg = yaml.safe_load(open(CFG))
    model = pHNN(CFG)
    sd = torch.load(checkpoint, weights_only=True)
    model.load_state_dict(sd)
    np.savez(os.path.join(OUT, "weights_phnn_cartpole_trained.npz"), **mg.sd_numpy(model))

    Qc, Rc = np.diag([10.0, 200.0, 1.0, 10.0]), np.diag([0.01])
    xlo = np.array([-1.0, -0.3, -0.5, -0.5])
    # the same seeded inputs as golden_phnn_cartpole.npz (seed 101, same cases)
    blk = mg.model_block("phnn_cartpole_trained", model, 4, 1, 0.02, Qc, Rc, np.zeros(4), -15.0, 15.0, xlo, -xlo, 5.0, 101,
                         [(1, 20), (8, 50), (4, 100), (2, 200)])

    # ---- G5 on the trained weights: MPCController.compute_control with the mapping of scripts/run_cartpole_mpc.py:57-88
    mpc = cfg["mpc"]

    def controller():
        return MPCController(phnn_model=model, horizon=mpc["horizon"], dt=cfg["cartpole"]["dt"], Q=mpc["Q_diag"],
                             R=mpc["R_diag"][0], target_state=mpc["x_target"], u_min=mpc["u_min"], u_max=mpc["u_max"],
                             optimizer_type="Adam", lr=mpc["learning_rate"], max_iterations=mpc["optimizer_steps"])

    x_init = np.array([0.0, 0.1, 0.0, 0.0], np.float32)
    c = controller()
    costs, orig = [], c.compute_cost

    def logged(states, controls):
        v = orig(states, controls)
        costs.append(float(v.item()))
        return v

    c.compute_cost = logged
    u0 = c.compute_control(x_init.copy())
    blk["mpc_x0"], blk["mpc_u0"], blk["mpc_costs"] = x_init, np.asarray(u0), np.asarray(costs)

    # ---- the reference's closed loop, by the reference's own driver functions (scripts/run_cartpole_mpc.py:57-182):
    # create_mpc_from_config + run_mpc_control on CartPoleSimulator; "stability achieved" as that function evaluates it
    import run_cartpole_mpc as RC
    x_cl = np.array([0.0, 0.1, 0.0, 0.0])  # the script's initial state family (small pole angle)
    sim = CartPoleSimulator(dt=cfg["cartpole"]["dt"])
    states, controls, hams, achieved, duration = RC.run_mpc_control(sim, RC.create_mpc_from_config(model, cfg), x_cl,
                                                                    mpc["simulation_steps"], cfg, verbose=False)
    blk["cl_x0"], blk["cl_states"], blk["cl_controls"], blk["cl_H"] = x_cl, states, controls, hams
    blk["cl_stability_achieved"], blk["cl_stable_duration"] = np.bool_(achieved), np.float64(duration)
    blk["cl_tolerance"] = np.asarray(cfg["stability"]["tolerance"], np.float64)
    blk["cl_min_duration"] = np.float64(cfg["stability"]["min_duration"])
    blk["train_epochs"] = np.int32(epochs_label)
    np.savez(os.path.join(OUT, "golden_phnn_cartpole_trained.npz"), **blk)
    print(f"exported: epochs {epochs_label}, compute_control -> {np.asarray(u0)}, closed loop: {len(controls)} steps, "
          f"stability achieved {achieved} (stable for {duration:.2f} s), final state {states[-1]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("train")
    a.add_argument("--epochs", type=int, default=860)
    a.add_argument("--threads", type=int, default=2)
    b = sub.add_parser("export")
    b.add_argument("--checkpoint", required=True)
    b.add_argument("--epochs-label", type=int, default=None)
    args = ap.parse_args()
    if args.cmd == "train":
        train(args.epochs, args.threads)
    else:
        import re
        lab = args.epochs_label
        if lab is None:
            m = re.search(r"epoch_(\d+)", args.checkpoint)
            lab = int(m.group(1)) if m else -1
        export(os.path.abspath(args.checkpoint), lab)
