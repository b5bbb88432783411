#!/usr/bin/env python3
"""bench.py -- the headline measurement: pHNN-MPC rollouts+grads/sec, cart-pole pHNN, H=50, batch 65536 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks ITSELF: the parent
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process before anything touches
the GPU (it never imports torch), relays the child's output and exits with its return code.  A WORLD_SIZE that
disagrees with --gpus is an error, so a scaling run can never silently report one GPU.

A "step" is one pass of the hot path over one batch resident in HBM: K1 (fused forward march: clamp, pHNN
dynamics, Euler step, stage cost over the whole horizon) then K2 (adjoint march -> d cost / d u), and, for
N > 1, the one exchange the path has: an RCCL all-gather of the per-rollout costs.  Each rank works on its
own 65536 rollouts (weak scaling; rollouts are independent).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      for the dominant kernel (K2, k_rollout_grad): algorithmic FLOPs per launch (SURVEY.md 8d:
                74.2 kFLOP per rollout-step of VJP work) / average launch duration measured live with events on
                the launch stream.  Peak: the dense MFMA peak of the matrix dtype divided by the number of split
                products one f32 product costs (f16x2: 2500/3 TFLOP/s; bf16x3: 2500/6; f32: 157.3); the fraction
                of the f32 MFMA/vector peak is given next to it (vs_f32_mfma_peak, > 1 means the f32 ALUs could
                not have done it).  traffic = measured HBM bytes per K2 launch (separate --pmc passes), mostly
                the K1->K2 activation stash; hbm_* fields compare algorithmic and measured bytes with the 8 TB/s
                roof, as north_star asks.
  cpu_baseline  the CPU oracle (plain-C port of the reference algorithm, float32, OpenMP over rollouts) timed
                on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per rollout-step, cart-pole pHNN
FLOP_FWD = 73.0e3  # f(x,u): H_net value+grad, R_net, combine
FLOP_VJP = 72.7e3 + 1.5e3  # (df/dx)^T lam incl. Hessian-vector product
PEAK_F32_MFMA = 157.3e12   # dense f32 MFMA = f32 vector peak (MI355X_MICROARCH.md)
PEAK_16BIT_MFMA = 2.5e15   # dense bf16 / f16 MFMA peak
PEAK_HBM = 8.0e12
# matrix instructions issued per f32-equivalent product in each matmul mode
SPLIT_PRODUCTS = {"f32": 1, "bf16x3": 6, "f16x2": 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preheat", type=int, default=0,
                    help="extra untimed passes before the W warmup steps (same count on every rank; reported as "
                         "preheat_steps).  Off by default: the first ~0.5 s after the idle clock run a few percent above "
                         "the rate the part sustains at its power cap (28.4 M/s over 30 s), so preheating a short run "
                         "would flatter it")
    ap.add_argument("--batch", type=int, default=65536, help="rollouts per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--model", default="phnn_cartpole",
                    choices=["phnn_cartpole", "canonical_cartpole", "phnn_pendulum", "odefunc_pendulum"])
    ap.add_argument("--integrator", default="euler", choices=["euler", "rk4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stash", action="store_true", help="K2 recomputes the forward tape instead of reading K1's")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="rollouts timed on the host for cpu_baseline")
    ap.add_argument("--matmul", default=None, choices=["default", "f32", "bf16x3", "f16x2"],
                    help="how the hidden x hidden products are evaluated (default: PHNN_MATMUL or the library default)")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the short timed loops of the other matmul modes (other_modes on the JSON line)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch rollouts per GPU (default); strong: --global-batch rollouts split over the GPUs")
    ap.add_argument("--global-batch", type=int, default=65536, help="total rollouts with --scaling strong")
    ap.add_argument("--rehearsal-engine", default=None, metavar="MODULE:FACTORY",
                    help="REHEARSAL ONLY (tests/test_bench_spawn.py): run the multi-rank flow on CPU tensors with the engine "
                         "FACTORY(weights) of MODULE (a stand-in that serves the engine protocol) and the gloo backend; the "
                         "line is marked rehearsal and carries no roofline -- it is not a measurement")
    ap.add_argument("--config", default=None, choices=["headline", "config4"],
                    help="named workloads: headline = H=50, 65536 rollouts per GPU; config4 = BASELINE config 4, "
                         "2^20 rollouts over 8 GPUs = 131072 per GPU")
    return ap.parse_args()


def synthetic_inputs(n, B, H, rank, u_amp):
    """x0 ranges of scripts/run_cartpole_mpc_enhanced.py:137-140, u ~ U(-5,5)."""
    rng_x = np.random.default_rng(1234 + rank)
    rng_u = np.random.default_rng(5678 + rank)
    scale = np.array([1.0, 0.3, 0.5, 0.5][:n]) if n == 4 else np.array([np.pi, 1.0])
    x0 = (rng_x.uniform(-1, 1, size=(B, n)) * scale).astype(np.float32)
    U = rng_u.uniform(-u_amp, u_amp, size=(B, H, 1)).astype(np.float32)
    return x0, U


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as a CHILD torch.distributed.run and exit with its code.

    Runs before torch is imported, so the parent never initialises the GPU (replacing or forking a process that has
    is not allowed on the GPU boxes).  The child's stdout is relayed line by line: rank 0's JSON line is the last
    line that parses as JSON; it is re-printed last so that a consumer reading the final line finds it."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(len(os.sched_getaffinity(0)) // args.gpus, 1)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line_json = None
    for line in proc.stdout:
        try:
            if isinstance(json.loads(line), dict):
                line_json = line
                continue
        except ValueError:
            pass
        sys.stdout.write(line)
    rc = proc.wait()
    if line_json is not None:
        sys.stdout.write(line_json)
    sys.stdout.flush()
    if rc == 0 and line_json is None:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # a launcher started another number of ranks than the line would claim: refuse
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it with "
                 f"--nproc-per-node {args.gpus} (or plain `python bench.py --gpus {args.gpus}`, which launches the ranks itself)")
    import torch
    import torch.distributed as dist

    rehearsal = args.rehearsal_engine is not None
    # PHNN_BENCH_BACKEND=gloo lets the N>1 flow be rehearsed with several ranks on ONE GPU (RCCL refuses two ranks
    # on a device); the driver's runs use the default, nccl (= RCCL) with one GPU per rank.
    backend = "gloo" if rehearsal else os.environ.get("PHNN_BENCH_BACKEND", "nccl")
    dev_index = 0 if rehearsal else local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not rehearsal:
            torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", dev_index)
    collective = {"nccl": "RCCL all-gather of costs", "gloo": "gloo all-gather of costs through host copies (rehearsal backend)"
                  }.get(backend, f"{backend} all-gather of costs")

    from phnn_mpc_amd import _capi
    from phnn_mpc_amd.distributed import shard_bounds
    from phnn_mpc_amd.engine import RolloutEngine
    import ctypes as C
    gold = os.path.join(ROOT, "tests", "golden")
    with np.load(os.path.join(gold, f"weights_{args.model}.npz")) as z:
        w = {k: z[k] for k in z.files}
    if rehearsal:  # a CPU stand-in behind the engine protocol: exercises the launch / shard / gather flow only
        import importlib
        mod, fac = args.rehearsal_engine.split(":")
        eng = getattr(importlib.import_module(mod), fac)(w)
    else:
        eng = RolloutEngine(w, dev, matmul=args.matmul)
    if args.config == "config4":
        args.batch = 131072
    if args.scaling == "strong":  # fixed total work: the global batch is split contiguously over the ranks
        lo, hi = shard_bounds(args.global_batch, world, rank)
        B, B_pad, total = hi - lo, -(-args.global_batch // world), args.global_batch
    else:
        B, B_pad, total = args.batch, args.batch, world * args.batch
    n, H = eng.n, args.horizon
    cart = n == 4
    dt = 0.02 if cart else 0.05
    umax = 15.0 if cart else 2.0
    cost = _capi.make_cost(n, 1, [10.0, 200.0, 1.0, 10.0] if cart else [10.0, 1.0], [0.01], None, -umax, umax)
    cost_ref = C.byref(cost)
    integ = _capi.INTEGRATORS[args.integrator]
    x0_h, U_h = synthetic_inputs(n, max(B, 1), H, rank, 5.0 if cart else 2.0)
    x0 = torch.from_numpy(x0_h).to(dev)
    U = torch.from_numpy(U_h).to(dev)

    class Workload:
        """One rank's share of the hot path: K1, (all-gather of the costs), K2 on Bk rollouts resident in HBM."""

        def __init__(self, engine, Bk, Bk_pad, stash=True):
            self.eng, self.B = engine, Bk
            self.traj = torch.empty(Bk, H + 1, n, dtype=torch.float32, device=dev)
            self.gu = torch.empty(Bk, H, 1, dtype=torch.float32, device=dev)
            self.c_pad = torch.zeros(Bk_pad, dtype=torch.float32, device=dev)  # K1 writes the first Bk entries
            self.cost = self.c_pad[:Bk]
            nst = engine.workspace_bytes(Bk, H, integ) if (engine.use_stash and stash) else 0
            self.stash = torch.empty(nst, dtype=torch.uint8, device=dev) if nst > 0 else None
            self.gathered = torch.empty(world * Bk_pad, dtype=torch.float32, device=dev) if world > 1 else None

        def step(self, ev=None):
            e, Bk = self.eng, self.B
            if ev is not None:
                ev[0].record()
            e.lib.phnn_rollout_fwd(e.h, e._p(x0), e._p(U), Bk, H, cost_ref, integ, float(dt), e._p(self.cost),
                                   e._p(self.traj), e._p(self.stash), e._stream())
            if ev is not None:
                ev[1].record()
            work = None
            if world > 1 and backend == "nccl":  # costs are final after K1: the gather runs on RCCL's stream under K2
                work = dist.all_gather_into_tensor(self.gathered, self.c_pad, async_op=True)
            e.lib.phnn_rollout_grad(e.h, e._p(x0), e._p(U), Bk, H, cost_ref, integ, float(dt), e._p(self.traj),
                                    e._p(self.stash), e._p(self.gu), None, e._stream())
            if ev is not None:
                ev[2].record()
            if world > 1:
                if work is not None:
                    work.wait()  # stream-level wait: the next K1 must not overwrite the costs before the gather read them
                else:  # rehearsal backend: collectives on host copies
                    parts = [torch.empty(self.c_pad.numel(), dtype=torch.float32) for _ in range(world)]
                    dist.all_gather(parts, self.c_pad.cpu())
                    self.gathered.copy_(torch.cat(parts))
            return self.cost

        def timed(self, steps, warmup):
            """W untimed steps, then exactly `steps` steps bracketed by barrier + synchronize; max over ranks."""
            for _ in range(warmup):
                self.step()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
            t0 = time.perf_counter()
            for k in range(steps):
                c_last = self.step(evs[k])
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            k1 = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
            k2 = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
            assert torch.isfinite(c_last).all(), "non-finite cost in the bench workload"
            return el, k1, k2

    class RehearsalWorkload(Workload):
        """The same flow -- K1, all-gather of the costs, K2, barriers, max over ranks -- on a CPU stand-in engine
        (engine protocol: rollout_cost_grad(..., after_forward=)).  Host clocks; not a measurement."""

        def __init__(self, engine, Bk, Bk_pad, stash=True):
            self.eng, self.B, self.stash = engine, Bk, None
            self.c_pad = torch.zeros(Bk_pad, dtype=torch.float32)
            self.gathered = torch.empty(world * Bk_pad, dtype=torch.float32) if world > 1 else None
            self.mark = [0.0, 0.0, 0.0]

        def step(self, ev=None):
            self.mark[0] = time.perf_counter()

            def gather(c):
                self.mark[1] = time.perf_counter()
                self.c_pad[: self.B] = c
                if world > 1:
                    dist.all_gather_into_tensor(self.gathered, self.c_pad)

            c, _ = self.eng.rollout_cost_grad(x0[: self.B], U[: self.B], cost, args.integrator, dt, after_forward=gather)
            self.mark[2] = time.perf_counter()
            if ev is not None:
                ev.extend(self.mark)
            return c

        def timed(self, steps, warmup):
            for _ in range(warmup):
                self.step()
            if world > 1:
                dist.barrier()
            marks = [[] for _ in range(steps)]
            t0 = time.perf_counter()
            for k in range(steps):
                c_last = self.step(marks[k])
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            assert torch.isfinite(c_last).all()
            return el, 1e3 * float(np.mean([m[1] - m[0] for m in marks])), 1e3 * float(np.mean([m[2] - m[1] for m in marks]))

    WL = RehearsalWorkload if rehearsal else Workload
    main_wl = WL(eng, B, B_pad, stash=not args.no_stash)
    ws_stash = main_wl.stash
    elapsed, k1_ms, k2_ms = main_wl.timed(args.steps, args.preheat + args.warmup)

    # strong-scaling figure next to the weak one (N > 1): the SAME global batch of 65536 split over the ranks
    strong = None
    if world > 1 and args.scaling == "weak":
        lo, hi = shard_bounds(args.global_batch, world, rank)
        Bs, Bs_pad = hi - lo, -(-args.global_batch // world)
        if Bs <= B:
            del main_wl
            torch.cuda.empty_cache()
            ks = max(min(args.steps, 50), 1)
            el_s, k1s, k2s = WL(eng, Bs, Bs_pad, stash=not args.no_stash).timed(ks, 3)
            strong = {"global_batch": args.global_batch, "batch_per_gpu": Bs_pad, "steps": ks,
                      "value": round(args.global_batch * ks / el_s, 1), "ms_per_step": round(1e3 * el_s / ks, 4),
                      "k1_launch_ms": round(k1s, 4), "k2_launch_ms": round(k2s, 4)}
    # the 24-bit-exact product modes, short timed loops in the same run (single GPU only)
    other_modes = None
    if world == 1 and not rehearsal and not args.no_other_modes and args.model in ("phnn_cartpole", "canonical_cartpole"):
        other_modes = {}
        for mode in ("bf16x3", "f32"):
            if mode == eng.matmul_mode:
                continue
            e2 = RolloutEngine(w, dev, matmul=mode)
            ko = max(min(args.steps, 50), 1)  # same warm-up as the main loop (first launches load the code objects)
            el_o, k1o, k2o = Workload(e2, B, B_pad, stash=not args.no_stash).timed(ko, args.preheat + args.warmup)
            pk = PEAK_F32_MFMA if mode == "f32" else PEAK_16BIT_MFMA / SPLIT_PRODUCTS[mode]
            fl = B * H * (4 if args.integrator == "rk4" else 1) * FLOP_VJP
            other_modes[mode] = {"value": round(B * ko / el_o, 1), "steps": ko, "warmup": args.preheat + args.warmup,
                                 "k1_launch_ms": round(k1o, 4), "k2_launch_ms": round(k2o, 4), "kernel_variant": e2.variant,
                                 "roofline": {"kernel": "k_rollout_grad", "achieved": round(fl / (k2o * 1e-3) / 1e12, 3),
                                              "peak": round(pk / 1e12, 1), "unit": "TFLOP/s", "frac": round(fl / (k2o * 1e-3) / pk, 4)}}
            e2.close()

    if rank == 0:
        stages = 4 if args.integrator == "rk4" else 1
        value = total * args.steps / elapsed
        flop_k2 = B * H * stages * FLOP_VJP  # algorithmic, per launch
        flop_job = B * H * stages * (FLOP_FWD + FLOP_VJP)
        bytes_k2 = B * (4 * (n + 2 * H) + 4)  # 4(n + 2 H m) + 4 per rollout with gradient
        ach = flop_k2 / (k2_ms * 1e-3)
        # HBM bytes per K2 launch from the PMC counters: measured in separate rocprofv3 --pmc passes of this same
        # command (profiles/traffic_measured.json), reported only for the configuration it was measured on
        traffic = traffic_k1 = None
        mode = "stash" if ws_stash is not None else "recompute"
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_measured.json")) as f:
                tm = json.load(f)
            entry = tm.get(f"{args.model}:{args.integrator}:B{B}:H{H}:{mode}", {})
            traffic = entry.get("K2", {}).get("hbm_bytes_per_launch")
            traffic_k1 = entry.get("K1", {}).get("hbm_bytes_per_launch")
        except OSError:
            pass
        mm = getattr(eng, "matmul_mode", "f32")
        peak = PEAK_F32_MFMA if mm == "f32" else PEAK_16BIT_MFMA / SPLIT_PRODUCTS[mm]
        roof = {
            "bound": "mfma", "kernel": "k_rollout_grad", "achieved": round(ach / 1e12, 3), "peak": round(peak / 1e12, 1),
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "peak_basis": ("dense f32 MFMA peak" if mm == "f32" else
                           f"dense 16-bit MFMA peak 2500 TFLOP/s / {SPLIT_PRODUCTS[mm]} split products per f32 product ({mm})"),
            "vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA, 4),
            "traffic_unit": "HBM bytes per K2 launch (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": "profiles/traffic_measured.json (separate rocprofv3 --pmc passes of this command; not observed in this run)",
            "launch_ms": round(k2_ms, 4), "k1_launch_ms": round(k1_ms, 4),
            "job_tflops": round(flop_job / ((k1_ms + k2_ms) * 1e-3) / 1e12, 3),
            "job_frac": round(flop_job / ((k1_ms + k2_ms) * 1e-3) / peak, 4),
            "job_vs_f32_mfma_peak": round(flop_job / ((k1_ms + k2_ms) * 1e-3) / PEAK_F32_MFMA, 4),
            "hbm_algorithmic_GBps": round(bytes_k2 / (k2_ms * 1e-3) / 1e9, 3),
            "hbm_algorithmic_frac": round(bytes_k2 / (k2_ms * 1e-3) / PEAK_HBM, 6),
            "hbm_traffic_frac": None if traffic is None else round(traffic / (k2_ms * 1e-3) / PEAK_HBM, 4),
            # K1 in stash mode streams the adjoint's workspace out: mostly HBM writes, and it is that stream -- not
            # its arithmetic (0.87 ms without the stash) -- that sets K1's time
            "k1_traffic": traffic_k1,
            "k1_hbm_traffic_frac": None if traffic_k1 is None else round(traffic_k1 / (k1_ms * 1e-3) / PEAK_HBM, 4),
        }
        cpu = cpu_torch = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(w, cost, x0_h, U_h, args.integrator, dt, args.cpu_sample)
            if args.model == "phnn_cartpole" and args.integrator == "euler":
                cpu_torch = cpu_baseline_torch(w, x0_h, U_h, dt, umax)
        # the labels follow the workload actually run (the default run reproduces BASELINE.json's metric wording)
        short = {"phnn_cartpole": "cartpole", "canonical_cartpole": "cartpole pHNN_canonical", "phnn_pendulum": "pendulum",
                 "odefunc_pendulum": "pendulum ODEFunc"}[args.model]
        batch_lbl = f"batch={B}" if args.scaling == "weak" else f"global batch={total} (strong scaling, {B_pad}/GPU)"
        out = {
            "metric": f"pHNN-MPC rollouts+grads/sec, {short}{'' if args.integrator == 'euler' else ' RK4'} H={H} {batch_lbl}",
            "value": round(value, 1),
            "unit": "rollouts+grads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_steps": args.preheat,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} (seed-0 fixture weights) {args.integrator} H={H} "
                                   f"B={B}/GPU: rollout + stage cost (K1) + control gradient (K2)"
                                   + (f" + {collective}" if world > 1 else ""),
                       "k2_mode": "stash" if ws_stash is not None else "recompute", "matmul": mm,
                       "kernel_variant": getattr(eng, "variant", type(eng).__name__), "horizon": H, "batch_per_gpu": B_pad,
                       "global_batch": total, "parallelism": f"shard{world}", "collective_backend": backend if world > 1 else None},
            "roofline": roof,
        }
        try:  # which product mode the parity margin measured on TRAINED weights allows (tests/parity_margin_trained.py)
            with open(os.path.join(ROOT, "profiles", "parity_margin_trained.json")) as f:
                pm = json.load(f)
            out["config"]["trained_weights_margin"] = {"mode": mm, "cost_grad": pm.get(mm), "unit": pm.get("unit"),
                                                        "reading": pm.get("reading")}
        except (OSError, ValueError):
            pass
        if rehearsal:  # the flow ran on a CPU stand-in: the line proves the launch / shard / gather plumbing, nothing else
            out["rehearsal"] = True
            out["data"] = "synthetic (REHEARSAL: CPU stand-in engine, host clocks -- not a measurement)"
            out["roofline"] = None
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if cpu_torch is not None:
            out["cpu_baseline_torch"] = cpu_torch
        if other_modes:
            out["other_modes"] = other_modes
        if strong is not None:
            out["strong_scaling"] = strong
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def host_cpu():
    """(usable cores, CPU model string).  Cores = scheduler affinity, capped by the cgroup CPU quota when one is set
    (the GPU box gives one GPU's job a share of the host); no hard-coded cap."""
    cores = len(os.sched_getaffinity(0))
    src = "sched_getaffinity"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            quota = max(int(int(q) / int(per)), 1)
            if quota < cores:
                cores, src = quota, "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return cores, src, model


def cpu_baseline(w, cost, x0_h, U_h, integ, dt, sample):
    """Time the float32 CPU oracle (test infrastructure, used here only as the measured baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores, src, model = host_cpu()
    m = ol.OracleModel(w, "f32")
    t = time.perf_counter()
    m.rollout(x0_h[:256], U_h[:256], cost, integ, dt, traj=False, nthreads=cores)  # warm + pilot
    pilot = 256 / max(time.perf_counter() - t, 1e-6)
    S = int(min(max(sample, 12.0 * pilot), x0_h.shape[0]))  # aim at ~12 s of CPU work, bounded by the batch
    reps = int(min(max(np.ceil(10.0 * pilot / S), 1), 4))  # a fast host: repeat the (bounded) sample to reach ~10 s
    t = time.perf_counter()
    for _ in range(reps):
        m.rollout(x0_h[:S], U_h[:S], cost, integ, dt, traj=False, nthreads=cores)
    el = time.perf_counter() - t
    return {"value": round(reps * S / el, 1), "unit": "rollouts+grads/s", "cores": cores, "cores_source": src,
            "cpu_model": model, "kind": "port",
            "sample": f"first {S} rollouts of the same batch x{reps}, float32 C oracle, OpenMP static over rollouts, {el:.1f} s"}


def cpu_baseline_torch(w, x0_h, U_h, dt, umax):
    """BASELINE.md section 4, second baseline: the stock PyTorch-CPU restatement of the batched oracle
    (oracle/torch_oracle.py: torch ops + autograd, as the reference computes it), all host cores."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from torch_oracle import TorchPhnn
    cores, src, model = host_cpu()
    torch.set_num_threads(cores)
    m = TorchPhnn(w, torch.float32)
    args = ([10.0, 200.0, 1.0, 10.0], 0.01, [0.0, 0.0, 0.0, 0.0], -umax, umax, dt)
    m.rollout_cost_grad(x0_h[:256], U_h[:256], *args)  # warm-up (thread pool, allocator)
    t = time.perf_counter()
    m.rollout_cost_grad(x0_h[:1024], U_h[:1024], *args)
    pilot = 1024 / max(time.perf_counter() - t, 1e-6)
    S = int(min(8192, x0_h.shape[0]))  # one pass bounded by the autograd graph's memory (~2 GB at 8192 x 50 steps)
    reps = int(min(max(np.ceil(8.0 * pilot / S), 1), 24))  # ~8 s of CPU work
    t = time.perf_counter()
    for _ in range(reps):
        m.rollout_cost_grad(x0_h[:S], U_h[:S], *args)
    el = time.perf_counter() - t
    return {"value": round(reps * S / el, 1), "unit": "rollouts+grads/s", "cores": cores, "cores_source": src, "cpu_model": model,
            "kind": "port", "sample": f"first {S} rollouts of the same batch x{reps}, float32 torch ops + autograd "
                                      f"(oracle/torch_oracle.py), torch.set_num_threads({cores}), {el:.1f} s"}


if __name__ == "__main__":
    main()
