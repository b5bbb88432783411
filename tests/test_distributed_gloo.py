"""Multi-process CPU test of the N > 1 path (world_size 2, gloo): the batch is sharded, each rank computes its own
rollouts, one all-gather returns every rollout's cost on every rank, bit-identical to the single-process result.
The compute engine here is the CPU oracle behind the engine protocol; on the GPU box the same code runs with
RolloutEngine and backend "nccl" (bench.py --gpus N).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from oracle_engine import OracleEngine
    from phnn_mpc_amd.distributed import ShardedRollout, shard_bounds
    w = ol.load_weights("phnn_cartpole")
    g = ol.load_golden("phnn_cartpole")
    rng = np.random.default_rng(11)
    x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1, 0.3, 0.5, 0.5]).astype(np.float32))
    U = torch.tensor(rng.uniform(-5, 5, size=(B, 12, 1)).astype(np.float32))
    cost = ol.cost_from_golden(g)
    sh = ShardedRollout(OracleEngine(w))
    c_all, g_loc, (lo, hi) = sh.rollout_cost_grad(x0, U, cost, "euler", 0.02)
    assert (lo, hi) == shard_bounds(B, world, rank)
    # same thing with pre-sharded inputs
    c2, g2, _ = sh.rollout_cost_grad(x0[lo:hi], U[lo:hi], cost, "euler", 0.02, B_total=B, sharded_inputs=True)
    assert torch.equal(c_all, c2) and torch.equal(g_loc, g2)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cost=c_all.numpy(), grad=g_loc.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [32, 37])
def test_sharded_rollout_world2_gloo(B, tmp_path):
    sys.path.insert(0, HERE)
    import oracle_lib as ol
    from oracle_engine import OracleEngine
    from phnn_mpc_amd.distributed import shard_bounds
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    w = ol.load_weights("phnn_cartpole")
    g = ol.load_golden("phnn_cartpole")
    rng = np.random.default_rng(11)
    x0 = torch.tensor((rng.uniform(-1, 1, size=(B, 4)) * [1, 0.3, 0.5, 0.5]).astype(np.float32))
    U = torch.tensor(rng.uniform(-5, 5, size=(B, 12, 1)).astype(np.float32))
    c_ref, g_ref = OracleEngine(w).rollout_cost_grad(x0, U, ol.cost_from_golden(g), "euler", 0.02)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(z["cost"], c_ref.numpy())  # every rank holds the whole batch's costs, in order
        lo, hi = int(z["lo"]), int(z["hi"])
        assert (lo, hi) == shard_bounds(B, world, r)
        assert np.array_equal(z["grad"], g_ref.numpy()[lo:hi])


def test_shard_bounds_cover_batch():
    from phnn_mpc_amd.distributed import shard_bounds
    for B in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
