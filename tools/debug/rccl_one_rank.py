"""Debug: the RCCL path of bench.py / phnn_mpc_amd.distributed with a one-rank group on the GPU box (what can be
checked without a second GPU): init with device_id, all_gather_into_tensor (sync + async on RCCL's stream), barrier,
all_reduce(MAX) on a float64 CUDA tensor.  Run: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/debug/rccl_one_rank.py"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phnn_mpc_amd.distributed import all_gather_costs, all_gather_costs_async
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
c = torch.arange(1000, dtype=torch.float32, device=dev)
g = all_gather_costs(c, 1000)
assert torch.equal(g, c)
h = all_gather_costs_async(c * 2, 1000)
g2 = h.result()
assert torch.equal(g2, c * 2)
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 3.5
print("rccl one-rank ok: backend", dist.get_backend(), "world", dist.get_world_size())
dist.destroy_process_group()
