"""Multi-GPU: shard the rollout batch, replicate the weights, all-gather the per-rollout costs.

The path partitions into independent units (rollout b depends only on x0[b], u[b] and the weights), so ranks
never exchange states, controls or gradients.  The single collective is one all-gather of cost[B/G] float32 per
pass (RCCL over xGMI when the backend is "nccl"; 512 KB per rank at B = 2^20 over 8 GPUs, latency-bound); it is
started as soon as K1 is enqueued and runs on RCCL's stream underneath K2.
One process per GPU, launched with torch.distributed.run; the reference has no counterpart (no distributed code).
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world, rank):
    """Contiguous split of B rollouts over `world` ranks, sizes differing by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_costs(local_cost, B, group=None):
    """local_cost: this rank's (B_local,) costs -> (B,) costs of the whole batch, in batch order, on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local_cost
    sizes = [shard_bounds(B, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    if len(set(counts)) == 1:
        out = torch.empty(B, dtype=local_cost.dtype, device=local_cost.device)
        dist.all_gather_into_tensor(out, local_cost.contiguous(), group=group)
        return out
    # ragged split: pad to the largest shard (one collective, then drop the padding)
    mx = max(counts)
    padded = torch.zeros(mx, dtype=local_cost.dtype, device=local_cost.device)
    padded[: local_cost.numel()] = local_cost
    buf = torch.empty(world * mx, dtype=local_cost.dtype, device=local_cost.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)])


class _PendingGather:
    """An all-gather of costs in flight (async_op): result() waits for it and returns the (B,) costs in batch order."""

    def __init__(self, work, buf, counts, mx):
        self.work, self.buf, self.counts, self.mx = work, buf, counts, mx

    def result(self):
        self.work.wait()  # the caller's stream waits for the collective; the host does not block on the GPU
        if self.mx is None:
            return self.buf
        return torch.cat([self.buf[r * self.mx: r * self.mx + n] for r, n in enumerate(self.counts)])


def all_gather_costs_async(local_cost, B, group=None):
    """Start the all-gather of all_gather_costs and return a handle; the collective runs on the backend's own stream
    (RCCL) while the caller keeps enqueueing work -- K2 -- on its stream."""
    world = dist.get_world_size(group)
    counts = [hi - lo for lo, hi in (shard_bounds(B, world, r) for r in range(world))]
    if len(set(counts)) == 1:
        out = torch.empty(B, dtype=local_cost.dtype, device=local_cost.device)
        return _PendingGather(dist.all_gather_into_tensor(out, local_cost.contiguous(), group=group, async_op=True), out, counts, None)
    mx = max(counts)
    padded = torch.zeros(mx, dtype=local_cost.dtype, device=local_cost.device)
    padded[: local_cost.numel()] = local_cost
    buf = torch.empty(world * mx, dtype=local_cost.dtype, device=local_cost.device)
    return _PendingGather(dist.all_gather_into_tensor(buf, padded, group=group, async_op=True), buf, counts, mx)


class ShardedRollout:
    """engine: this rank's rollout engine (RolloutEngine bound to the rank's GPU).

    rollout_cost_grad(x0, u, ...) takes the FULL batch description on every rank (or this rank's shard with
    sharded_inputs=True), computes the local shard, and returns (cost of the whole batch (B,), local grad_u,
    (lo, hi)) -- gradients stay local because each rank owns the Adam state of its own rollouts.
    """

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def rollout_cost_grad(self, x0, u, cost, integrator="euler", dt=0.02, B_total=None, sharded_inputs=False,
                          workspace=None):
        if sharded_inputs:
            if B_total is None:
                raise ValueError("B_total is required with sharded_inputs=True")
            lo, hi = shard_bounds(B_total, self.world, self.rank)
            x0l, ul = x0, u
        else:
            B_total = x0.shape[0]
            lo, hi = shard_bounds(B_total, self.world, self.rank)
            x0l, ul = x0[lo:hi], u[lo:hi]
        pending = []

        def start_gather(local_cost):  # called between K1 and K2: the costs are final, the gather overlaps the adjoint
            if self.world > 1:
                pending.append(all_gather_costs_async(local_cost, B_total, self.group))

        c, g = self.engine.rollout_cost_grad(x0l, ul, cost, integrator, dt, workspace=workspace, after_forward=start_gather)
        if pending:
            c = pending[0].result()
        return c, g, (lo, hi)
