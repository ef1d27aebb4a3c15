"""Multi-GPU: one process per GPU, models/chains block-partitioned over ranks.

The forward path has no exchange step -- every evaluation is independent and chains never talk
while sampling (reference src/mcmcOptimizer.py:208-216, src/SingleChain.py:511-589) -- so there is
no collective on the data path.  The only communication is the gather of fixed-shape result blocks
(the reference's per-chain shared arrays, src/mcmcOptimizer.py:92-125, merged offline by
src/Plotting.py:161-262): an all-gather of equal-size row blocks over RCCL (backend "nccl" on
ROCm) or gloo on CPU.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_rows(local, n_total, group=None):
    """All ranks contribute their block of rows (shard_range order); every rank gets the full
    [n_total, ...] tensor.  Blocks are padded to the largest shard so that a single fixed-shape
    all_gather suffices (direct peer writes on xGMI, no ring of variable-size sends)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def max_over_ranks(value, device=None, group=None):
    """MAX-reduce a Python float over ranks (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
