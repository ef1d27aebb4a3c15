"""Multi-GPU: one process per GPU, models/chains block-partitioned over ranks.

The forward path has no exchange step -- every evaluation is independent and chains never talk
while sampling (reference src/mcmcOptimizer.py:208-216, src/SingleChain.py:511-589) -- so there is
no collective on the data path.  The only communication is the gather of fixed-shape result blocks
(the reference's per-chain shared arrays, src/mcmcOptimizer.py:92-125, merged offline by
src/Plotting.py:161-262).  Only rank 0 needs them (it writes the files / plots), so the default is a
gather TO THE ROOT: every other rank sends its block once, point to point -- direct peer writes over
its xGMI link to the root, which receives from its 7 neighbours concurrently -- and nothing is
replicated.  `gather_rows` (all-gather, every rank ends up with everything) stays for callers that
want the merged result everywhere.  Backend "nccl" is RCCL on ROCm; gloo on CPU.

Bytes: a rank sends exactly rows_local * row_bytes (no padding); the root receives
(n_total - rows_root) * row_bytes.  SURVEY section 5's example (512 chains x 44 236 rows x 51 float32
= 4.6 GB): 0.58 GB leaves each GPU over one link (~3.8 ms at 153 GB/s), instead of 4.6 GB arriving
at every GPU with the all-gather.  Thinning to the reference's `maxmodels` before the gather
(ChainPool.gather_final) cuts that to <= maxmodels rows per chain.
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
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def _global(group, r):
    """Rank r of `group` as the world rank that send / irecv / gather address."""
    return r if group is None else dist.get_global_rank(group, r)


def gather_rows_to_root(local, n_total, dst=0, group=None):
    """Rank `dst` gets the full [n_total, ...] tensor (blocks in shard_range order), every other
    rank returns None.  Point-to-point: each rank sends its own block, unpadded, once.  `dst` and
    the shard order are ranks WITHIN `group` (a sub-group's ranks need not be 0..n-1 of the world)."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, local.shape[0], sizes[rank]))
    local = local.contiguous()
    if rank != dst:
        if sizes[rank]:
            dist.send(local, dst=_global(group, dst), group=group)
        return None
    full = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    lo = 0
    reqs = []
    for r, n in enumerate(sizes):
        if r == dst:
            full[lo:lo + n] = local
        elif n:
            reqs.append(dist.irecv(full[lo:lo + n], src=_global(group, r), group=group))     # all peers at once
        lo += n
    for q in reqs:
        q.wait()
    return full


def gather_ragged_to_root(local, dst=0, group=None):
    """Row blocks of different, a-priori unknown length (e.g. thinned chains): the root gets the list
    of every rank's block (rank order), the others None.  Lengths travel first (one int64 each).
    `dst` is a rank within `group`."""
    if not dist.is_initialized():
        return [local]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)] if rank == dst else None
    dist.gather(n, counts, dst=_global(group, dst), group=group)
    local = local.contiguous()
    if rank != dst:
        if local.shape[0]:
            dist.send(local, dst=_global(group, dst), group=group)
        return None
    out, reqs = [], []
    for r in range(world):
        k = int(counts[r].item())
        if r == dst:
            out.append(local)
            continue
        buf = torch.empty((k,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if k:
            reqs.append(dist.irecv(buf, src=_global(group, r), group=group))
        out.append(buf)
    for q in reqs:
        q.wait()
    return out


def max_over_ranks(value, device=None, group=None):
    """MAX-reduce a Python float over ranks (bench timing)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
