"""CPU tier: the N>1 path (block sharding + gather of result blocks) with world_size-2 gloo.
The per-rank compute is stood in for by the oracle (no GPU here); what is checked is that the
sharded run gathers to exactly the single-process result, independent of the rank count."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayhunter_amd.distributed import gather_rows, max_over_ranks, shard_range, shard_sizes
from bayhunter_amd.synthetic import draw_models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1000, 131072):
        for world in (1, 2, 3, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            s = shard_sizes(n, world)
            assert sum(s) == n and max(s) - min(s) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import pyoracle as po
    H, VP, VS, RHO, nl = draw_models(n, (2, 8), seed=123)        # same global batch on every rank
    lo, hi = shard_range(n, rank, world)
    per = np.linspace(1, 41, 11)
    out, err, _ = po.swd_batch(H[lo:hi], VP[lo:hi], VS[lo:hi], RHO[lo:hi], nl[lo:hi], per, 2, 0)
    rows = torch.from_numpy(np.concatenate([out, err[:, None].astype(np.float64)], axis=1))
    full = gather_rows(rows, n)
    t = max_over_ranks(float(rank + 1))
    if rank == 0:
        q.put((full.numpy(), t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [37, 64])
def test_gloo_world2_gather_equals_single_process(oracle, n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, t = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H, VP, VS, RHO, nl = draw_models(n, (2, 8), seed=123)
    out, err, _ = oracle.swd_batch(H, VP, VS, RHO, nl, np.linspace(1, 41, 11), 2, 0)
    assert full.shape == (n, 12)
    assert np.array_equal(full[:, :11], out) and np.array_equal(full[:, 11], err)
    assert t == 2.0


def _chain_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import pyoracle as po
    from chain_scenario import CASES, OraclePlugin, joint_target, oracle_evaluator
    from bayhunter_amd.chains import ChainPool
    case = CASES['fixednoise']
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    joint = joint_target(data, lambda xs, xr: (OraclePlugin(po, xs, 'swd'), OraclePlugin(po, xr, 'rf')))
    ip = dict(case['initparams'], iter_burnin=case['burnin'], iter_main=case['main'])
    pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=[5, 6, 7], shard=(rank, world),
                     evaluator=oracle_evaluator(joint)).run()
    full = pool.gather()
    if rank == 0:
        q.put((pool.first, pool.nchains, full))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_chain_pool_shards_and_gathers(golden_chains):
    """Three chains over two ranks (2 + 1): every rank runs its block with the global seeds, the
    gathered sample blocks equal the reference's chains, chain by chain."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    first, mine, full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (first, mine) == (0, 2) and full['likes'].shape[0] == 3
    for i, seed in enumerate((5, 6, 7)):
        n = int(golden_chains['fixednoise/%d/n' % seed])
        assert int(full['naccepted'][i]) == n
        for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(full[k][i, :n], golden_chains['fixednoise/%d/%s' % (seed, k)], equal_nan=True), (seed, k)
