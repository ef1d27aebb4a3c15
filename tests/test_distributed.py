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


# ---- gather to the root: ragged shards, nothing replicated, thinning before the exchange --------
def _root_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bayhunter_amd.distributed import gather_ragged_to_root, gather_rows_to_root
    n = 5                                                   # 5 rows over 4 ranks: shards 2, 1, 1, 1
    lo, hi = shard_range(n, rank, world)
    rows = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)[lo:hi]
    full = gather_rows_to_root(rows, n, dst=0)
    rag = gather_ragged_to_root(torch.full((rank * 2, 2), float(rank)), dst=0)     # 0, 2, 4, 6 rows
    from chain_scenario import CASES, joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['fixednoise']
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    joint = joint_target(data, lambda xs, xr: (None, None))
    rng = np.random.RandomState(1)

    def ev(packed, nlay, noise):                            # made-up likelihood: only the plumbing is tested
        return -60.0 * (packed[:, 2, 0] - 3.1) ** 2 - 1e-3 * nlay, np.zeros((packed.shape[0], 3))
    ip = dict(case['initparams'], iter_burnin=60, iter_main=40, acceptance=(40, 100), maxmodels=17)
    seeds = [5, 6, 7, 8, 9]
    pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=seeds, shard=(rank, world),
                     evaluator=ev).run()
    blocks = pool.gather()
    everywhere = pool.gather(everywhere=True)
    final = pool.gather_final()
    q.put((rank, None if full is None else full.numpy(), None if rag is None else [r.numpy() for r in rag],
           pool.first, pool.nchains, blocks, everywhere['likes'], final))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world4_gather_to_root_ragged_and_thinned():
    """5 chains on 4 ranks (2+1+1+1): only the root holds the gathered blocks; they equal a single
    process's; gather_final delivers what save() would write, thinned to maxmodels rows per chain."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_root_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(4):
        r = q.get(timeout=240)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, rag = got[0][0], got[0][1]
    assert np.array_equal(full, np.arange(15, dtype=np.float32).reshape(5, 3))
    assert [a.shape for a in rag] == [(0, 2), (2, 2), (4, 2), (6, 2)] and all((a == i).all() for i, a in enumerate(rag))
    assert [got[r][2:4] for r in range(4)] == [(0, 2), (2, 1), (3, 1), (4, 1)]
    for r in (1, 2, 3):                                     # nothing is replicated onto the other ranks
        assert got[r][0] is None and got[r][1] is None and got[r][4] is None and got[r][6] is None
    # the same five chains in one process
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
    from chain_scenario import CASES, joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['fixednoise']
    joint = joint_target(os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed'), lambda xs, xr: (None, None))
    ev = lambda packed, nlay, noise: (-60.0 * (packed[:, 2, 0] - 3.1) ** 2 - 1e-3 * nlay, np.zeros((packed.shape[0], 3)))
    ip = dict(case['initparams'], iter_burnin=60, iter_main=40, acceptance=(40, 100), maxmodels=17)
    one = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=[5, 6, 7, 8, 9], evaluator=ev).run()
    blocks = got[0][4]
    for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
        assert np.array_equal(blocks[k], getattr(one, k), equal_nan=True), k
    assert np.array_equal(blocks['naccepted'], one.counters()[0])
    for r in range(4):
        assert np.array_equal(got[r][5], one.likes, equal_nan=True)          # everywhere=True: all ranks
    final = got[0][6]
    assert len(final) == 5
    for i in range(5):
        want = one.final(i)
        assert 0 < want.shape[0] <= 17 and np.array_equal(final[i], want, equal_nan=True)
        w = one.weighted(i)[2]
        thin = int(np.ceil(w[1].size / 17.0))
        assert np.array_equal(want[:, :one.models.shape[2]], w[0][::thin].astype(np.float32), equal_nan=True)


# ---- a proper sub-group: its ranks are not 0..n-1 of the world -----------------------------------
def _subgroup_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bayhunter_amd.distributed import gather_ragged_to_root, gather_rows_to_root
    members = [1, 3]                                       # world ranks 1 and 3 are ranks 0 and 1 of the group
    grp = dist.new_group(members)                          # (every process must take part in new_group)
    res = None
    if rank in members:
        gr = members.index(rank)
        n = 5
        lo, hi = shard_range(n, gr, 2)
        rows = torch.arange(n * 2, dtype=torch.float32).reshape(n, 2)[lo:hi]
        full = gather_rows_to_root(rows, n, dst=1, group=grp)            # root = group rank 1 = world rank 3
        rag = gather_ragged_to_root(torch.full((gr + 2, 3), float(rank)), dst=0, group=grp)
        t = max_over_ranks(float(rank), group=grp)
        res = (None if full is None else full.numpy(), None if rag is None else [r.numpy() for r in rag], t)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_subgroup_gathers_address_the_right_peers():
    """Gathers inside a sub-group {1, 3} of a 4-rank world: group ranks must be translated to world
    ranks for send / irecv / gather, otherwise the blocks go to (or hang on) the wrong peers."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(4))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] is None and got[2] is None
    full1, rag1, t1 = got[1]
    full3, rag3, t3 = got[3]
    assert full1 is None and np.array_equal(full3, np.arange(10, dtype=np.float32).reshape(5, 2))
    assert rag3 is None and [a.shape for a in rag1] == [(2, 3), (3, 3)]
    assert (rag1[0] == 1).all() and (rag1[1] == 3).all()
    assert t1 == 3.0 and t3 == 3.0


# ---- bench.py's N > 1 legs (sharded sampler + timed gather) over gloo, with a made-up likelihood ----------
def _made_up_likelihood(packed, nlay, noise):
    return -60.0 * (packed[:, 2, 0] - 3.1) ** 2 - 1e-3 * nlay, np.zeros((packed.shape[0], 3))


_TINY = {'cfg4': dict(layers=(1, 14), chains_per_gpu=5, burnin=30, main=20),
         'cfg5': dict(layers=(1, 30), chains_per_gpu=9, burnin=30, main=20)}


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    bench.BLAS_SETTLE_S = 0.0
    ranks = bench.Ranks(world, 'gloo', 0)
    pools, gather = bench.sharded_chain_pools(rank, world, ranks, 'gloo', workloads=_TINY, evaluator=_made_up_likelihood)
    if rank == 0:
        q.put((pools, gather))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_bench_sharded_sampler_and_gather():
    """What `bench.py --gpus N` adds to its line for N > 1 (bench.py: sharded_chain_pools) -- BASELINE cfg4 / cfg5 as
    chain pools block-sharded over the ranks, then the pool's sample blocks gathered over the process group, to the
    root and as an all-gather, raw and thinned -- run on the CPU tier by two gloo ranks with a made-up likelihood in
    place of the device: the record has every field, the byte counts are the payload's, and the gathered chains are
    the chains ONE process produces for the same seeds (digest)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pools, gather = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import bench
    from bayhunter_amd.chains import ChainPool
    for name, wl in _TINY.items():
        total = 2 * wl['chains_per_gpu']
        sh, g = pools[name], gather[name]
        assert sh['nchains'] == total and sh['value'] > 0 and sh['iterations'] == 50 and sh['models_evaluated'] > 0
        nmodels = 50                                           # int(iterations * max(acceptance) / 100), acceptance (40, 100)
        width = 2 * (wl['layers'][1] + 1) + 3 + 1 + 4 + 1
        assert g['bytes_per_rank'] == wl['chains_per_gpu'] * nmodels * width * 4 == g['bytes_received_by_root']
        assert g['backend'] == 'gloo' and g['to_root']['ms'] > 0 and g['all_gather']['GB/s'] > 0
        assert g['ChainPool.gather_final']['rows_at_root'] > 0
        joint, priors = bench.chain_setup(wl['layers'])
        ip = dict(bench.CHAIN_IP, iter_burnin=wl['burnin'], iter_main=wl['main'])
        with ChainPool(joint, initparams=ip, modelpriors=priors, seeds=np.arange(total) % 1000,
                       evaluator=_made_up_likelihood) as one:
            one.run()
        blocks = dict(models=one.models, likes=one.likes, iter=one.iter, noise=one.noise, vpvs=one.vpvs,
                      naccepted=one.counters()[0])
        assert bench.chain_digest(blocks) == g['chains_sha256'] and int(blocks['naccepted'].sum()) == g['accepted_total']
