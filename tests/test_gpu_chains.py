"""Chain pool on the GPU (`-m gpu`): bh_chains_* on the host cores, proposals evaluated by
bh_swd_batch / bh_rf_batch / bh_likelihood_batch.  Compared with the committed chains of the
reference's own sampler (tests/golden/chains_golden.npz, made by make_golden_chains.py).

The device forward values differ from the reference's in the last bits (receiver functions <= 1e-15,
LVZ dispersion <= the reference's own root bracket), so a chain is expected to make the SAME
decisions -- identical accepted models and acceptance iterations -- with likelihoods and misfits
equal to float32 storage precision.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
from chain_scenario import CASES, make_pool  # noqa: E402

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLDEN, 'tutorial_observed')


def gpu_evaluator(joint):
    from bayhunter_amd.chains import GpuEvaluator
    return GpuEvaluator(joint)


@pytest.mark.parametrize('name', sorted(CASES))
def test_gpu_pool_reproduces_reference_chains(golden_chains, name):
    case = CASES[name]
    seeds = [int(s) for s in golden_chains['%s/seeds' % name]]
    pool = make_pool(None, DATA, case, seeds=seeds, groups=2, evaluator=gpu_evaluator).run()
    for i, seed in enumerate(seeds):
        got = pool.chain(i)
        ref = {k: golden_chains['%s/%d/%s' % (name, seed, k)] for k in
               ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter', 'n')}
        assert got['n'] == int(ref['n']), (name, seed)
        for k in ('models', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(ref[k], got[k], equal_nan=True), (name, seed, k)
        # float32 rows of float64 values that agree to ~1e-12: at most one float32 ulp apart
        assert np.allclose(ref['likes'], got['likes'], rtol=3e-7, atol=0), (name, seed)
        assert np.allclose(ref['misfits'], got['misfits'], rtol=3e-7, atol=0), (name, seed)
        assert np.mean(ref['likes'] == got['likes']) > 0.99


def test_gpu_pool_many_chains_lockstep_and_files(tmp_path):
    """A pool large enough for two groups, threads and the lane kernel: chains with equal seeds are
    equal whatever their position in the pool, and save() writes one file set per chain."""
    case = dict(CASES['tutorial'], burnin=40, main=24)
    case['initparams'] = dict(case['initparams'], acceptance=(40, 100))     # room for every iteration
    seeds = list(range(100, 100 + 640)) + [100, 101, 739]
    pool = make_pool(None, DATA, case, seeds=seeds, evaluator=gpu_evaluator).run()
    assert len(pool.groups) == 2
    n = pool.counters()[0]
    assert n.min() >= 1 and pool.evaluated > 0.5 * len(seeds) * 64
    for a, b in ((0, 640), (1, 641), (639, 642)):
        ca, cb = pool.chain(a), pool.chain(b)
        assert ca['n'] == cb['n']
        for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(ca[k], cb[k], equal_nan=True), (a, b, k)
    small = make_pool(None, DATA, case, seeds=seeds[:3], evaluator=gpu_evaluator).run()
    assert small.save(str(tmp_path)) >= 15
    w = small.weighted(0)
    assert w[1][0].shape[0] == 40 and w[2][0].shape[0] == 24       # one row per iteration
    assert np.load(str(tmp_path / 'data' / 'c000_p2likes.npy')).shape == (24,)
    for k in ('models', 'likes'):                                   # chain 0 of both pools: same seed
        assert np.array_equal(small.chain(0)[k], pool.chain(0)[k], equal_nan=True)


def test_gpu_pool_lookahead_same_chains_in_fewer_calls():
    """The default pool on the GPU looks ahead (a call of a small pool is bound by its latency, not its size:
    bh_chains_set_lookahead): bit for bit the chains of the one-proposal-per-call pool -- every kernel form gives a
    model the same values wherever it stands in a batch -- in a fraction of the device calls."""
    case = dict(CASES['tutorial'], burnin=1250, main=250)
    case['initparams'] = dict(case['initparams'], acceptance=(40, 100))
    seeds = list(range(40, 52))
    one = make_pool(None, DATA, case, seeds=seeds, evaluator=gpu_evaluator, lookahead=1).run()
    auto = make_pool(None, DATA, case, seeds=seeds, evaluator=gpu_evaluator).run()
    assert one.lookahead == 1 and auto.lookahead == 32          # min(32, 128 / sqrt(12 chains))
    for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
        assert np.array_equal(getattr(one, k), getattr(auto, k), equal_nan=True), k
    for a, b in zip(one.counters(), auto.counters()):
        assert np.array_equal(a, b)
    c1, i1, r1 = one.advance()
    c2, i2, r2 = auto.advance()
    assert c1 == 1500 and i1 == i2 == 12 * 1500 and c2 < c1 / 4 and r2 > r1
    one.close()
    auto.close()
    assert auto.advance() == (c2, i2, r2)                           # (kept by a closed pool)


def test_gpu_pool_result_files_match_the_reference_files(tmp_path):
    """SURVEY 8(f4) on the tier the driver runs: ChainPool.save() after a GPU-evaluated run against
    the ten files the reference's own chain writes (tests/golden/chain_files_golden.npz: weighting by
    residence time, thinning to maxmodels, the reference's dtypes).  Models, noise and vp/vs files
    exactly; likelihoods and misfits to one float32 ulp (device forward values differ in the last bits)."""
    want = np.load(os.path.join(GOLDEN, 'chain_files_golden.npz'))
    case = dict(CASES['tutorial'])
    case['initparams'] = dict(case['initparams'], maxmodels=int(want['maxmodels']))
    pool = make_pool(None, DATA, case, seeds=[int(want['seed'])], evaluator=gpu_evaluator).run()
    assert pool.save(str(tmp_path)) == 10
    names = sorted(k for k in want.files if k.startswith('c000_'))
    assert sorted(f[:-4] for f in os.listdir(str(tmp_path / 'data')) if f.endswith('.npy')) == names
    for k in names:
        a, b = want[k], np.load(str(tmp_path / 'data' / (k + '.npy')))
        assert a.dtype == b.dtype and a.shape == b.shape, k
        if k.endswith('likes') or k.endswith('misfits'):
            assert np.allclose(a, b, rtol=3e-7, atol=0), k
            assert np.mean(a == b) > 0.99, k
        else:
            assert np.array_equal(a, b, equal_nan=True), k
    assert os.path.exists(str(tmp_path / 'data' / 'test_config.pkl'))


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from chain_scenario import CASES, joint_target
    from bayhunter_amd.chains import ChainPool, GpuEvaluator
    case = CASES['constrained']
    joint = joint_target(DATA)
    ip = dict(case['initparams'], iter_burnin=case['burnin'], iter_main=case['main'])
    pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=[21, 22, 23], shard=(rank, world),
                     evaluator=GpuEvaluator(joint)).run()
    full = pool.gather()
    if rank == 0:
        q.put({k: v for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_gpu_pool_sharded_over_two_ranks(golden_chains):
    """One process per rank (both on this box's single GPU; the gather runs over gloo, on a node
    it is RCCL): rank r runs its block of the chains, the gathered blocks are the golden chains."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for i, seed in enumerate((21, 22, 23)):
        n = int(golden_chains['constrained/%d/n' % seed])
        assert int(full['naccepted'][i]) == n
        for k in ('models', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(full[k][i, :n], golden_chains['constrained/%d/%s' % (seed, k)], equal_nan=True), (seed, k)
        assert np.allclose(full['likes'][i, :n], golden_chains['constrained/%d/likes' % seed], rtol=3e-7, atol=0)


def test_hunt_script_on_the_gpu(tmp_path):
    """tools/hunt.py with its default (GPU) evaluator: config.ini + data files -> result files, and
    the same seeds give the same chains as a pool built by hand."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import hunt
    ini = tmp_path / 'config.ini'
    ini.write_text("""[modelpriors]
vpvs = 1.4, 2.1
layers = 1, 20
vs = 2, 5
z = 0, 60
mohoest = None
rfnoise_corr = 0.9
swdnoise_corr = 0.
rfnoise_sigma = 1e-5, 0.05
swdnoise_sigma = 1e-5, 0.05

[initparams]
nchains = 6
iter_burnin = 60
iter_main = 30
propdist = 0.015, 0.015, 0.015, 0.005, 0.005
acceptance = 40, 45
thickmin = 0.1
lvz = None
hvz = None
rcond = 1e-5
station = 'st3'
savepath = '%s'
maxmodels = 50000
""" % str(tmp_path / 'out'))
    pool = hunt.main([str(ini), '--target', 'rdispph=' + os.path.join(DATA, 'st3_rdispph.dat'),
                      '--target', 'prf=' + os.path.join(DATA, 'st3_prf.dat'), '--seed', '11', '--full-storage'])
    files = os.listdir(str(tmp_path / 'out' / 'data'))
    assert 'st3_config.pkl' in files and sum(f.endswith('.npy') for f in files) == 60
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool
    priors, ip = hunt.load_params(str(ini))
    ref = ChainPool(joint_target(DATA), initparams=ip, modelpriors=priors, seeds=[int(s) for s in pool.seeds],
                    nmodels=91).run()
    assert list(pool.seeds) == [np.random.RandomState(11).randint(1000)] + list(pool.seeds[1:])
    for i in range(6):
        assert np.array_equal(pool.chain(i)['models'], ref.chain(i)['models'], equal_nan=True)


def test_rccl_branches_execute_on_one_gpu(tmp_path):
    """The "nccl" (= RCCL) code paths of the package on this box's single GPU: a one-rank process
    group in a child process runs the device-side collectives of bench.py's timing protocol (barrier
    with device_ids, MAX all-reduce on the device) and of the gathers (all_gather, gather-to-root,
    ragged gather, ChainPool.gather / gather_final with device tensors)."""
    import subprocess
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from bayhunter_amd.distributed import gather_ragged_to_root, gather_rows, gather_rows_to_root, max_over_ranks
dist.barrier(device_ids=[0])
assert max_over_ranks(2.5, device='cuda') == 2.5
x = torch.arange(12, dtype=torch.float32, device='cuda').reshape(4, 3)
assert torch.equal(gather_rows(x, 4), x) and torch.equal(gather_rows_to_root(x, 4), x)
assert torch.equal(gather_ragged_to_root(x)[0], x)
from chain_scenario import CASES, joint_target
from bayhunter_amd.chains import ChainPool, GpuEvaluator
case = CASES['fixednoise']
joint = joint_target(%r)
ip = dict(case['initparams'], iter_burnin=40, iter_main=30, acceptance=(40, 100), maxmodels=11)
pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=[5, 6, 7], shard=(0, 1), evaluator=GpuEvaluator(joint)).run()
full = pool.gather()
assert np.array_equal(full['likes'], pool.likes, equal_nan=True) and full['models'].shape[0] == 3
fin = pool.gather_final()
assert len(fin) == 3 and all(np.array_equal(a, pool.final(i), equal_nan=True) for i, a in enumerate(fin))
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print('rccl ok')
""" % (ROOT, os.path.join(ROOT, 'tests', 'scenarios'), DATA)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, '-c', code, str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- the evaluation plan of the library (bh_eval_*): what GpuEvaluator submits a batch through --------
def _random_batch(B, Lmax, T, seed, ragged=(2, 12)):
    from bayhunter_amd.synthetic import draw_models
    H, VP, VS, RHO, nl = draw_models(B, ragged, seed=seed, sorted_vs=False, Lmax=Lmax)
    rs = np.random.RandomState(seed + 1)
    noise = np.column_stack([f(B) for _ in range(T) for f in (lambda n: rs.uniform(0.0, 0.9, n), lambda n: rs.uniform(0.005, 0.05, n))])
    return np.stack([H, VP, VS, RHO], axis=1), nl, np.ascontiguousarray(noise)


@pytest.mark.parametrize('B', [5, 700, 20000])
def test_eval_plan_equals_the_engine_path(B):
    """One bh_eval_submit = ForwardEngine.run + likelihood_batch through torch, bit for bit: small batches
    (team kernels), one with more rows than the plan was sized for half of, and one large enough for the
    processing order (device radix sort instead of torch.argsort: results land in the caller's rows either
    way).  Receiver-function target with the dense Gaussian covariance, i.e. the matrix-core path."""
    import torch
    from chain_scenario import joint_target
    joint = joint_target(DATA)
    joint.set_target_covariance([True, True], [0.0, 0.9], 1e-5)
    Lmax, T = 12, joint.ntargets
    packed, nl, noise = _random_batch(B, Lmax, T, seed=B)
    plan = joint.eval_plan(B + 7, Lmax)
    for rep in range(2):                                   # a plan is reused iteration after iteration
        plan.packed[:B], plan.nlay[:B], plan.noise[:B] = packed, nl, noise
        plan.submit(B)
        logL, mis = (a.copy() for a in plan.wait())
        wl, wm = joint.evaluate_batch(packed[:, 0], packed[:, 1], packed[:, 2], nl, noise, RHO=packed[:, 3])
        torch.cuda.synchronize()
        assert np.array_equal(logL, wl.cpu().numpy(), equal_nan=True), (B, rep)
        assert np.array_equal(mis, wm.cpu().numpy(), equal_nan=True), (B, rep)
        assert logL.shape == (B,) and mis.shape == (B, T + 1) and np.isfinite(logL).mean() > 0.9
        packed, nl, noise = _random_batch(B, Lmax, T, seed=B + 100)
    plan.submit(0)                                         # an iteration without a valid proposal
    assert plan.wait()[0].shape == (0,)
    plan.close()


def test_eval_plan_more_than_60_periods_and_swd_only():
    """A dispersion target with 75 observed periods (solved on 60, interpolated on the device by the plan's
    own kernel) next to an ordinary one, no receiver function: equal to the engine's torch interpolation."""
    import torch
    from bayhunter_amd import targets as T
    x75, x21 = np.linspace(2, 80, 75), np.linspace(1, 41, 21)
    joint = T.JointTarget([T.RayleighDispersionPhase(x75, np.full(75, 3.5)), T.LoveDispersionGroup(x21, np.full(21, 3.3))])
    joint.set_target_covariance([True, False], [0.0, 0.3], None)
    packed, nl, noise = _random_batch(300, 10, 2, seed=9, ragged=(2, 10))
    plan = joint.eval_plan(300, 10)
    plan.packed[:], plan.nlay[:], plan.noise[:] = packed, nl, noise
    plan.submit(300)
    logL, mis = plan.wait()
    wl, wm = joint.evaluate_batch(packed[:, 0], packed[:, 1], packed[:, 2], nl, noise, RHO=packed[:, 3])
    torch.cuda.synchronize()
    assert np.array_equal(logL, wl.cpu().numpy(), equal_nan=True) and np.array_equal(mis, wm.cpu().numpy(), equal_nan=True)
    assert np.isfinite(logL).mean() > 0.5


def test_pools_and_plans_made_run_and_closed_back_to_back():
    """the driver's bench died in its chain-pool leg with `hipEventQuery(queue slot):
    operation not permitted when stream is capturing` -- work-queue slot events of the library outlived the
    evaluation plans' streams they were recorded on.  Five pools (two plans each) made, run and closed one after
    the other in ONE process, forward launches on other streams in between, a caller-owned stream retired and
    destroyed, with a ring of only 8 slots so that every launch re-claims a slot an earlier pool used
    (tests/scenarios/pool_lifecycle.py; a child process: the old failure could as well be a crash)."""
    import json
    import subprocess
    env = dict(os.environ, BH_SWD_QUEUE_SLOTS='8')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'scenarios', 'pool_lifecycle.py'), '5', '600', '40'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['ok'] and rec['pools'] == 5 and rec['plans_closed'] == 10 and rec['forward_checks'] == 13 and rec['slots'] == '8'


def test_eval_plan_context_manager_and_use_after_close():
    from bayhunter_amd._lib import BayHunterAmdError
    from chain_scenario import joint_target
    joint = joint_target(DATA)
    joint.set_target_covariance([True, True], [0.0, 0.9], 1e-5)
    packed, nl, noise = _random_batch(40, 12, joint.ntargets, seed=4)
    with joint.eval_plan(64, 12) as plan:
        plan.packed[:40], plan.nlay[:40], plan.noise[:40] = packed, nl, noise
        plan.submit(40)
        logL = plan.wait()[0].copy()
    assert plan.closed and plan.packed is None and np.isfinite(logL).mean() > 0.9
    plan.close()                                           # idempotent
    with pytest.raises(BayHunterAmdError, match='closed'):
        plan.submit(1)
    with pytest.raises(BayHunterAmdError, match='closed'):
        plan.wait()


def test_rf_only_plan_flags_bad_depths_like_the_engine():
    """without a dispersion kernel the plan only zeroed the flags, so a model deeper than the plan
    reached the likelihood as a NaN row with flag 0 where ForwardEngine raises BH_MODEL_BAD_DEPTH (2)."""
    import torch
    from bayhunter_amd import targets as T
    x = np.linspace(-5, 35, 201)
    joint = T.JointTarget([T.PReceiverFunction(x, np.zeros(201))])
    joint.set_target_covariance([True], [0.0], None)
    packed, nl, noise = _random_batch(64, 10, 1, seed=11, ragged=(2, 10))
    nl = nl.copy()
    nl[3], nl[17], nl[40] = 0, 11, -2                      # outside 1..Lmax
    with joint.eval_plan(64, 10) as plan:
        plan.packed[:], plan.nlay[:], plan.noise[:] = packed, nl, noise
        plan.submit(64)
        logL, mis = (a.copy() for a in plan.wait())
    wl, wm = joint.evaluate_batch(packed[:, 0], packed[:, 1], packed[:, 2], nl, noise, RHO=packed[:, 3])
    torch.cuda.synchronize()
    assert np.array_equal(logL, wl.cpu().numpy(), equal_nan=True) and np.array_equal(mis, wm.cpu().numpy(), equal_nan=True)
    bad = np.zeros(64, dtype=bool)
    bad[[3, 17, 40]] = True
    assert np.all(logL[bad] == -1e15) and np.all(np.isfinite(logL[~bad])) and np.all(logL[~bad] > -1e15)


def test_plans_made_used_and_closed_by_several_threads_at_once():
    """Four host threads, each making, using and closing twelve evaluation plans while the others launch, on a ring of
    8 work-queue slots (tests/scenarios/thread_stress.py; a child process): slots are claimed across threads, and a
    plan's streams are retired while other threads hold slots whose guard events were recorded on them -- the case
    retire_stream waits for (capi.hip).  Every one of the 288 batches returns the first batch's bits."""
    import json
    import subprocess
    env = dict(os.environ, BH_SWD_QUEUE_SLOTS='8')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'scenarios', 'thread_stress.py'), '4', '12', '6'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['ok'] and rec['batches'] == 4 * 12 * 6 and rec['slots'] == '8'
