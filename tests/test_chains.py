"""Lock-step chain pool (bayhunter_amd/chains.py + csrc/chains.cpp), CPU tier.

* the per-chain random streams are numpy.random.RandomState's, draw for draw;
* with the same seed and the same forward values a pool chain IS the reference chain: where the
  reference tree is available its own, unmodified SingleChain.run_chain() runs next to the pool
  (both on the CPU oracle) and the stored samples, acceptance counters and adapted proposal widths
  must be identical; the committed fixture tests/golden/chains_golden.npz (made by
  tests/golden/make_golden_chains.py from the reference) repeats this without the reference tree;
* the result files equal the ones the reference chain writes.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import reference_chain as rc  # noqa: E402
from chain_scenario import CASES, OraclePlugin, make_pool, oracle_evaluator  # noqa: E402

DATA = os.path.join(GOLDEN, 'tutorial_observed')


def _tiny_pool(lib, nchains=3, seeds=(0, 1, 12345)):
    from bayhunter_amd import _lib
    cfg = _lib.ChainConfig()
    cfg.ntargets, cfg.layers_min, cfg.layers_max = 1, 1, 5
    cfg.vs_min, cfg.vs_max, cfg.z_min, cfg.z_max = 2, 5, 0, 60
    cfg.vpvs_fixed, cfg.vpvs_min, cfg.vpvs_max = 1, 1.73, 1.73
    cfg.iter_burnin, cfg.iter_main = 10, 10
    nm = 9
    arrs = [np.full((nchains, nm, w), np.nan, dtype=np.float32) for w in (12, 2, 1, 2, 1)]
    it = np.full((nchains, nm), np.nan)
    st = _lib.ChainStorage()
    st.nmodels = nm
    for name, a in zip(('models', 'misfits', 'likes', 'noise', 'vpvs'), arrs):
        setattr(st, name, a.ctypes.data)
    st.iter = it.ctypes.data
    seeds = np.asarray(seeds, dtype=np.uint32)
    h = C.c_void_p()
    _lib.check(lib.bh_chains_create(C.byref(cfg), nchains, seeds.ctypes.data, C.byref(st), C.byref(h)))
    return h, (arrs, it, seeds)


def test_random_streams_are_numpys(lib):
    from bayhunter_amd import _lib
    seeds = (0, 1, 12345, 999, 2**32 - 1)
    h, keep = _tiny_pool(lib, len(seeds), seeds)
    try:
        for ci, seed in enumerate(seeds):
            rs = np.random.RandomState(seed)
            # interleave the three generators like a chain does; odd counts exercise the cached gaussian
            for kind, a, b, n in ((0, 2.0, 5.0, 7), (1, 0.0, 0.015, 5), (2, 0, 6, 9), (1, 31.0, 2.5, 4),
                                  (0, 0.0, 1.0, 3), (2, 4, 8, 5), (2, 0, 1, 2), (2, 0, 2**33 + 5, 6),
                                  (1, 0.0, 1.0, 1001), (0, 1e-5, 0.05, 700)):
                out = np.zeros(n)
                _lib.check(lib.bh_chains_draw(h, ci, kind, float(a), float(b), n, out.ctypes.data))
                if kind == 0:
                    ref = np.array([rs.uniform(a, b) for _ in range(n)])
                elif kind == 1:
                    ref = np.array([rs.normal(a, b) for _ in range(n)])
                else:
                    ref = np.array([rs.randint(a, b) for _ in range(n)], dtype=np.float64)
                assert np.array_equal(out, ref), (seed, kind, a, b)
            # the state itself, and adopting a numpy state
            key = np.zeros(624, dtype=np.uint32)
            pos, hg, g = C.c_int(), C.c_int(), C.c_double()
            _lib.check(lib.bh_chains_get_rng(h, ci, key.ctypes.data, C.byref(pos), C.byref(hg), C.byref(g)))
            _, nkey, npos, nhg, ng = rs.get_state()
            assert np.array_equal(key, nkey) and pos.value == npos and hg.value == nhg and g.value == ng
        rs = np.random.RandomState(4242)
        rs.normal(size=3)
        _, nkey, npos, nhg, ng = rs.get_state()
        nkey = np.ascontiguousarray(nkey, dtype=np.uint32)
        _lib.check(lib.bh_chains_set_rng(h, 0, nkey.ctypes.data, int(npos), int(nhg), float(ng)))
        out = np.zeros(5)
        _lib.check(lib.bh_chains_draw(h, 0, 1, 0.0, 1.0, 5, out.ctypes.data))
        assert np.array_equal(out, rs.normal(size=5))
    finally:
        lib.bh_chains_destroy(h)


def test_protocol_errors(lib):
    from bayhunter_amd import _lib
    h, keep = _tiny_pool(lib)
    try:
        packed, nlay = np.zeros((3, 4, 8)), np.zeros(3, dtype=np.int32)
        noise, chain, cnt = np.zeros((3, 2)), np.zeros(3, dtype=np.int32), C.c_int()
        args = (packed.ctypes.data, nlay.ctypes.data, noise.ctypes.data, chain.ctypes.data, C.byref(cnt))
        assert lib.bh_chains_accept(h, None, None) == _lib.BH_ERR_ARG             # nothing outstanding
        assert lib.bh_chains_propose(h, 4, *args) == _lib.BH_ERR_ARG              # Lmax < layers_max + 1
        assert b'Lmax' in lib.bh_last_error()
        assert lib.bh_chains_propose(h, 8, *args) == _lib.BH_OK and cnt.value == 3
        assert lib.bh_chains_propose(h, 8, *args) == _lib.BH_ERR_ARG              # results still due
        assert list(nlay) == [2, 2, 2] and np.all(packed[:, 0, 1] == 0) and np.all(packed[:, 0, 0] > 0)
        assert np.array_equal(packed[:, 3, :2], packed[:, 1, :2] * 0.32 + 0.77)
        assert lib.bh_chains_done(h) == 0 and lib.bh_chains_iteration(h) == -10
    finally:
        lib.bh_chains_destroy(h)


needs_ref = pytest.mark.skipif(not rc.available(), reason='reference tree not present')


def _same(ref, got, name):
    for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
        assert np.array_equal(ref[k], got[k], equal_nan=True), (name, k)


@needs_ref
@pytest.mark.parametrize('name', sorted(CASES))
def test_pool_chain_is_the_reference_chain(oracle, name):
    case = CASES[name]
    ref = rc.run_chain(None, refs=case.get('refs', ('rdispph', 'prf')), plugin_for=lambda ref, x: OraclePlugin(oracle, x, ref),
                       seed=case['seed'], burnin=case['burnin'], main=case['main'], data_dir=DATA,
                       priors=case['priors'], initparams=case['initparams'])
    pool = make_pool(oracle, DATA, case, seeds=[case['seed']]).run()
    got = pool.chain(0)
    assert ref['n'] == got['n'] and got['n'] > 20
    _same(ref, got, name)
    n, propdist, accepted, proposed = pool.counters()
    assert np.array_equal(propdist[0], ref['propdist'])
    assert np.array_equal(accepted[0], ref['accepted']) and np.array_equal(proposed[0], ref['proposed'])


def test_pool_chains_match_golden(oracle, golden_chains):
    """Same comparison from the committed fixture (the reference tree is not needed): several
    chains in ONE pool, two groups, so that lock step and grouping are covered too."""
    for name in sorted(CASES):
        case = CASES[name]
        seeds = [int(s) for s in golden_chains['%s/seeds' % name]]
        pool = make_pool(oracle, DATA, case, seeds=seeds, groups=2).run()
        counts = pool.counters()[0]
        for i, seed in enumerate(seeds):
            got = pool.chain(i)
            assert got['n'] == int(golden_chains['%s/%d/n' % (name, seed)]) == counts[i]
            _same({k: golden_chains['%s/%d/%s' % (name, seed, k)] for k in
                   ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter')}, got, (name, seed))


def test_pool_lifetime_close_and_context_manager(oracle):
    """A pool's native resources end with close() / the with-block, not whenever the interpreter collects it;
    its results stay readable, a closed pool refuses to run, several pools in a row are the same inversion
    (the reference: MCMC_Optimizer.mp_inversion can be called again, src/mcmcOptimizer.py:202-283)."""
    from bayhunter_amd._lib import BayHunterAmdError
    case = dict(CASES['fixednoise'], burnin=30, main=20)
    first = None
    for k in range(3):
        with make_pool(oracle, DATA, case, seeds=[5, 6], groups=2) as pool:
            assert not pool.closed
            pool.run()
        assert pool.closed and all(g.handle is None for g in pool.groups)
        pool.close()                                                     # idempotent
        n = pool.counters()[0]
        got = [pool.chain(i) for i in range(2)]
        assert got[0]['n'] == n[0] >= 1 and pool.weighted(0)[1] is not None
        if first is None:
            first = got
        for a, b in zip(first, got):
            _same(a, b, k)
    unrun = make_pool(oracle, DATA, case, seeds=[5])
    unrun.close()
    with pytest.raises(BayHunterAmdError, match='closed'):
        unrun.run()


def test_storage_overflow_is_reported(oracle):
    case = dict(CASES['tutorial'], burnin=30, main=10)
    case['initparams'] = dict(case['initparams'], acceptance=(1, 2))    # room for int(40*0.02) = 0 -> refuse
    with pytest.raises(ValueError):
        make_pool(oracle, DATA, case, seeds=[3])
    case['initparams'] = dict(case['initparams'], acceptance=(3, 5))    # two rows: fills up quickly
    from bayhunter_amd._lib import BayHunterAmdError
    with pytest.raises(BayHunterAmdError, match='storage'):
        make_pool(oracle, DATA, case, seeds=[3]).run()


def test_result_files_equal_the_committed_reference_files(oracle, tmp_path):
    """SURVEY 8(f4) without the reference tree: the ten files the reference's own chain writes for the
    tutorial set-up (tests/golden/chain_files_golden.npz, made by make_golden_chains.py from the
    unmodified SingleChain.save_finalmodels) against ChainPool.save(): dtype, shape and every value."""
    want = np.load(os.path.join(os.path.dirname(DATA), 'chain_files_golden.npz'))
    case = dict(CASES['tutorial'])
    case['initparams'] = dict(case['initparams'], maxmodels=int(want['maxmodels']))
    pool = make_pool(oracle, DATA, case, seeds=[int(want['seed'])]).run()
    assert pool.save(str(tmp_path)) == 10
    names = sorted(k for k in want.files if k.startswith('c000_'))
    assert len(names) == 10
    assert sorted(f[:-4] for f in os.listdir(str(tmp_path / 'data')) if f.endswith('.npy')) == names
    for k in names:
        a, b = want[k], np.load(str(tmp_path / 'data' / (k + '.npy')))
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True), k


@needs_ref
def test_result_files_equal_the_reference_chains(oracle, tmp_path):
    case = CASES['tutorial']
    refdir, mydir = str(tmp_path / 'ref'), str(tmp_path / 'mine')
    rc.run_chain(None, refs=case.get('refs', ('rdispph', 'prf')), plugin_for=lambda ref, x: OraclePlugin(oracle, x, ref),
                 seed=case['seed'], burnin=case['burnin'], main=case['main'], data_dir=DATA,
                 priors=case['priors'], initparams=dict(case['initparams'], maxmodels=150), savepath=refdir)
    case = dict(case, initparams=dict(case['initparams'], maxmodels=150))
    pool = make_pool(oracle, DATA, case, seeds=[case['seed']]).run()
    assert pool.save(mydir) == 10
    files = sorted(f for f in os.listdir(os.path.join(refdir, 'data')) if f.endswith('.npy'))
    assert len(files) == 10
    for f in files:
        a, b = np.load(os.path.join(refdir, 'data', f)), np.load(os.path.join(mydir, 'data', f))
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True), f
    import pickle
    with open(os.path.join(mydir, 'data', 'test_config.pkl'), 'rb') as fh:
        cfg = pickle.load(fh)
    assert cfg['targetrefs'] == ['rdispph', 'prf'] and cfg['initparams']['maxmodels'] == 150
    assert np.array_equal(cfg['targets'][0].obsdata.x, pool.targets.targets[0].obsdata.x)


def test_helper_threads_survive_a_fork(lib):
    """The broker forks chain processes; a child of a process that already used the helper threads
    must get its own (threads do not survive fork) instead of waiting for the parent's."""
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool

    def run():
        case = CASES['tutorial']
        rng = np.random.RandomState(0)

        def ev(packed, nlay, noise):
            return -50 + rng.rand(packed.shape[0]), np.zeros((packed.shape[0], 3))
        ip = dict(case['initparams'], iter_burnin=12, iter_main=6, acceptance=(40, 100))
        pool = ChainPool(joint_target(DATA), initparams=ip, modelpriors=case['priors'],
                         seeds=np.arange(1024) % 1000, evaluator=ev, nthreads=4).run()
        return int(pool.counters()[0].sum())
    want = run()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.write(w, str(run()).encode())
        finally:
            os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 60)
    got = os.read(r, 64).decode() if ready else 'timeout'
    os.waitpid(pid, 0)
    assert got == str(want) and run() == want


@needs_ref
def test_reference_plotfromstorage_reads_pool_files(oracle, tmp_path):
    """the reference's own PlotFromStorage (src/Plotting.py, unmodified) opens a
    result directory written by ChainPool.save(): finds all chain files, detects outlier chains and
    merges the main-phase samples into the final distribution files."""
    case = dict(CASES['fixednoise'])
    pool = make_pool(oracle, DATA, case, seeds=[5, 6, 7, 8]).run()
    pool.save(str(tmp_path))
    data = str(tmp_path / 'data')
    PlotFromStorage = rc.load_plot_from_storage()
    obj = PlotFromStorage(os.path.join(data, 'test_config.pkl'))
    assert obj.ntargets == 2 and obj.refs == ['rdispph', 'prf', 'joint']
    assert len(obj.likefiles[1]) == 4 and len(obj.modfiles[0]) == 4
    chains, nmodels = obj._get_chaininfo()
    # one row per main-phase iteration from the first acceptance of that phase on
    assert chains == [0, 1, 2, 3] and nmodels == [pool.weighted(i)[2][1].size for i in range(4)]
    assert all(case['main'] // 2 < n <= case['main'] for n in nmodels)
    obj.save_final_distribution(maxmodels=200, dev=0.5)
    likes = np.load(os.path.join(data, 'c_likes.npy'))
    models = np.load(os.path.join(data, 'c_models.npy'))
    keep = 4 - len(obj.outliers)
    mpc = 200 // keep
    assert keep >= 1 and likes.size == sum(min(n, mpc) for i, n in enumerate(nmodels) if i not in obj.outliers)
    assert models.shape[0] == likes.size
    # every merged sample is a stored sample of one of the chains
    stored = np.concatenate([pool.weighted(i)[2][1] for i in range(4)])
    assert np.all(np.isin(likes, stored))


def test_threads_and_pool_interleaving_do_not_change_chains(lib):
    """Chains depend on their seed only: the same chains come out single-threaded, on the helper
    threads, and when two pools of different size (different thread counts per job) are stepped
    alternately in one process."""
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['constrained']

    def toy(packed, nlay, noise):                  # deterministic in the proposal, no forward model
        vs, h = packed[:, 2, :], packed[:, 0, :]
        d = vs[:, 0] - 3.1 + 0.01 * nlay + 0.002 * h.sum(axis=1)
        return -40. * d * d - 3. * noise[:, 3], np.stack([np.abs(d), np.abs(d), 2 * np.abs(d)], axis=1)

    def pool(n, nthreads, groups):
        ip = dict(case['initparams'], iter_burnin=30, iter_main=15, acceptance=(40, 100))
        return ChainPool(joint_target(DATA), initparams=ip, modelpriors=case['priors'], seeds=(np.arange(n) * 7) % 1000,
                         evaluator=toy, nthreads=nthreads, groups=groups)
    ref = pool(1500, 1, 1).run()
    thr = pool(1500, 8, 2).run()
    a, b = pool(1500, 8, 1), pool(300, 3, 1)       # stepped alternately: 8-part and 2-part jobs interleave
    for p in (a, b):
        p._launch(p.groups[0])
    while not (a.groups[0].done() and b.groups[0].done()):
        for p in (a, b):
            g = p.groups[0]
            if not g.done():
                p._land(g)
                if not g.done():
                    p._launch(g)
    for other, n in ((thr, 1500), (a, 1500), (b, 300)):
        assert np.array_equal(other.counters()[0], ref.counters()[0][:n])
        for k in ('models', 'likes', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(getattr(other, k), getattr(ref, k)[:n], equal_nan=True), k


def test_lookahead_does_not_change_chains(lib):
    """bh_chains_set_lookahead: with 2 .. 64 proposals per chain and call -- the following iterations for the likeliest
    outcomes -- every chain stores exactly the samples of the one-proposal-per-call chain, ends with the same counters,
    proposal widths and random stream, and the pool needs fewer calls.  Long enough to cross the iterations at which the
    proposal widths adapt (every 1000th; the look-ahead stops there) and the end of the early phase (1 %)."""
    from bayhunter_amd import _lib
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['tutorial']

    def toy(packed, nlay, noise):
        vs, h = packed[:, 2, :], packed[:, 0, :]
        d = vs[:, 0] - 3.1 + 0.01 * nlay + 0.002 * h.sum(axis=1)
        return -40. * d * d - 3. * noise[:, 3], np.stack([np.abs(d), np.abs(d), 2 * np.abs(d)], axis=1)

    def run(lookahead, groups=1, n=24):
        ip = dict(case['initparams'], iter_burnin=1700, iter_main=700, acceptance=(40, 45))
        pool = ChainPool(joint_target(DATA), initparams=ip, modelpriors=case['priors'], seeds=(np.arange(n) * 13) % 1000,
                         evaluator=toy, groups=groups, lookahead=lookahead, nmodels=2401)
        pool.run()
        rng = []
        for g in pool.groups:
            for ci in range(g.n):
                key, pos, hg, gs = np.zeros(624, dtype=np.uint32), C.c_int(), C.c_int(), C.c_double()
                _lib.check(lib.bh_chains_get_rng(g.handle, ci, key.ctypes.data, C.byref(pos), C.byref(hg), C.byref(gs)))
                rng.append((key.tobytes(), pos.value, hg.value, gs.value))
        return pool, rng

    ref, ref_rng = run(1)
    calls, iters, rows = ref.advance()
    assert calls == 2400 and iters == 24 * 2400 and rows == ref.evaluated - 24 and ref.lookahead == 1
    assert len(set(ref.counters()[0])) > 5                      # (the chains differ)
    for la, groups in ((2, 1), (5, 2), (16, 1), (64, 3)):
        pool, rng = run(la, groups)
        assert pool.lookahead == la and pool.iteration == 700
        for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
            assert np.array_equal(getattr(pool, k), getattr(ref, k), equal_nan=True), (la, k)
        for a, b in zip(pool.counters(), ref.counters()):
            assert np.array_equal(a, b), la
        assert rng == ref_rng, la
        c, i, r = pool.advance()
        assert i == iters and c < calls * groups / min(la, 1.5) and r <= c * la * 24 / groups + 24
    # the staging arrays hold nchains * lookahead rows; the setting is refused while results are due and out of range
    g = pool.groups[0]
    assert lib.bh_chains_rows(g.handle) == g.n * 64 == g.packed.shape[0] and lib.bh_chains_lookahead(g.handle) == 64
    assert lib.bh_chains_set_lookahead(g.handle, 0) == _lib.BH_ERR_ARG and lib.bh_chains_set_lookahead(g.handle, 65) == _lib.BH_ERR_ARG
    with pytest.raises(ValueError):
        ChainPool(joint_target(DATA), initparams=case['initparams'], modelpriors=case['priors'], seeds=[1], evaluator=toy, lookahead=65)


def test_lookahead_chain_is_the_golden_chain(oracle, golden_chains):
    """The committed reference chains (real forward values from the oracle) with a look-ahead of 6 proposals."""
    for name in sorted(CASES):
        case = CASES[name]
        seeds = [int(s) for s in golden_chains['%s/seeds' % name]]
        pool = make_pool(oracle, DATA, case, seeds=seeds, groups=1, lookahead=6).run()
        counts = pool.counters()[0]
        for i, seed in enumerate(seeds):
            got = pool.chain(i)
            assert got['n'] == int(golden_chains['%s/%d/n' % (name, seed)]) == counts[i]
            _same({k: golden_chains['%s/%d/%s' % (name, seed, k)] for k in
                   ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter')}, got, (name, seed))
        calls, iters, rows = pool.advance()
        assert iters == len(seeds) * (case['burnin'] + case['main']) and calls < 0.6 * (case['burnin'] + case['main'])


def test_lookahead_changed_mid_run_and_short_runs(lib):
    """The look-ahead may be changed whenever no results are outstanding, a tree never reaches past the last
    iteration (2 iterations with 16 proposals per call), and a chain that overflows its storage inside a tree stops
    the pool like it does without one."""
    from bayhunter_amd import _lib
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['constrained']

    def toy(packed, nlay, noise):
        vs, h = packed[:, 2, :], packed[:, 0, :]
        d = vs[:, 0] - 3.1 + 0.01 * nlay + 0.002 * h.sum(axis=1)
        return -40. * d * d - 3. * noise[:, 3], np.stack([np.abs(d), np.abs(d), 2 * np.abs(d)], axis=1)

    def pool(burnin, main, lookahead, n=9, **kw):
        ip = dict(case['initparams'], iter_burnin=burnin, iter_main=main, acceptance=(40, 100))
        return ChainPool(joint_target(DATA), initparams=ip, modelpriors=case['priors'], seeds=np.arange(n) + 5, evaluator=toy,
                         groups=1, lookahead=lookahead, **kw)
    ref = pool(260, 140, 1).run()
    # 16 rows per chain are staged; the look-ahead switches 16 -> 3 -> 1 -> 9 while the pool runs
    p = pool(260, 140, 16)
    g = p.groups[0]
    p._launch(g)
    step = 0
    while True:
        p._land(g)
        if g.done():
            break
        step += 1
        if step in (5, 11, 40):
            assert lib.bh_chains_set_lookahead(g.handle, {5: 3, 11: 1, 40: 9}[step]) == _lib.BH_OK
        g.propose()                                       # ... and refused while results are due
        assert lib.bh_chains_set_lookahead(g.handle, 2) == _lib.BH_ERR_ARG
        n = g.count
        logL, mis = toy(g.packed[:n], g.nlay[:n], g.noise[:n])
        g.accept(np.ascontiguousarray(logL), mis)
        if g.done():
            break
        p._launch(g)
    for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter'):
        assert np.array_equal(getattr(p, k), getattr(ref, k), equal_nan=True), k
    for a, b in zip(p.counters(), ref.counters()):
        assert np.array_equal(a, b)
    # two iterations in all: the tree stops at the last one
    short1, short16 = pool(1, 1, 1, nmodels=3).run(), pool(1, 1, 16, nmodels=3).run()
    assert short16.advance()[1] == short1.advance()[1] == 9 * 2 and short16.advance()[0] <= 2
    for k in ('models', 'likes', 'iter'):
        assert np.array_equal(getattr(short1, k), getattr(short16, k), equal_nan=True), k
    # storage for three rows per chain: a chain overflows inside a tree
    with pytest.raises(_lib.BayHunterAmdError, match='storage'):
        pool(40, 20, 8, nmodels=3).run()


def test_move_and_acceptance_queries(lib):
    """bh_chains_moves / bh_chains_accepted: per model of the last batch, which move produced it and
    whether it became the chain's current model; consistent with the stored samples."""
    from bayhunter_amd import _lib
    from chain_scenario import joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['tutorial']
    seen_moves, taken = set(), 0

    def toy(packed, nlay, noise):
        d = packed[:, 2, 0] - 3.0
        return -30. * d * d - noise[:, 3], np.zeros((packed.shape[0], 3))
    ip = dict(case['initparams'], iter_burnin=60, iter_main=20, acceptance=(40, 100))
    pool = ChainPool(joint_target(DATA), initparams=ip, modelpriors=case['priors'], seeds=np.arange(40), evaluator=toy,
                     groups=1)
    g = pool.groups[0]
    first = True
    while not g.done():
        n = g.propose()
        moves = np.full(n, -9, dtype=np.int32)
        _lib.check(lib.bh_chains_moves(g.handle, moves.ctypes.data))
        assert np.all(moves == -1) if first else (np.all(moves >= 0) and np.all(moves <= 5))
        seen_moves.update(moves.tolist())
        logL, mis = toy(g.packed[:n], g.nlay[:n], g.noise[:n])
        g.accept(np.ascontiguousarray(logL), mis)
        flags = np.full(n, -9, dtype=np.int32)
        _lib.check(lib.bh_chains_accepted(g.handle, flags.ctypes.data))
        assert set(flags.tolist()) <= {0, 1} and (not first or np.all(flags == 1))
        taken += int(flags.sum())
        first = False
    assert seen_moves == {-1, 0, 1, 2, 3, 4, 5}
    assert taken == int(pool.counters()[0].sum())              # every stored sample was reported as taken


def test_hunt_script_reads_reference_config_and_writes_results(oracle, tmp_path):
    """tools/hunt.py: a BayHunter config.ini (the tutorial's values) + observed data files -> chain
    pool -> the reference's result files; the forward model is the oracle here (no GPU)."""
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
nchains = 3
iter_burnin = (10 * 4)
iter_main = (10 * 2)
propdist = 0.015, 0.015, 0.015, 0.005, 0.005
acceptance = 40, 45
thickmin = 0.1
lvz = None
hvz = None
rcond= 1e-5
station = 'st3'
savepath = '%s'
maxmodels = 50000
""" % str(tmp_path / 'results'))
    priors, ip = hunt.load_params(str(ini))
    assert priors['vpvs'] == (1.4, 2.1) and priors['mohoest'] is None and priors['swdnoise_corr'] == 0.0
    assert ip['iter_burnin'] == 40 and ip['station'] == 'st3' and ip['acceptance'] == (40, 45)
    joint = [None]

    def evaluator(packed, nlay, noise):
        return oracle_evaluator(joint[0])(packed, nlay, noise)
    real_build = hunt.build_targets

    def build(specs, gauss=None, p=None):
        j = real_build(specs, gauss, p)
        for t in j.targets:
            t.update_plugin(OraclePlugin(oracle, t.obsdata.x, t.ref))
        joint[0] = j
        return j
    hunt.build_targets = build
    try:
        pool = hunt.main([str(ini), '--target', 'rdispph=' + os.path.join(DATA, 'st3_rdispph.dat'),
                          '--target', 'prf=' + os.path.join(DATA, 'st3_prf.dat'), '--seed', '5', '--full-storage'],
                         evaluator=evaluator)
    finally:
        hunt.build_targets = real_build
    files = sorted(os.listdir(str(tmp_path / 'results' / 'data')))
    assert 'st3_config.pkl' in files and sum(f.endswith('.npy') for f in files) == 30
    assert pool.nchains == 3 and list(pool.seeds) == [np.random.RandomState(5).randint(1000) for _ in range(1)] + list(pool.seeds[1:])


def test_chain_pool_argument_errors(lib):
    """Misuse of the bh_chains_* entry points is reported, not executed."""
    from bayhunter_amd import _lib
    ERR = _lib.BH_ERR_ARG
    cfg = _lib.ChainConfig()
    cfg.ntargets, cfg.layers_min, cfg.layers_max = 1, 1, 5
    cfg.vs_min, cfg.vs_max, cfg.z_min, cfg.z_max = 2, 5, 0, 60
    cfg.vpvs_fixed, cfg.vpvs_min, cfg.vpvs_max = 1, 1.73, 1.73
    cfg.iter_burnin, cfg.iter_main = 5, 5
    arrs = [np.full((2, 4, w), np.nan, dtype=np.float32) for w in (12, 2, 1, 2, 1)]
    it = np.full((2, 4), np.nan)
    st = _lib.ChainStorage()
    st.nmodels = 4
    for name, a in zip(('models', 'misfits', 'likes', 'noise', 'vpvs'), arrs):
        setattr(st, name, a.ctypes.data)
    st.iter = it.ctypes.data
    seeds = np.array([1, 2], dtype=np.uint32)
    h = C.c_void_p()

    def create(c=cfg, n=2, s=st):
        return lib.bh_chains_create(C.byref(c), n, seeds.ctypes.data, C.byref(s), C.byref(h))
    assert create(n=0) == ERR
    bad = _lib.ChainConfig.from_buffer_copy(cfg)
    bad.ntargets = 0
    assert create(c=bad) == ERR
    bad = _lib.ChainConfig.from_buffer_copy(cfg)
    bad.layers_max = 100                                   # more nuclei than NL = 100 layers
    assert create(c=bad) == ERR and b'layer' in lib.bh_last_error()
    bad = _lib.ChainConfig.from_buffer_copy(cfg)
    bad.iter_main = -1
    assert create(c=bad) == ERR
    nost = _lib.ChainStorage()
    nost.nmodels = 4
    assert create(s=nost) == ERR and b'storage' in lib.bh_last_error()
    assert create() == _lib.BH_OK
    try:
        assert lib.bh_chains_set_threads(h, 0) == ERR
        assert lib.bh_chains_current(h, 2, None, None, None, None, None, None) == ERR
        assert lib.bh_chains_current(h, 1, None, None, None, None, None, None) == _lib.BH_OK
        out = np.zeros(3)
        assert lib.bh_chains_draw(h, 0, 3, 0.0, 1.0, 3, out.ctypes.data) == ERR
        assert lib.bh_chains_draw(h, 5, 0, 0.0, 1.0, 3, out.ctypes.data) == ERR
        key = np.zeros(624, dtype=np.uint32)
        assert lib.bh_chains_set_rng(h, 0, key.ctypes.data, 625, 0, 0.0) == ERR
        assert lib.bh_chains_propose(h, 8, None, None, None, None, None) == ERR
        assert lib.bh_chains_moves(h, None) == ERR and lib.bh_chains_accepted(h, None) == ERR
    finally:
        lib.bh_chains_destroy(h)
