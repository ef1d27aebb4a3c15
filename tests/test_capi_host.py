"""CPU tier: the C-ABI library loads, exports every symbol the header declares, refuses to compute
without a GPU (no CPU fallback), and the Python host side mirrors the reference's plugin contract."""
import os
import sys
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'bayhunter_amd.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(bh_[a-z0-9_]+|surfdisp96_|synrf_cwrap)\s*\(', txt)))


def test_header_symbols_exported(lib):
    from bayhunter_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 15
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if ' T ' in l)
    for s in declared:
        assert s in exported, s
        assert hasattr(lib, s)
    assert sorted(_lib.EXPORTS) == declared


def test_no_torch_types_in_abi():
    txt = open(os.path.join(ROOT, 'include', 'bayhunter_amd.h')).read()
    assert 'torch' not in txt and 'at::' not in txt and 'hipStream_t' in txt  # mentioned in prose only
    assert re.search(r'void \*stream', txt)


def test_product_never_touches_oracle():
    """The shipped package and library must not import, link or call the oracle / host simulator."""
    pkg = os.path.join(ROOT, 'bayhunter_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert 'pyoracle' not in src and 'liboracle' not in src and 'bho_' not in src, f
                assert 'import oracle' not in src and 'from oracle' not in src, f
                assert 'libhostsim' not in src, f
    from bayhunter_amd import _lib
    out = subprocess.run(['ldd', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'oracle' not in out and 'hostsim' not in out
    # helper scripts are not test code either: the diagnostics that need the oracle live in tests/
    for f in os.listdir(os.path.join(ROOT, 'tools')):
        if f.endswith(('.py', '.sh')):
            src = open(os.path.join(ROOT, 'tools', f)).read()
            assert 'pyoracle' not in src and 'import oracle' not in src and 'from oracle' not in src, f
    # bench.py: only the cpu_baseline leg (its worker function) may name the oracle
    import ast
    tree = ast.parse(open(os.path.join(ROOT, 'bench.py')).read())
    users = {fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             and any('oracle' in ast.dump(n) for n in ast.walk(fn) if isinstance(n, (ast.Import, ast.ImportFrom)))}
    assert users and users <= {'_cpu_worker', 'cpu_baseline'}, users
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and 'oracle' in ast.dump(n)]
    assert not top


def test_arg_validation_without_gpu(lib):
    from bayhunter_amd import _lib
    import ctypes as C
    tg = (_lib.SwdTarget * 1)(_lib.SwdTarget(3, 0, 1, 0, 21, 0, 0, 0))   # iwave 3 is invalid
    rc = lib.bh_swd_batch(4, 101, 101, 1, 1, 1, 1, 1, 1, tg, 1, 1, 21, 1, None, 0, None)
    assert rc == _lib.BH_ERR_ARG
    tg[0].iwave = 2
    rc = lib.bh_swd_batch(4, 10, 9, 1, 1, 1, 1, 1, 1, tg, 1, 1, 21, 1, None, 0, None)   # stride < Lmax
    assert rc == _lib.BH_ERR_ARG
    assert lib.bh_swd_workspace_bytes(100, 1, tg) == 0
    tg[0].mode = 2
    assert lib.bh_swd_workspace_bytes(100, 1, tg) == 100 * 2 * 60 * 8
    n = C.c_int(-1)
    assert lib.bh_device_count(C.byref(n)) == 0 and n.value >= 0


def test_fails_loudly_without_device(lib):
    import bayhunter_amd as bh
    if bh.device_count() > 0:
        pytest.skip('a GPU is present')
    sd = bh.SurfDisp(np.linspace(1, 41, 21), 'rdispph')
    with pytest.raises(bh.BayHunterAmdError):
        sd.run_model(np.array([5., 0.]), np.array([5., 7.]), np.array([3., 4.]), np.array([2.4, 3.]))
    rf = bh.RFminiModRF(np.linspace(-5, 35, 201), 'prf')
    with pytest.raises(bh.BayHunterAmdError):
        rf.run_model(np.array([5., 0.]), np.array([5., 7.]), np.array([3., 4.]), np.array([2.4, 3.]))


def test_eval_plan_refuses_without_a_device_and_validates_arguments(lib):
    """bh_eval_create: argument errors before any device work; with valid arguments and no GPU the plan is not
    created (BH_ERR_NO_DEVICE) -- the chain pool's evaluator has no CPU path to fall back to."""
    import ctypes as C
    import bayhunter_amd as bh
    from bayhunter_amd import _lib
    from bayhunter_amd import targets as T
    x, t = np.linspace(1, 41, 21), np.linspace(-5, 35, 201)
    joint = T.JointTarget([T.RayleighDispersionPhase(x, np.full(21, 3.5)), T.PReceiverFunction(t, np.zeros(201))])
    joint.set_target_covariance([True, True], [0.0, 0.0], None)
    bl = joint.batch_layout()
    lay = bl['layout']
    assert lay.row == 222 and bl['nflags'] == 1 and [d.n for d in bl['desc']] == [21, 201]
    rfp = (_lib.RfParams * 1)(*lay.rfp)
    handle = C.c_void_p()

    def create(rows=8, Lmax=10, nflags=1, ntargets=2):
        return lib.bh_eval_create(rows, Lmax, lay.row, 1, lay.tg, lay.periods.ctypes.data, lay.periods.size, 1, rfp,
                                  ntargets, bl['desc'], nflags, bl['yobs'].ctypes.data, bl['aux'].ctypes.data,
                                  bl['aux'].size, 0, None, 1, C.byref(handle))
    assert create(rows=0) == _lib.BH_ERR_ARG and create(Lmax=101) == _lib.BH_ERR_ARG
    assert create(nflags=3) == _lib.BH_ERR_ARG and create(ntargets=0) == _lib.BH_ERR_ARG
    assert lib.bh_eval_submit(None, 1) == _lib.BH_ERR_ARG and lib.bh_eval_wait(None, None) == _lib.BH_ERR_ARG
    if bh.device_count() > 0:
        pytest.skip('a GPU is present: the rest is the GPU tier')
    assert create() == _lib.BH_ERR_NO_DEVICE and not handle.value
    with pytest.raises(bh.BayHunterAmdError):
        joint.eval_plan(8, 10)
    from bayhunter_amd.chains import ChainPool
    with pytest.raises(bh.BayHunterAmdError):
        ChainPool(joint, initparams=dict(iter_burnin=4, iter_main=4), modelpriors=dict(swdnoise_corr=0., rfnoise_corr=0.),
                  seeds=[1, 2])


def test_plugin_contract():
    """ctor (obsx, ref), ref -> (iwave, igr) map, ReferenceError, modelparams keys, obs params
    (reference: surf96_modsw.py:24-66, rfmini_modrf.py:17-62)."""
    import bayhunter_amd as bh
    x = np.linspace(1, 41, 21)
    tags = {'rdispgr': (2, 1), 'ldispgr': (1, 1), 'rdispph': (2, 0), 'ldispph': (1, 0)}
    for ref, t in tags.items():
        sd = bh.SurfDisp(x, ref)
        assert (sd.wavetype, sd.veltype) == t and sd.kmax == 21
        assert sd.modelparams == {'mode': 1, 'flsph': 0}
    with pytest.raises(ReferenceError):
        bh.SurfDisp(x, 'xdisp')
    sd = bh.SurfDisp(np.linspace(2, 80, 75), 'rdispph')
    assert sd.obsx_int.size == 60 and sd.obsx_int[0] == 2 and sd.obsx_int[-1] == 80
    sd.set_modelparams(mode=2)
    assert sd.modelparams['mode'] == 2

    rf = bh.RFminiModRF(np.linspace(-5, 35, 201), 'prf')
    assert rf.modelparams == {'wtype': 'P', 'gauss': 1.0, 'p': 6.4, 'water': 0.001, 'nsv': None}
    assert abs(rf.fsamp - 5.0) < 1e-12 and rf.tshft == 5.0 and rf.nsamp == 512
    assert bh.RFminiModRF(np.linspace(-5, 35, 201), 'srf').modelparams['wtype'] == 'SV'
    with pytest.raises(ValueError):
        bh.RFminiModRF(np.array([0., 0.2, 0.5, 0.6]), 'prf')


def test_bench_evaluation_counts_are_the_oracles():
    """bench.py normalises its flop figure by the reference path's evaluation count; the
    committed table must be what the oracle's counter says for the default workload."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
    import bench
    import count_dltar
    counts, lmean = count_dltar.count(bench.WORKLOADS['joint10'])
    assert (counts, lmean) == bench.N_DLTAR['joint10']


def test_form_plan_follows_the_measured_table():
    """bh_swd_plan_forms (host only): one form per call by the measured table (swd_form_table.h) -- a handful of models
    on wide teams, a few thousand on the narrow ones (one trial per lane, round 4), a saturated chip on the lane kernel
    -- with the call's work counted in table searches (BASELINE cfg3: four targets x 40 periods are 7.9 of them per
    model) and its heaviest target bounding the latency."""
    import ctypes as C
    from bayhunter_amd import _lib
    lib = _lib.load()

    def plan(B, L, specs):
        tg = (_lib.SwdTarget * len(specs))()
        off = 0
        for i, (iwave, igr, nper) in enumerate(specs):
            tg[i] = _lib.SwdTarget(iwave, igr, 1, 0, nper, off, off, 0)
            off += nper
        forms = (C.c_int * len(specs))()
        _lib.check(lib.bh_swd_plan_forms(B, L, len(specs), tg, 256, forms))
        return list(forms)
    cfg3 = [(2, 0, 40), (2, 1, 40), (1, 0, 40), (1, 1, 40)]
    assert plan(8192, 10, cfg3) == [8] * 4 and plan(8192, 5, cfg3) == [8] * 4
    assert plan(2048, 10, cfg3) == [16] * 4 and plan(16384, 10, cfg3) == [0] * 4
    assert plan(64, 10, cfg3) == [512] * 4 and plan(524288, 10, cfg3) == [0] * 4
    assert plan(12288, 10, [(2, 0, 21)]) == [16] and plan(524288, 10, [(2, 0, 21)]) == [0]
    # a sampler's ragged batch (proposals of a tutorial pool: 2-14 layers, 4.8 on average): priced by its mean depth and
    # by the number of batches in flight once the caller says so (bh_swd_hint), by its deepest model otherwise
    one = [(2, 0, 21)]
    assert plan(4096, 14, one) == [16]
    _lib.check(lib.bh_swd_hint(4.8, 2))
    assert plan(4096, 14, one) == [32]
    assert plan(4096, 14, one) == [16]                       # a hint is about one call
    _lib.check(lib.bh_swd_hint(4.8, 1))
    assert plan(8192, 14, one) == [32]
    _lib.check(lib.bh_swd_hint(4.8, 1))
    assert plan(2048, 14, one) == [128]                      # the deepest model's latency still bounds a small batch
    _lib.check(lib.bh_swd_hint(16.75, 1))
    assert plan(8192, 31, one) == [16]                       # BASELINE cfg5: 16 lanes per search, one trial each
    assert lib.bh_swd_hint(-1.0, 1) != 0 and lib.bh_swd_hint(3.0, 0) != 0
    for B in (1, 64, 1024, 8192, 65536):
        for L in (3, 10, 30):
            f = plan(B, L, cfg3)
            assert len(set(f)) <= 3 and all(w in (0, 8, 16, 32, 64, 128, 256, 512) for w in f)
    assert lib.bh_swd_plan_forms(8192, 10, 0, None, 256, None) != 0
