"""GPU tier (-m gpu): the large-sample parity campaigns and the exact BASELINE.json shapes.

  * the bench workload's own rank-0 models: 131 072 ten-layer models, Rayleigh phase + P-RF, every
    value against the oracle (2.75 M dispersion values: bit-identical; RF <= 1e-10)
  * low-velocity-zone sets of 2 048 models per depth (5, 10, 15, ragged 2..31), four dispersion
    targets: derived bounds of tests/tolerances.py, >= 99 % of the values bit-identical
  * BASELINE cfg2 exactly: Rayleigh phase, 5 layers x 20 periods x 1 024 models
  * BASELINE cfg4 exactly: Rayleigh phase (21 periods) + P-RF, 15 layers x 64 models
  * BASELINE cfg3 / cfg5 shapes at full size against the oracle
  * models under a water layer (vs[0] = 0, surfdisp96.f:850-867) against goldens from the reference
Every kernel form the launcher can pick is covered where the batch size allows it.
"""
import os

import numpy as np
import pytest

from bayhunter_amd.synthetic import draw_models
from conftest import REFS
from tolerances import MIN_IDENTICAL_LVZ, TOL_RF
from test_gpu_parity import _check_swd, _engine

pytestmark = pytest.mark.gpu
THREADS = min(len(os.sched_getaffinity(0)), 16)


def _run(eng, H, VP, VS, RHO, nl, kernel='auto'):
    from bayhunter_amd import _lib
    _lib.set_swd_kernel(kernel)
    try:
        out, err = eng.run(H, VP, VS, RHO, nl)
        return out.cpu().numpy(), err.cpu().numpy()
    finally:
        _lib.set_swd_kernel('auto')


def test_bench_models_every_value(lib, oracle):
    """parity campaign of round 1 (tests/scenarios/parity_campaign.py) as a test: bench.py's rank-0
    models (seed 6000), throughput kernel in processing order."""
    B = 131072
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(B, 10, seed=6000, sorted_vs=True)
    eng = _engine(['rdispph'], per, rf=True)
    out, err = _run(eng, H, VP, VS, RHO, nl, 'lane')
    want, werr, nc = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0, nthreads=THREADS)
    assert np.array_equal(err[:, 0], werr) and werr.sum() == 0
    ndiff = int((out[:, :21] != want).sum())
    assert ndiff == 0, '%d of %d dispersion values differ' % (ndiff, want.size)
    assert 650 < nc / B < 750                        # evaluations per search bench.py normalises by
    wrf = oracle.rf_batch(H, VP, VS, RHO, nl, nthreads=THREADS)
    assert np.abs(out[:, 21:] - wrf).max() <= TOL_RF


@pytest.mark.parametrize('L', [5, 10, 15, (2, 31)])
def test_lvz_sets_2048(lib, oracle, L):
    """2 048 models with low-velocity zones per depth: Love bit-identical, Rayleigh within the
    reference's stopping bracket and >= 99 % bit-identical, err flags and zero fill identical;
    the automatic kernel choice and the throughput kernel."""
    B = 2048
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(B, L, seed=52000 + (sum(L) if isinstance(L, tuple) else L), sorted_vs=False)
    eng = _engine([r[0] for r in REFS], per, rf=True)
    res = {k: _run(eng, H, VP, VS, RHO, nl, k) for k in ('auto', 'lane')}
    assert np.array_equal(res['auto'][0], res['lane'][0], equal_nan=True) and np.array_equal(res['auto'][1], res['lane'][1])
    out, err = res['auto']
    for t, (name, iw, ig) in enumerate(REFS):
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=THREADS)
        _check_swd(name, out[:, eng.slices[t]], want, err[:, t], werr, min_identical=MIN_IDENTICAL_LVZ)
    wrf = oracle.rf_batch(H, VP, VS, RHO, nl, nthreads=THREADS)
    assert np.array_equal(np.isnan(out[:, eng.slices[4]]), np.isnan(wrf))
    assert np.nanmax(np.abs(out[:, eng.slices[4]] - wrf)) <= TOL_RF * max(1.0, np.nanmax(np.abs(wrf)))


@pytest.mark.parametrize('kernel', ['auto', 'lane', 'team', 'team128', 'team32', 'team16', 'team8'])
def test_cfg2_exact_shape(lib, oracle, kernel):
    """BASELINE.json configs[1]: Rayleigh phase only, 5 layers, 20 periods, 1 024 models (bench.py
    --workload cfg2, its seed)."""
    per = np.linspace(1, 41, 20)
    H, VP, VS, RHO, nl = draw_models(1024, 5, seed=2000, sorted_vs=True)
    assert H.shape == (1024, 5)
    eng = _engine(['rdispph'], per)
    out, err = _run(eng, H, VP, VS, RHO, nl, kernel)
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0, nthreads=THREADS)
    assert out.shape == (1024, 20)
    assert np.array_equal(err[:, 0], werr) and np.array_equal(out, want)


def test_cfg2_shape_with_lvz(lib, oracle):
    per = np.linspace(1, 41, 20)
    H, VP, VS, RHO, nl = draw_models(1024, 5, seed=2001, sorted_vs=False)
    eng = _engine(['rdispph'], per)
    out, err = _run(eng, H, VP, VS, RHO, nl)
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0, nthreads=THREADS)
    _check_swd('rdispph', out, want, err[:, 0], werr, min_identical=MIN_IDENTICAL_LVZ)


@pytest.mark.parametrize('kernel', ['auto', 'lane', 'team', 'team128', 'team256', 'team8'])
def test_cfg4_exact_shape(lib, oracle, kernel):
    """BASELINE.json configs[3], one GPU's share: Rayleigh phase (21 periods) + P receiver function,
    15 layers, 64 models (bench.py --workload cfg4, its seed)."""
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(64, 15, seed=4000, sorted_vs=True)
    assert H.shape == (64, 15)
    eng = _engine(['rdispph'], per, rf=True)
    out, err = _run(eng, H, VP, VS, RHO, nl, kernel)
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0)
    assert out.shape == (64, 222)
    assert np.array_equal(err[:, 0], werr) and np.array_equal(out[:, :21], want)
    assert np.abs(out[:, 21:] - oracle.rf_batch(H, VP, VS, RHO, nl)).max() <= TOL_RF


def test_cfg3_full_size_every_value(lib, oracle):
    """BASELINE.json configs[2]: Rayleigh + Love x phase + group, 10 layers, 40 periods, 8 192
    models (bench.py --workload cfg3, its seed): all 1.3 M values against the oracle."""
    per = np.linspace(1, 41, 40)
    H, VP, VS, RHO, nl = draw_models(8192, 10, seed=3000, sorted_vs=True)
    eng = _engine([r[0] for r in REFS], per)
    out, err = _run(eng, H, VP, VS, RHO, nl)
    for t, (name, iw, ig) in enumerate(REFS):
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=THREADS)
        _check_swd(name, out[:, eng.slices[t]], want, err[:, t], werr, monotone=True)


def test_cfg5_ragged_full_size(lib, oracle):
    """BASELINE.json configs[4], one GPU's pool: ragged 2..31 layers, Rayleigh phase + P-RF, 8 192
    models (bench.py --workload cfg5, its seed)."""
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(8192, (2, 31), seed=5000, sorted_vs=True)
    assert nl.min() == 2 and nl.max() == 31
    eng = _engine(['rdispph'], per, rf=True)
    out, err = _run(eng, H, VP, VS, RHO, nl)
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0, nthreads=THREADS)
    _check_swd('rdispph', out[:, :21], want, err[:, 0], werr, monotone=True)
    assert np.abs(out[:, 21:] - oracle.rf_batch(H, VP, VS, RHO, nl, nthreads=THREADS)).max() <= TOL_RF


@pytest.mark.parametrize('kernel', ['lane', 'team', 'team256', 'team32', 'team16', 'team8'])
@pytest.mark.parametrize('tag', ['L5_sorted', 'L5_lvz', 'L10_sorted', 'L10_lvz'])
def test_water_layer_golden(lib, golden, tag, kernel):
    """vs[0] = 0 -> llw = 2: the layer loop stops above the water layer and the tail of dltar4
    (surfdisp96.f:850-867) closes the period equation; Love ignores the water layer (:732).
    Goldens from the reference binary (tests/golden/make_golden.py water)."""
    g = golden['swd_water']
    H, VP, VS, RHO = g[tag + '_model']
    nl = np.array([1 + int((m[1:] > 0).sum()) for m in VS], dtype=np.int32)
    eng = _engine([r[0] for r in REFS], g['periods'])
    out, err = _run(eng, H, VP, VS, RHO, nl, kernel)
    for t, (name, _, _) in enumerate(REFS):
        _check_swd(name, out[:, eng.slices[t]], g[tag + '_' + name], err[:, t], g[tag + '_' + name + '_err'],
                   monotone=tag.endswith('sorted'))


def test_water_layer_large_sample(lib, oracle):
    """1 024 water-covered models with low-velocity zones against the oracle (all four targets)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_golden import water_models
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = water_models(1024, 8, 77001, False)
    eng = _engine([r[0] for r in REFS], per)
    out, err = _run(eng, H, VP, VS, RHO, nl)
    for t, (name, iw, ig) in enumerate(REFS):
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=THREADS)
        _check_swd(name, out[:, eng.slices[t]], want, err[:, t], werr, min_identical=MIN_IDENTICAL_LVZ)
