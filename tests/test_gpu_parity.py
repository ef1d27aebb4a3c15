"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden
vectors.

Tolerances: tests/tolerances.py (derived from the reference's own stopping criterion, with the
evidence that the reference differs from itself by as much when only glibc's libm variant changes,
profiles/r02_libm_selfdiff.txt).  In short:
  * velocity increasing with depth (tutorial, bench workloads): dispersion bit-identical
  * low-velocity zones: Rayleigh phase |dc|/c <= 1.2e-6, group |dU|/U <= 2.5e-4, >= 99 % of the
    values bit-identical on large sets; Love bit-identical
  * err flags identical; receiver function |diff| <= 1e-10
  * north_star: RMS misfit against the tutorial dataset within 1e-6 of the reference's
The large-sample campaigns (131 072 bench models, 2 048-model LVZ sets, exact BASELINE shapes, water
layer) are in tests/test_gpu_campaign.py.
"""
import os

import numpy as np
import pytest

from bayhunter_amd.synthetic import draw_models, tutorial_model
from conftest import GOLDEN, REFS, SETS

pytestmark = pytest.mark.gpu

from tolerances import (MIN_IDENTICAL_LVZ_SMALL, MIN_IDENTICAL_MONOTONE, TOL_GROUP, TOL_GROUP_REL,
                        TOL_MISFIT, TOL_PHASE, TOL_PHASE_REL, TOL_RF)


def _nlay(model):
    return np.array([int((m > 0).sum()) for m in model[2]], dtype=np.int32)


def _engine(refs, per, rf=False, **kw):
    from bayhunter_amd.engine import ForwardEngine, SwdSpec, RfSpec
    return ForwardEngine(swd=[SwdSpec(r, per, **kw) for r in refs],
                         rf=[RfSpec('prf', np.linspace(-5, 35, 201))] if rf else [])


def _check_swd(name, got, want, err_got, err_want, monotone=False, min_identical=MIN_IDENTICAL_LVZ_SMALL):
    """err flags equal; solved rows within the derived bounds (tolerances.py); Love and monotone
    models bit-identical; zero-filled tails (failed searches) identical."""
    assert np.array_equal(err_got, err_want), name
    bad = err_want != 0
    assert np.array_equal(got[bad], want[bad]), name             # zero fill from the failing period on
    ok = ~bad
    got, want = got[ok], want[ok]
    if got.size == 0:
        return 1.0
    frac = float((got == want).mean())
    if monotone or name.startswith('l'):
        assert frac >= MIN_IDENTICAL_MONOTONE, (name, frac)
        return frac
    rel = np.abs(got - want) / np.abs(want)
    assert rel.max() <= (TOL_GROUP_REL if name.endswith('gr') else TOL_PHASE_REL), (name, rel.max())
    assert frac >= min_identical, (name, frac)
    return frac


@pytest.mark.parametrize('tag', SETS + ['ragged'])
def test_golden_sets(lib, golden, tag):
    g = golden['swd_rf_random']
    H, VP, VS, RHO = g[tag + '_model']
    nl = g['ragged_nlay'] if tag == 'ragged' else _nlay(g[tag + '_model'])
    eng = _engine([r[0] for r in REFS], g['periods'], rf=True)
    out, err = eng.run(H, VP, VS, RHO, nl)
    out, err = out.cpu().numpy(), err.cpu().numpy()
    for t, (name, _, _) in enumerate(REFS):
        _check_swd(name, out[:, eng.slices[t]], g[tag + '_' + name], err[:, t], g[tag + '_' + name + '_err'],
                   monotone=tag.endswith('sorted') or tag == 'ragged')
    rf = out[:, eng.slices[4]]
    want = g[tag + '_prf']
    assert np.array_equal(np.isnan(rf), np.isnan(want))
    assert np.nanmax(np.abs(rf - want)) <= TOL_RF


def test_swd_variants(lib, golden):
    g = golden['swd_variants']
    H, VP, VS, RHO = g['model']
    nl = _nlay(g['model'])
    per = np.linspace(1, 41, 21)
    for mode in (1, 2, 3):
        for fl in (0, 1):
            eng = _engine([r[0] for r in REFS], per, mode=mode, flsph=fl)
            out, err = eng.run(H, VP, VS, RHO, nl)
            out, err = out.cpu().numpy(), err.cpu().numpy()
            for t, (name, _, _) in enumerate(REFS):
                key = '%s_m%d_f%d' % (name, mode, fl)
                got, want = out[:, eng.slices[t]], g[key]
                assert np.array_equal(err[:, t], g[key + '_err']), key
                # (flsph=1: btp**(-2.275) is evaluated in fp64 and rounded once, reproducing glibc's powf)
                tol = TOL_GROUP if name.endswith('gr') else TOL_PHASE
                assert np.abs(got - want).max() <= tol, (key, np.abs(got - want).max())
    for P in (20, 40, 60):
        eng = _engine([r[0] for r in REFS], np.linspace(1, 41, P))
        out, err = eng.run(H, VP, VS, RHO, nl)
        out, err = out.cpu().numpy(), err.cpu().numpy()
        for t, (name, _, _) in enumerate(REFS):
            _check_swd(name, out[:, eng.slices[t]], g['%s_P%d' % (name, P)], err[:, t], g['%s_P%d_err' % (name, P)],
                       monotone=True)


def test_rf_variants(lib, golden):
    from bayhunter_amd.engine import ForwardEngine, RfSpec
    g = golden['rf_variants']
    H, VP, VS, RHO = g['model']
    nl = _nlay(g['model'])
    worst = 0.0
    for key in g.files:
        if key == 'model':
            continue
        w, gs, p, n, nsv = key.split('_')
        n = int(n[1:])
        x = np.arange(n // 2) / 5.0 - 5.0            # n/2 samples at 5 Hz -> nsamp = n
        eng = ForwardEngine(rf=[RfSpec('prf' if w == 'w0' else 'srf', x, float(gs[1:]), float(p[1:]),
                                       3.0 if nsv == 'nsv' else None)])
        assert int(eng.rf[0].nsamp) == n
        out, _ = eng.run(H, VP, VS, RHO, nl)
        d = np.abs(out.cpu().numpy() - g[key]).max()
        worst = max(worst, d)
        assert d <= TOL_RF, (key, d)
    print('rf variants worst |diff| = %.3e' % worst)


def test_more_than_60_periods_in_the_batched_engine(lib):
    """75 periods: the engine solves on 60 evenly spaced ones and interpolates on the device, which
    must be SurfDisp.run_model's numpy.interp of the single-model solver, bit for bit; the fused
    likelihood sees the interpolated columns."""
    import torch
    from bayhunter_amd import targets as T
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
    from bayhunter_amd.plugins import SurfDisp
    per = np.sort(np.concatenate([np.linspace(1.0, 41.0, 60), np.linspace(1.3, 40.2, 15)]))   # some on the grid
    H, VP, VS, RHO, nl = draw_models(40, (2, 9), seed=91, sorted_vs=False)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', per), SwdSpec('ldispgr', np.linspace(2, 30, 9))],
                        rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    assert eng.ncols == 75 + 9 + 201 and eng.row == eng.ncols + 60
    out, err = eng.run(H, VP, VS, RHO, nl)
    out, err = out.cpu().numpy(), err.cpu().numpy()
    plug = SurfDisp(per, 'rdispph')
    for b in range(40):
        n = nl[b]
        x, y = plug.run_model(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n])
        if err[b, 0]:
            assert np.isnan(y) if np.isscalar(y) else False
        else:
            assert np.array_equal(x, per) and np.array_equal(out[b, eng.slices[0]], y), b
    # through JointTarget.evaluate_batch against the one-model path
    rs = np.random.RandomState(3)
    t1 = T.RayleighDispersionPhase(per, 3.0 + 0.02 * per + 0.01 * rs.randn(75))
    t2 = T.PReceiverFunction(np.linspace(-5, 35, 201), 0.05 * rs.randn(201))
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([True, False], [0.0, 0.9])
    noise = np.tile([0.0, 0.02, 0.85, 0.01], (40, 1))
    logL, mis = joint.evaluate_batch(H, VP, VS, nl, noise)
    logL, mis = logL.cpu().numpy(), mis.cpu().numpy()
    for b in range(0, 40, 7):
        n = nl[b]
        joint.evaluate(h=H[b, :n], vp=VP[b, :n], vs=VS[b, :n], noise=noise[b])
        assert np.isclose(joint.proposallikelihood, logL[b], rtol=1e-11, atol=1e-6)
        assert np.allclose(joint.proposalmisfits, mis[b], rtol=1e-11)


def test_batch_depth_hint_changes_nothing(lib):
    """Rows allocated for 31 layers holding models of at most 9: with the depth known the kernels
    size their LDS for 10 layers (and may pick another team width); values are the same bits."""
    import torch
    from bayhunter_amd.engine import DeviceModels, ForwardEngine, RfSpec, SwdSpec
    H, VP, VS, RHO, nl = draw_models(3000, (2, 9), seed=5, sorted_vs=False)
    pad = lambda a: np.concatenate([a, np.zeros((a.shape[0], 31 - a.shape[1]))], axis=1)
    H, VP, VS, RHO = pad(H), pad(VP), pad(VS), pad(RHO)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, 21)), SwdSpec('ldispgr', np.linspace(2, 30, 9))],
                        rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    with_hint = eng.upload(H, VP, VS, RHO, nl)
    assert with_hint.depth == 9
    plain = DeviceModels(with_hint.packed, with_hint.nlay)
    a, ea = eng.run(with_hint)
    b, eb = eng.run(plain)
    torch.cuda.synchronize()
    assert torch.equal(ea, eb) and torch.equal(a.nan_to_num(nan=-1.0), b.nan_to_num(nan=-1.0))


def test_rf_post_critical_slowness(lib, oracle):
    """Slowness 14 s/deg (0.126 s/km) is post-critical for vp > 7.9 km/s: the interface coefficient
    matrices of such models are complex (the general form of the recursion), those of the slower
    models real (the specialised form).  For an incident P wave the reference's delay sum then
    takes the root of a negative number and the whole trace is NaN (reproduced); an incident SV
    wave stays finite and exercises the complex form.  Both against the oracle."""
    from bayhunter_amd.engine import ForwardEngine, RfSpec
    H, VP, VS, RHO, nl = draw_models(96, (3, 9), seed=77)
    x = np.linspace(-5, 35, 201)
    post = (VP.max(axis=1) * 14.0 * 0.00899) > 1.0
    assert 10 < post.sum() < 86                                   # both kinds present
    for ref, wn in (('prf', 0), ('srf', 1)):
        eng = ForwardEngine(rf=[RfSpec(ref, x, 1.0, 14.0)])
        out, _ = eng.run(H, VP, VS, RHO, nl)
        out = out.cpu().numpy()
        want = np.stack([oracle.rf_model(H[b, :nl[b]], VP[b, :nl[b]], VS[b, :nl[b]], RHO[b, :nl[b]], p=14.0,
                                         waveno=wn, nout=201) for b in range(96)])
        ok = np.isfinite(want).all(axis=1)
        # (where the rotation of the top layer is itself post-critical the reference returns NaN)
        assert np.array_equal(np.isfinite(out).all(axis=1), ok)
        if wn == 1:
            assert (ok & post).sum() >= 5 and (ok & ~post).sum() >= 5, ((ok & post).sum(), (ok & ~post).sum())
        else:
            assert not (ok & post).any() and (~ok).sum() >= 5
        scale = np.maximum(1.0, np.abs(want[ok]).max(axis=1, keepdims=True))
        assert (np.abs(out[ok] - want[ok]) / scale).max() <= TOL_RF, ref


def test_tutorial_dataset_single_model_dropins(lib, golden):
    """north_star: misfit within 1e-6 of SURF96/rfmini on the tutorial dataset, through the
    single-model drop-ins (bh_surfdisp96 / bh_synrf) behind the plugin classes."""
    import bayhunter_amd as bh
    h, vp, vs, rho = tutorial_model()
    per = np.linspace(1, 41, 21)
    full = golden['tutorial_full']
    for name, _, _ in REFS:
        obs = np.loadtxt(os.path.join(GOLDEN, 'tutorial_observed', 'st3_%s.dat' % name))
        x, y = bh.SurfDisp(per, name).run_model(h, vp, vs, rho)
        assert np.array_equal(x, per)
        assert np.array_equal(y, full[name]), name                   # the tutorial model: bit-identical
        mis_ref = np.sqrt(np.mean((full[name] - obs[:, 1]) ** 2))
        mis_gpu = np.sqrt(np.mean((y - obs[:, 1]) ** 2))
        assert abs(mis_ref - mis_gpu) <= TOL_MISFIT
        assert np.abs(y - obs[:, 1]).max() <= 5.1e-5
    for name in ('prf', 'srf'):
        obs = np.loadtxt(os.path.join(GOLDEN, 'tutorial_observed', 'st3_%s.dat' % name))
        t, y = bh.RFminiModRF(obs[:, 0], name).run_model(h, vp, vs, rho)
        assert np.allclose(t, obs[:, 0])
        assert np.abs(y - full[name]).max() <= TOL_RF
        mis_ref = np.sqrt(np.mean((full[name] - obs[:, 1]) ** 2))
        mis_gpu = np.sqrt(np.mean((y - obs[:, 1]) ** 2))
        assert abs(mis_ref - mis_gpu) <= TOL_MISFIT


def test_failure_semantics_and_nan(lib, oracle):
    """err flags, zero fill, (nan, nan) from run_model, NaN model terminates."""
    import bayhunter_amd as bh
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(512, 6, seed=77, sorted_vs=False)
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0)
    sd = bh.SurfDisp(per, 'rdispph')
    x, Y, e = sd.run_models(H, VP, VS, RHO, nl)
    assert np.array_equal(e, werr) and 0 < e.sum() < 256
    assert np.all(np.isnan(Y[e != 0])) and np.abs(Y[e == 0] - want[werr == 0]).max() <= TOL_PHASE
    b = int(np.nonzero(werr)[0][0])
    n = nl[b]
    xm, ym = sd.run_model(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n])
    assert np.isnan(xm) and np.isnan(ym)
    # NaN velocity: must end (no root), not hang
    vs = np.array([3.0, np.nan, 4.5])
    xm, ym = sd.run_model(np.array([5., 10., 0.]), vs * 1.73, vs, vs * 1.73 * .32 + .77)
    assert np.isnan(xm)


def test_edge_shapes(lib, oracle):
    """B=1, B not a multiple of the workgroup, 1- and 2-layer models, 100 layers, 1 and 60 periods."""
    per = np.linspace(1, 41, 21)
    for B, L in ((1, 4), (63, 3), (65, 2), (130, 7)):
        H, VP, VS, RHO, nl = draw_models(B, L, seed=B * 10 + L)
        eng = _engine(['rdispph', 'ldispgr'], per, rf=True)
        out, err = eng.run(H, VP, VS, RHO, nl)
        out, err = out.cpu().numpy(), err.cpu().numpy()
        for t, (iw, ig) in enumerate(((2, 0), (1, 1))):
            want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, ig)
            assert np.array_equal(err[:, t], werr)
            assert np.abs(out[:, eng.slices[t]] - want).max() <= (TOL_GROUP if ig else TOL_PHASE)
        assert np.abs(out[:, eng.slices[2]] - oracle.rf_batch(H, VP, VS, RHO, nl)).max() <= TOL_RF
    # half-space only: Love has no root (err=1), RF is NaN (greens.cpp:208,572)
    H, VP, VS, RHO, nl = draw_models(3, 1, seed=5)
    eng = _engine(['rdispph', 'ldispph'], per, rf=True)
    out, err = eng.run(H, VP, VS, RHO, nl)
    out, err = out.cpu().numpy(), err.cpu().numpy()
    for t, iw in enumerate((2, 1)):
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, 0)
        assert np.array_equal(err[:, t], werr) and np.abs(out[:, eng.slices[t]] - want).max() <= TOL_PHASE
    assert np.all(np.isnan(out[:, eng.slices[2]]))
    # 100 layers (NL), 60 periods (NP), single period
    H, VP, VS, RHO, nl = draw_models(5, 100, seed=6, zmax=300.0, thickmin=0.05)
    for P in (1, 60):
        p = np.linspace(2, 50, P)
        eng = _engine(['rdispph'], p, rf=(P == 1))
        out, err = eng.run(H, VP, VS, RHO, nl)
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, p, 2, 0)
        assert np.array_equal(err.cpu().numpy()[:, 0], werr)
        assert np.abs(out.cpu().numpy()[:, :P] - want).max() <= TOL_PHASE
        if P == 1:
            assert np.abs(out.cpu().numpy()[:, 1:] - oracle.rf_batch(H, VP, VS, RHO, nl)).max() <= TOL_RF


def test_more_than_60_periods_interpolation(lib, oracle):
    import bayhunter_amd as bh
    h, vp, vs, rho = tutorial_model()
    x = np.linspace(2, 40, 75)
    sd = bh.SurfDisp(x, 'rdispph')
    xm, ym = sd.run_model(h, vp, vs, rho)
    want, err = oracle.swd(h, vp, vs, rho, sd.obsx_int, 2, 0)
    assert err == 0 and np.array_equal(xm, x)
    assert np.abs(ym - np.interp(x, sd.obsx_int, want)).max() <= TOL_PHASE


def test_full_size_properties(lib, oracle):
    """BASELINE cfg3 size (8192 models x 10 layers x 4 targets x 40 periods) + RF: properties that
    do not need the oracle on every model, plus an oracle spot check on a slice."""
    import torch
    per = np.linspace(1, 41, 40)
    H, VP, VS, RHO, nl = draw_models(8192, 10, seed=3000)
    eng = _engine([r[0] for r in REFS], per, rf=True)
    out, err = eng.run(H, VP, VS, RHO, nl)
    out2, err2 = eng.run(H, VP, VS, RHO, nl)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(err, err2)          # deterministic / idempotent
    o, e = out.cpu().numpy(), err.cpu().numpy()
    assert e.sum() == 0                                               # sorted-Vs models always solve
    vsmin, vsmax = VS[:, 0], VS[:, -1]
    for t in range(4):                                                # physical range of the roots
        c = o[:, eng.slices[t]]
        if t in (0, 2):   # phase velocity: between the search floor 0.95*0.9*c_R(min layer) and max Vs
            assert np.all(c > 0.8 * 0.9 * vsmin[:, None]) and np.all(c <= vsmax[:, None] * 1.0001)
        else:             # group velocity (fp32 finite difference) has no simple bound: finite only
            assert np.all(np.isfinite(c))
    assert np.all(o[:, eng.slices[0]] == o[:, eng.slices[0]].astype(np.float32))  # dble(sngl(c))
    # permutation equivariance: a model's result does not depend on its slot / neighbours
    perm = np.random.RandomState(1).permutation(8192)
    outp, _ = eng.run(H[perm], VP[perm], VS[perm], RHO[perm], nl[perm])
    assert np.array_equal(outp.cpu().numpy(), o[perm], equal_nan=True)
    sl = slice(4000, 4064)
    for t, (name, iw, ig) in enumerate(REFS):
        want, werr, _ = oracle.swd_batch(H[sl], VP[sl], VS[sl], RHO[sl], nl[sl], per, iw, ig)
        _check_swd(name, o[sl, eng.slices[t]], want, e[sl, t], werr, monotone=True)
    assert np.abs(o[sl, eng.slices[4]] - oracle.rf_batch(H[sl], VP[sl], VS[sl], RHO[sl], nl[sl])).max() <= TOL_RF


def test_headline_size_order_invariance(lib, oracle):
    """The bench workload at its full size (524 288 ten-layer models, Rayleigh phase + P-RF): the
    processing order (work queue in sorted order) changes no bit of any row, every model solves, and
    a slice agrees with the oracle."""
    import torch
    from bayhunter_amd.engine import DeviceModels
    B = 524288
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(B, 10, seed=6000)
    eng = _engine(['rdispph'], per, rf=True)
    models = eng.upload(H, VP, VS, RHO, nl)
    assert models.order is not None
    out, err = eng.run(models)
    eng.sort_ragged = False                                            # no order: queue in row order
    out2, err2 = eng.run(DeviceModels(models.packed, models.nlay))
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(err, err2)
    assert int(err.sum().item()) == 0
    del out2, err2
    sl = slice(300000, 300064)
    o = out[sl].cpu().numpy()
    want, werr, _ = oracle.swd_batch(H[sl], VP[sl], VS[sl], RHO[sl], nl[sl], per, 2, 0)
    assert np.array_equal(o[:, :21], want)                             # monotone models: bit-identical
    assert np.abs(o[:, 21:] - oracle.rf_batch(H[sl], VP[sl], VS[sl], RHO[sl], nl[sl])).max() <= TOL_RF


@pytest.fixture(params=['team', 'team32', 'team16', 'team8', 'team128', 'team256', 'team512'])
def team_mode(lib, request):
    from bayhunter_amd import _lib
    _lib.set_swd_kernel(request.param)
    yield
    _lib.set_swd_kernel('auto')


@pytest.mark.parametrize('tag', ['L2_sorted', 'L10_sorted', 'L10_lvz', 'L31_lvz', 'ragged'])
def test_team_kernel_golden_sets(lib, golden, team_mode, tag):
    """The latency kernels (64, 32 or 16 lanes per search) against the golden vectors, same
    tolerances."""
    g = golden['swd_rf_random']
    H, VP, VS, RHO = g[tag + '_model']
    nl = g['ragged_nlay'] if tag == 'ragged' else _nlay(g[tag + '_model'])
    eng = _engine([r[0] for r in REFS], g['periods'])
    out, err = eng.run(H, VP, VS, RHO, nl)
    out, err = out.cpu().numpy(), err.cpu().numpy()
    for t, (name, _, _) in enumerate(REFS):
        _check_swd(name, out[:, eng.slices[t]], g[tag + '_' + name], err[:, t], g[tag + '_' + name + '_err'],
                   monotone=tag.endswith('sorted') or tag == 'ragged')


def test_team_and_lane_kernels_agree_bitwise(lib, oracle):
    """Both kernels run the same arithmetic per search -> identical bits, incl. modes, flsph, the
    failure path, 1-layer and 100-layer models."""
    from bayhunter_amd import _lib
    per = np.linspace(1, 41, 21)
    cases = [(draw_models(257, (1, 12), seed=11, sorted_vs=False), dict()),
             (draw_models(64, 10, seed=12), dict(mode=2)),
             (draw_models(64, 7, seed=13), dict(mode=3, flsph=1)),
             (draw_models(6, 100, seed=14, zmax=300.0, thickmin=0.05), dict())]
    for (H, VP, VS, RHO, nl), kw in cases:
        res = {}
        for mode in ('lane', 'team', 'team32', 'team16', 'team8', 'team128', 'team256', 'team512'):
            _lib.set_swd_kernel(mode)
            try:
                eng = _engine([r[0] for r in REFS], per, **kw)
                out, err = eng.run(H, VP, VS, RHO, nl)
                res[mode] = (out.cpu().numpy(), err.cpu().numpy())
            finally:
                _lib.set_swd_kernel('auto')
        for mode in ('team', 'team32', 'team16', 'team8', 'team128', 'team256', 'team512'):
            assert np.array_equal(res['lane'][1], res[mode][1]), mode
            assert np.array_equal(res['lane'][0], res[mode][0]), mode
        if not kw:
            want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0)
            assert np.array_equal(res['team'][1][:, 0], werr)


def test_work_queue_path(lib, oracle):
    """The persistent-lane work queue only engages above ~4e5 searches per call; force it with a
    tiny resident-wave budget (test hook BH_SWD_RESIDENT_WAVES) in a child process and compare with
    the oracle: every lane then runs many searches back to back, on ragged models."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models
H, VP, VS, RHO, nl = draw_models(3000, (2, 9), seed=21, sorted_vs=False)
per = np.linspace(1, 41, 21)
_lib.set_swd_kernel('lane')
eng = ForwardEngine(swd=[SwdSpec('rdispph', per), SwdSpec('ldispgr', per)])
out, err = eng.run(H, VP, VS, RHO, nl)
np.savez(sys.argv[1], out=out.cpu().numpy(), err=err.cpu().numpy())
""" % ROOT
    path = os.path.join(ROOT, 'gpurun_out', 'queue_test.npz')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    env = dict(os.environ, BH_SWD_RESIDENT_WAVES='6')      # 3 waves per target -> ~16 searches per lane
    subprocess.run([sys.executable, '-c', code, path], check=True, env=env)
    got = np.load(path)
    H, VP, VS, RHO, nl = draw_models(3000, (2, 9), seed=21, sorted_vs=False)
    per = np.linspace(1, 41, 21)
    for t, (iw, ig) in enumerate(((2, 0), (1, 1))):
        want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=8)
        assert np.array_equal(got['err'][:, t], werr)
        ok = werr == 0
        d = np.abs(got['out'][ok][:, 21 * t:21 * (t + 1)] - want[ok])
        assert d.max() <= (TOL_GROUP if ig else TOL_PHASE)
        assert (d == 0).mean() >= 0.97


def test_engine_corner_cases(lib, oracle):
    """RF-only engine, empty batch, nsamp 4096 with 100 layers (one model per workgroup, 102 KB of
    LDS in the SWD kernel), SV receiver function through the batched path."""
    import torch
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
    H, VP, VS, RHO, nl = draw_models(5, 100, seed=31, zmax=300.0, thickmin=0.05)
    x = np.arange(1500) / 20.0 - 5.0                               # 1500 samples @ 20 Hz -> nsamp 4096
    eng = ForwardEngine(rf=[RfSpec('srf', x, gauss=2.0, p=5.0)])
    assert int(eng.rf[0].nsamp) == 4096
    out, err = eng.run(H, VP, VS, RHO, nl)
    want = oracle.rf_batch(H, VP, VS, RHO, nl, p=5.0, gauss=2.0, nsamp=4096, fsamp=20.0, tshift=5.0,
                           waveno=1, nout=1500)
    assert np.abs(out.cpu().numpy() - want).max() <= TOL_RF * max(1.0, np.abs(want).max())
    assert int(err.sum().item()) == 0
    # empty batch
    eng2 = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, 21))], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    out, err = eng2.run(np.zeros((0, 4)), np.zeros((0, 4)), np.zeros((0, 4)), np.zeros((0, 4)), np.zeros(0, dtype=np.int32))
    torch.cuda.synchronize()
    assert out.shape == (0, 222) and err.shape == (0, 1)
    # model on the wrong device / bad shapes are host errors, not kernel faults
    with pytest.raises(Exception):
        eng2.run(np.zeros((3, 101)), np.zeros((3, 101)), np.zeros((3, 101)), np.zeros((3, 101)),
                 np.full(3, 101, dtype=np.int32))                     # more than NL = 100 layers


def test_division_selftest(lib):
    """The shared-reciprocal division of the SWD kernels is bit-identical to the IEEE `/` operator:
    2 x 2^26 random quotients at moderate exponents (the period equation's range) and 2 x 2^24 over
    2^-300..2^300."""
    import ctypes as C
    from bayhunter_amd import _lib
    for n, max_exp in ((1 << 26, 40), (1 << 24, 300)):
        bad = C.c_long(-1)
        _lib.check(lib.bh_selftest_division(n, 12345 + max_exp, max_exp, C.byref(bad)))
        assert bad.value == 0, (max_exp, bad.value)


def test_synrf_dropin_returns_all_three_traces(lib, oracle):
    """bh_synrf with fz/fr buffers = the full return value of rfmini.synrf (fz, fr, rf), P and SV;
    one Q for all layers (models 0, 1: the shared-factor form of the recursion) and Q varying
    from layer to layer (models 2, 3: the general form)."""
    from bayhunter_amd import _lib
    H, VP, VS, RHO, nl = draw_models(4, (3, 9), seed=17, sorted_vs=False)
    for b in range(4):
        n = nl[b]
        z = np.ascontiguousarray(np.concatenate(([0], np.cumsum(H[b, :n])[:-1])))
        vp, vs, rh = (np.ascontiguousarray(a[b, :n]) for a in (VP, VS, RHO))
        qp, qs = np.full(n, 450.), np.full(n, 200.)
        if b >= 2:
            qp, qs = qp + 35. * np.arange(n), qs - 12. * np.arange(n)
        for wn, nsamp in ((0, 512), (1, 256)):
            want = oracle.synrf(z, vp, vs, rh, qp, qs, 6.4, 1.5, nsamp, 5.0, 5.0, 3.1, 0.27, wn)
            fz, fr, rf = np.zeros(nsamp), np.zeros(nsamp), np.zeros(nsamp)
            _lib.check(lib.bh_synrf(nsamp, 5.0, 5.0, 6.4, 1.5, 3.1, 0.27, wn, n, z.ctypes.data,
                                    vp.ctypes.data, vs.ctypes.data, rh.ctypes.data, qp.ctypes.data,
                                    qs.ctypes.data, fz.ctypes.data, fr.ctypes.data, rf.ctypes.data))
            for got, w in zip((fz, fr, rf), want):
                assert np.abs(got - w).max() <= TOL_RF * max(1.0, np.abs(w).max())
            rf2 = np.zeros(nsamp)
            _lib.check(lib.bh_synrf(nsamp, 5.0, 5.0, 6.4, 1.5, 3.1, 0.27, wn, n, z.ctypes.data,
                                    vp.ctypes.data, vs.ctypes.data, rh.ctypes.data, qp.ctypes.data,
                                    qs.ctypes.data, None, None, rf2.ctypes.data))
            # (without fz/fr the kernel leaves the unit factor exp(i w t0) out of the spectral ratio, in
            # which it cancels: same trace up to rounding)
            assert np.abs(rf - rf2).max() <= 1e-14 * max(1.0, np.abs(rf).max())


def test_rf_frequency_cutoff_on_resonant_low_q_models(lib, oracle):
    """bh_synrf against the oracle on models with extreme spectral ratios (thin slow surface layer, small
    Q, Gauss factor <= 1.2: tests/rf_extreme.py).  The kernel drops the frequencies whose filter weight is
    below 3e-19, the reference computes all 257: the result must still be within TOL_RF."""
    from bayhunter_amd import _lib
    from rf_extreme import resonant_models
    worst, finite = 0.0, 0
    for m in resonant_models(120):
        want = oracle.synrf(m['z'], m['vp'], m['vs'], m['rho'], m['qp'], m['qs'], m['p'], m['gauss'], 512, 5.0, 5.0,
                            m['vs'][0], m['sigma'], m['waveno'])[2]
        a = [np.ascontiguousarray(m[k]) for k in ('z', 'vp', 'vs', 'rho', 'qp', 'qs')]
        rf = np.zeros(512)
        _lib.check(lib.bh_synrf(512, 5.0, 5.0, m['p'], m['gauss'], m['vs'][0], m['sigma'], m['waveno'], a[0].size,
                                *[x.ctypes.data for x in a], None, None, rf.ctypes.data))
        assert np.array_equal(np.isfinite(want), np.isfinite(rf))
        if np.isfinite(want).all():
            finite += 1
            worst = max(worst, np.abs(rf - want).max() / max(1.0, np.abs(want).max()))
    assert finite >= 90 and worst <= TOL_RF, (finite, worst)


def test_rf_ill_conditioned_models_are_bounded_by_the_references_own_error(lib):
    """The worst models of the random RF campaigns (tests/rf_extreme.py: ill_conditioned_models): the fp64 oracle is
    2e-9 ... 1e-7 from the extended-precision evaluation of the same algorithm; the kernel must be within
    tolerances.rf_bound of the oracle (observed 0.8 - 1.25e-10)."""
    from bayhunter_amd.engine import ForwardEngine, RfSpec
    from oracle import pyoracle as po
    from rf_extreme import ill_conditioned_models, oracle_error
    from tolerances import rf_bound
    for m in ill_conditioned_models():
        want, scale, ref_error = oracle_error(po, m)
        x = np.arange(m['nout']) / m['fsamp'] - m['tshift']
        eng = ForwardEngine(rf=[RfSpec('srf', x, m['gauss'], m['p'], None)])
        a = [np.ascontiguousarray(m[k][None, :]) for k in ('h', 'vp', 'vs', 'rho')]
        out, err = eng.run(*a, np.array([m['h'].size], dtype=np.int32))
        d = np.abs(out.cpu().numpy()[0] - want).max() / scale
        assert int(err.sum()) == 0 and d <= rf_bound(ref_error) and d < 0.2 * ref_error, (d, ref_error)


@pytest.mark.parametrize('case', ['cfg3', 'modes'])
def test_targets_of_one_call_on_different_kernel_forms(lib, case):
    """A latency-bound call with several targets (BASELINE cfg3: 8192 models, Rayleigh + Love, phase + group, 40
    periods) can run its targets on different kernel forms, each on a stream of its own (capi.hip: plan_forms,
    bh_swd_set_forms).  The values must not depend on that: same rows, bit for bit, as with every target on the lane
    kernel -- also for a target with higher modes (its workspace block and err column keep their index)."""
    import ctypes as C
    from bayhunter_amd import _lib
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    per = np.linspace(1, 41, 40)
    if case == 'cfg3':
        specs = [SwdSpec(r, per) for r in ('rdispph', 'rdispgr', 'ldispph', 'ldispgr')]
    else:
        specs = [SwdSpec('rdispph', per[::2]), SwdSpec('rdispgr', per, mode=2), SwdSpec('ldispph', per[::2])]
    H, VP, VS, RHO, nl = draw_models(8192, 10, seed=3100, sorted_vs=(case == 'cfg3'))
    eng = ForwardEngine(swd=specs)
    models = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.run(models)
    forms = (C.c_int * len(specs))()
    _lib.check(lib.bh_swd_last_forms(forms, len(specs)))
    if len(set(forms)) == 1:
        # (with the round-4 table -- narrow teams of one trial per lane -- the planner takes ONE team form for all
        # targets of these calls, eight lanes per search for cfg3; the mixed call, the forms on their own streams and
        # a two-mode target beside them are what this test is about, so it is asked for)
        mixed = ['team8', 'team128', 'lane', 'team8'] if case == 'cfg3' else ['lane', 'team128', 'lane']
        _lib.set_swd_forms(mixed)
        try:
            out, err = eng.run(models)
            _lib.check(lib.bh_swd_last_forms(forms, len(specs)))
        finally:
            _lib.set_swd_forms(None)
        assert list(forms) == ([8, 128, 0, 8] if case == 'cfg3' else [0, 128, 0])
    out, err = out.cpu().numpy(), err.cpu().numpy()
    _lib.set_swd_kernel('lane')
    try:
        want, werr = eng.run(models)
        _lib.check(lib.bh_swd_last_forms(forms, len(specs)))
        assert set(forms) == {0}
    finally:
        _lib.set_swd_kernel('auto')
    assert np.array_equal(out, want.cpu().numpy(), equal_nan=True) and np.array_equal(err, werr.cpu().numpy())
    assert np.count_nonzero(out) > 0.5 * out.size          # (the first higher mode does not exist at every period)


def test_forced_forms_per_target_small_batches(lib):
    """bh_swd_set_forms: three forms in one call (three concurrent launches), ragged models, more targets than
    forms, a multimode target -- the rows of the lane kernel, bit for bit; bad arguments refused."""
    import ctypes as C
    from bayhunter_amd import _lib
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    per = np.linspace(2, 30, 9)
    specs = [SwdSpec('rdispph', per), SwdSpec('ldispgr', per), SwdSpec('rdispgr', per, mode=2), SwdSpec('ldispph', per, flsph=1)]
    eng = ForwardEngine(swd=specs)
    for B, L, seed in ((1, 4, 1), (77, (2, 13), 2), (1500, 7, 3)):
        H, VP, VS, RHO, nl = draw_models(B, L, seed=3200 + seed, sorted_vs=False)
        _lib.set_swd_kernel('lane')
        try:
            want, werr = [x.cpu().numpy() for x in eng.run(H, VP, VS, RHO, nl)]
        finally:
            _lib.set_swd_kernel('auto')
        for assign in (['team8', 'lane', 'team', 'lane'], ['team512', 'team512', 'team16', 'team32'], ['lane', 'team128', 'lane', 'lane']):
            _lib.set_swd_forms(assign)
            try:
                out, err = [x.cpu().numpy() for x in eng.run(H, VP, VS, RHO, nl)]
                forms = (C.c_int * 4)()
                _lib.check(lib.bh_swd_last_forms(forms, 4))
            finally:
                _lib.set_swd_forms(None)
            names = {0: 'lane', 64: 'team'}
            assert [names.get(f, 'team%d' % f) for f in forms] == assign
            assert np.array_equal(out, want, equal_nan=True) and np.array_equal(err, werr), (B, assign)
    bad = (C.c_int * 4)(0, 8, 16, 32)                                   # four different forms
    assert lib.bh_swd_set_forms(bad, 4) != 0 and lib.bh_swd_set_forms((C.c_int * 1)(7), 1) != 0


def test_ragged_batch_is_reordered_transparently(lib, oracle):
    """Above 8192 models the searches are processed deepest first and by S travel time within a
    depth (a permutation handed to bh_swd_batch_ordered); results land in the caller's rows,
    identical to the unordered run, with and without caller-provided buffers, lane and team kernels."""
    import torch
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
    H, VP, VS, RHO, nl = draw_models(9000, (2, 12), seed=41, sorted_vs=False)
    per = np.linspace(1, 41, 11)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', per)], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    models = eng.upload(H, VP, VS, RHO, nl)
    order = models.order.cpu().numpy()
    assert order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(9000))
    assert np.all(np.diff(nl[order]) <= 0)                       # deepest first
    out, err = eng.run(models)
    buf_out, buf_err = eng.alloc_out(9000)
    eng.run(models, out=buf_out, err=buf_err)
    eng.sort_ragged = False
    ref_out, ref_err = eng.run(H, VP, VS, RHO, nl)
    torch.cuda.synchronize()
    for o, e in ((out, err), (buf_out, buf_err)):
        assert torch.equal(o.nan_to_num(nan=-1.0), ref_out.nan_to_num(nan=-1.0)) and torch.equal(e, ref_err)
    from bayhunter_amd import _lib
    for mode in ('lane', 'team16', 'team8'):
        _lib.set_swd_kernel(mode)
        try:
            o, e = eng.run(models)
            torch.cuda.synchronize()
        finally:
            _lib.set_swd_kernel('auto')
        assert torch.equal(o.nan_to_num(nan=-1.0), ref_out.nan_to_num(nan=-1.0)) and torch.equal(e, ref_err), mode
    sl = slice(100, 132)
    want, werr, _ = oracle.swd_batch(H[sl], VP[sl], VS[sl], RHO[sl], nl[sl], per, 2, 0)
    assert np.array_equal(err.cpu().numpy()[sl, 0], werr)
    ok = werr == 0
    assert np.abs(out.cpu().numpy()[sl][ok][:, :11] - want[ok]).max() <= TOL_PHASE


def test_more_launches_in_flight_than_queue_slots(lib, oracle):
    """The work-queue counters live in a ring of slots; a slot is re-used only after the launch that
    used it has finished (event per slot).  600 small launches on two streams with the ring shrunk to
    4 slots (BH_SWD_QUEUE_SLOTS, child process): every launch must still produce the oracle's rows
    (a shared counter would hand models to the wrong launch or skip some)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models
per = np.linspace(1, 41, 7)
_lib.set_swd_kernel(sys.argv[2])
eng = ForwardEngine(swd=[SwdSpec('rdispph', per)])
eng.sort_ragged = False
sets = [eng.upload(*draw_models(200 + 13 * i, (2, 6), seed=300 + i, sorted_vs=False)) for i in range(6)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = []
for it in range(100):
    for i, m in enumerate(sets):
        st = streams[(it + i) %% 2]
        with torch.cuda.stream(st):
            outs.append((i, eng.run(m, stream=st)))
torch.cuda.synchronize()
ref = [None] * 6
bad = 0
for i, (o, e) in outs:
    o, e = o.cpu().numpy(), e.cpu().numpy()
    if ref[i] is None:
        ref[i] = (o, e)
    bad += int(not (np.array_equal(o, ref[i][0]) and np.array_equal(e, ref[i][1])))
np.savez(sys.argv[1], bad=bad, **{'o%%d' %% i: ref[i][0] for i in range(6)}, **{'e%%d' %% i: ref[i][1] for i in range(6)})
""" % ROOT
    per = np.linspace(1, 41, 7)
    for kernel, waves in (('lane', '4'), ('team16', '')):
        path = os.path.join(ROOT, 'gpurun_out', 'slots_test_%s.npz' % kernel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ, BH_SWD_QUEUE_SLOTS='4')
        if waves:
            env['BH_SWD_RESIDENT_WAVES'] = waves                      # the queue path of the lane kernel
        subprocess.run([sys.executable, '-c', code, path, kernel], check=True, env=env, timeout=600)
        got = np.load(path)
        assert int(got['bad']) == 0, kernel
        for i in range(6):
            H, VP, VS, RHO, nl = draw_models(200 + 13 * i, (2, 6), seed=300 + i, sorted_vs=False)
            want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0)
            assert np.array_equal(got['e%d' % i][:, 0], werr), (kernel, i)
            ok = werr == 0
            assert np.abs(got['o%d' % i][ok] - want[ok]).max() <= TOL_PHASE, (kernel, i)


def test_model_deeper_than_the_launch_is_flagged_not_truncated(lib, oracle):
    """nlay outside 1..Lmax (here: a depth hint that understates three models, and a zero): the row
    is NaN and the flag 2 (BH_MODEL_BAD_DEPTH) in every kernel form, the likelihood of such a model is
    the failure value, and the other models of the batch are untouched."""
    import torch
    from bayhunter_amd import _lib
    from bayhunter_amd.engine import DeviceModels, ForwardEngine, RfSpec, SwdSpec
    per = np.linspace(1, 41, 21)
    H, VP, VS, RHO, nl = draw_models(300, (2, 6), seed=61, Lmax=8)
    bad = np.array([7, 130, 299])
    nl2 = nl.copy()
    nl2[bad[:2]] = 8                                     # deeper than the hint below
    nl2[bad[2]] = 0
    eng = ForwardEngine(swd=[SwdSpec('rdispph', per), SwdSpec('ldispgr', per)], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    m = eng.upload(H, VP, VS, RHO, nl2)
    hinted = DeviceModels(m.packed, m.nlay, depth=6)     # kernels size their LDS for 6 layers
    want, werr, _ = oracle.swd_batch(H, VP, VS, RHO, nl, per, 2, 0)
    good = np.setdiff1d(np.arange(300), bad)
    for mode in ('lane', 'team', 'team256', 'team16', 'team8'):
        _lib.set_swd_kernel(mode)
        try:
            out, err = eng.run(hinted)
            torch.cuda.synchronize()
        finally:
            _lib.set_swd_kernel('auto')
        out, err = out.cpu().numpy(), err.cpu().numpy()
        assert np.all(err[bad] == 2) and np.all(np.isnan(out[bad])), mode
        assert np.array_equal(err[good, 0], werr[good]) and np.array_equal(out[good, :21][werr[good] == 0], want[good][werr[good] == 0]), mode
    # a receiver-function-only engine has no dispersion kernel to raise the flag: same contract
    rf_only = ForwardEngine(rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    out, err = rf_only.run(hinted)
    torch.cuda.synchronize()
    out, err = out.cpu().numpy(), err.cpu().numpy()
    assert err.shape == (300, 1) and np.all(err[bad, 0] == 2) and np.all(err[good] == 0)
    assert np.all(np.isnan(out[bad])) and not np.isnan(out[good]).any()


@pytest.mark.parametrize('kernel', ['lane', 'team', 'team256', 'team16'])
def test_nan_models_end_in_every_kernel_form(lib, kernel):
    """A NaN model: one with a NaN top layer (finite start value, NaN period equation) and one that is
    NaN throughout (NaN start value, so every trial velocity is NaN and equals nothing).  The reference
    would scan forever on the second; every kernel form must run into the bracketing step cap, report
    "no root" (err = 1, zero row), and leave the neighbours in the batch untouched."""
    from bayhunter_amd import _lib
    per = np.linspace(1, 41, 5)
    H, VP, VS, RHO, nl = draw_models(6, 4, seed=71)
    clean = [a.copy() for a in (H, VP, VS, RHO)]
    VS[2, 0] = np.nan; VP[2, 0] = np.nan
    VS[4, :] = np.nan; VP[4, :] = np.nan; RHO[4, :] = np.nan
    eng = _engine(['rdispph', 'ldispph'], per)
    _lib.set_swd_kernel(kernel)
    try:
        out, err = eng.run(H, VP, VS, RHO, nl)
        ref, rerr = eng.run(*clean, nl)
        out, err, ref, rerr = (x.cpu().numpy() for x in (out, err, ref, rerr))
    finally:
        _lib.set_swd_kernel('auto')
    for b in (2, 4):                 # "no root" (zero row, err 1) or NaN values: never a plausible curve
        for t in (0, 1):
            row = out[b, 5 * t:5 * t + 5]
            assert (err[b, t] == 1 and np.all(row == 0.0)) or np.isnan(row).any(), (b, t, row, err[b])
    assert np.all(err[4] == 1) and np.all(out[4] == 0.0)          # all-NaN model: the step cap ends the scan
    ok = [0, 1, 3, 5]
    assert np.array_equal(out[ok], ref[ok]) and np.array_equal(err[ok], rerr[ok])


@pytest.mark.parametrize('Lmax,B,packed', [(5, 1000, True), (10, 4099, True), (31, 700, False), (100, 130, True), (2, 65, False)])
def test_order_keys_staged_through_lds_equal_the_per_model_formula(lib, Lmax, B, packed):
    """The processing-order keys (like_kernel.hip: order_key_kernel; a wave stages the rows of its 64 models in LDS
    with coalesced loads since round 4) against the formula evaluated per model in numpy: depth field exact, travel
    time to one unit of 1/256 s, length class to one class (the device uses its fast exponential) -- for packed and
    separate arrays, row lengths that divide 64 and that do not, more layers than lanes, ragged and partial tiles."""
    import ctypes as C
    import torch
    from bayhunter_amd import _lib
    kw = dict(zmax=600.0, thickmin=0.01) if Lmax > 40 else {}
    H, VP, VS, RHO, nl = draw_models(B, (1, Lmax) if Lmax > 2 else 2, seed=90 + Lmax, sorted_vs=False, Lmax=Lmax, **kw)
    VS = VS.copy()
    VS[::7, 0] = 0.0                                                     # some water-covered models
    dev = torch.device('cuda')
    if packed:
        blk = torch.from_numpy(np.ascontiguousarray(np.stack([H, VP, VS, RHO], axis=1))).to(dev)
        hp, vp_, stride = blk[:, 0, :].data_ptr(), blk[:, 2, :].data_ptr(), 4 * Lmax
    else:
        th, tv = torch.from_numpy(np.ascontiguousarray(H)).to(dev), torch.from_numpy(np.ascontiguousarray(VS)).to(dev)
        hp, vp_, stride = th.data_ptr(), tv.data_ptr(), Lmax
    tn = torch.from_numpy(nl.astype(np.int32)).to(dev)
    keys = torch.empty(B, dtype=torch.int32, device=dev)
    tmax = 41.0
    _lib.check(lib.bh_swd_order_keys(B, Lmax, stride, tn.data_ptr(), hp, vp_, tmax, 1, keys.data_ptr(), None))
    torch.cuda.synchronize()
    got = keys.cpu().numpy()
    f = np.float32
    reach = f(0.35 * 3.5 * tmax)
    for b in range(B):
        n = int(nl[b])
        tt = z = sw = svw = f(0)
        vmin = f(1e9)
        for i in range(n):
            hi, vi = f(H[b, i]), f(VS[b, i])
            if not vi > 0:
                z = f(z + hi)
                continue
            tt = f(tt + hi / vi)
            hs = i == n - 1
            zmid = f(z + f(10)) if hs else f(z + f(0.5) * hi)
            w = f(np.exp(-zmid / reach) * (reach if hs else hi))
            sw, svw, vmin, z = f(sw + w), f(svw + vi * w), min(vmin, vi), f(z + hi)
        cls = int(min(max(np.floor((f(6.0) - (f(0.92) * svw / sw - f(0.79) * vmin)) / f(0.15)), 0), 63)) if sw > 0 else 0
        t18 = int(min(max(tt * f(256), 0), 262143))
        k = int(got[b])
        assert k >> 24 == 127 - n, (b, n)
        assert abs(((k >> 18) & 63) - cls) <= 1 and abs((k & 0x3ffff) - t18) <= 1, (b, k, cls, t18)


def test_frequency_table_cache_is_bounded(lib):
    """the per-(nsamp, fsamp, gauss, tshift) tables of the receiver-function kernel were cached for the
    life of the process -- a caller of the single-model drop-in that varies the Gauss width or the time shift per call
    grew the cache without bound.  It is a least-recently-used cache of 64 tables now; an evicted parameter set is
    rebuilt on its next use and gives the same trace."""
    from bayhunter_amd import _lib
    H, VP, VS, RHO, nl = draw_models(1, 5, seed=23)
    n = int(nl[0])
    z = np.ascontiguousarray(np.concatenate(([0], np.cumsum(H[0, :n])[:-1])))
    vp, vs, rh = (np.ascontiguousarray(a[0, :n]) for a in (VP, VS, RHO))
    qp, qs = np.full(n, 450.), np.full(n, 200.)

    def trace(gauss, tshift):
        rf = np.zeros(64)
        _lib.check(lib.bh_synrf(64, 5.0, tshift, 6.4, gauss, -1.0, float('nan'), 0, n, z.ctypes.data, vp.ctypes.data,
                                vs.ctypes.data, rh.ctypes.data, qp.ctypes.data, qs.ctypes.data, None, None, rf.ctypes.data))
        return rf
    first = trace(0.8, 5.0)
    for k in range(150):
        trace(0.8 + 0.001 * (k + 1), 5.0 + 0.01 * k)
    assert 1 <= lib.bh_rf_cached_tables() <= 64
    again = trace(0.8, 5.0)                               # long evicted: rebuilt
    assert np.array_equal(first, again) and np.isfinite(first).all() and lib.bh_rf_cached_tables() <= 64
