"""CPU tier: the device solver cores (bayhunter_amd/csrc/*_core.h) compiled with g++ and replayed
lane by lane, against the oracle.  Proves the kernels' control-flow transformation (one period-
equation call site, task-parallel reflectivity, LDS FFT) without a GPU: with the host libm the SWD
search is bit-identical (values, err flags and the count of period-equation evaluations)."""
import numpy as np
import pytest

from bayhunter_amd.synthetic import draw_models
from conftest import REFS


@pytest.mark.parametrize('L,srt', [(2, True), (5, True), (10, True), (10, False), (31, False)])
def test_swd_state_machine_bitexact(oracle, hostsim, L, srt):
    H, VP, VS, RHO, nl = draw_models(12, L, seed=500 + L + int(srt), sorted_vs=srt)
    per = np.linspace(1, 41, 21)
    for name, iw, ig in REFS:
        for mode, fl in ((1, 0), (2, 0), (3, 1)):
            for b in range(12):
                n = nl[b]
                a, e1, n1 = oracle.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig,
                                       mode, fl, count=True)
                r, e2, n2 = hostsim.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig, mode, fl)
                assert e1 == e2 and n1 == n2
                assert np.array_equal(a, r)


@pytest.mark.parametrize('L,srt', [(1, True), (2, True), (5, True), (10, True), (10, False), (31, False), (80, True)])
def test_swd_team_replay_bitexact(oracle, hostsim, L, srt):
    """swd_team.h (speculative bracketing + layer-parallel assembly): same values, err flags and
    number of CONSUMED period-equation evaluations as the reference; far fewer rounds."""
    kw = dict(zmax=300.0, thickmin=0.05) if L > 40 else {}
    H, VP, VS, RHO, nl = draw_models(6, L, seed=700 + L + int(srt), sorted_vs=srt, **kw)
    per = np.linspace(1, 41, 21)
    tot_calls = tot_rounds = 0
    for name, iw, ig in REFS:
        for mode, fl in ((1, 0), (2, 0), (3, 1)):
            for b in range(6):
                n = nl[b]
                a, e1, n1 = oracle.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig,
                                       mode, fl, count=True)
                for nlanes in (64, 32, 16, 8, 7):   # the four kernel widths + an odd one
                    r, e2, n2, nspec, nrounds = hostsim.swd_team(H[b, :n], VP[b, :n], VS[b, :n],
                                                                  RHO[b, :n], per, iw, ig, mode, fl, nlanes)
                    assert e1 == e2 and n1 == n2 and nspec >= n2
                    assert np.array_equal(a, r)
                    if nlanes == 64:
                        tot_calls += n2
                        tot_rounds += nrounds
    if L <= 10:
        assert tot_rounds < 0.62 * tot_calls      # speculation pays when >= 6 trials fit a round


@pytest.mark.parametrize('L,srt', [(1, True), (2, True), (5, True), (10, False), (15, True), (31, False), (80, True)])
def test_swd_wide_team_replay_bitexact(oracle, hostsim, L, srt):
    """Wide teams (swd_teamw_*: slot layout, bisection candidates, speculation across the end of a root
    search, value-matched consumption with run fast-forward, Neville tables in memory): same values,
    err flags and number of CONSUMED evaluations as the reference for 64/128/256 lanes -- and for the plan with 8, 16,
    32 and 64 slots per round, which is what the teams of one trial per lane run (kernels.hip: swd_tpl_body and the
    one-wave team; nlanes <= 64 in the replay); the replay itself checks the plan's invariants.  Rounds per period
    roughly halve against the round-1 scheme."""
    kw = dict(zmax=300.0, thickmin=0.05) if L > 40 else {}
    H, VP, VS, RHO, nl = draw_models(5, L, seed=1700 + L + int(srt), sorted_vs=srt, **kw)
    per = np.linspace(1, 41, 21)
    new = old = 0
    for name, iw, ig in REFS:
        for mode, fl in ((1, 0), (2, 0), (3, 1)):
            for b in range(5):
                n = nl[b]
                a, e1, n1 = oracle.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig, mode, fl, count=True)
                for nlanes in (8, 16, 32, 64, 128, 256):
                    r, e2, n2, nspec, nrounds = hostsim.swd_team(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per,
                                                                  iw, ig, mode, fl, nlanes, wide=True)
                    assert e2 >= 0, 'plan invariant %d violated' % e2
                    assert e1 == e2 and n1 == n2 and nspec >= n2
                    assert np.array_equal(a, r)
                    if nlanes == 64 and mode == 1 and fl == 0:
                        new += nrounds
                        old += hostsim.swd_team(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig, 1, 0, 64)[4]
    if 2 <= L <= 5:
        assert new < 0.6 * old


def test_swd_nan_model_terminates(hostsim):
    """A NaN model must not spin forever (the reference would); it ends as 'no root'."""
    h = np.array([5., 10., 0.])
    vs = np.array([3.0, np.nan, 4.5])
    cg, err, ncalls = hostsim.swd(h, vs * 1.73, vs, vs * 1.73 * .32 + .77, np.linspace(1, 41, 5), 2, 0)
    assert err == 1 and np.all(cg == 0.0) and ncalls <= 100003
    # NaN throughout: the start value and every trial velocity are NaN (equal to nothing); the wide-team
    # replay must still consume its pending evaluation every round and run into the step cap
    nanv = np.full(3, np.nan)
    for wide, nlanes in ((False, 64), (True, 64), (True, 256)):
        cg, err, ncalls, nspec, nrounds = hostsim.swd_team(h, nanv, nanv, nanv, np.linspace(1, 41, 3), 2, 0,
                                                           nlanes=nlanes, wide=wide)
        assert err == 1 and np.all(cg == 0.0) and 100000 <= ncalls <= 100003, (wide, nlanes, err, ncalls)


@pytest.mark.parametrize('wn', [0, 1])
def test_rf_workgroup_program(oracle, hostsim, wn):
    H, VP, VS, RHO, nl = draw_models(10, (2, 15), seed=600 + wn, sorted_vs=False)
    for b in range(10):
        n = nl[b]
        for gauss, p, nsamp, nsv in ((1.0, 6.4, 512, None), (2.5, 4.0, 256, 3.0)):
            a = oracle.rf_model(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], p, gauss, nsamp, 5.0,
                                5.0, nsv, wn, 100)
            r = hostsim.rf(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], p, gauss, nsamp, 5.0, 5.0,
                           nsv, wn, 100)
            assert np.abs(a - r).max() <= 1e-12 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize('nsamp', [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_rf_every_transform_length(oracle, hostsim, nsamp):
    """All power-of-two lengths: the swizzled FFT buffer (rf_swz) keeps the spectrum clear of the
    parameter region during phase 3 (the replay checks that and returns NaN otherwise) and the
    transform equals the reference's."""
    H, VP, VS, RHO, nl = draw_models(3, (2, 12), seed=640 + nsamp, sorted_vs=False)
    for b in range(3):
        n = nl[b]
        a = oracle.rf_model(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], 6.0, 1.3, nsamp, 10.0, 2.0, None, 0, nsamp // 2)
        r = hostsim.rf(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], 6.0, 1.3, nsamp, 10.0, 2.0, None, 0, nsamp // 2)
        assert np.all(np.isfinite(r)) and np.abs(a - r).max() <= 1e-12 * max(1.0, np.abs(a).max())


def test_scan_cells_in_closed_form_equal_repeated_addition(hostsim):
    """swd_scan_cell gives cell i of a bracketing scan as base + i dc, base + (i+1) dc when the scan stays
    inside one binade, where the reference's repeated addition of dc = dble(0.005) (surfdisp96.f:448-452)
    does not round; the bits must be those of the repeated addition for every start value, also across
    powers of two, for fp32-valued starts (the first period's) and for values far outside any model."""
    rs = np.random.RandomState(3)
    n = 400000
    base = np.concatenate([rs.uniform(0.05, 9.0, n), rs.uniform(0.3, 5.3, n).astype(np.float32).astype(np.float64),
                           rs.choice([0.5, 1., 2., 4., 8.], n) - 0.35 * rs.rand(n) ** 2,
                           np.ldexp(rs.rand(n), rs.randint(-30, 10, n)), [0.0, -1.0, np.nan, np.inf]])
    cell = rs.randint(0, 64, base.size)
    dc = np.float64(np.float32(0.005))
    x, c = base.copy(), base + dc
    for k in range(1, 64):
        step = cell >= k
        x = np.where(step, c, x)
        c = np.where(step, x + dc, c)
    b, cn = hostsim.scan_cell(base, cell)
    assert np.array_equal(b, x, equal_nan=True) and np.array_equal(cn, c, equal_nan=True)


def test_rf_frequency_cutoff_on_resonant_low_q_models(oracle, hostsim):
    """The kernel zero-fills the frequencies whose Gauss-filter weight is below 3e-19 (rf_host.h: 213 of
    257 at a = 1, 5 Hz); the reference computes all of them.  What is dropped is cutoff x |R/Z| per bin,
    so the models to fear are those with large deconvolved spectral ratios at high frequency: a thin,
    very slow surface layer, small Q, a Gauss factor at or below 1.  400 such models through the
    replay of the device program against the oracle (which computes every frequency): the deviation
    stays at rounding level (observed 2.9e-15), five orders below TOL_RF."""
    from rf_extreme import resonant_models
    worst, finite = 0.0, 0
    for m in resonant_models(400):
        want = oracle.synrf(m['z'], m['vp'], m['vs'], m['rho'], m['qp'], m['qs'], m['p'], m['gauss'], 512, 5.0, 5.0,
                            m['vs'][0], m['sigma'], m['waveno'])[2]
        got = hostsim.rf(m['h'], m['vp'], m['vs'], m['rho'], m['p'], m['gauss'], 512, 5.0, 5.0, None, m['waveno'], 512,
                         qp=m['qp'], qs=m['qs'])
        assert np.array_equal(np.isfinite(want), np.isfinite(got))
        if np.isfinite(want).all():
            finite += 1
            worst = max(worst, np.abs(got - want).max() / max(1.0, np.abs(want).max()))
    assert finite >= 300 and worst <= 1e-13, (finite, worst)


def test_rf_ill_conditioned_models_are_bounded_by_the_references_own_error(hostsim, hostsim_devmath):
    """tests/rf_extreme.py: ill_conditioned_models -- the models of the random campaigns where TOL_RF is not the
    right yardstick: the fp64 oracle is 2e-9 ... 1e-7 from the extended-precision evaluation of the same algorithm
    (hp_oracle), a typical model 1e-12.  Both replays (glibc math, device math) stay within tolerances.rf_bound,
    i.e. far closer to the oracle than the oracle is to the exact trace."""
    from oracle import pyoracle as po
    from rf_extreme import ill_conditioned_models, oracle_error
    from tolerances import rf_bound
    for m in ill_conditioned_models():
        want, scale, ref_error = oracle_error(po, m)
        assert 1e-9 < ref_error < 1e-6, ref_error
        for hs in (hostsim, hostsim_devmath):
            got = hs.rf(m['h'], m['vp'], m['vs'], m['rho'], m['p'], m['gauss'], m['nsamp'], m['fsamp'], m['tshift'], None,
                        m['waveno'], m['nout'])
            d = np.abs(got - want).max() / scale
            assert d <= rf_bound(ref_error) and d < 0.2 * ref_error, (d, ref_error)
    # and a well-conditioned model for contrast: the oracle within 1e-11 of the extended-precision trace
    m = dict(h=np.array([2., 8., 25., 0.]), vp=np.array([3.5, 6.0, 6.8, 8.1]), vs=np.array([2.0, 3.5, 3.9, 4.5]),
             rho=np.array([2.3, 2.7, 2.9, 3.3]), gauss=1.0, p=6.4, waveno=0, nsamp=512, fsamp=5.0, tshift=5.0, nout=201)
    assert oracle_error(po, m)[2] < 1e-11


def _ulp_err(got, x, fn):
    ref = fn(x.astype(np.longdouble))
    u = np.spacing(np.abs(ref.astype(np.float64)))
    return np.abs((got.astype(np.longdouble) - ref) / u).astype(np.float64)


def test_math_accuracy(hostsim_devmath):
    """bh_math.h: < 1 ulp over the argument ranges the solvers produce, sane special values."""
    rs = np.random.RandomState(0)
    for lo, hi, bound in ((-1, 1, 0.85), (-300, 300, 0.85), (-1e4, 1e4, 1.1)):
        x = rs.uniform(lo, hi, 400000)
        s, c = hostsim_devmath.sincos(x)
        assert _ulp_err(s, x, np.sin).max() <= bound and _ulp_err(c, x, np.cos).max() <= bound
    # 1e4 .. 1e12: the exact-product reduction (no device-library call); also next to multiples of
    # pi/2, where the reduced argument cancels almost completely
    for lo, hi in ((1e4, 1.6e6), (1.6e6, 1e9), (1e9, 0.999e12)):
        x = rs.uniform(lo, hi, 300000) * rs.choice([-1.0, 1.0], 300000)
        k = np.round(x[:100000] / (np.pi / 2))
        x[:100000] = np.nextafter(k * (np.pi / 2), np.inf) + k * 6.123233995736766e-17 * rs.choice([0, 1], 100000)
        s, c = hostsim_devmath.sincos(x)
        assert _ulp_err(s, x, np.sin).max() <= 1.1 and _ulp_err(c, x, np.cos).max() <= 1.1, (lo, hi)
    x = np.array([0.0, 1e-300, 1e7, -1e9, np.pi / 2, 9.99e11])
    s, c = hostsim_devmath.sincos(x)
    assert np.allclose(s, np.sin(x), rtol=0, atol=1e-15) and np.allclose(c, np.cos(x), rtol=0, atol=1e-15)
    assert all(np.isnan(v).all() for v in hostsim_devmath.sincos(np.array([np.inf, np.nan, 1e12, -3e300])))
    for lo, hi in ((-60, 0), (-700, 700)):
        x = rs.uniform(lo, hi, 400000)
        assert _ulp_err(hostsim_devmath.exp(x), x, np.exp).max() <= 0.95
    with np.errstate(over='ignore'):
        x = np.array([0.0, -0.0, -745.5, -800, 710, 800, np.inf, -np.inf])
        assert np.array_equal(hostsim_devmath.exp(x), np.exp(x))
    assert np.isnan(hostsim_devmath.exp(np.array([np.nan]))[0])


def test_swd_device_math_parity(oracle, hostsim_devmath):
    """With the device math the search still lands on the reference's values: bit-identical for
    monotone models, inside the reference's 1e-6 stopping bracket for LVZ models."""
    per = np.linspace(1, 41, 21)
    for srt in (True, False):
        H, VP, VS, RHO, nl = draw_models(40, 8, seed=900 + int(srt), sorted_vs=srt)
        same = tot = 0
        for name, iw, ig in REFS:
            for b in range(40):
                a, e1 = oracle.swd(H[b], VP[b], VS[b], RHO[b], per, iw, ig)
                r, e2, _ = hostsim_devmath.swd(H[b], VP[b], VS[b], RHO[b], per, iw, ig)
                assert e1 == e2
                tol = 1.5e-3 if ig else 1.5e-6 * 5
                assert np.abs(a - r).max() <= tol
                same += int((a == r).sum())
                tot += a.size
        assert same / tot >= (0.999 if srt else 0.98)
