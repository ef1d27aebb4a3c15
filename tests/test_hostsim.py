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


def test_swd_nan_model_terminates(hostsim):
    """A NaN model must not spin forever (the reference would); it ends as 'no root'."""
    h = np.array([5., 10., 0.])
    vs = np.array([3.0, np.nan, 4.5])
    cg, err, ncalls = hostsim.swd(h, vs * 1.73, vs, vs * 1.73 * .32 + .77, np.linspace(1, 41, 5), 2, 0)
    assert err == 1 and np.all(cg == 0.0) and ncalls <= 100003


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
