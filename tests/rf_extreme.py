"""Receiver-function models with extreme spectral ratios: a thin, very slow surface layer (strong
reverberations -> large |R/Z| at high frequency), small Q, Gauss factors around the value (1.21) below
which the kernel zero-fills the frequencies whose filter weight is under 3e-19 (rf_host.h) -- the
reference computes every frequency.  Used by the CPU replay (test_hostsim.py) and the GPU tier."""
import numpy as np


def resonant_models(count, seed=5):
    rs = np.random.RandomState(seed)
    for _ in range(count):
        n = rs.randint(2, 8)
        h = np.concatenate((rs.uniform(0.05, 0.6, 1), rs.uniform(0.5, 15, max(0, n - 2)), [0.]))
        vs = np.concatenate(([rs.uniform(0.2, 1.0)], np.sort(rs.uniform(2.5, 4.8, n - 1))))
        vp = vs * rs.uniform(1.6, 2.5)
        rho = 0.77 + 0.32 * vp
        qs = float(rs.choice([5., 10., 25., 225.]))
        yield dict(h=h, vp=vp, vs=vs, rho=rho, qp=np.full(n, 2 * qs), qs=np.full(n, qs),
                   z=np.concatenate(([0], np.cumsum(h)[:-1])), gauss=float(rs.choice([0.8, 1.0, 1.2])),
                   p=float(rs.uniform(4, 8)), waveno=int(rs.randint(0, 2)),
                   sigma=float((2 - (vp[0] / vs[0]) ** 2) / (2 - 2 * (vp[0] / vs[0]) ** 2)))


def ill_conditioned_model():
    """The worst model of the round-3 random campaign (tests/scenarios/rf_fuzz.py, seed 79, configuration
    4219, model 112 of 1000): 20 layers with strong velocity inversions, SV incidence, p = 6.71 s/deg,
    a = 2.56.  A change of the slowness by ONE ulp moves the oracle's own trace by 6.6e-11 of its scale
    (typical models: 2e-15), so no evaluation order can be expected closer than that to another."""
    from bayhunter_amd.synthetic import draw_models
    H, VP, VS, RHO, nl = draw_models(1000, 20, seed=1061471874, sorted_vs=False)
    return dict(h=H[112], vp=VP[112], vs=VS[112], rho=RHO[112], gauss=2.5578946714018787, p=6.711044087493648,
                waveno=1, nsamp=512, fsamp=5.0, tshift=2.0, nout=201)


def oracle_spread(po, m):
    """How far the oracle's trace moves under +-1 ulp of the slowness, relative to the trace's scale."""
    a = [np.ascontiguousarray(m[k][None, :]) for k in ('h', 'vp', 'vs', 'rho')]
    nl = np.array([m['h'].size], dtype=np.int32)
    run = lambda p: po.rf_batch(*a, nl, p, m['gauss'], m['nsamp'], m['fsamp'], m['tshift'], None, m['waveno'],
                                nout=m['nout'], nthreads=1)[0]
    want = run(m['p'])
    scale = max(1.0, np.abs(want).max())
    return want, scale, max(np.abs(run(np.nextafter(m['p'], s)) - want).max() / scale for s in (0.0, 99.0))
