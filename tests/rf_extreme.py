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


def ill_conditioned_models():
    """The two worst models of the round-3 random campaigns (tests/scenarios/rf_fuzz.py), both SV incidence on
    ~20 layers with strong velocity inversions:
      seed 79, configuration 4219, model 112 of 1000 (p = 6.71 s/deg, a = 2.56): device 8.0e-11 from the oracle;
          one ulp of SLOWNESS moves the oracle's own trace by 6.6e-11 of its scale (typical models: 2e-15);
      seed 81, configuration 14547, model 660 of 1000 (p = 7.85, a = 2.92): device 1.25e-10, the CPU replay of the
          device program with glibc math 1.8e-10; one ulp of slowness moves the oracle by 4e-12 only, one ulp of
          the second layer's vs by 1.2e-10.
    No evaluation order can be expected closer to another than the oracle is to itself under such a change."""
    from bayhunter_amd.synthetic import draw_models
    H, VP, VS, RHO, nl = draw_models(1000, 20, seed=1061471874, sorted_vs=False)
    yield dict(h=H[112], vp=VP[112], vs=VS[112], rho=RHO[112], gauss=2.5578946714018787, p=6.711044087493648,
               waveno=1, nsamp=512, fsamp=5.0, tshift=2.0, nout=201)
    H, VP, VS, RHO, nl = draw_models(1000, (21, 27), seed=978415244, sorted_vs=False, zmax=200.0, thickmin=0.05)
    k = int(nl[660])
    yield dict(h=H[660, :k], vp=VP[660, :k], vs=VS[660, :k], rho=RHO[660, :k], gauss=2.921833794173314,
               p=7.846671088611139, waveno=1, nsamp=1024, fsamp=2.0, tshift=2.0, nout=400)


def oracle_spread(po, m, nsv=None):
    """(trace, scale, spread): the oracle's trace for model m and how far it moves, relative to its scale, when ONE
    input -- the slowness, or the thickness, vp, vs or density of one layer -- changes by one ulp (the largest
    such response).  This is the yardstick for a deviation beyond TOL_RF (tolerances.rf_bound)."""
    n = m['h'].size
    nl = np.array([n], dtype=np.int32)

    def run(arrs, p):
        a = [np.ascontiguousarray(arrs[k][None, :], dtype=np.float64) for k in ('h', 'vp', 'vs', 'rho')]
        return po.rf_batch(*a, nl, p, m['gauss'], m['nsamp'], m['fsamp'], m['tshift'], nsv, m['waveno'],
                           nout=m['nout'], nthreads=1)[0]
    want = run(m, m['p'])
    scale = max(1.0, np.abs(want).max())
    spread = max(np.abs(run(m, np.nextafter(m['p'], s)) - want).max() / scale for s in (0.0, 99.0))
    for key in ('h', 'vp', 'vs', 'rho'):
        for i in range(n - 1 if key == 'h' else n):          # (the half-space has no thickness)
            for s in (0.0, 1e9):
                arrs = dict(m)
                arrs[key] = m[key].copy()
                arrs[key][i] = np.nextafter(arrs[key][i], s)
                spread = max(spread, np.abs(run(arrs, m['p']) - want).max() / scale)
    return want, scale, spread
