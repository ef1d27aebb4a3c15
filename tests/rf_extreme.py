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
    """The worst models of the round-3 random campaigns (tests/scenarios/rf_fuzz.py), all SV incidence on ~20 layers
    with strong velocity inversions.  Deviation of the device from the fp64 oracle | of the fp64 oracle from the
    same algorithm in extended precision (tests/hp_oracle.py), relative to the trace's scale:
      seed 79, configuration 4219, model 112 of 1000 (p = 6.71 s/deg, a = 2.56):   8.0e-11 | 1.0e-7
      seed 81, configuration 14547, model 660 of 1000 (p = 7.85, a = 2.92):         1.25e-10 | 3.7e-9
      seed 82, model 29 of 64 (p = 6.67, a = 2.91, 256 samples):                    1.05e-10 | 2.5e-9
    (their neighbours in the same batches: 1e-13 | 3e-13 ... 7e-12).  The reference's own double-precision result is
    good to seven to nine digits there; the device sits 20 to 1000 times closer to it than it is to the exact one."""
    from bayhunter_amd.synthetic import draw_models
    H, VP, VS, RHO, nl = draw_models(1000, 20, seed=1061471874, sorted_vs=False)
    yield dict(h=H[112], vp=VP[112], vs=VS[112], rho=RHO[112], gauss=2.5578946714018787, p=6.711044087493648,
               waveno=1, nsamp=512, fsamp=5.0, tshift=2.0, nout=201)
    H, VP, VS, RHO, nl = draw_models(1000, (21, 27), seed=978415244, sorted_vs=False, zmax=200.0, thickmin=0.05)
    k = int(nl[660])
    yield dict(h=H[660, :k], vp=VP[660, :k], vs=VS[660, :k], rho=RHO[660, :k], gauss=2.921833794173314,
               p=7.846671088611139, waveno=1, nsamp=1024, fsamp=2.0, tshift=2.0, nout=400)
    H, VP, VS, RHO, nl = draw_models(64, (17, 29), seed=533734618, sorted_vs=False, zmax=200.0, thickmin=0.05)
    k = int(nl[29])
    yield dict(h=H[29, :k], vp=VP[29, :k], vs=VS[29, :k], rho=RHO[29, :k], gauss=2.9084908042736477,
               p=6.666963374321674, waveno=1, nsamp=256, fsamp=5.0, tshift=5.0, nout=100)


def oracle_error(po, m, nsv=None):
    """(trace, scale, error): the fp64 oracle's trace for model m and its distance, relative to the trace's scale,
    from the same algorithm evaluated in extended precision (hp_oracle) -- the rounding error of the reference's own
    double-precision result on this model, and the yardstick for a deviation beyond TOL_RF (tolerances.rf_bound)."""
    import hp_oracle
    n = m['h'].size
    a = [np.ascontiguousarray(m[k][None, :], dtype=np.float64) for k in ('h', 'vp', 'vs', 'rho')]
    want = po.rf_batch(*a, np.array([n], dtype=np.int32), m['p'], m['gauss'], m['nsamp'], m['fsamp'], m['tshift'], nsv,
                       m['waveno'], nout=m['nout'], nthreads=1)[0]
    exact = hp_oracle.rf_model_ld(m['h'], m['vp'], m['vs'], m['rho'], m['p'], m['gauss'], m['nsamp'], m['fsamp'],
                                  m['tshift'], nsv, m['waveno'], m['nout'])
    scale = max(1.0, np.abs(want).max())
    return want, scale, float(np.abs(want - exact).max() / scale)
