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
