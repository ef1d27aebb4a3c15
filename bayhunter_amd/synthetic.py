"""Synthetic batched layer models for parity tests and bench.py.

Mimics SingleChain.draw_initmodel (reference src/SingleChain.py:94-123) followed by
Model.get_vp_vs_h (src/Models.py:40-52): vs ~ U(2,5) sorted (or unsorted for the low-velocity-zone
variant), Voronoi nuclei z ~ U(0,60) sorted, interfaces midway between nuclei, half-space h = 0,
models with a layer thinner than `thickmin` (tutorial/config.ini:18) redrawn, vp = vpvs*vs,
rho = 0.77 + 0.32*vp (src/Targets.py:319).
"""
import numpy as np


def draw_models(B, nlayers, seed, sorted_vs=True, vpvs=1.73, zmax=60.0, vsmin=2.0, vsmax=5.0,
                thickmin=0.1, Lmax=None):
    """Return (H, VP, VS, RHO, nlay): fp64 [B, Lmax] zero-padded arrays and int32 [B].

    `nlayers` is an int (all models have that many layers incl. the half-space) or a (lo, hi)
    tuple for ragged batches (layer count drawn uniformly in [lo, hi]).
    """
    rs = np.random.RandomState(seed)
    if isinstance(nlayers, (tuple, list)):
        lo, hi = nlayers
        nlay = rs.randint(lo, hi + 1, size=B).astype(np.int32)
    else:
        nlay = np.full(B, int(nlayers), dtype=np.int32)
    if Lmax is None:
        Lmax = int(nlay.max())
    H = np.zeros((B, Lmax))
    VS = np.zeros((B, Lmax))
    for b in range(B):
        n = int(nlay[b])
        while True:
            vs = rs.uniform(vsmin, vsmax, size=n)
            if sorted_vs:
                vs.sort()
            z = np.sort(rs.uniform(0.0, zmax, size=n))
            z_disc = (z[:n - 1] + z[1:n]) / 2.
            h_lay = z_disc - np.concatenate(([0], z_disc[:-1]))
            if n == 1 or np.all(h_lay >= thickmin):
                break
        H[b, :n - 1] = h_lay
        VS[b, :n] = vs
    VP = VS * vpvs
    RHO = np.where(VS > 0, VP * 0.32 + 0.77, 0.0)
    return H, VP, VS, RHO, nlay


TUTORIAL_MODEL = dict(h=np.array([5., 23., 8., 0.]), vs=np.array([2.7, 3.6, 3.8, 4.4]), vpvs=1.73)


def tutorial_model():
    """The st3 model of tutorial/create_testdata.py:13-17 as (h, vp, vs, rho)."""
    h = TUTORIAL_MODEL['h'].copy()
    vs = TUTORIAL_MODEL['vs'].copy()
    vp = vs * TUTORIAL_MODEL['vpvs']
    rho = vp * 0.32 + 0.77
    return h, vp, vs, rho
