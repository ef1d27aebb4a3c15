"""Golden vectors for the step in front of the forward path, produced by the REFERENCE's own
src/Models.py (Model.get_vp_vs_h) and src/SingleChain.py (SingleChain._validmodel), loaded
file-wise in the development container.

    python tests/golden/make_golden_models.py        (needs /root/reference)

SingleChain.py is loaded with a stand-in `BayHunter` package that only provides the reference's
own Model/ModelMatrix (from Models.py) and an empty `utils`; `_validmodel` is called unbound on a
plain namespace carrying the attributes it reads.  `np.int` (removed in NumPy 2, touched by a no-op
statement at src/Models.py:35) is aliased to `int`.
Output voronoi.npz: nuclei, per-model vpvs, the reference's h/vp and validity flags for several
prior set-ups.
"""
import importlib.util
import os
import sys
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'

SETUPS = {
    # name: (priors, thickmin, lvz, hvz, mantle)
    'defaults': (dict(layers=(1, 20), vs=(1, 5), z=(0, 60)), 0., None, None, None),
    'tutorial': (dict(layers=(1, 20), vs=(2, 5), z=(0, 60)), 0.1, None, None, (4.3, 1.8)),
    'zones': (dict(layers=(2, 8), vs=(1.5, 4.8), z=(1, 55)), 0.5, 0.1, 0.3, None),
}


def load(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    if not hasattr(np, 'int'):
        np.int = int
    models = load('ref_models', 'Models.py')
    pkg = types.ModuleType('BayHunter')
    pkg.Model, pkg.ModelMatrix, pkg.utils = models.Model, models.ModelMatrix, types.ModuleType('BayHunter.utils')
    sys.modules['BayHunter'] = pkg
    sys.modules['BayHunter.utils'] = pkg.utils
    chain = load('ref_singlechain', 'SingleChain.py')

    rs = np.random.RandomState(77)
    B, Lmax = 400, 12
    nlay = rs.randint(1, Lmax + 1, size=B).astype(np.int32)
    VSN = np.full((B, Lmax), np.nan)
    ZV = np.full((B, Lmax), np.nan)
    vpvs = rs.uniform(1.5, 2.1, size=B)
    for b in range(B):
        n = nlay[b]
        vs = rs.uniform(0.8, 5.2, size=n)
        if b % 3:
            vs.sort()
        VSN[b, :n] = vs
        ZV[b, :n] = np.sort(rs.uniform(-1, 62, size=n) if b % 7 == 0 else rs.uniform(0, 60, size=n))
    d = dict(VSN=VSN, ZV=ZV, nlay=nlay, vpvs=vpvs)
    for name, (priors, thickmin, lvz, hvz, mantle) in SETUPS.items():
        H = np.zeros((B, Lmax)); VP = np.zeros((B, Lmax)); valid = np.zeros(B, dtype=np.int32)
        for b in range(B):
            n = nlay[b]
            model = np.concatenate((VSN[b, :n], ZV[b, :n]))
            vp, vs, h = models.Model.get_vp_vs_h(model, vpvs[b], list(mantle) if mantle else None)
            H[b, :n], VP[b, :n] = h, vp
            me = types.SimpleNamespace(priors=priors, thickmin=thickmin, lowvelperc=lvz,
                                       highvelperc=hvz, currentvpvs=vpvs[b],
                                       mantle=list(mantle) if mantle else None, chainidx=0)
            valid[b] = int(chain.SingleChain._validmodel(me, model))
        d[name + '_H'], d[name + '_VP'], d[name + '_valid'] = H, VP, valid
        print(name, 'valid', valid.sum(), 'of', B)
    np.savez_compressed(os.path.join(OUT, 'voronoi.npz'), **d)


if __name__ == '__main__':
    main()
