"""RFminiModRF: forward plugin for receiver functions, same surface as the reference's
src/rfmini_modrf.py (constructor (obsx, ref), set_modelparams with keys gauss/p/water/nsv/wtype,
run_model -> (time, rf)), computed by the HIP engine instead of the Cython-wrapped C++.
Adds run_models (batched).  `water` is accepted and ignored exactly like the reference does
(rfmini_modrf.py:114 reads it, :134-137 never passes it on).
"""
import numpy as np

from . import _lib
from .engine import rf_obsparams


class RFminiModRF(object):
    """Forward modeling of receiver functions on MI355X (drop-in for BayHunter.RFminiModRF)."""

    def __init__(self, obsx, ref):
        self.ref = ref
        self.obsx = obsx
        self._init_obsparams()

        if self.ref in ['prf', 'seis']:
            self.modelparams = {'wtype': 'P'}
        elif self.ref in ['srf']:
            self.modelparams = {'wtype': 'SV'}

        self.modelparams.update(
            {'gauss': 1.0,
             'p': 6.4,
             'water': 0.001,
             'nsv': None
             })
        self._engine = None

    def _init_obsparams(self):
        self.fsamp, self.tshft, self.nsamp = rf_obsparams(self.obsx, self.ref)

    def set_modelparams(self, **mparams):
        self.modelparams.update(mparams)
        self._engine = None

    def _waveno(self):
        wtype = self.modelparams['wtype']
        try:
            return ["P", "SV", "SH"].index(wtype)      # rfmini.pyx:91-94
        except ValueError:
            raise ValueError("wave must be 'P', 'SV' or 'SH', not '%s'" % wtype)

    def compute_rf(self, h, vp, vs, rho, **params):
        """One model through the C-ABI drop-in of synrf_cwrap (bh_synrf)."""
        lib = _lib.load()
        gauss = self.modelparams['gauss']
        p = self.modelparams['p']
        nsv = self.modelparams['nsv']

        qp = np.ascontiguousarray(params.get('qp', np.ones(h.size) * 500.), dtype=np.float64)
        qs = np.ascontiguousarray(params.get('qs', np.ones(h.size) * 225.), dtype=np.float64)

        z = np.cumsum(h)
        z = np.ascontiguousarray(np.concatenate(([0], z[:-1])))

        nsvp, nsvs = float(vp[0]), float(vs[0])
        vpvs = nsvp / nsvs
        poisson = (2 - vpvs**2)/(2 - 2 * vpvs**2)

        if nsv is None:
            nsv = nsvs

        nsamp = int(self.nsamp)
        time = np.arange(nsamp) / self.fsamp - self.tshft
        waveno = self._waveno()
        if waveno == 2:
            raise ValueError("SH receiver functions are not computed by BayHunter's rfmini path")
        vp, vs, rho = (np.ascontiguousarray(x, dtype=np.float64) for x in (vp, vs, rho))
        qrf = np.zeros(nsamp)
        _lib.check(lib.bh_synrf(nsamp, self.fsamp, self.tshft, p, gauss, nsv, poisson, waveno,
                                h.size, z.ctypes.data, vp.ctypes.data, vs.ctypes.data,
                                rho.ctypes.data, qp.ctypes.data, qs.ctypes.data, None, None,
                                qrf.ctypes.data))
        return time[:self.obsx.size], qrf[:self.obsx.size]

    def run_model(self, h, vp, vs, rho, **params):

        assert h.size == vp.size == vs.size == rho.size

        h = h.astype(float)
        vp = vp.astype(float)
        vs = vs.astype(float)
        rho = rho.astype(float)

        time, qrf = self.compute_rf(h, vp, vs, rho, **params)
        return time, qrf

    def run_models(self, H, VP, VS, RHO, nlay):
        """Batched: [B, Lmax] arrays -> (time, RF[B, obsx.size])."""
        from .engine import ForwardEngine, RfSpec
        if self._engine is None:
            self._engine = ForwardEngine(rf=[RfSpec(self.ref, self.obsx, self.modelparams['gauss'],
                                                    self.modelparams['p'], self.modelparams['nsv'])])
        out, _ = self._engine.run(H, VP, VS, RHO, nlay)
        nsamp = int(self.nsamp)
        time = np.arange(nsamp) / self.fsamp - self.tshft
        return time[:self.obsx.size], out.cpu().numpy()
