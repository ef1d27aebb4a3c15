"""Forward-modelling plugins with BayHunter's plugin contract, computed on MI355X.

Contract (reference: src/Targets.py:33-82,201-214, templates/myfwd.py): an object built as
``Plugin(obsx, ref)`` with ``set_modelparams(**kw)`` and
``run_model(h, vp, vs, rho, **kw) -> (xmod, ymod)``, where ``xmod`` equals ``obsx`` and ``ymod`` is
an ndarray of the same length -- or ``(nan, nan)`` when no solution exists; it never raises for a
model it cannot solve.  ``SurfDisp`` and ``RFminiModRF`` are stand-ins for the reference classes of
the same names (src/surf96_modsw.py, src/rfmini_modrf.py) and accept the same parameters; both add
``run_models`` for batches.  All numerics happen in libbayhunter_amd (no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import RF_REFS, SWD_REFS, ForwardEngine, RfSpec, SwdSpec, rf_obsparams

NP_MAX = _lib.MAX_PERIODS   # surfdisp96.f:62


class _Plugin(object):
    """Shared plumbing: observed axis, model-parameter dict, lazily built batched engine."""

    defaults = {}

    def __init__(self, obsx, ref):
        self.obsx = obsx
        self.ref = ref
        self.modelparams = dict(self.defaults)
        self._engine = None

    def set_modelparams(self, **mparams):
        self.modelparams.update(mparams)
        self._engine = None          # parameters are baked into the engine's launch descriptors

    def __getstate__(self):
        # targets are pickled into <station>_config.pkl (src/utils.py:127-153); device handles stay out
        state = dict(self.__dict__)
        state['_engine'] = None
        return state


class SurfDisp(_Plugin):
    """Dispersion curves (Rayleigh/Love, phase/group) for layered models.

    ref            'rdispph' | 'rdispgr' | 'ldispph' | 'ldispgr'  (surf96_modsw.py:48-59)
    modelparams    mode (1 = fundamental), flsph (0 = flat earth)  (surf96_modsw.py:29-32)
    More than 60 periods are handled like the reference does (surf96_modsw.py:35-43,106-122): solve
    on 60 linearly spaced periods over the same span and interpolate back linearly.
    """
    defaults = {'mode': 1, 'flsph': 0}

    def __init__(self, obsx, ref):
        if ref not in SWD_REFS:
            raise ReferenceError(
                "SurfDisp has no forward model for ref '%s' (known: %s). Give the target one of "
                "these refs or install your own plugin with target.update_plugin(...)."
                % (ref, ', '.join(sorted(SWD_REFS))))
        _Plugin.__init__(self, obsx, ref)
        self.kmax = obsx.size
        self.wavetype, self.veltype = SWD_REFS[ref]
        if self.kmax > NP_MAX:
            self.obsx_int = np.linspace(obsx.min(), obsx.max(), NP_MAX)
            print("SurfDisp(%s): %d periods exceed the solver limit of %d; solving on %d evenly "
                  "spaced periods between %g and %g s and interpolating linearly to the observed "
                  "ones." % (ref, self.kmax, NP_MAX, NP_MAX, obsx.min(), obsx.max()))

    def get_surftags(self, ref):
        return SWD_REFS[ref]

    def _solver_periods(self):
        if self.kmax > NP_MAX:
            return np.ascontiguousarray(self.obsx_int, dtype=np.float64)
        return np.ascontiguousarray(self.obsx, dtype=np.float64)

    def _to_observed_axis(self, pers, vel):
        if self.kmax > NP_MAX:
            return self.obsx, np.interp(self.obsx, pers, vel)
        return pers, vel

    def run_model(self, h, vp, vs, rho, **params):
        """One model through bh_surfdisp96, the C-ABI stand-in for the f2py symbol."""
        lib = _lib.load()
        model32 = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))
                   for a in (h, vp, vs, rho)]                  # the cast f2py applies
        pers = self._solver_periods()
        vel = np.zeros(pers.size)
        flag = C.c_int(0)
        _lib.check(lib.bh_surfdisp96(
            model32[0].ctypes.data, model32[1].ctypes.data, model32[2].ctypes.data,
            model32[3].ctypes.data, len(h), self.modelparams['flsph'], self.wavetype,
            self.modelparams['mode'], self.veltype, pers.size, pers.ctypes.data, vel.ctypes.data,
            C.byref(flag)))
        if flag.value != 0:
            return np.nan, np.nan                               # surf96_modsw.py:126
        return self._to_observed_axis(pers, vel)

    def run_models(self, H, VP, VS, RHO, nlay):
        """Batch of models ([B, Lmax] arrays) -> (x, Y[B, nobs], err[B]); unsolved rows are NaN."""
        pers = self._solver_periods()
        if self._engine is None:
            self._engine = ForwardEngine(swd=[SwdSpec(self.ref, pers, self.modelparams['mode'],
                                                      self.modelparams['flsph'])])
        out, err = self._engine.run(H, VP, VS, RHO, nlay)
        Y, flags = out.cpu().numpy(), err.cpu().numpy()[:, 0]
        if self.kmax > NP_MAX:
            Y = np.stack([np.interp(self.obsx, pers, y) for y in Y])
        Y[flags != 0] = np.nan
        return (self.obsx if self.kmax > NP_MAX else pers), Y, flags


class RFminiModRF(_Plugin):
    """P or S receiver functions for layered models.

    ref            'prf' | 'seis' (incident P) or 'srf' (incident SV)   (rfmini_modrf.py:21-24)
    modelparams    gauss, p [s/deg], nsv (None: top-layer Vs), wtype, and `water`, which is accepted
                   and has no effect -- exactly as in the reference, whose native code never
                   receives it (rfmini_modrf.py:114 vs :134-137, greens.cpp:384)
    The sampling frequency, time shift and FFT length follow from the observed time axis
    (rfmini_modrf.py:41-62); a non-uniform axis raises ValueError.
    """

    def __init__(self, obsx, ref):
        _Plugin.__init__(self, obsx, ref)
        self.fsamp, self.tshft, self.nsamp = rf_obsparams(obsx, ref)
        if ref in RF_REFS:
            self.modelparams['wtype'] = 'SV' if RF_REFS[ref] == 1 else 'P'
        self.modelparams.update({'gauss': 1.0, 'p': 6.4, 'water': 0.001, 'nsv': None})

    def _waveno(self):
        wtype = self.modelparams['wtype']
        if wtype not in ('P', 'SV', 'SH'):                      # rfmini.pyx:91-94
            raise ValueError("wave must be 'P', 'SV' or 'SH', not '%s'" % wtype)
        if wtype == 'SH':
            raise ValueError("SH receiver functions are not part of BayHunter's forward path")
        return 0 if wtype == 'P' else 1

    def _time_axis(self):
        return (np.arange(int(self.nsamp)) / self.fsamp - self.tshft)[:self.obsx.size]

    def compute_rf(self, h, vp, vs, rho, **params):
        """One model through bh_synrf, the C-ABI stand-in for rfmini.synrf / synrf_cwrap."""
        lib = _lib.load()
        n = h.size
        qp = np.ascontiguousarray(params.get('qp', np.full(n, 500.)), dtype=np.float64)
        qs = np.ascontiguousarray(params.get('qs', np.full(n, 225.)), dtype=np.float64)
        depth = np.ascontiguousarray(np.concatenate(([0], np.cumsum(h)[:-1])))   # layer tops
        top_vp, top_vs = float(vp[0]), float(vs[0])
        k = top_vp / top_vs
        sigma = (2 - k**2) / (2 - 2 * k**2)                     # Poisson ratio of the top layer
        nsv = self.modelparams['nsv']
        nsv = top_vs if nsv is None else nsv
        vp, vs, rho = (np.ascontiguousarray(a, dtype=np.float64) for a in (vp, vs, rho))
        nsamp = int(self.nsamp)
        trace = np.zeros(nsamp)
        _lib.check(lib.bh_synrf(nsamp, self.fsamp, self.tshft, self.modelparams['p'],
                                self.modelparams['gauss'], nsv, sigma, self._waveno(), n,
                                depth.ctypes.data, vp.ctypes.data, vs.ctypes.data, rho.ctypes.data,
                                qp.ctypes.data, qs.ctypes.data, None, None, trace.ctypes.data))
        return self._time_axis(), trace[:self.obsx.size]

    def run_model(self, h, vp, vs, rho, **params):
        if not (h.size == vp.size == vs.size == rho.size):
            raise AssertionError("h, vp, vs, rho must have the same length")
        h, vp, vs, rho = (np.asarray(a, dtype=float) for a in (h, vp, vs, rho))
        return self.compute_rf(h, vp, vs, rho, **params)

    def run_models(self, H, VP, VS, RHO, nlay):
        """Batch of models ([B, Lmax] arrays) -> (time, RF[B, nobs])."""
        if self._engine is None:
            self._waveno()
            self._engine = ForwardEngine(rf=[RfSpec(self.ref, self.obsx, self.modelparams['gauss'],
                                                    self.modelparams['p'], self.modelparams['nsv'],
                                                    wtype=self.modelparams['wtype'])])
        out, _ = self._engine.run(H, VP, VS, RHO, nlay)
        return self._time_axis(), out.cpu().numpy()
