"""SurfDisp: forward plugin for surface-wave dispersion, same surface as the reference's
src/surf96_modsw.py (constructor (obsx, ref), set_modelparams, run_model -> (x, y) or (nan, nan)),
computed by the HIP engine instead of the f2py-wrapped Fortran.  Adds run_models (batched).
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import SWD_REFS


class SurfDisp(object):
    """Forward modeling of dispersion curves on MI355X (drop-in for BayHunter.SurfDisp)."""

    def __init__(self, obsx, ref):
        self.obsx = obsx
        self.kmax = obsx.size
        self.ref = ref

        self.modelparams = {
            'mode': 1,  # mode, 1 fundamental, 2 first higher
            'flsph': 0  # flat earth model
            }

        self.wavetype, self.veltype = self.get_surftags(ref)

        if self.kmax > 60:  # surf96_modsw.py:35-43
            message = "Your observed data vector exceeds the maximum of 60 \
periods that is allowed in SurfDisp. For forward modeling SurfDisp will \
reduce the samples to 60 by linear interpolation within the given period \
span.\nFrom this data, the dispersion velocities to your observed periods \
will be determined. The precision of the data will depend on the distribution \
of your samples and the complexity of the input velocity-depth model."
            self.obsx_int = np.linspace(obsx.min(), obsx.max(), 60)
            print(message)
        self._engine = None

    def set_modelparams(self, **mparams):
        self.modelparams.update(mparams)
        self._engine = None

    def get_surftags(self, ref):
        if ref in SWD_REFS:
            return SWD_REFS[ref]
        tagerror = "Reference is not available in SurfDisp. If you defined \
a user Target, assign the correct reference (target.ref) or update the \
forward modeling plugin with target.update_plugin(MyForwardClass()).\n \
* Your ref was: %s\nAvailable refs are: rdispgr, ldispgr, rdispph, ldispph\n \
(r=rayleigh, l=love, gr=group, ph=phase)" % ref
        raise ReferenceError(tagerror)

    def _periods(self):
        if self.kmax > 60:
            return 60, self.obsx_int
        return self.kmax, np.ascontiguousarray(self.obsx, dtype=np.float64)

    def run_model(self, h, vp, vs, rho, **params):
        """One model through the C-ABI drop-in of the f2py symbol (bh_surfdisp96)."""
        lib = _lib.load()
        nlayer = len(h)
        f32 = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32))
               for x in (h, vp, vs, rho)]
        kmax, pers = self._periods()
        pers = np.ascontiguousarray(pers, dtype=np.float64)
        dispvel = np.zeros(kmax)
        error = C.c_int(0)
        _lib.check(lib.bh_surfdisp96(
            f32[0].ctypes.data, f32[1].ctypes.data, f32[2].ctypes.data, f32[3].ctypes.data,
            nlayer, self.modelparams['flsph'], self.wavetype, self.modelparams['mode'],
            self.veltype, kmax, pers.ctypes.data, dispvel.ctypes.data, C.byref(error)))
        if error.value == 0:
            if self.kmax > 60:
                disp_int = np.interp(self.obsx, pers, dispvel)
                return self.obsx, disp_int
            return pers[:kmax], dispvel[:kmax]
        return np.nan, np.nan

    def run_models(self, H, VP, VS, RHO, nlay):
        """Batched: [B, Lmax] arrays -> (x, Y[B, kmax], err[B]); rows with err != 0 are NaN."""
        from .engine import ForwardEngine, SwdSpec
        kmax, pers = self._periods()
        if self._engine is None:
            self._engine = ForwardEngine(swd=[SwdSpec(self.ref, pers, self.modelparams['mode'],
                                                      self.modelparams['flsph'])])
        out, err = self._engine.run(H, VP, VS, RHO, nlay)
        Y = out.cpu().numpy()
        e = err.cpu().numpy()[:, 0]
        if self.kmax > 60:
            Y = np.stack([np.interp(self.obsx, pers, y) for y in Y])
        Y[e != 0] = np.nan
        return (self.obsx if self.kmax > 60 else pers[:kmax]), Y, e
