"""Targets and joint likelihood -- the host side of the hot path's entry point.

Stand-in for the parts of the reference's src/Targets.py that sit on the hot path
(ObservedData, ModeledData as plugin host, Valuation, SingleTarget and its six typed subclasses,
JointTarget.evaluate); plotting is out of scope.  Same constructor arguments, attribute names and
results; two differences by design:

* nothing here builds an n x n inverse covariance per evaluation (the reference does,
  src/Targets.py:113,127,136,171): the diagonal and exponential models are evaluated in closed
  form, the fixed Gaussian model keeps its R^-1 once;
* `JointTarget.evaluate_batch` evaluates many models at once on the GPU: forward kernels + the
  fused likelihood kernel (bh_likelihood_batch), nothing but logL and misfits comes back.

Covariance selection follows SingleChain.set_target_covariance (src/SingleChain.py:159-205) and is
exposed as `JointTarget.set_target_covariance`.
"""
import ctypes as C
import logging

import numpy as np

from . import _lib
from .engine import RF_REFS, SWD_REFS
from .plugins import RFminiModRF, SurfDisp

logger = logging.getLogger()

LOG_2PI = np.log(2 * np.pi)


class ObservedData(object):
    """x, y (and optionally yerr) of one observed curve (src/Targets.py:16-30)."""

    def __init__(self, x, y, yerr=None):
        self.x = x
        self.y = y
        usable = yerr is not None and not (np.any(yerr <= 0.) or np.any(np.isnan(yerr)))
        self.yerr = yerr if usable else np.ones(x.size) * np.nan


class ModeledData(object):
    """Holds the forward plugin of a target and its latest synthetic (src/Targets.py:33-82)."""

    def __init__(self, obsx, ref):
        if ref in RF_REFS and ref != 'seis':
            self.plugin = RFminiModRF(obsx, ref)
            self.xlabel = 'Time in s'
        elif ref in SWD_REFS:
            self.plugin = SurfDisp(obsx, ref)
            self.xlabel = 'Period in s'
        else:
            logger.info("No built-in forward model for ref '%s': install one with "
                        "target.update_plugin(MyForwardClass())" % ref)
            self.plugin = None
            self.xlabel = 'x'
        self.x = np.nan
        self.y = np.nan

    def update(self, plugin):
        self.plugin = plugin

    def calc_synth(self, h, vp, vs, **kwargs):
        rho = kwargs.pop('rho')
        self.x, self.y = self.plugin.run_model(h, vp, vs, rho=rho, **kwargs)


class Valuation(object):
    """Misfit and likelihood pieces (src/Targets.py:85-183).

    The `get_covariance_*` methods return (c_inv, logc_det) like the reference for callers that
    want the matrices; `quadratic_form` is what the evaluation path uses."""

    def __init__(self):
        self.corr_inv = None
        self.logcorr_det = None
        self.misfit = None
        self.likelihood = None

    @staticmethod
    def get_rms(yobs, ymod):
        return np.sqrt(np.mean((ymod - yobs)**2))

    @staticmethod
    def get_covariance_nocorr(sigma, size, yerr=None, corr=0):
        return np.eye(size) / (sigma**2), (2 * size) * np.log(sigma)

    @staticmethod
    def get_covariance_nocorr_scalederr(sigma, size, yerr, corr=0):
        scaled_err = yerr / yerr.min()
        c_inv = np.diag(1.0 / (scaled_err * sigma**2))
        return c_inv, (2 * size) * np.log(sigma) + np.log(np.prod(scaled_err))

    @staticmethod
    def get_corr_inv(corr, size):
        main = np.full(size, 1.0 + corr**2)
        main[0] = main[-1] = 1
        off = np.full(size - 1, -corr)
        return np.diag(main) + np.diag(off, k=1) + np.diag(off, k=-1)

    def get_covariance_exp(self, corr, sigma, size, yerr=None):
        c_inv = self.get_corr_inv(corr, size) / (sigma**2 * (1 - corr**2))
        return c_inv, (2 * size) * np.log(sigma) + (size - 1) * np.log(1 - corr**2)

    def init_covariance_gauss(self, corr, size, rcond=None):
        # The factorisation depends on (corr, size, rcond) only and is kept: every ChainPool of a process comes
        # through here.  Not just for the milliseconds -- numpy's OpenBLAS hands the 201 x 201 SVD to a pool of one
        # thread per core (64 on a GPU box) whose idle workers SPIN for about a tenth of a second afterwards: 6
        # CPU-seconds that a sampler started right behind the set-up pays for with a stall of 50-80 ms when the
        # box's CPU quota (16 cores per 100 ms) runs out (tools/host_phase_probe.py: cgroup nr_throttled; 1 024
        # chains 8.5e5 -> 1.29e6 chain iterations/s without the spin).  The calls themselves stay as the reference
        # makes them (src/Targets.py:140-150): with rcond near the small singular values the pseudo-inverse, and
        # with it the likelihood, depends on how the BLAS splits its work -- a single-threaded call gives
        # logL = 2838.56 where the reference, and the golden vectors, have 2835.83.
        key = (float(corr), int(size), None if rcond is None else float(rcond))
        hit = _GAUSS_COVARIANCE.get(key)
        if hit is None:
            lag = np.abs(np.subtract.outer(np.arange(size), np.arange(size))).astype(float)
            rmatrix = corr**(lag**2)
            corr_inv = np.linalg.pinv(rmatrix, rcond=rcond) if rcond is not None else np.linalg.inv(rmatrix)
            hit = _GAUSS_COVARIANCE[key] = (corr_inv, np.linalg.slogdet(rmatrix)[1])
            if len(_GAUSS_COVARIANCE) > 32:
                _GAUSS_COVARIANCE.pop(next(iter(_GAUSS_COVARIANCE)))
        self.corr_inv, self.logcorr_det = hit[0].copy(), hit[1]

    def get_covariance_gauss(self, sigma, size, yerr=None, corr=None):
        return self.corr_inv / (sigma**2), (2 * size) * np.log(sigma) + self.logcorr_det

    @staticmethod
    def get_likelihood(yobs, ymod, c_inv, logc_det):
        ydiff = ymod - yobs
        madist = ydiff.dot(c_inv).dot(ydiff)
        return -0.5 * (yobs.size * LOG_2PI + logc_det) - madist / 2.


# (corr, size, rcond) -> (corr_inv, logcorr_det): see Valuation.init_covariance_gauss
_GAUSS_COVARIANCE = {}


class SingleTarget(object):
    """Observed + modelled data + valuation of one data type (src/Targets.py:186-247)."""
    noiseref = None

    def __init__(self, x, y, ref, yerr=None):
        self.ref = ref
        self.obsdata = ObservedData(x=x, y=y, yerr=yerr)
        self.moddata = ModeledData(obsx=x, ref=ref)
        self.valuation = Valuation()
        self.covmodel = _lib.COV_NOCORR
        self.get_covariance = self.valuation.get_covariance_nocorr
        logger.info("Initiated target: %s (ref: %s)" % (self.__class__.__name__, self.ref))

    def update_plugin(self, plugin):
        self.moddata.update(plugin)

    def _moddata_valid(self):
        mx, my = self.moddata.x, self.moddata.y
        return (type(mx) == np.ndarray and len(self.obsdata.x) == len(mx)
                and bool(np.sum(self.obsdata.x - mx) <= 1e-5) and len(self.obsdata.y) == len(my))

    def calc_misfit(self):
        self.valuation.misfit = self.valuation.get_rms(self.obsdata.y, self.moddata.y) \
            if self._moddata_valid() else 1e15

    def calc_likelihood(self, c_inv, logc_det):
        self.valuation.likelihood = self.valuation.get_likelihood(
            self.obsdata.y, self.moddata.y, c_inv, logc_det) if self._moddata_valid() else -1e15

    # closed-form Mahalanobis distance + log-determinant for the selected covariance model
    def quadratic_form(self, ydiff, corr, sigma):
        n = ydiff.size
        logdet = (2 * n) * np.log(sigma)
        if self.covmodel == _lib.COV_NOCORR:
            return ydiff.dot(ydiff) / sigma**2, logdet
        if self.covmodel == _lib.COV_NOCORR_SCALED:
            se = self.obsdata.yerr / self.obsdata.yerr.min()
            return np.sum(ydiff**2 / se) / sigma**2, logdet + np.log(np.prod(se))
        if self.covmodel == _lib.COV_EXP:
            w = np.full(n, 1.0 + corr**2)
            w[0] = w[-1] = 1.0
            q = np.sum(w * ydiff**2) - 2.0 * corr * np.sum(ydiff[:-1] * ydiff[1:])
            return q / (sigma**2 * (1 - corr**2)), logdet + (n - 1) * np.log(1 - corr**2)
        q = ydiff.dot(self.valuation.corr_inv).dot(ydiff)
        return q / sigma**2, logdet + self.valuation.logcorr_det


def _typed(ref, noiseref, name):
    def __init__(self, x, y, yerr=None):
        SingleTarget.__init__(self, x, y, ref, yerr=yerr)
    return type(name, (SingleTarget,), {'__init__': __init__, 'noiseref': noiseref})


RayleighDispersionPhase = _typed('rdispph', 'swd', 'RayleighDispersionPhase')
RayleighDispersionGroup = _typed('rdispgr', 'swd', 'RayleighDispersionGroup')
LoveDispersionPhase = _typed('ldispph', 'swd', 'LoveDispersionPhase')
LoveDispersionGroup = _typed('ldispgr', 'swd', 'LoveDispersionGroup')
PReceiverFunction = _typed('prf', 'rf', 'PReceiverFunction')
SReceiverFunction = _typed('srf', 'rf', 'SReceiverFunction')


class JointTarget(object):
    """List of SingleTargets + joint likelihood (src/Targets.py:298-347)."""

    def __init__(self, targets):
        self.targets = targets
        self.ntargets = len(targets)
        self._batch = None
        self.use_mfma = True     # dense Gaussian covariance product on the FP64 matrix cores

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_batch'] = None       # device tensors and launch descriptors are rebuilt on demand
        return state

    def get_misfits(self):
        misfits = [target.valuation.misfit for target in self.targets]
        return np.concatenate((misfits, [np.sum(misfits)]))

    def set_target_covariance(self, corrfix, noise_corr, rcond=None):
        """Choose each target's covariance model like SingleChain.set_target_covariance
        (src/SingleChain.py:159-205): corr free -> exponential; corr fixed at 0 -> diagonal (scaled
        by yerr when given); corr fixed != 0 -> Gaussian for 'rf' targets, exponential otherwise."""
        for i, target in enumerate(self.targets):
            v = target.valuation
            if not corrfix[i]:
                target.covmodel, target.get_covariance = _lib.COV_EXP, v.get_covariance_exp
            elif noise_corr[i] == 0 and np.any(np.isnan(target.obsdata.yerr)):
                target.covmodel, target.get_covariance = _lib.COV_NOCORR, v.get_covariance_nocorr
            elif noise_corr[i] == 0:
                target.covmodel = _lib.COV_NOCORR_SCALED
                target.get_covariance = v.get_covariance_nocorr_scalederr
            elif target.noiseref == 'rf':
                v.init_covariance_gauss(noise_corr[i], target.obsdata.x.size, rcond=rcond)
                target.covmodel, target.get_covariance = _lib.COV_GAUSS, v.get_covariance_gauss
            else:
                target.covmodel, target.get_covariance = _lib.COV_EXP, v.get_covariance_exp
        self._batch = None

    def evaluate(self, h, vp, vs, noise, **kwargs):
        """One model: sets proposallikelihood and proposalmisfits (src/Targets.py:314-347)."""
        rho = kwargs.pop('rho', vp * 0.32 + 0.77)
        logL = 0
        for n, target in enumerate(self.targets):
            target.moddata.calc_synth(h=h, vp=vp, vs=vs, rho=rho, **kwargs)
            if not target._moddata_valid():
                self.proposallikelihood = -1e15
                self.proposalmisfits = [1e15] * (self.ntargets + 1)
                return
            target.calc_misfit()
            corr, sigma = noise[2 * n:2 * n + 2]
            ydiff = target.moddata.y - target.obsdata.y
            madist, logc_det = target.quadratic_form(ydiff, corr, sigma)
            logL += -0.5 * (ydiff.size * LOG_2PI + logc_det) - madist / 2.
        self.proposallikelihood = logL
        self.proposalmisfits = self.get_misfits()

    # ------------------------------------------------------------------ batched, on the GPU
    def batch_layout(self):
        """What a batched evaluation needs besides device memory (no torch, no device): the row layout of
        the forward kernels (layout.RowLayout), the likelihood descriptors, and the observed data / auxiliary
        arrays (scaled errors, dense R^-1) as host arrays."""
        from .layout import RfSpec, RowLayout, SwdSpec
        swd, rf, order = [], [], []
        for t in self.targets:
            p = t.moddata.plugin
            if isinstance(p, SurfDisp):
                order.append(('swd', len(swd)))
                swd.append(SwdSpec(t.ref, p.obsx, p.modelparams['mode'], p.modelparams['flsph']))
            elif isinstance(p, RFminiModRF):
                p._waveno()
                order.append(('rf', len(rf)))
                rf.append(RfSpec(t.ref, p.obsx, p.modelparams['gauss'], p.modelparams['p'],
                                 p.modelparams['nsv'], wtype=p.modelparams['wtype']))
            else:
                raise TypeError("evaluate_batch supports the built-in SurfDisp / RFminiModRF plugins")
        lay = RowLayout(swd, rf)
        slices = [lay.slices[i if kind == 'swd' else len(swd) + i] for kind, i in order]
        yobs = np.zeros(lay.row)
        aux, desc = [], (_lib.LikeTarget * self.ntargets)()
        aux_off = 0
        for n, (t, sl) in enumerate(zip(self.targets, slices)):
            yobs[sl] = t.obsdata.y
            extra, a = 0.0, None
            if t.covmodel == _lib.COV_NOCORR_SCALED:
                a = t.obsdata.yerr / t.obsdata.yerr.min()
                extra = float(np.log(np.prod(a)))
            elif t.covmodel == _lib.COV_GAUSS:
                a = np.ascontiguousarray(t.valuation.corr_inv, dtype=np.float64).ravel()
                extra = float(t.valuation.logcorr_det)
            desc[n] = _lib.LikeTarget(sl.stop - sl.start, sl.start, t.covmodel, aux_off, extra)
            if a is not None:
                aux.append(np.asarray(a, dtype=np.float64))
                aux_off += aux[-1].size
        return dict(layout=lay, desc=desc, nflags=max(1, len(swd)), yobs=yobs,
                    aux=np.ascontiguousarray(np.concatenate(aux) if aux else np.zeros(1), dtype=np.float64))

    def _build_batch(self):
        import torch
        from .engine import ForwardEngine
        bl = self.batch_layout()
        eng = ForwardEngine(swd=bl['layout'].swd, rf=bl['layout'].rf)
        dev = eng.device
        self._batch = dict(eng=eng, desc=bl['desc'], nflags=bl['nflags'],
                           yobs=torch.from_numpy(bl['yobs']).to(dev), aux=torch.from_numpy(bl['aux']).to(dev))
        return self._batch

    def eval_plan(self, max_models, Lmax):
        """An evaluation plan of the library for batches of up to `max_models` proposals of at most `Lmax`
        layers (evalplan.EvalPlan: proposals -> (logL, misfits) in one C call, no torch)."""
        from .evalplan import EvalPlan
        return EvalPlan(self.batch_layout(), max_models, Lmax, use_mfma=self.use_mfma)

    def evaluate_batch(self, H, VP=None, VS=None, nlay=None, noise=None, RHO=None, stream=None):
        """Many models at once: `evaluate_batch(models, noise=...)` with resident
        engine.DeviceModels, or `evaluate_batch(H, VP, VS, nlay, noise)` with [B, Lmax] arrays (RHO
        defaults to 0.77 + 0.32*VP like src/Targets.py:319).  noise: [B, 2*ntargets] (corr, sigma
        per target).  Returns (logL[B], misfits[B, ntargets+1]) as device tensors (asynchronous)."""
        import torch
        from .engine import DeviceModels
        bt = self._batch or self._build_batch()
        eng = bt['eng']
        if isinstance(H, DeviceModels):
            out, err = eng.run(H, stream=stream)
        else:
            if RHO is None:
                RHO = (torch.as_tensor(VP) * 0.32 + 0.77) * (torch.as_tensor(VS) > 0) \
                    if isinstance(VP, torch.Tensor) else np.where(np.asarray(VS) > 0, np.asarray(VP) * 0.32 + 0.77, 0.0)
            out, err = eng.run(H, VP, VS, RHO, nlay, stream=stream)
        return self.likelihood_batch(out, err, noise, stream=stream)

    def likelihood_batch(self, out, err, noise, stream=None):
        """The fused likelihood on modelled data that are already on the device: out[B, row] and
        err[B, nswd] as ForwardEngine.run returns them (e.g. rows kept from an earlier launch),
        noise[B, 2*ntargets].  Returns (logL[B], misfits[B, ntargets+1])."""
        import torch
        bt = self._batch or self._build_batch()
        eng = bt['eng']
        B = out.shape[0]
        st = torch.cuda.current_stream(eng.device) if stream is None else stream
        need = eng.lib.bh_likelihood_workspace_bytes(B, self.ntargets, bt['desc']) if self.use_mfma else 0
        ws = bt.setdefault('ws', {}).get(st.cuda_stream)        # one workspace per launch stream
        with torch.cuda.stream(st):                             # allocations belong to the launch stream
            noise = eng._as_dev(noise, torch.float64)
            logL = torch.empty(B, dtype=torch.float64, device=eng.device)
            misfits = torch.empty((B, self.ntargets + 1), dtype=torch.float64, device=eng.device)
            if need and (ws is None or ws.numel() * 8 < need):
                ws = bt['ws'][st.cuda_stream] = torch.empty((need + 7) // 8, dtype=torch.float64, device=eng.device)
        with torch.cuda.device(eng.device):
            _lib.check(eng.lib.bh_likelihood_batch(
                B, self.ntargets, bt['desc'], out.data_ptr(), eng.row, err.data_ptr(), bt['nflags'],
                bt['yobs'].data_ptr(), noise.data_ptr(), bt['aux'].data_ptr(), logL.data_ptr(),
                misfits.data_ptr(), ws.data_ptr() if need else None, need,
                C.c_void_p(st.cuda_stream)))
        return logL, misfits
