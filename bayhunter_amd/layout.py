"""Targets of a joint inversion and the layout of one model's output row -- no torch, no device.

Shared by the batched engine (engine.ForwardEngine: torch tensors and streams) and by the evaluation plan
of the chain pool (evalplan.EvalPlan: everything inside libbayhunter_amd).
"""
import numpy as np

from . import _lib

# Batches above this many models get a processing order (deepest first, ...: engine.reorder).  It is not only
# the lane kernel that needs it: the team kernels take one search per workgroup in launch order, a thousand or
# two are resident at a time, and a deep model started late is a tail of its own -- 8 192 ragged 2-31-layer
# models 14.3 -> 9.8 ms, 2 048 models of 2-21 layers 2.9 -> 1.9 ms with the order (profiles/r03_order_small.txt).
# Up to 1 024 searches every team is resident at once and the order is irrelevant.
ORDER_MIN = 1024

SWD_REFS = {'rdispgr': (2, 1), 'ldispgr': (1, 1), 'rdispph': (2, 0), 'ldispph': (1, 0)}
RF_REFS = {'prf': 0, 'seis': 0, 'srf': 1}


def rf_obsparams(obsx, ref='prf'):
    """fsamp, tshft, nsamp from the observed time axis (src/rfmini_modrf.py:41-62)."""
    obsx = np.asarray(obsx, dtype=np.float64)
    steps = np.unique(np.round(np.diff(obsx), 4))
    if steps.size != 1:
        raise ValueError("receiver-function target '%s': the time axis must be uniformly sampled" % ref)
    fsamp = 1. / float(steps[0])
    nsamp = 2.**int(np.ceil(np.log2(obsx.size * 2)))       # a float, like the reference's
    return fsamp, float(-obsx[0]), nsamp


class SwdSpec(object):
    def __init__(self, ref, periods, mode=1, flsph=0):
        if ref not in SWD_REFS:
            raise ReferenceError("no dispersion forward model for ref '%s'" % ref)
        self.ref = ref
        self.iwave, self.igr = SWD_REFS[ref]
        self.obsx = np.ascontiguousarray(periods, dtype=np.float64)
        # more than NP = 60 periods (surfdisp96.f:62): solve on 60 evenly spaced periods over the same
        # span and interpolate linearly to the observed ones, like SurfDisp (surf96_modsw.py:35-43,
        # 106-122); ForwardEngine does the interpolation on the device
        self.resample = self.obsx.size > _lib.MAX_PERIODS
        self.periods = np.linspace(self.obsx.min(), self.obsx.max(), _lib.MAX_PERIODS) if self.resample \
            else self.obsx
        self.mode, self.flsph = int(mode), int(flsph)


class RfSpec(object):
    def __init__(self, ref, obsx, gauss=1.0, p=6.4, nsv=None, wtype=None):
        if wtype is None:
            if ref not in RF_REFS:
                raise ReferenceError("no receiver-function forward model for ref '%s'" % ref)
            self.waveno = RF_REFS[ref]
        else:
            self.waveno = {'P': 0, 'SV': 1}[wtype]
        self.ref = ref
        self.obsx = np.ascontiguousarray(obsx, dtype=np.float64)
        self.fsamp, self.tshft, self.nsamp = rf_obsparams(self.obsx, ref)
        self.gauss, self.p, self.nsv = float(gauss), float(p), nsv



class RowLayout(object):
    """Output row of model b: [swd target 0 | swd target 1 | ... | rf target 0 | ...], fp64, `ncols`
    values; `slices[t]` is target t's column range.  Rows are `row` doubles apart: with a dispersion
    target of more than 60 periods `row` > `ncols` (the 60 solved values live behind the visible
    columns and are interpolated into them after the kernel).  `tg` / `rfp` are the descriptor arrays
    bh_swd_batch / bh_rf_batch take, `periods` the concatenated solved periods, `resampled` the
    (target index, visible slice, scratch offset, SwdSpec) of every target with more than 60 periods."""

    def __init__(self, swd=(), rf=()):
        self.swd, self.rf = list(swd), list(rf)
        if len(self.swd) > _lib.MAX_TARGETS:
            raise ValueError("too many SWD targets")
        off = 0
        self.slices = []
        for sp in self.swd:
            self.slices.append(slice(off, off + sp.obsx.size))
            off += sp.obsx.size
        rf_off = []
        for r in self.rf:
            rf_off.append(off)
            self.slices.append(slice(off, off + r.obsx.size))
            off += r.obsx.size
        self.ncols = off
        per_off = 0
        self.tg = (_lib.SwdTarget * max(1, len(self.swd)))()
        pers = []
        self.resampled = []
        for t, sp in enumerate(self.swd):
            koff = self.slices[t].start
            if sp.resample:                                # the kernel writes behind the visible columns
                koff = off
                off += sp.periods.size
                self.resampled.append((t, self.slices[t], koff, sp))
            self.tg[t] = _lib.SwdTarget(sp.iwave, sp.igr, sp.mode, sp.flsph, sp.periods.size, per_off, koff, 0)
            pers.append(sp.periods)
            per_off += sp.periods.size
        self.rfp = [_lib.RfParams(r.p, r.gauss, r.fsamp, r.tshft, -1.0 if r.nsv is None else float(r.nsv),
                                  int(r.nsamp), r.waveno, r.obsx.size, o) for r, o in zip(self.rf, rf_off)]
        self.row = off
        self.periods = np.ascontiguousarray(np.concatenate(pers) if pers else np.zeros(1), dtype=np.float64)
