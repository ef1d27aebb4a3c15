"""Batched forward engine: many layered models per launch, inputs and outputs resident in HBM.

This is the entry point the reference does not have (it evaluates one model per call,
src/Targets.py:314-347); the single-model plugin classes in plugins.py are
thin views on it.  torch is used only for device memory and streams.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .layout import ORDER_MIN, RF_REFS, SWD_REFS, RfSpec, RowLayout, SwdSpec, rf_obsparams  # noqa: F401 (re-exported)


class DeviceModels(object):
    """A batch of layered models resident in HBM: `packed` is [B, 4, Lmax] fp64 (h, vp, vs, rho
    rows of each model contiguous, zero padded), `nlay` int32 [B].  H/VP/VS/RHO are views."""

    def __init__(self, packed, nlay, order=None, depth=None, mean_depth=None):
        assert packed.dim() == 3 and packed.shape[1] == 4 and packed.is_contiguous()
        self.packed, self.nlay = packed, nlay
        self.B, self.Lmax = packed.shape[0], packed.shape[2]
        # mean layer count when the host knows it: a ragged batch is priced by it, not by its deepest model
        # (bh_swd_hint)
        self.mean_depth = None if mean_depth is None else float(mean_depth)
        # deepest model of the batch when the host knows it (<= Lmax): the kernels size their LDS
        # images and pick the team width by it instead of by the allocated row length
        self.depth = None if depth is None else max(1, min(int(depth), self.Lmax))
        self.H, self.VP, self.VS, self.RHO = (packed[:, i, :] for i in range(4))
        # Processing order of the dispersion searches (int32 permutation, ForwardEngine.reorder):
        # the data stay where they are and results land in the caller's rows.
        self.order = order


class ForwardEngine(object):
    """All targets of a joint inversion for a batch of models in (at most) two launches.

    Output row of model b: [swd target 0 | swd target 1 | ... | rf target 0 | ...], fp64, `ncols`
    values; `slices[t]` is target t's column range.  Rows are `row` doubles apart: with a dispersion
    target of more than 60 periods `row` > `ncols` (the 60 solved values live behind the visible
    columns and are interpolated into them after the kernel).
    """

    def __init__(self, swd=(), rf=(), device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.BayHunterAmdError("no HIP device visible to torch; there is no CPU fallback")
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self.layout = lay = RowLayout(swd, rf)
        self.swd, self.rf = lay.swd, lay.rf
        self.slices, self.ncols, self.row = lay.slices, lay.ncols, lay.row
        self._tg, self._rfp = lay.tg, lay.rfp
        # (user slice, scratch offset, device tables) of every target with more than 60 periods
        self._interp = [(sl, koff, self._interp_tables(sp)) for _, sl, koff, sp in lay.resampled]
        with torch.cuda.device(self.device):
            self.periods = torch.from_numpy(lay.periods).to(self.device)
        # per launch stream (batches may be in flight on several streams at once, chains.GpuEvaluator):
        self._ws = {}            # workspace of bh_swd_batch
        self._side = {}          # side stream: RF back-fills the SIMDs the SWD tail leaves idle
        self.overlap = True
        self.sort_ragged = True  # re-order ragged batches by layer count at upload
        self.order_by_length = os.environ.get('BH_ORDER_BY_LENGTH', '1') != '0'   # (A/B switch)

    # -- helpers
    def _interp_tables(self, sp):
        """numpy.interp(obsx, periods, values) as gather + two flops: left index j, x - xp[j],
        xp[j+1] - xp[j], and which observed periods sit exactly on the last solved one."""
        xp, x = sp.periods, sp.obsx
        j = np.clip(np.searchsorted(xp, x, side='right') - 1, 0, xp.size - 2)
        with torch.cuda.device(self.device):
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            return dict(j=dev(j.astype(np.int64)), xm=dev(x - xp[j]), dx=dev(xp[j + 1] - xp[j]),
                        last=dev(x == xp[-1]), n=xp.size)

    def _as_dev(self, x, dtype):
        if isinstance(x, torch.Tensor):
            if x.device != self.device or x.dtype != dtype or not x.is_contiguous():
                x = x.to(device=self.device, dtype=dtype).contiguous()
            return x
        return torch.from_numpy(np.ascontiguousarray(x)).to(device=self.device, dtype=dtype)

    def upload(self, H, VP, VS, RHO, nlay):
        """[B, Lmax] arrays (+ int nlay[B]) -> DeviceModels: one packed [B, 4, Lmax] device tensor,
        i.e. one contiguous 32*Lmax-byte block per model, which is what a lane fetches from the
        work queue."""
        f64 = torch.float64
        parts = [self._as_dev(x, f64) for x in (H, VP, VS, RHO)]
        packed = torch.stack(parts, dim=1).contiguous()
        known = isinstance(nlay, np.ndarray) and nlay.size
        depth = int(np.max(nlay)) if known else None
        mean = float(np.mean(np.clip(nlay, 0, None))) if known else None
        nlay = self._as_dev(nlay, torch.int32)
        return self.reorder(packed, nlay, depth=depth, mean_depth=mean)

    def reorder(self, packed, nlay, ragged=None, depth=None, mean_depth=None):
        """DeviceModels for a packed [B, 4, Lmax] device tensor, with a processing order for large
        batches.  Three keys, one sort (no data move; bh_swd_batch_ordered takes the permutation):
          1. deepest models first: the layer loop of a wave runs to its deepest model;
          2. longest searches first: persistent lanes run a handful of searches each, so a launch ends
             with a tail in which lanes wait for the last searches -- shorter if those are short ones.
             A search costs ~11 evaluations per period plus one per 0.005 km/s between the start value
             (0.855 c_R of the slowest layer) and the phase velocity at the longest period; the
             predictor is a depth-kernel average of vs at that period minus 0.79 min(vs), in classes of
             0.15 km/s (5 % at 524 288 ten-layer models, tools/divergence_probe.py);
          3. within a class by the S-wave travel time through the stack: the lanes of a wave run in
             lock step, so neighbours should be alike (10 % over a random order,
             profiles/r01_divergence_probe.txt).
        (`ragged` is accepted for compatibility.)"""
        order = None
        if self.sort_ragged and self.swd and packed.shape[0] > ORDER_MIN:
            B, L = packed.shape[0], packed.shape[2]
            keys = torch.empty(B, dtype=torch.int32, device=self.device)
            tmax = max(float(sp.periods.max()) for sp in self.swd)
            st = torch.cuda.current_stream(self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.bh_swd_order_keys(
                    B, L, 4 * L, nlay.data_ptr(), packed[:, 0, :].data_ptr(), packed[:, 2, :].data_ptr(),
                    tmax, 1 if self.order_by_length else 0, keys.data_ptr(), C.c_void_p(st.cuda_stream)))
            order = torch.argsort(keys).to(torch.int32)
        return DeviceModels(packed, nlay, order, depth, mean_depth)

    def alloc_out(self, B):
        out = torch.empty((B, self.row), dtype=torch.float64, device=self.device)
        err = torch.empty((B, max(1, len(self.swd))), dtype=torch.int32, device=self.device)
        return out, err

    def run(self, H, VP=None, VS=None, RHO=None, nlay=None, out=None, err=None, stream=None):
        """Launch all targets for the batch (asynchronous).  Either `run(models)` with resident
        DeviceModels (from `upload` / models.layers_from_voronoi) or `run(H, VP, VS, RHO, nlay)`
        with host or device arrays, which are packed first.  Returns (out[B,row], err[B,nswd])."""
        models = H if isinstance(H, DeviceModels) else self.upload(H, VP, VS, RHO, nlay)
        if models.packed.device != self.device:
            raise ValueError("models live on %s, engine on %s" % (models.packed.device, self.device))
        if models.order is None and self.sort_ragged and self.swd and models.B > ORDER_MIN:
            # resident models that came without a processing order (e.g. models.layers_from_voronoi):
            # computed once and kept with them
            models.order = self.reorder(models.packed, models.nlay).order
        H, VP, VS, RHO, nlay = models.H, models.VP, models.VS, models.RHO, models.nlay
        B = models.B
        mstride = 4 * models.Lmax
        Lmax = models.Lmax
        if models.depth is not None:                 # even, so that rows can be fetched two layers at a time
            Lmax = min(models.Lmax, models.depth + (models.depth % 2))
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        sp = C.c_void_p(st.cuda_stream)
        if out is None or err is None:
            # allocated on the launch stream (the caching allocator recycles a block for the stream
            # it was allocated on); a buffer the caller passed is kept, only the missing one is made
            with torch.cuda.stream(st):
                new_out, new_err = self.alloc_out(B)
            out = new_out if out is None else out
            err = new_err if err is None else err
        with torch.cuda.device(self.device):
            # The two kernels are independent.  swd_kernel ends with a tail in which its persistent
            # waves retire one by one; launched on a second stream, rf_kernel's workgroups take over
            # the freed SIMDs instead of waiting for the last search to finish.
            side = None
            if self.swd and self._rfp and self.overlap:
                side = self._side.get(st.cuda_stream)
                if side is None:
                    side = self._side[st.cuda_stream] = torch.cuda.Stream(device=self.device)
                side.wait_stream(st)
            if self.swd:
                need = self.lib.bh_swd_workspace_bytes(B, len(self.swd), self._tg)
                ws_ptr = None
                if need:
                    ws = self._ws.get(st.cuda_stream)
                    if ws is None or ws.numel() * 8 < need:
                        with torch.cuda.stream(st):      # the old block returns to this stream's pool
                            ws = self._ws[st.cuda_stream] = torch.empty((need + 7) // 8, dtype=torch.float64,
                                                                        device=self.device)
                    ws_ptr = ws.data_ptr()
                if models.mean_depth is not None:
                    _lib.check(self.lib.bh_swd_hint(models.mean_depth, 1))
                _lib.check(self.lib.bh_swd_batch_ordered(
                    B, Lmax, mstride, nlay.data_ptr(), H.data_ptr(), VP.data_ptr(), VS.data_ptr(),
                    RHO.data_ptr(), len(self.swd), self._tg, self.periods.data_ptr(),
                    out.data_ptr(), self.row, err.data_ptr(),
                    models.order.data_ptr() if models.order is not None else None, ws_ptr, need, sp))
                with torch.cuda.stream(st):
                    for sl, koff, tb in self._interp:      # > 60 periods: numpy.interp on the device
                        f = out[:, koff:koff + tb['n']]
                        f0, f1 = f.index_select(1, tb['j']), f.index_select(1, tb['j'] + 1)
                        y = ((f1 - f0) / tb['dx']) * tb['xm'] + f0
                        out[:, sl] = torch.where(tb['last'], f[:, -1:].expand(-1, y.shape[1]), y)
            else:
                # no dispersion kernel to raise BH_MODEL_BAD_DEPTH: the receiver-function kernel NaNs the
                # row of a model whose nlay is outside 1..Lmax, the flag is set here
                with torch.cuda.stream(st):
                    err.zero_()
                    err[:, 0] = torch.where((nlay < 1) | (nlay > Lmax), 2, 0).to(torch.int32)
            rsp = sp if side is None else C.c_void_p(side.cuda_stream)
            for rp in self._rfp:
                _lib.check(self.lib.bh_rf_batch(
                    B, Lmax, mstride, nlay.data_ptr(), H.data_ptr(), VP.data_ptr(), VS.data_ptr(),
                    RHO.data_ptr(), None, None, C.byref(rp), out.data_ptr(), self.row, None, 0, rsp))
            if side is not None:
                st.wait_stream(side)
                for t in (H, VP, VS, RHO, nlay, out):
                    t.record_stream(side)
        return out, err
