"""Evaluation plan: a batch of proposals -> (logL, misfits) in ONE call into libbayhunter_amd.

The hand-over between the sampler and the device (reference: one `JointTarget.evaluate` per chain and
iteration, src/SingleChain.py:545-552 -> src/Targets.py:314-347).  `bh_eval_submit` copies the pinned
staging block the proposals were written into, orders the batch, launches the dispersion, receiver-function
and likelihood kernels on the plan's own streams and copies 8*(ntargets+2) bytes per model back; the
plan owns every buffer (include/bayhunter_amd.h, "evaluation plan").  Nothing here imports torch.
"""
import ctypes as C

import numpy as np

from . import _lib


def _view(ptr, ctype, count):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))


class EvalPlan(object):
    """`layout` = JointTarget.batch_layout().  Host views of the plan's pinned staging block:
    packed[rows, 4, Lmax], nlay[rows], noise[rows, 2*ntargets], chain[rows]; `submit(count)` evaluates
    the first `count` rows, `wait()` returns (logL[count], misfits[count, ntargets+1]) -- views that stay
    valid until the next submit.

    A plan owns device memory, pinned host memory, two streams and events: `close()` it (or use it as a
    context manager) when the sampler is done with it.  `__del__` only backs that up -- when the interpreter
    collects an object is not something to hang device resources on."""

    def __init__(self, layout, max_models, Lmax, use_mfma=True):
        self.lib = _lib.load()
        lay, desc = layout['layout'], layout['desc']
        self.rows, self.Lmax, self.T = int(max_models), int(Lmax), len(desc)
        self._keep = [lay.periods, layout['yobs'], layout['aux']]
        interp = (_lib.EvalInterp * max(1, len(lay.resampled)))()
        for i, (t, sl, _, sp) in enumerate(lay.resampled):
            self._keep.append(sp.obsx)
            interp[i] = _lib.EvalInterp(t, sl.start, sl.stop - sl.start, 0, sp.obsx.ctypes.data)
        rfp = (_lib.RfParams * max(1, len(lay.rfp)))(*lay.rfp)
        self.handle = C.c_void_p()
        _lib.check(self.lib.bh_eval_create(
            self.rows, self.Lmax, lay.row, len(lay.swd), lay.tg, lay.periods.ctypes.data, lay.periods.size,
            len(lay.rfp), rfp, self.T, desc, layout['nflags'], layout['yobs'].ctypes.data, layout['aux'].ctypes.data,
            layout['aux'].size, len(lay.resampled), interp, 1 if use_mfma else 0, C.byref(self.handle)))
        p = [C.c_void_p() for _ in range(5)]
        _lib.check(self.lib.bh_eval_buffers(self.handle, *[C.byref(x) for x in p]))
        R, L, T = self.rows, self.Lmax, self.T
        self.packed = _view(p[0], C.c_double, R * 4 * L).reshape(R, 4, L)
        self.nlay = _view(p[1], C.c_int32, R)
        self.noise = _view(p[2], C.c_double, R * 2 * T).reshape(R, 2 * T)
        self.chain = _view(p[3], C.c_int32, R)
        self._results = _view(p[4], C.c_double, R * (T + 2))

    def _live(self):
        if not self.handle:
            raise _lib.BayHunterAmdError("this evaluation plan has been closed")
        return self.handle

    def submit(self, count):
        _lib.check(self.lib.bh_eval_submit(self._live(), int(count)))

    def set_concurrency(self, plans_in_flight):
        """How many plans take turns on the device (the chain groups of a pool): the library chooses its kernel
        forms for that load."""
        _lib.check(self.lib.bh_eval_set_concurrency(self._live(), int(plans_in_flight)))

    def wait(self):
        self._live()
        n = C.c_int(0)
        _lib.check(self.lib.bh_eval_wait(self.handle, C.byref(n)))
        n, T = n.value, self.T
        return self._results[:n], self._results[n:n * (T + 2)].reshape(n, T + 1)

    def close(self):
        """Wait for the plan's streams, then free everything it owns (idempotent).  The host views
        (packed, nlay, noise, chain, results) point into freed pinned memory afterwards and are dropped."""
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            self.packed = self.nlay = self.noise = self.chain = self._results = None
            self.lib.bh_eval_destroy(h)

    @property
    def closed(self):
        return not self.handle

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
