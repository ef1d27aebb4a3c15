"""Batching broker: many one-model-at-a-time chain processes share one GPU.

The reference runs one OS process per Markov chain (src/mcmcOptimizer.py:248-252); each calls
`plugin.run_model(h, vp, vs, rho)` once per target and per iteration (src/Targets.py:78-82,323) and
blocks on the result.  One model per launch leaves an MI355X idle (a single search is a 2.5-14 ms
dependent chain), so a server process owns the device and coalesces the requests that are pending
at the same moment into one batched launch:

    chain process i                          server process (owns the GPU)
    ---------------                          -----------------------------
    BrokerSession.evaluate(h,vp,vs,rho)
      write model into slot i                wait until a request is pending, then until every
      state[i] = REQUEST; post `wake`        connected chain has one (or `window` seconds passed)
      block on done[i]                       gather slots -> one ForwardEngine.run (all targets)
                                             scatter rows/flags, state[i] = DONE, post done[i]
      read row; (x, y) per target

All targets of a joint inversion are computed by the first `run_model` of an iteration; the
plugins of the other targets find the row in the session cache (same model bytes), so one
iteration costs one round trip.  Chains stay exactly as they are: plugins are installed with the
reference's own hook, `target.update_plugin(session.plugin(target.ref, target.obsdata.x))`.

Shared state lives in `multiprocessing` RawArrays/semaphores created before the chains fork.  The
compute back end is a factory called inside the server process: the default builds a
ForwardEngine (GPU, no CPU fallback); tests inject their own.
"""
import multiprocessing as mp
import os
import sys
import time
import traceback

import numpy as np

IDLE, REQUEST, DONE = 0, 1, 2


def gpu_backend(swd, rf):
    """Default back end, constructed in the server process: (H, VP, VS, RHO, nlay) -> (out, err)."""
    import torch
    from .engine import ForwardEngine, RfSpec, SwdSpec
    eng = ForwardEngine(swd=[SwdSpec(*s) for s in swd], rf=[RfSpec(*r) for r in rf])

    def run(H, VP, VS, RHO, nlay):
        out, err = eng.run(H, VP, VS, RHO, nlay)
        torch.cuda.synchronize()
        return out[:, :eng.ncols].cpu().numpy(), err.cpu().numpy()
    return run


class BrokerError(RuntimeError):
    """The broker's server process failed or is gone; the chain cannot be evaluated any further."""


def _fail(S, text):
    """Server side: publish the failure and wake every client that is (or will be) waiting."""
    raw = text.encode('utf-8', 'replace')[-(len(S['errtext']) - 1):]
    S['errtext'][:len(raw)] = raw
    S['failed'].value = 1
    S['ready'].release()
    for sem in S['done']:
        sem.release()


def _server_alive(pid):
    """Is process `pid` running?  (A child that died but was not reaped yet is a zombie: not alive.)"""
    try:
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[0] not in ('Z', 'X')
    except (IOError, OSError, IndexError):
        return False


def _serve(shared, backend_factory, swd, rf):
    S = shared
    S['server_pid'].value = os.getpid()
    # A server that could not create its back end, or whose launch failed, ends with a non-zero exit
    # code (a supervisor / `_proc.exitcode` must not see a clean exit); KeyboardInterrupt and SystemExit
    # are reported to the waiting clients and then re-raised, not swallowed.
    try:
        run = backend_factory(swd, rf)
    except Exception:
        _fail(S, 'broker back end could not be created:\n' + traceback.format_exc())
        sys.exit(1)
    except BaseException:
        _fail(S, 'broker server interrupted while creating its back end:\n' + traceback.format_exc())
        raise
    model = np.frombuffer(S['model'], dtype=np.float64).reshape(S['n'], 4, S['Lmax'])
    rows = np.frombuffer(S['rows'], dtype=np.float64).reshape(S['n'], S['row'])
    flags = np.frombuffer(S['flags'], dtype=np.int32).reshape(S['n'], S['nflags'])
    nlay = np.frombuffer(S['nlay'], dtype=np.int32)
    state = np.frombuffer(S['state'], dtype=np.int32)
    stats = np.frombuffer(S['stats'], dtype=np.float64)      # launches, models, busy seconds
    S['ready'].release()
    while True:
        S['wake'].acquire()                                   # at least one request (or stop)
        if S['stop'].value:
            break
        t_first = time.perf_counter()
        while True:                                           # coalescing window
            pending = np.nonzero(state == REQUEST)[0]
            if pending.size >= max(1, S['connected'].value):
                break
            if time.perf_counter() - t_first > S['window']:
                break
            S['wake'].acquire(timeout=S['window'] / 4)
        pending = np.nonzero(state == REQUEST)[0]
        if pending.size == 0:
            continue
        t0 = time.perf_counter()
        m = model[pending]
        try:
            out, err = run(m[:, 0], m[:, 1], m[:, 2], m[:, 3], nlay[pending].copy())
            rows[pending] = out
            flags[pending] = err
        except Exception:
            # A failed launch (HIP error, out of memory, bad shapes) ends the server: a process that
            # has touched the GPU is never restarted in place.  Clients raise BrokerError; recovery
            # is a fresh ForwardBroker started from a process that has not used the GPU.
            _fail(S, 'broker launch of %d models failed:\n' % pending.size + traceback.format_exc())
            sys.exit(1)
        except BaseException:
            _fail(S, 'broker server interrupted during a launch of %d models:\n' % pending.size
                  + traceback.format_exc())
            raise
        stats[0] += 1
        stats[1] += pending.size
        stats[2] += time.perf_counter() - t0
        for i in pending:
            state[i] = DONE
            S['done'][i].release()


class ForwardBroker(object):
    """Server side.  swd: list of (ref, periods[, mode, flsph]); rf: list of
    (ref, obsx[, gauss, p, nsv]) -- the argument tuples of engine.SwdSpec / engine.RfSpec."""

    def __init__(self, swd=(), rf=(), max_clients=64, Lmax=32, window=3e-4,
                 backend_factory=gpu_backend):
        self.swd = [tuple(s) for s in swd]
        self.rf = [tuple(r) for r in rf]
        self.sizes = [len(s[1]) for s in self.swd] + [len(r[1]) for r in self.rf]
        self.refs = [s[0] for s in self.swd] + [r[0] for r in self.rf]
        self.axes = [np.asarray(s[1], dtype=np.float64) for s in self.swd] + \
                    [np.asarray(r[1], dtype=np.float64) for r in self.rf]
        row = int(sum(self.sizes))
        n = int(max_clients)
        ctx = mp.get_context('fork')
        self._ctx = ctx
        self.shared = dict(
            n=n, Lmax=int(Lmax), row=row, nflags=max(1, len(self.swd)), window=float(window),
            model=ctx.RawArray('d', n * 4 * int(Lmax)), rows=ctx.RawArray('d', n * row),
            flags=ctx.RawArray('i', n * max(1, len(self.swd))), nlay=ctx.RawArray('i', n),
            state=ctx.RawArray('i', n), stats=ctx.RawArray('d', 3),
            wake=ctx.Semaphore(0), ready=ctx.Semaphore(0), done=[ctx.Semaphore(0) for _ in range(n)],
            connected=ctx.Value('i', 0), stop=ctx.Value('i', 0), next_slot=ctx.Value('i', 0),
            failed=ctx.Value('i', 0), server_pid=ctx.Value('i', 0), errtext=ctx.RawArray('c', 4096))
        self._factory = backend_factory
        self._proc = None
        self.exitcode = None

    def start(self):
        """Fork the server.  Nothing in this process may have touched the GPU before: a forked child
        inherits a HIP runtime it cannot use (first launch faults or hangs)."""
        torch = sys.modules.get('torch')
        if self._factory is gpu_backend and torch is not None and torch.cuda.is_initialized():
            raise BrokerError("ForwardBroker.start() after this process initialised the GPU: start the "
                              "broker first (or from a fresh process), then create engines / chain pools")
        self._proc = self._ctx.Process(target=_serve, args=(self.shared, self._factory, self.swd, self.rf),
                                       daemon=True)
        self._proc.start()
        if not self.shared['ready'].acquire(timeout=300):
            raise BrokerError("broker server did not come up")
        if self.shared['failed'].value:
            raise BrokerError(self.error())
        return self

    def error(self):
        """The server's failure report ('' while it is healthy)."""
        return bytes(self.shared['errtext'][:]).split(b'\0', 1)[0].decode('utf-8', 'replace')

    def alive(self):
        return self._proc is not None and self._proc.is_alive() and not self.shared['failed'].value

    def session(self):
        """A handle for one chain; create it in the parent (or the chain) process, use it in one."""
        with self.shared['next_slot'].get_lock():
            slot = self.shared['next_slot'].value
            if slot >= self.shared['n']:
                raise RuntimeError("more sessions than max_clients")
            self.shared['next_slot'].value = slot + 1
        return BrokerSession(self.shared, slot, self.refs, self.axes, self.sizes, len(self.swd))

    def stats(self):
        launches, models, busy = np.frombuffer(self.shared['stats'], dtype=np.float64)
        return dict(launches=int(launches), models=int(models), busy_s=float(busy),
                    mean_batch=float(models / launches) if launches else 0.0)

    def stop(self):
        if self._proc is not None:
            self.shared['stop'].value = 1
            self.shared['wake'].release()
            self._proc.join(timeout=30)
            self.exitcode = self._proc.exitcode      # 0: stopped on request; non-zero: the server failed
            self._proc = None


class BrokerSession(object):
    """Client side, one per chain process."""

    def __init__(self, shared, slot, refs, axes, sizes, nswd):
        self.S, self.slot = shared, slot
        self.refs, self.axes, self.nswd = refs, axes, nswd
        self.offsets = np.concatenate(([0], np.cumsum(sizes)))
        self._key = None
        self._row = None
        self._flags = None
        self._open = False
        self.poll = 0.5          # seconds between liveness checks while waiting for a launch

    def _views(self):
        S = self.S
        model = np.frombuffer(S['model'], dtype=np.float64).reshape(S['n'], 4, S['Lmax'])
        rows = np.frombuffer(S['rows'], dtype=np.float64).reshape(S['n'], S['row'])
        flags = np.frombuffer(S['flags'], dtype=np.int32).reshape(S['n'], S['nflags'])
        nlay = np.frombuffer(S['nlay'], dtype=np.int32)
        state = np.frombuffer(S['state'], dtype=np.int32)
        return model, rows, flags, nlay, state

    def open(self):
        if not self._open:
            with self.S['connected'].get_lock():
                self.S['connected'].value += 1
            self._open = True

    def close(self):
        """Tell the server not to wait for this chain any more (call when the chain ends)."""
        if self._open:
            with self.S['connected'].get_lock():
                self.S['connected'].value -= 1
            self._open = False
            self.S['wake'].release()

    def evaluate(self, h, vp, vs, rho):
        """All targets for one model: (row, flags).  Blocks until the server's next launch."""
        h, vp, vs, rho = (np.ascontiguousarray(a, dtype=np.float64) for a in (h, vp, vs, rho))
        key = h.tobytes() + vp.tobytes() + vs.tobytes() + rho.tobytes()
        if key == self._key:
            return self._row, self._flags
        n = h.size
        if n > self.S['Lmax']:
            raise ValueError("model has more layers than the broker's Lmax")
        self.open()
        model, rows, flags, nlay, state = self._views()
        model[self.slot, :, :] = 0.0
        model[self.slot, 0, :n], model[self.slot, 1, :n] = h, vp
        model[self.slot, 2, :n], model[self.slot, 3, :n] = vs, rho
        nlay[self.slot] = n
        state[self.slot] = REQUEST
        self.S['wake'].release()
        # never wait blindly: if the server died (HIP error, OOM kill, ...) this chain must end with
        # an error, not hang -- the reference's mp_inversion would wait for it forever
        while not self.S['done'][self.slot].acquire(timeout=self.poll):
            if self.S['failed'].value or not _server_alive(self.S['server_pid'].value):
                break
        if self.S['failed'].value or state[self.slot] != DONE:
            state[self.slot] = IDLE
            text = bytes(self.S['errtext'][:]).split(b'\0', 1)[0].decode('utf-8', 'replace')
            raise BrokerError(text or "the broker's server process is gone (killed?)")
        state[self.slot] = IDLE
        self._key, self._row, self._flags = key, rows[self.slot].copy(), flags[self.slot].copy()
        return self._row, self._flags

    def plugin(self, ref, obsx=None):
        """A forward plugin (BayHunter's contract) for the target with this ref."""
        t = self.refs.index(ref)
        if obsx is not None and not np.array_equal(np.asarray(obsx, dtype=np.float64), self.axes[t]):
            raise ValueError("target '%s': observed axis differs from the broker's" % ref)
        return BrokerPlugin(self, t)


class BrokerPlugin(object):
    """run_model(h, vp, vs, rho) -> (x, y) through the broker; (nan, nan) when the solver found no
    root for this model -- the reference's failure convention (src/surf96_modsw.py:126)."""

    def __init__(self, session, t):
        self.session, self.t = session, t
        self.obsx = session.axes[t]
        self.ref = session.refs[t]
        self.modelparams = {}

    def set_modelparams(self, **mparams):
        raise RuntimeError("set the model parameters on the ForwardBroker's target specs")

    def run_model(self, h, vp, vs, rho, **params):
        row, flags = self.session.evaluate(h, vp, vs, rho)
        if self.t < self.session.nswd and flags[self.t] != 0:
            return np.nan, np.nan
        lo, hi = self.session.offsets[self.t], self.session.offsets[self.t + 1]
        return self.obsx, row[lo:hi].copy()
