"""Lock-step chain pool: many Markov chains advanced together, one batch of proposals per iteration.

The reference runs `SingleChain.run_chain()` in one process per chain (src/mcmcOptimizer.py:202-283)
and every `iterate()` (src/SingleChain.py:511-589) evaluates ONE model.  Here all chains of a pool
make their move at the same time:

    host (libbayhunter_amd, bh_chains_propose)    each chain draws its move from its own
                                                  RandomState stream, builds the proposal, checks priors
    GPU  (JointTarget.evaluate_batch)             forward models + likelihood of every valid proposal
    host (bh_chains_accept)                       u = log(uniform) per chain, accept / store / adapt

The per-chain arithmetic and the order of random draws are the reference's, so chain i of a pool is
the chain the reference would have produced with the same seed (tests/test_chains.py drives the
reference's own SingleChain next to it).  The pool is split into groups that alternate between the
host and the GPU: while one group's proposals are on the device, the other group is being accepted
and re-proposed on the host cores.

`ChainPool.save()` writes the reference's result files (src/SingleChain.py:608-690:
`c%03d_p{1,2}{models,likes,misfits,noise,vpvs}.npy` + `<station>_config.pkl`), so its
`PlotFromStorage` reads a GPU-produced inversion unchanged.
"""
import ctypes as C
import os
import pickle
import time

import numpy as np

from . import _lib

# src/defaults/defaults.ini of the reference (what utils.load_params returns for it)
DEFAULT_PRIORS = dict(mantle=None, vpvs=(1.5, 2.1), layers=(1, 20), vs=(1, 5), z=(0, 60), mohoest=None,
                      rfnoise_corr=(0.35, 0.75), rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0.,
                      swdnoise_sigma=(1e-5, 0.1))
DEFAULT_INITPARAMS = dict(nchains=3, iter_burnin=2048 * 2, iter_main=2048, propdist=(0.025, 0.025, 0.015, 0.005, 0.005),
                          acceptance=(40, 45), thickmin=0., lvz=None, hvz=None, rcond=None,
                          station='test', savepath='results/', maxmodels=50000)


LOOKAHEAD_MAX = 64        # BH_CHAIN_MAX_LOOKAHEAD
GROUPS_MIN_CHAINS = 32        # two chain groups (one batch on the device while the host works on the other) from here on:
                              # +11 ... +20 % from 32 to 1 024 chains, nothing below (profiles/r04_chain_groups.txt)
LOOKAHEAD_DEFAULT_MAX = 32    # (5 chains: 32 proposals each -- 130 models a call -- 2.36 s, 57 of them 2.84 s: profiles/r04_lookahead_tiny.txt)
LOOKAHEAD_ROWS, LOOKAHEAD_MID_MAX = 6500, 8
LOOKAHEAD_SCALE = 128.    # default look-ahead = LOOKAHEAD_SCALE / sqrt(chains per group) (profiles/r04_lookahead_sweep.txt)


def default_lookahead(per_group):
    """Proposals per chain and device call for a group of `per_group` chains on the GPU.  Small groups are bound by the
    latency of a call: LOOKAHEAD_SCALE / sqrt(chains), at most LOOKAHEAD_DEFAULT_MAX.  From a few hundred chains on the
    narrow team kernels are at their best with LOOKAHEAD_ROWS models a call: that many rows,
    at most LOOKAHEAD_MID_MAX proposals per chain; from ~8 000 chains per group a call is bound by throughput: 1.
    (profiles/r04_lookahead_sweep.txt, r04_lookahead_groups.txt: measured best at 32 / 128 / 512 / 1 024 / 2 048 /
    8 192 chains per group: 16-22 / 5-11 / 8 / 8 / 4 / 1.)"""
    if per_group <= 256:
        return max(1, min(LOOKAHEAD_DEFAULT_MAX, int(LOOKAHEAD_SCALE / np.sqrt(per_group))))
    return max(1, min(LOOKAHEAD_MID_MAX, -(-LOOKAHEAD_ROWS // per_group)))


def _is_number(x):
    return isinstance(x, (int, float, np.floating, np.integer))


class _CallEvaluator(object):
    """Adapter for a plain function (packed, nlay, noise) -> (logL, misfits)."""

    def __init__(self, fn):
        self.fn = fn

    def buffers(self, rows, Lmax, ntargets):
        return (np.zeros((rows, 4, Lmax)), np.zeros(rows, dtype=np.int32), np.zeros((rows, 2 * ntargets)),
                np.zeros(rows, dtype=np.int32))

    def submit(self, group, packed, nlay, noise):
        return self.fn(packed, nlay, noise)

    def collect(self, ticket):
        logL, misfits = ticket
        return np.ascontiguousarray(logL, dtype=np.float64), np.ascontiguousarray(misfits, dtype=np.float64)

    def release(self, packed):
        pass


class GpuEvaluator(object):
    """Proposals -> (logL, misfits) on the device, one evaluation plan of the library per chain group
    (evalplan.EvalPlan / bh_eval_*): the library writes the proposals straight into the plan's pinned
    staging block, `submit` is ONE call -- upload, processing order, dispersion / receiver-function /
    likelihood kernels and the download of 8*(ntargets+2) bytes per model are queued on the plan's own
    streams -- and `collect` waits for its event, so the batches of different groups overlap on the device
    as far as it has room.  (Until round 3 this was a sequence of torch calls per batch: as much host time
    as the device needed for the arithmetic at 4 096 chains.)"""

    def __init__(self, joint, device=None):
        self.joint = joint
        if device is not None:
            dev = device if isinstance(device, int) else getattr(device, 'index', None)
            if dev is None:
                dev = int(str(device).split(':')[1]) if ':' in str(device) else 0
            _lib.check(_lib.load().bh_set_device(int(dev)))
        self._plans = {}

    def buffers(self, rows, Lmax, ntargets):
        plan = self.joint.eval_plan(rows, Lmax)
        assert plan.T == ntargets
        self._plans[plan.packed.ctypes.data] = plan
        return plan.packed, plan.nlay, plan.noise, plan.chain

    def submit(self, group, packed, nlay, noise):
        if packed.shape[0] == 0:
            return None
        plan = self._plans[packed.ctypes.data]
        plan.submit(packed.shape[0])
        return plan

    def collect(self, ticket):
        if ticket is None:
            return np.zeros(0), np.zeros((0, self.joint.ntargets + 1))
        return ticket.wait()

    def release(self, packed):
        """Close the plan behind the staging block `packed` (a chain group is done with it)."""
        for key, plan in list(self._plans.items()):
            if plan.packed is packed:
                del self._plans[key]
                plan.close()

    def set_concurrency(self, n):
        for plan in self._plans.values():
            plan.set_concurrency(n)

    def close(self):
        for plan in self._plans.values():
            plan.close()
        self._plans = {}


class _Group(object):
    """A contiguous range of chains behind one bh_chain_pool handle."""

    def __init__(self, lib, cfg, seeds, arrays, first, last, Lmax, evaluator, ntargets, lookahead=1):
        self.lib, self.first, self.last = lib, first, last
        n = last - first
        self.n, self.Lmax = n, Lmax
        st = _lib.ChainStorage()
        st.nmodels = arrays['likes'].shape[1]
        keep = []
        for name in ('models', 'misfits', 'likes', 'noise', 'vpvs', 'iter'):
            part = arrays[name][first:last]
            assert part.flags['C_CONTIGUOUS']
            keep.append(part)
            setattr(st, name, part.ctypes.data)
        self._keep = keep
        self.seeds = np.ascontiguousarray(seeds[first:last], dtype=np.uint32)
        self.handle = C.c_void_p()
        _lib.check(lib.bh_chains_create(C.byref(cfg), n, self.seeds.ctypes.data, C.byref(st), C.byref(self.handle)))
        if lookahead != 1:
            _lib.check(lib.bh_chains_set_lookahead(self.handle, int(lookahead)))
        self.lookahead = int(lookahead)
        self.packed, self.nlay, self.noise, self.chain = evaluator.buffers(n * self.lookahead, Lmax, ntargets)
        self.count = 0
        self.ticket = None

    def propose(self):
        cnt = C.c_int(0)
        _lib.check(self.lib.bh_chains_propose(self.handle, self.Lmax, self.packed.ctypes.data, self.nlay.ctypes.data,
                                              self.noise.ctypes.data, self.chain.ctypes.data, C.byref(cnt)))
        self.count = cnt.value
        return self.count

    def accept(self, logL, misfits):
        assert logL.shape[0] == self.count and logL.dtype == np.float64 and misfits.dtype == np.float64
        logL, misfits = np.ascontiguousarray(logL), np.ascontiguousarray(misfits)
        _lib.check(self.lib.bh_chains_accept(self.handle, logL.ctypes.data, misfits.ctypes.data))

    def done(self):
        return bool(self.lib.bh_chains_done(self.handle))

    def close(self, evaluator=None):
        if self.handle:
            self.lib.bh_chains_destroy(self.handle)
            self.handle = None           # (not C.c_void_p(): at interpreter shutdown the module globals are gone)
        if evaluator is not None and self.packed is not None:
            if self.ticket is not None:                  # a batch still in flight: let it land before its plan goes
                try:
                    evaluator.collect(self.ticket)
                except Exception:
                    pass
                self.ticket = None
            if hasattr(evaluator, 'release'):
                evaluator.release(self.packed)
            self.packed = self.nlay = self.noise = self.chain = None


class ChainPool(object):
    """`ChainPool(targets, initparams, modelpriors, random_seed=...)`: the arguments of the
    reference's MCMC_Optimizer (src/mcmcOptimizer.py:35-75); `targets` is a bayhunter_amd JointTarget.

    nchains      overrides initparams['nchains']
    seeds        per-chain RandomState seeds; default: drawn like the reference's optimizer does,
                 `RandomState(random_seed).randint(1000)` per chain (:133-137)
    evaluator    object with buffers/submit/collect (default GpuEvaluator) or a plain function
                 (packed[B,4,Lmax], nlay[B], noise[B,2*ntargets]) -> (logL[B], misfits[B,ntargets+1])
    groups       number of chain groups alternating between host and GPU (default 2 when the pool
                 has at least 32 chains, else 1)
    nmodels      rows of sample storage per chain.  Default: the reference's
                 int(iterations * max(acceptance) / 100) (src/mcmcOptimizer.py:87-89) -- a chain that
                 accepts more than that overflows (IndexError there, an error from bh_chains_accept
                 here), which short runs do easily; `iterations + 1` can never overflow.
    lookahead    proposals per chain and device call (1 .. 64; bh_chains_set_lookahead): with more than one, a
                 chain also hands in the proposals of its following iterations for the likeliest outcomes of
                 the ones before and advances by as many iterations as the likelihoods confirm -- the same
                 samples in fewer, larger device calls.  Default: default_lookahead(chains per group)
                 when the evaluator is the GPU's (small pools are bound by the latency of a call more than by
                 its size: 32 proposals per chain up to 16 chains, 22 for 64, 8 for 1 024, 4 for 4 096, 1 from
                 16 384 chains on), 1 for any other evaluator.
    shard        (rank, world): this process runs only its contiguous block of the nchains chains
                 (distributed.shard_range), one process per GPU.  Seeds are drawn for ALL chains
                 first, so chain c is the same chain whatever the number of ranks; chains never
                 talk while sampling, `gather()` collects the sample blocks afterwards.

    Lifetime: a pool holds host threads, pinned memory, device buffers and streams (one evaluation
    plan per group).  `close()` releases them -- the sample arrays, `chain()`, `weighted()`, `final()`,
    `save()` and the gathers keep working on a closed pool -- and `with ChainPool(...) as pool:` does it
    at the end of the block.  The reference's equivalent is MCMC_Optimizer.mp_inversion returning
    (src/mcmcOptimizer.py:202-283): it can be called again, and so can this.
    """

    def __init__(self, targets, initparams=None, modelpriors=None, random_seed=None, nchains=None,
                 seeds=None, evaluator=None, groups=None, nthreads=None, shard=None, nmodels=None, lookahead=None):
        self.lib = _lib.load()
        self.targets = targets
        self.priors = dict(DEFAULT_PRIORS)
        self.priors.update(modelpriors or {})
        self.initparams = dict(DEFAULT_INITPARAMS)
        self.initparams.update(initparams or {})
        if nchains is None and seeds is not None:
            nchains = len(seeds)
        if nchains is not None:
            self.initparams['nchains'] = int(nchains)
        self.nchains = int(self.initparams['nchains'])
        self.ntargets = targets.ntargets
        if self.ntargets > _lib.MAX_TARGETS:
            raise ValueError("at most %d targets" % _lib.MAX_TARGETS)
        self.iter_burnin, self.iter_main = int(self.initparams['iter_burnin']), int(self.initparams['iter_main'])
        self.iterations = self.iter_burnin + self.iter_main
        self.maxlayers = int(self.priors['layers'][1]) + 1
        self.Lmax = max(8, self.maxlayers)
        if seeds is None:
            rstate = np.random.RandomState(random_seed)
            seeds = [rstate.randint(1000) for _ in range(self.nchains)]
        self.seeds = np.asarray(seeds, dtype=np.uint32)
        if self.seeds.size != self.nchains:
            raise ValueError("one seed per chain")
        self.nchains_total, self.first = self.nchains, 0
        if shard is not None:
            from .distributed import shard_range
            lo, hi = shard_range(self.nchains, int(shard[0]), int(shard[1]))
            if hi <= lo:
                raise ValueError("rank %d of %d has no chain to run (nchains = %d)" % (shard[0], shard[1], self.nchains))
            self.first, self.nchains, self.seeds = lo, hi - lo, self.seeds[lo:hi]
        self.cfg, corrfix, corr = self._config()
        # SingleChain._init_model_and_currentvalues -> set_target_covariance(corrfix[::2], inoise[::2], rcond):
        # a fixed correlation is the same number for every chain, a free one selects the exponential law
        targets.set_target_covariance(corrfix, corr, self.initparams['rcond'])
        # the reference's shared arrays (src/mcmcOptimizer.py:77-125)
        self.nmodels = int(self.iterations * np.max(self.initparams['acceptance']) / 100.) if nmodels is None \
            else int(nmodels)
        if self.nmodels < 1:
            raise ValueError("iterations * max(acceptance) / 100 leaves no room for accepted models")
        f32 = np.float32
        shape = (self.nchains, self.nmodels)
        self.models = np.full(shape + (self.maxlayers * 2,), np.nan, dtype=f32)
        self.misfits = np.full(shape + (self.ntargets + 1,), np.nan, dtype=f32)
        self.likes = np.full(shape, np.nan, dtype=f32)
        self.noise = np.full(shape + (self.ntargets * 2,), np.nan, dtype=f32)
        self.vpvs = np.full(shape, np.nan, dtype=f32)
        self.iter = np.full(shape, np.nan, dtype=np.float64)
        arrays = dict(models=self.models, misfits=self.misfits, likes=self.likes, noise=self.noise,
                      vpvs=self.vpvs, iter=self.iter)
        if evaluator is None:
            evaluator = GpuEvaluator(targets)
        elif not hasattr(evaluator, 'submit'):
            evaluator = _CallEvaluator(evaluator)
        self.evaluator = evaluator
        if groups is None:
            groups = 2 if self.nchains >= GROUPS_MIN_CHAINS else 1
        groups = max(1, min(int(groups), self.nchains))
        bounds = [(g * self.nchains) // groups for g in range(groups + 1)]
        if lookahead is None:
            per_group = max(1, self.nchains // groups)
            lookahead = default_lookahead(per_group) if isinstance(evaluator, GpuEvaluator) else 1
        self.lookahead = int(lookahead)
        if not 1 <= self.lookahead <= LOOKAHEAD_MAX:
            raise ValueError("lookahead: 1 .. %d proposals per chain and call" % LOOKAHEAD_MAX)
        self.groups = [_Group(self.lib, self.cfg, self.seeds, arrays, bounds[g], bounds[g + 1], self.Lmax,
                              evaluator, self.ntargets, self.lookahead) for g in range(groups)]
        if hasattr(evaluator, 'set_concurrency'):          # the groups' batches alternate on the device
            evaluator.set_concurrency(len(self.groups))
        if nthreads is not None:
            for g in self.groups:
                _lib.check(self.lib.bh_chains_set_threads(g.handle, int(nthreads)))
        self.evaluated = 0
        # where run() spends its wall time: host proposal / launch calls / waiting for the device /
        # host acceptance
        self.seconds = dict(propose=0.0, submit=0.0, wait=0.0, accept=0.0)
        self._finished = False
        self._closed_counters = None
        self._closed_advance = None

    def close(self):
        """Release the pool's native resources (idempotent); results stay readable."""
        groups = getattr(self, 'groups', ())
        if self._closed_counters is None and groups and all(g.handle for g in groups):
            self._closed_counters = self.counters()
            self._closed_advance = self.advance()
        for g in groups:
            g.close(getattr(self, 'evaluator', None))

    @property
    def closed(self):
        return not any(g.handle for g in getattr(self, 'groups', ()))

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

    # -- configuration -----------------------------------------------------------------------
    def _config(self):
        pr, ip = self.priors, self.initparams
        c = _lib.ChainConfig()
        c.ntargets = self.ntargets
        c.layers_min, c.layers_max = int(pr['layers'][0]), int(pr['layers'][1])
        c.vs_min, c.vs_max = float(pr['vs'][0]), float(pr['vs'][1])
        c.z_min, c.z_max = float(pr['z'][0]), float(pr['z'][1])
        if _is_number(pr['vpvs']):
            c.vpvs_fixed, c.vpvs_min, c.vpvs_max = 1, float(pr['vpvs']), float(pr['vpvs'])
        else:
            c.vpvs_fixed, c.vpvs_min, c.vpvs_max = 0, float(pr['vpvs'][0]), float(pr['vpvs'][1])
        if pr.get('mantle') is not None:
            c.has_mantle, c.mantle_vs, c.mantle_vpvs = 1, float(pr['mantle'][0]), float(pr['mantle'][1])
        if pr.get('mohoest') is not None:
            c.has_mohoest, c.moho_mean, c.moho_std = 1, float(pr['mohoest'][0]), float(pr['mohoest'][1])
        c.thickmin = float(ip['thickmin'])
        if ip.get('lvz') is not None:
            c.has_lvz, c.lvz = 1, float(ip['lvz'])
        if ip.get('hvz') is not None:
            c.has_hvz, c.hvz = 1, float(ip['hvz'])
        for i, v in enumerate(ip['propdist']):
            c.propdist[i] = float(v)
        c.acceptance[0], c.acceptance[1] = float(ip['acceptance'][0]), float(ip['acceptance'][1])
        c.iter_burnin, c.iter_main = self.iter_burnin, self.iter_main
        corrfix, corr = [], []
        for t, target in enumerate(self.targets.targets):
            for j, name in enumerate(('noise_corr', 'noise_sigma')):
                prior = pr[target.noiseref + name]
                k = 2 * t + j
                if _is_number(prior):
                    c.noise_fixed[k], c.noise_lo[k], c.noise_hi[k] = 1, float(prior), float(prior)
                else:
                    c.noise_fixed[k], c.noise_lo[k], c.noise_hi[k] = 0, float(prior[0]), float(prior[1])
                if j == 0:
                    corrfix.append(bool(c.noise_fixed[k]))
                    corr.append(c.noise_lo[k])
        return c, corrfix, corr

    # -- running -----------------------------------------------------------------------------
    def _launch(self, g):
        t0 = time.perf_counter()
        n = g.propose()
        t1 = time.perf_counter()
        self.evaluated += n
        g.ticket = self.evaluator.submit(g, g.packed[:n], g.nlay[:n], g.noise[:n])
        t2 = time.perf_counter()
        self.seconds['propose'] += t1 - t0
        self.seconds['submit'] += t2 - t1

    def _land(self, g):
        t0 = time.perf_counter()
        logL, misfits = self.evaluator.collect(g.ticket)
        t1 = time.perf_counter()
        g.ticket = None
        g.accept(logL, misfits)
        self.seconds['wait'] += t1 - t0
        self.seconds['accept'] += time.perf_counter() - t1

    def run(self, progress=None):
        """Initial models, then iter_burnin + iter_main iterations of every chain."""
        if self._finished:
            return self
        if self.closed:
            raise _lib.BayHunterAmdError("this chain pool has been closed")
        for g in self.groups:
            self._launch(g)                       # initial models of every group in flight
        live = list(self.groups)
        step = 0
        while live:
            for g in list(live):
                self._land(g)
                if g.done():
                    live.remove(g)
                else:
                    self._launch(g)
            step += 1
            if progress is not None and step % progress[0] == 0:
                progress[1](self)
        self._finished = True
        return self

    @property
    def iteration(self):
        """The iteration every chain has reached (with a look-ahead chains advance at their own pace)."""
        return min(self.lib.bh_chains_iteration(g.handle) for g in self.groups)

    def advance(self):
        """(device calls, chain iterations completed, models evaluated) of the iterations so far, initial models
        not counted; iterations / calls / nchains-per-group is what a call advances a chain by on average."""
        if self._closed_advance is not None and self.closed:
            return self._closed_advance
        tot = [0, 0, 0]
        for g in self.groups:
            v = [C.c_long(0) for _ in range(3)]
            _lib.check(self.lib.bh_chains_advance(g.handle, *[C.byref(x) for x in v]))
            tot = [a + x.value for a, x in zip(tot, v)]
        return tuple(tot)

    # -- results -----------------------------------------------------------------------------
    def counters(self):
        """naccepted[nchains], propdist / accepted / proposed [nchains, 5]."""
        if self._closed_counters is not None and self.closed:
            return self._closed_counters
        n = np.zeros(self.nchains, dtype=np.int64)
        pd, acc, pro = (np.zeros((self.nchains, 5)) for _ in range(3))
        for g in self.groups:
            sl = slice(g.first, g.last)
            parts = [np.ascontiguousarray(a[sl]) for a in (n, pd, acc, pro)]
            _lib.check(self.lib.bh_chains_counters(g.handle, *[p.ctypes.data for p in parts]))
            n[sl], pd[sl], acc[sl], pro[sl] = parts
        return n, pd, acc, pro

    def chain(self, i):
        """What a finished reference chain holds in chainmodels/chainlikes/... (rows up to n)."""
        n = int(self.counters()[0][i])
        return dict(n=n, models=self.models[i, :n], likes=self.likes[i, :n], misfits=self.misfits[i, :n],
                    noise=self.noise[i, :n], vpvs=self.vpvs[i, :n], iter=self.iter[i, :n])

    def weighted(self, i):
        """Residence-time weighting of one chain, SingleChain.run_chain's tail (:608-652) and
        ModelMatrix.get_weightedvalues (src/Models.py:228-274): an accepted model is repeated once
        per iteration it stayed current.  -> {1: (models, likes, misfits, noise, vpvs) | None, 2: ...}"""
        ch = self.chain(i)
        out = {}
        for phase, sel, final in ((1, ch['iter'] < 0, 0), (2, ch['iter'] >= 0, self.iter_main)):
            if not np.any(sel):
                out[phase] = None
                continue
            w = np.diff(np.concatenate((ch['iter'][sel], [final]))).astype(int)
            rep = lambda a, wide: np.repeat(a[sel].astype(np.float64) if wide else a[sel], w, axis=0)
            # the reference fills float64 matrices for models/misfits/noise and repeats the float32
            # vectors of likes and vpvs
            out[phase] = (rep(ch['models'], True), rep(ch['likes'], False), rep(ch['misfits'], True),
                          rep(ch['noise'], True), rep(ch['vpvs'], False))
        return out

    def gather(self, group=None, root=0, everywhere=False):
        """All ranks' sample blocks -> dict of [nchains_total, nmodels, ...] arrays (+ 'naccepted') on
        rank `root` (None on the others), or on every rank with everywhere=True.  The one exchange of
        a multi-GPU run (the reference keeps these blocks in shared memory, src/mcmcOptimizer.py:
        77-125): each rank sends its block once, point to point (distributed.gather_rows_to_root) --
        RCCL over xGMI with backend "nccl", gloo on CPU.  Bytes per rank: nchains_local * nmodels *
        (2*(maxlayers+1) + (ntargets+1) + 1 + 2*ntargets + 1 + 2) * 4 (iter is float64)."""
        import torch
        import torch.distributed as dist
        from .distributed import gather_rows, gather_rows_to_root
        blocks = dict(models=self.models, misfits=self.misfits, likes=self.likes, noise=self.noise,
                      vpvs=self.vpvs, iter=self.iter, naccepted=self.counters()[0])
        if not dist.is_initialized():
            return blocks
        dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else 'cpu'
        out = {}
        for k, v in blocks.items():
            t = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
            g = gather_rows(t, self.nchains_total, group) if everywhere else \
                gather_rows_to_root(t, self.nchains_total, root, group)
            out[k] = None if g is None else g.cpu().numpy()
        return None if out['likes'] is None else out

    def final(self, i, maxmodels=None):
        """What save() writes for chain i's main phase: residence-time weighted samples thinned to at
        most `maxmodels` rows (SingleChain.py:646-663, :665-690), as float32 [rows, width] with the
        columns models | misfits | likes | noise | vpvs -- or an empty array if nothing was accepted
        after the burn-in."""
        w = self.weighted(i)[2]
        width = self.models.shape[2] + self.misfits.shape[2] + 1 + self.noise.shape[2] + 1
        if w is None:
            return np.zeros((0, width), dtype=np.float32)
        maxmodels = maxmodels or self.initparams['maxmodels']
        thin = int(np.ceil(float(w[1].size) / float(maxmodels)))
        models, likes, misfits, noise, vpvs = (a[::thin] for a in w)
        return np.concatenate([models, misfits, likes[:, None], noise, vpvs[:, None]], axis=1).astype(np.float32)

    def gather_final(self, maxmodels=None, group=None, root=0):
        """Thin BEFORE the exchange: every rank weights and thins its chains like save() does and
        sends only those rows (<= maxmodels per chain; src/Plotting.py:198-219 keeps no more for the
        merged posterior).  Rank `root` gets one float32 [rows, width] array per chain (global chain
        order; columns as in `final`), the others None."""
        import torch
        import torch.distributed as dist
        from .distributed import gather_ragged_to_root
        mine = [self.final(i, maxmodels) for i in range(self.nchains)]
        if not dist.is_initialized():
            return mine
        dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else 'cpu'
        width = mine[0].shape[1] if mine else 0
        rows = np.array([m.shape[0] for m in mine], dtype=np.int64)
        flat = np.concatenate(mine, axis=0) if mine else np.zeros((0, width), dtype=np.float32)
        counts = gather_ragged_to_root(torch.from_numpy(rows).to(dev), root, group)
        data = gather_ragged_to_root(torch.from_numpy(np.ascontiguousarray(flat)).to(dev), root, group)
        if data is None:
            return None
        out = []
        for c, d in zip(counts, data):
            d, lo = d.cpu().numpy(), 0
            for n in c.cpu().numpy():
                out.append(d[lo:lo + int(n)])
                lo += int(n)
        return out

    def save(self, savepath=None, chainidx_offset=None):
        """Write <savepath>/data/c%03d_p{1,2}{models,likes,misfits,noise,vpvs}.npy like
        SingleChain.save_finalmodels (:654-690), thinned to initparams['maxmodels'] main-phase
        models per chain, and <station>_config.pkl like utils.save_config (src/utils.py:127-153).
        (The reference's PlotFromStorage globs `c???_...`: it sees chain numbers up to 999.)"""
        savepath = savepath or self.initparams['savepath']
        if chainidx_offset is None:
            chainidx_offset = self.first          # a rank names its files by the global chain index
        data = os.path.join(savepath, 'data')
        os.makedirs(data, exist_ok=True)
        names = ('models', 'likes', 'misfits', 'noise', 'vpvs')
        written = 0
        for i in range(self.nchains):
            w = self.weighted(i)
            thinning = 1
            if w[2] is not None:
                thinning = int(np.ceil(float(w[2][1].size) / float(self.initparams['maxmodels'])))
            for phase in (1, 2):
                if w[phase] is None:
                    continue
                for name, arr in zip(names, w[phase]):
                    np.save(os.path.join(data, 'c%.3d_p%d%s' % (i + chainidx_offset, phase, name)), arr[::thinning])
                    written += 1
        for target in self.targets.targets:
            target.get_covariance = None
        cfg = dict(targets=self.targets.targets, targetrefs=[t.ref for t in self.targets.targets],
                   priors=self.priors, initparams=self.initparams)
        with open(os.path.join(data, '%s_config.pkl' % self.initparams['station']), 'wb') as f:
            pickle.dump(cfg, f)
        corrfix = [bool(self.cfg.noise_fixed[2 * t]) for t in range(self.ntargets)]
        corr = [self.cfg.noise_lo[2 * t] for t in range(self.ntargets)]
        self.targets.set_target_covariance(corrfix, corr, self.initparams['rcond'])
        return written
