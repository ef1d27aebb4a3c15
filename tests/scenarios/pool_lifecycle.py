"""The sampler's real lifecycle in ONE process (GPU box): pools and evaluation plans made, run and closed
back to back with forward launches on other streams in between -- what `MCMC_Optimizer.mp_inversion`
being callable more than once is for the reference (src/mcmcOptimizer.py:202-283).

Run by tests/test_gpu_chains.py in a child process with a SMALL ring of work-queue slots
(BH_SWD_QUEUE_SLOTS), so that every launch claims a slot whose guard event was last recorded by an earlier
pool -- on a stream that no longer exists unless the plan retired it (capi.hip: retire_stream).  Round 3's
library failed exactly here ("hipEventQuery(queue slot): operation not permitted when stream is capturing"
in the driver's bench run); a crash of this script is the same bug.

Prints one JSON line.  usage: pool_lifecycle.py [npools] [nchains] [iterations]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(npools=5, nchains=600, iterations=40):
    import torch
    from bayhunter_amd import _lib
    from bayhunter_amd.chains import ChainPool, GpuEvaluator
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
    from bayhunter_amd.synthetic import draw_models
    from chain_scenario import CASES, joint_target
    lib = _lib.load()
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    case = CASES['tutorial']
    ip = dict(case['initparams'], iter_burnin=iterations - 10, iter_main=10, acceptance=(40, 100))
    seeds = np.arange(nchains) % 1000
    per = np.linspace(1, 41, 21)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', per)], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    H, VP, VS, RHO, nl = draw_models(20000, 10, seed=3)          # enough searches for the queued lane kernel
    models = eng.upload(H, VP, VS, RHO, nl)
    ref_out, ref_err = (t.cpu().numpy() for t in eng.run(models))
    torch.cuda.synchronize()
    rec = dict(pools=0, plans_closed=0, forward_checks=0, forms=[])

    def forward_check(stream=None):
        out, err = eng.run(models, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref_out, equal_nan=True) and np.array_equal(err.cpu().numpy(), ref_err)
        rec['forward_checks'] += 1
        rec['forms'].append(int(lib.bh_swd_last_form()))

    # a closed pool refuses to run
    joint = joint_target(data)
    unrun = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=seeds[:8], evaluator=GpuEvaluator(joint))
    unrun.close()
    try:
        unrun.run()
        raise AssertionError('a closed pool ran')
    except _lib.BayHunterAmdError:
        pass

    first = None
    for k in range(npools):
        joint = joint_target(data)
        if k % 2 == 0:                                   # context manager ...
            with ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=seeds, groups=2,
                           evaluator=GpuEvaluator(joint)) as pool:
                pool.run()
                plans = list(pool.evaluator._plans.values())
                assert len(plans) == 2 and not any(p.closed for p in plans)
        else:                                            # ... or explicit close(), twice
            pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=seeds, groups=2,
                             evaluator=GpuEvaluator(joint))
            pool.run()
            plans = list(pool.evaluator._plans.values())
            pool.close()
            pool.close()
        assert pool.closed and all(p.closed for p in plans) and not pool.evaluator._plans
        rec['plans_closed'] += len(plans)
        # a closed pool keeps its results
        got = {k2: pool.chain(0)[k2].copy() for k2 in ('models', 'likes', 'iter')}
        assert pool.chain(0)['n'] >= 1 and pool.counters()[0].shape == (nchains,)
        if first is None:
            first = got
        else:                                            # the same seeds: every pool is the same inversion
            for k2 in got:
                assert np.array_equal(first[k2], got[k2], equal_nan=True), (k, k2)
        rec['pools'] += 1
        # forward launches between the pools: on torch's current stream and on a fresh side stream; with the
        # small ring each of them claims slots the pool's (now destroyed) streams used last
        forward_check()
        forward_check(torch.cuda.Stream())
        del pool, plans

    # the C ABI's own contract: a caller's stream, used, retired, destroyed; then the ring goes round again
    st = C.c_void_p()
    _lib.check(lib.bh_stream_create(C.byref(st)))
    B, L = 4096, 10
    sizes = dict(m=B * 4 * L * 8, nl=B * 4, per=21 * 8, out=B * 21 * 8, err=B * 4)
    d = {}
    for k2, n in sizes.items():
        p = C.c_void_p()
        _lib.check(lib.bh_malloc(C.byref(p), n))
        d[k2] = p
    packed = np.ascontiguousarray(np.stack([H[:B], VP[:B], VS[:B], RHO[:B]], axis=1))
    nl32 = np.ascontiguousarray(nl[:B], dtype=np.int32)
    _lib.check(lib.bh_memcpy_h2d(d['m'], packed.ctypes.data, sizes['m'], st))
    _lib.check(lib.bh_memcpy_h2d(d['nl'], nl32.ctypes.data, sizes['nl'], st))
    _lib.check(lib.bh_memcpy_h2d(d['per'], per.ctypes.data, sizes['per'], st))
    tg = (_lib.SwdTarget * 1)(_lib.SwdTarget(2, 0, 1, 0, 21, 0, 0, 0))
    base = d['m'].value
    for _ in range(12):                                  # more launches than the test's ring has slots
        _lib.check(lib.bh_swd_batch(B, L, 4 * L, d['nl'], base, base + 8 * L, base + 16 * L, base + 24 * L, 1, tg,
                                    d['per'], d['out'], 21, d['err'], None, 0, st))
    out = np.empty((B, 21))
    _lib.check(lib.bh_memcpy_d2h(out.ctypes.data, d['out'], sizes['out'], st))
    _lib.check(lib.bh_stream_retire(st))                 # waits for the stream, drops the library's events on it
    _lib.check(lib.bh_stream_destroy(st))                # (retires once more -- nothing left -- and destroys)
    assert np.array_equal(out, ref_out[:B, :21], equal_nan=True)
    for _ in range(3):
        forward_check()
    for p in d.values():
        _lib.check(lib.bh_free(p))
    _lib.check(lib.bh_stream_retire(None))               # the null stream: nothing to do
    rec['slots'] = os.environ.get('BH_SWD_QUEUE_SLOTS')
    rec['ok'] = True
    print(json.dumps(rec))


if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:4]])
