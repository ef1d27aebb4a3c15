"""Broker scenario run as a separate program (tests/test_broker.py): N chain-like processes each
evaluate their own sequence of models one at a time through BrokerSession plugins installed on
JointTarget objects (bayhunter_amd.targets), while a server coalesces them.

usage: broker_scenario.py <backend: gpu|oracle> <nclients> <niter> <out.npz>

The parent never touches the GPU (the server forks from it), like a BayHunter driver script."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayhunter_amd import targets as T  # noqa: E402
from bayhunter_amd.broker import ForwardBroker, gpu_backend  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402

PER = np.linspace(1, 41, 21)
TRF = np.linspace(-5, 35, 201)


def oracle_backend(swd, rf):
    """TEST back end: the CPU oracle instead of the GPU engine (protocol tests without a GPU)."""
    from oracle import pyoracle as po
    tags = {'rdispgr': (2, 1), 'ldispgr': (1, 1), 'rdispph': (2, 0), 'ldispph': (1, 0)}

    def run(H, VP, VS, RHO, nlay):
        cols, flags = [], []
        for s in swd:
            out, err, _ = po.swd_batch(H, VP, VS, RHO, nlay, s[1], *tags[s[0]])
            cols.append(out)
            flags.append(err)
        for r in rf:
            cols.append(po.rf_batch(H, VP, VS, RHO, nlay, nout=len(r[1])))
        return np.concatenate(cols, axis=1), np.stack(flags, axis=1)
    return run


def failing_model():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'likelihood.npz'))
    m = np.zeros((4, 8))
    m[:, :4] = g['model'][:, 5]
    return m[0], m[1], m[2], m[3], 4


def chain(idx, session, niter, q):
    try:
        _chain(idx, session, niter, q)
    except BaseException as e:      # report instead of leaving the parent waiting
        session.close()
        q.put((idx, e))


def _chain(idx, session, niter, q):
    """What a SingleChain does per iteration, reduced to the calls that hit the forward path."""
    rs = np.random.RandomState(100 + idx)
    sw_obs, rf_obs = rs.normal(3.5, .1, 21), rs.normal(0, .1, 201)
    t1 = T.RayleighDispersionPhase(PER, sw_obs)
    t2 = T.PReceiverFunction(TRF, rf_obs)
    t1.update_plugin(session.plugin('rdispph', PER))
    t2.update_plugin(session.plugin('prf', TRF))
    joint = T.JointTarget([t1, t2])
    H, VP, VS, RHO, nl = draw_models(niter, (2, 8), seed=200 + idx, sorted_vs=(idx % 2 == 0), Lmax=8)
    H[3], VP[3], VS[3], RHO[3], nl[3] = failing_model()   # a model SURF96 cannot solve (err = 1)
    likes = []
    for it in range(niter):
        n = nl[it]
        joint.evaluate(h=H[it, :n], vp=VP[it, :n], vs=VS[it, :n], noise=np.array([0, .02, 0, .01]))
        likes.append(joint.proposallikelihood)
    session.close()
    q.put((idx, np.array(likes)))


def main():
    backend, nclients, niter, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    factory = gpu_backend if backend == 'gpu' else oracle_backend
    broker = ForwardBroker(swd=[('rdispph', PER)], rf=[('prf', TRF)], max_clients=nclients, Lmax=8,
                           window=(2e-3 if backend == 'gpu' else 5e-3), backend_factory=factory).start()
    ctx = mp.get_context('fork')
    q = ctx.Queue()
    sessions = [broker.session() for _ in range(nclients)]
    procs = [ctx.Process(target=chain, args=(i, sessions[i], niter, q)) for i in range(nclients)]
    import time
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for v in res.values():
        if isinstance(v, BaseException):
            broker.stop()
            raise v
    for p in procs:
        p.join(timeout=60)
    st = broker.stats()
    broker.stop()
    np.savez(out, likes=np.stack([res[i] for i in range(nclients)]), launches=st['launches'],
             models=st['models'], mean_batch=st['mean_batch'])
    wall = time.perf_counter() - t0
    print('broker stats', st, '| %d chains x %d iterations in %.2f s = %.0f iterations/s (%.2f ms per launch)'
          % (nclients, niter, wall, nclients * niter / wall, 1e3 * st['busy_s'] / max(1, st['launches'])))


if __name__ == '__main__':
    main()
