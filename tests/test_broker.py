"""Batching broker: chain-like processes evaluating one model at a time share
one engine.  CPU tier: protocol, caching, failure convention and coalescing with the oracle as the
injected back end.  GPU tier: the same scenario on the real engine, in a program whose parent
never touches the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

SCEN = os.path.join(ROOT, 'tests', 'scenarios', 'broker_scenario.py')


def _expected(oracle, nclients, niter):
    """The same evaluations without a broker: oracle synthetics + host JointTarget."""
    sys.path.insert(0, os.path.dirname(SCEN))
    from bayhunter_amd import targets as T
    from bayhunter_amd.synthetic import draw_models
    from broker_scenario import failing_model
    per, trf = np.linspace(1, 41, 21), np.linspace(-5, 35, 201)

    class Fixed(object):
        def __init__(self, x):
            self.x, self.y = x, None

        def run_model(self, h, vp, vs, rho, **kw):
            return (self.x, self.y) if self.y is not None else (np.nan, np.nan)
    want = np.zeros((nclients, niter))
    for idx in range(nclients):
        rs = np.random.RandomState(100 + idx)
        t1 = T.RayleighDispersionPhase(per, rs.normal(3.5, .1, 21))
        t2 = T.PReceiverFunction(trf, rs.normal(0, .1, 201))
        p1, p2 = Fixed(per), Fixed(trf)
        t1.update_plugin(p1)
        t2.update_plugin(p2)
        joint = T.JointTarget([t1, t2])
        H, VP, VS, RHO, nl = draw_models(niter, (2, 8), seed=200 + idx, sorted_vs=(idx % 2 == 0), Lmax=8)
        H[3], VP[3], VS[3], RHO[3], nl[3] = failing_model()
        for it in range(niter):
            n = nl[it]
            y, err = oracle.swd(H[it, :n], VP[it, :n], VS[it, :n], RHO[it, :n], per, 2, 0)
            p1.y = y if err == 0 else None
            p2.y = oracle.rf_model(H[it, :n], VP[it, :n], VS[it, :n], RHO[it, :n], nout=201)
            joint.evaluate(h=H[it, :n], vp=VP[it, :n], vs=VS[it, :n], noise=np.array([0, .02, 0, .01]))
            want[idx, it] = joint.proposallikelihood
    return want


def _run(backend, nclients, niter, tmp):
    out = os.path.join(tmp, 'broker_%s.npz' % backend)
    subprocess.run([sys.executable, SCEN, backend, str(nclients), str(niter), out], check=True, timeout=400)
    return np.load(out)


def test_broker_protocol_with_oracle_backend(oracle, tmp_path):
    nclients, niter = 6, 12
    got = _run('oracle', nclients, niter, str(tmp_path))
    want = _expected(oracle, nclients, niter)
    assert np.array_equal(got['likes'], want)            # same numbers, same -1e15 failures
    assert (want == -1e15).any()
    assert got['models'] == nclients * niter              # one round trip per iteration (row cache)
    assert got['mean_batch'] > 1.5                        # requests really were coalesced (loaded hosts: generous)


@pytest.mark.gpu
def test_broker_on_gpu(oracle, tmp_path):
    nclients, niter = 16, 10
    got = _run('gpu', nclients, niter, str(tmp_path))
    want = _expected(oracle, nclients, niter)
    fail = want == -1e15
    assert np.array_equal(got['likes'] == -1e15, fail)
    assert np.allclose(got['likes'][~fail], want[~fail], rtol=1e-6)
    assert got['mean_batch'] > 4.0
