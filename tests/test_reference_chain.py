"""north_star: "SingleChain.run_chain / mcmcOptimizer are unchanged".  Where the reference tree is
available (development container) its own SingleChain.run_chain() -- unmodified, loaded file-wise --
is driven (a) with plugins that call the oracle directly and (b) with bayhunter_amd's broker
plugins (server process + shared-memory round trips, oracle injected as the server's back end so
that this runs without a GPU).  Same seed -> the two chains must be identical, sample for sample.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import reference_chain as rc  # noqa: E402
from broker_scenario import oracle_backend  # noqa: E402

pytestmark = pytest.mark.skipif(not rc.available(), reason='reference tree not present')
DATA = os.path.join(GOLDEN, 'tutorial_observed')


class OraclePlugin(object):
    """The plugin contract on top of the CPU oracle (test only)."""

    def __init__(self, oracle, x, kind):
        self.oracle, self.obsx, self.kind = oracle, x, kind

    def run_model(self, h, vp, vs, rho, **kw):
        if self.kind == 'swd':
            y, err = self.oracle.swd(h, vp, vs, rho, self.obsx, 2, 0)
            return (self.obsx, y) if err == 0 else (np.nan, np.nan)
        return self.obsx, self.oracle.rf_model(h, vp, vs, rho, nout=self.obsx.size)


def test_unmodified_reference_chain_through_broker(oracle):
    from bayhunter_amd.broker import ForwardBroker
    direct = rc.run_chain(lambda xs, xr: (OraclePlugin(oracle, xs, 'swd'), OraclePlugin(oracle, xr, 'rf')),
                          data_dir=DATA)
    broker = None

    def broker_plugins(xs, xr):
        nonlocal broker
        broker = ForwardBroker(swd=[('rdispph', xs)], rf=[('prf', xr)], max_clients=1, Lmax=21,
                               backend_factory=oracle_backend).start()
        s = broker.session()
        return s.plugin('rdispph', xs), s.plugin('prf', xr)
    try:
        via = rc.run_chain(broker_plugins, data_dir=DATA)
        st = broker.stats()
    finally:
        broker.stop()
    assert direct['n'] == via['n'] and direct['n'] > 10
    for k in ('models', 'likes', 'misfits', 'noise', 'vpvs'):
        assert np.array_equal(direct[k], via[k], equal_nan=True), k
    # one broker round trip per evaluated model: the second target's call hits the row cache
    assert 100 < st['models'] <= 181
