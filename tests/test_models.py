"""Voronoi nuclei -> layers + prior checks (src/Models.py:26-52, src/SingleChain.py:330-392)
against golden vectors from the reference's own modules (tests/golden/make_golden_models.py).
CPU tier: the NumPy stand-in; GPU tier: bh_voronoi_to_layers, bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

SETUPS = {
    'defaults': (dict(layers=(1, 20), vs=(1, 5), z=(0, 60)), 0., None, None, None),
    'tutorial': (dict(layers=(1, 20), vs=(2, 5), z=(0, 60)), 0.1, None, None, (4.3, 1.8)),
    'zones': (dict(layers=(2, 8), vs=(1.5, 4.8), z=(1, 55)), 0.5, 0.1, 0.3, None),
}


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'voronoi.npz'))


@pytest.mark.parametrize('name', sorted(SETUPS))
def test_numpy_model_matches_reference(g, name):
    from bayhunter_amd.models import Model, valid_model
    priors, thickmin, lvz, hvz, mantle = SETUPS[name]
    for b in range(g['nlay'].size):
        n = g['nlay'][b]
        model = np.concatenate((g['VSN'][b, :n], g['ZV'][b, :n], [np.nan] * 3))   # NaN padding
        vp, vs, h = Model.get_vp_vs_h(model, g['vpvs'][b], mantle)
        assert np.array_equal(h, g[name + '_H'][b, :n]) and np.array_equal(vp, g[name + '_VP'][b, :n])
        assert np.array_equal(vs, g['VSN'][b, :n])
        assert int(valid_model(model, g['vpvs'][b], priors, thickmin, lvz, hvz, mantle)) == g[name + '_valid'][b]
    assert 0 < g[name + '_valid'].sum() < g['nlay'].size


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(SETUPS))
def test_gpu_voronoi_bitexact(lib, g, name):
    from bayhunter_amd.models import layers_from_voronoi
    priors, thickmin, lvz, hvz, mantle = SETUPS[name]
    VSN, ZV = np.nan_to_num(g['VSN'], nan=-7.0), np.nan_to_num(g['ZV'], nan=-7.0)   # padding is ignored
    models, valid = layers_from_voronoi(VSN, ZV, g['nlay'], g['vpvs'], priors, thickmin, lvz, hvz, mantle)
    H, VP, VS, RHO, valid = (t.cpu().numpy() for t in (models.H, models.VP, models.VS, models.RHO, valid))
    assert np.array_equal(valid, g[name + '_valid'])
    assert np.array_equal(H, g[name + '_H']) and np.array_equal(VP, g[name + '_VP'])
    live = np.arange(H.shape[1])[None, :] < g['nlay'][:, None]
    assert np.array_equal(VS[live], g['VSN'][live]) and np.all(VS[~live] == 0)
    assert np.array_equal(RHO[live], (g[name + '_VP'] * 0.32 + 0.77)[live]) and np.all(RHO[~live] == 0)


@pytest.mark.gpu
def test_gpu_voronoi_feeds_engine(lib, oracle):
    """The packed block goes straight into the forward kernels (no repacking)."""
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    from bayhunter_amd.models import layers_from_voronoi
    rs = np.random.RandomState(5)
    B, L = 64, 6
    VSN = np.sort(rs.uniform(2, 5, (B, L)), axis=1)
    ZV = np.sort(rs.uniform(0, 60, (B, L)), axis=1)
    nl = np.full(B, L, dtype=np.int32)
    models, valid = layers_from_voronoi(VSN, ZV, nl, np.full(B, 1.73),
                                        dict(layers=(1, 20), vs=(2, 5), z=(0, 60)), 0.1)
    per = np.linspace(1, 41, 21)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', per)])
    out, err = eng.run(models)
    want, werr, _ = oracle.swd_batch(models.H.cpu().numpy(), models.VP.cpu().numpy(),
                                     models.VS.cpu().numpy(), models.RHO.cpu().numpy(), nl, per, 2, 0)
    assert np.array_equal(err.cpu().numpy()[:, 0], werr)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.gpu
def test_full_device_pipeline_nuclei_to_loglikelihood(lib, oracle):
    """Voronoi nuclei in, log-likelihood out, nothing else crosses the PCIe bus: voronoi_kernel ->
    swd/rf kernels -> like_kernel, against the host path (NumPy Model + oracle synthetics +
    JointTarget.evaluate)."""
    from bayhunter_amd import targets as T
    from bayhunter_amd.models import Model, layers_from_voronoi
    rs = np.random.RandomState(9)
    B, L = 96, 7
    nl = rs.randint(2, L + 1, size=B).astype(np.int32)
    VSN = np.sort(rs.uniform(2, 5, (B, L)), axis=1)
    ZV = np.sort(rs.uniform(0, 60, (B, L)), axis=1)
    vpvs = rs.uniform(1.6, 1.9, size=B)
    per, trf = np.linspace(1, 41, 21), np.linspace(-5, 35, 201)
    t1 = T.RayleighDispersionPhase(per, rs.normal(3.5, .2, 21))
    t2 = T.PReceiverFunction(trf, rs.normal(0, .05, 201))
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([True, False], [0.0, 0.6])
    noise = np.stack([np.zeros(B), rs.uniform(.01, .05, B), rs.uniform(.3, .8, B), rs.uniform(.005, .02, B)], 1)
    models, valid = layers_from_voronoi(VSN, ZV, nl, vpvs, dict(layers=(1, 20), vs=(2, 5), z=(0, 60)), 0.1)
    logL, mis = joint.evaluate_batch(models, noise=noise)
    logL, mis = logL.cpu().numpy(), mis.cpu().numpy()

    class Fixed(object):
        def __init__(self, x):
            self.x, self.y = x, None

        def run_model(self, h, vp, vs, rho, **kw):
            return (self.x, self.y) if self.y is not None else (np.nan, np.nan)
    p1, p2 = Fixed(per), Fixed(trf)
    t1.update_plugin(p1)
    t2.update_plugin(p2)
    for b in range(0, B, 5):
        n = nl[b]
        vp, vs, h = Model.get_vp_vs_h(np.concatenate((VSN[b, :n], ZV[b, :n])), vpvs[b])
        rho = vp * 0.32 + 0.77
        y, err = oracle.swd(h, vp, vs, rho, per, 2, 0)
        p1.y = y if err == 0 else None
        p2.y = oracle.rf_model(h, vp, vs, rho, nout=201)
        joint.evaluate(h=h, vp=vp, vs=vs, noise=noise[b])
        assert np.isclose(logL[b], joint.proposallikelihood, rtol=1e-8)
        assert np.allclose(mis[b], joint.proposalmisfits, rtol=1e-7)
