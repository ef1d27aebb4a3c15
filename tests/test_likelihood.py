"""Likelihood tail of the hot path (JointTarget.evaluate, src/Targets.py:314-347) against golden
vectors produced by the reference's own Targets.py (tests/golden/make_golden_likelihood.py).

CPU tier: the host mirror bayhunter_amd.targets with the synthetics of the fixture fed through the
plugin hook -- checks the closed-form covariance algebra, the covariance selection rule and the
-1e15 failure convention.  GPU tier: evaluate_batch = forward kernels + fused likelihood kernel."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = {
    'nocorr_gauss': (False, [(True, 0.0), (True, 0.98)]),
    'scaled_exp': (True, [(True, 0.0), (False, 0.85)]),
    'exp_exp': (False, [(False, 0.3), (False, 0.92)]),
    'nocorr_nocorr': (False, [(True, 0.0), (True, 0.0)]),
}
RTOL = 1e-9     # numpy's dot in the reference sums in a different order; data differ by <= 1e-13


class Fixed(object):
    def __init__(self, x):
        self.x, self.y = x, None

    def run_model(self, h, vp, vs, rho, **kw):
        return (self.x, self.y) if self.y is not None else (np.nan, np.nan)


def _targets(g, with_yerr):
    from bayhunter_amd import targets as T
    t1 = T.RayleighDispersionPhase(g['sw_x'], g['sw_y'], yerr=g['yerr_sw'] if with_yerr else None)
    t2 = T.PReceiverFunction(g['rf_x'], g['rf_y'])
    return T, t1, t2


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'likelihood.npz'))


@pytest.mark.parametrize('case', sorted(CASES))
def test_host_joint_target_matches_reference(g, case):
    with_yerr, setup = CASES[case]
    T, t1, t2 = _targets(g, with_yerr)
    p1, p2 = Fixed(g['sw_x']), Fixed(g['rf_x'])
    t1.update_plugin(p1)
    t2.update_plugin(p2)
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([s[0] for s in setup], [s[1] for s in setup], rcond=1e-5)
    H, VP, VS, RHO = g['model']
    assert g['esw'][5] == 1 and g['esw'][:5].sum() == 0
    for b in range(6):
        p1.y = g['ysw'][b] if g['esw'][b] == 0 else None
        p2.y = g['yrf'][b]
        joint.evaluate(h=H[b], vp=VP[b], vs=VS[b], noise=g[case + '_noise'][b])
        want_l, want_m = g[case + '_logL'][b], g[case + '_misfits'][b]
        assert np.isclose(joint.proposallikelihood, want_l, rtol=RTOL, atol=0)
        assert np.allclose(np.asarray(joint.proposalmisfits, dtype=float), want_m, rtol=RTOL, atol=0)
    assert g[case + '_logL'][5] == -1e15 and np.all(g[case + '_misfits'][5] == 1e15)


def test_valuation_matrices_match_closed_forms():
    """get_covariance_* still hand out the reference's matrices; quadratic_form equals d^T C^-1 d."""
    from bayhunter_amd import targets as T
    rs = np.random.RandomState(3)
    x = np.linspace(0, 10, 37)
    d = rs.normal(size=37)
    yerr = rs.uniform(0.5, 2.0, size=37)
    t = T.RayleighDispersionPhase(x, np.zeros(37), yerr=yerr)
    v = t.valuation
    v.init_covariance_gauss(0.9, 37, rcond=1e-6)
    table = [(0, v.get_covariance_nocorr(0.3, 37), 0.0),
             (1, v.get_covariance_nocorr_scalederr(0.3, 37, yerr), 0.0),
             (2, v.get_covariance_exp(0.7, 0.3, 37), 0.7),
             (3, v.get_covariance_gauss(0.3, 37), 0.9)]
    for cov, (c_inv, logdet), corr in table:
        t.covmodel = cov
        q, ld = t.quadratic_form(d, corr, 0.3)
        assert np.isclose(q, d.dot(c_inv).dot(d), rtol=1e-12) and np.isclose(ld, logdet, rtol=1e-12)
    R = 0.7 ** np.abs(np.subtract.outer(np.arange(37), np.arange(37)))
    assert np.allclose(v.get_corr_inv(0.7, 37) / (1 - 0.49), np.linalg.inv(R), atol=1e-10)


def test_survey_anchor_loglikelihood(g, oracle):
    """tutorial model, noise [0, .012, .98, .005], nocorr SWD + gauss RF (rcond 1e-5)
    -> the unmodified reference printed logL = 3070.29144695828."""
    from bayhunter_amd.synthetic import tutorial_model
    T, t1, t2 = _targets(g, False)
    h, vp, vs, rho = tutorial_model()
    p1, p2 = Fixed(g['sw_x']), Fixed(g['rf_x'])
    p1.y = oracle.swd(h, vp, vs, rho, g['sw_x'], 2, 0)[0]
    p2.y = oracle.rf_model(h, vp, vs, rho, nout=201)
    t1.update_plugin(p1)
    t2.update_plugin(p2)
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([True, True], [0.0, 0.98], rcond=1e-5)
    joint.evaluate(h=h, vp=vp, vs=vs, noise=np.array([0, 0.012, 0.98, 0.005]))
    assert abs(joint.proposallikelihood - 3070.29144695828) < 1e-6
    assert np.allclose(joint.proposalmisfits, [2.92e-05, 3.08e-05, 6.01e-05], atol=5e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_gpu_evaluate_batch_matches_reference(lib, g, case):
    with_yerr, setup = CASES[case]
    T, t1, t2 = _targets(g, with_yerr)
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([s[0] for s in setup], [s[1] for s in setup], rcond=1e-5)
    H, VP, VS, RHO = g['model']
    nl = np.full(6, 4, dtype=np.int32)
    logL, mis = joint.evaluate_batch(H, VP, VS, nl, g[case + '_noise'])
    logL, mis = logL.cpu().numpy(), mis.cpu().numpy()
    assert np.allclose(logL, g[case + '_logL'], rtol=RTOL, atol=0)
    assert np.allclose(mis, g[case + '_misfits'], rtol=1e-7, atol=0)
    assert logL[5] == -1e15 and np.all(mis[5] == 1e15)


@pytest.mark.gpu
def test_gpu_evaluate_batch_large_consistent_with_host(lib, g):
    """Batch of 1000 (not a multiple of the workgroup tile), per-model noise, against the host
    mirror fed with the GPU's own synthetics."""
    import torch
    from bayhunter_amd.synthetic import draw_models
    T, t1, t2 = _targets(g, False)
    joint = T.JointTarget([t1, t2])
    joint.set_target_covariance([True, True], [0.0, 0.98], rcond=1e-5)
    H, VP, VS, RHO, nl = draw_models(1000, (2, 12), seed=8)
    rs = np.random.RandomState(2)
    noise = np.stack([np.zeros(1000), rs.uniform(.005, .05, 1000), np.full(1000, .98),
                      rs.uniform(.002, .02, 1000)], axis=1)
    logL, mis = joint.evaluate_batch(H, VP, VS, nl, noise)
    out, err = joint._batch['eng'].run(H, VP, VS, RHO, nl)
    torch.cuda.synchronize()
    out, logL, mis = out.cpu().numpy(), logL.cpu().numpy(), mis.cpu().numpy()
    p1, p2 = Fixed(g['sw_x']), Fixed(g['rf_x'])
    t1.update_plugin(p1)
    t2.update_plugin(p2)
    for b in range(0, 1000, 37):
        p1.y, p2.y = out[b, :21], out[b, 21:]
        joint.evaluate(h=H[b, :nl[b]], vp=VP[b, :nl[b]], vs=VS[b, :nl[b]], noise=noise[b])
        assert np.isclose(logL[b], joint.proposallikelihood, rtol=1e-11)
        assert np.allclose(mis[b], joint.proposalmisfits, rtol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize('n,B', [(5, 37), (64, 1), (130, 37), (201, 130), (208, 64), (209, 65), (300, 37), (1024, 37)])
def test_gpu_gauss_quadratic_form_all_tile_shapes(lib, n, B):
    """bh_likelihood_batch straight through the C ABI for dense-Gaussian targets of every MFMA tile
    variant (4/8/13 column tiles, several passes above 208 points; four waves = 64 models per workgroup,
    R^-1 staged through LDS in 16-row chunks), with and without the matrix-core workspace, no err-flag
    columns, batches that are not a multiple of 16 or 64; against NumPy with an ASYMMETRIC
    'R^-1' (catches a transposed operand) ."""
    import ctypes as C
    import torch
    from bayhunter_amd import _lib
    rs = np.random.RandomState(n)
    out = rs.normal(size=(B, n + 3))
    yobs = np.zeros(n + 3)
    yobs[3:] = rs.normal(size=n)
    Rinv = rs.normal(size=(n, n)) / n                      # deliberately not symmetric
    noise = np.stack([np.full(B, 0.9), rs.uniform(0.5, 2.0, B)], axis=1)
    d = out[:, 3:] - yobs[3:]
    q = np.einsum('bi,ij,bj->b', d, Rinv, d)
    want = -0.5 * (n * np.log(2 * np.pi) + 2 * n * np.log(noise[:, 1]) + 1.25) - q / noise[:, 1] ** 2 / 2
    want_mis = np.sqrt((d ** 2).mean(axis=1))
    dev = torch.device('cuda')
    t_out, t_yobs, t_noise, t_aux = (torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                                     for a in (out, yobs, noise, Rinv.ravel()))
    desc = (_lib.LikeTarget * 1)(_lib.LikeTarget(n, 3, _lib.COV_GAUSS, 0, 1.25))
    need = lib.bh_likelihood_workspace_bytes(B, 1, desc)
    assert need == B * 2 * 8 * (((n + 15) // 16 + 3) // 4)       # a (q, sum d^2) pair per model and group of four column tiles
    ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
    for use_ws in (True, False):
        logL = torch.zeros(B, dtype=torch.float64, device=dev)
        mis = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        _lib.check(lib.bh_likelihood_batch(B, 1, desc, t_out.data_ptr(), n + 3, None, 0, t_yobs.data_ptr(),
                                           t_noise.data_ptr(), t_aux.data_ptr(), logL.data_ptr(),
                                           mis.data_ptr(), ws.data_ptr() if use_ws else None,
                                           need if use_ws else 0, None))
        torch.cuda.synchronize()
        assert np.allclose(logL.cpu().numpy(), want, rtol=1e-11, atol=1e-9)
        assert np.allclose(mis.cpu().numpy()[:, 0], want_mis, rtol=1e-12)
        assert np.allclose(mis.cpu().numpy()[:, 1], want_mis, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [201, 250, 60])
def test_gpu_gauss_form_does_not_depend_on_the_batch(lib, n):
    """The dense Gaussian product has two decompositions (like_kernel.hip: gauss_q_kernel) -- a workgroup per group of
    four column tiles for small batches, all tiles in one workgroup for large ones -- that compute the very same
    partial sums: the likelihood of a model must be the same bits whether it is evaluated in a batch of 40 000 (fused
    form) or in one of 9 000 (split form).  Chains of a pool, and of the ranks of a sharded pool, rely on that."""
    import torch
    from bayhunter_amd import _lib
    B = 40000
    rs = np.random.RandomState(n)
    out = rs.normal(size=(B, n))
    yobs = rs.normal(size=n)
    Rinv = rs.normal(size=(n, n)) / n
    noise = np.stack([np.full(B, 0.9), rs.uniform(0.5, 2.0, B)], axis=1)
    dev = torch.device('cuda')
    t_yobs, t_aux = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (yobs, Rinv.ravel()))
    desc = (_lib.LikeTarget * 1)(_lib.LikeTarget(n, 0, _lib.COV_GAUSS, 0, 0.0))

    def run(lo, hi):
        nb = hi - lo
        t_out = torch.from_numpy(np.ascontiguousarray(out[lo:hi])).to(dev)
        t_noise = torch.from_numpy(np.ascontiguousarray(noise[lo:hi])).to(dev)
        need = lib.bh_likelihood_workspace_bytes(nb, 1, desc)
        ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
        logL = torch.zeros(nb, dtype=torch.float64, device=dev)
        mis = torch.zeros((nb, 2), dtype=torch.float64, device=dev)
        _lib.check(lib.bh_likelihood_batch(nb, 1, desc, t_out.data_ptr(), n, None, 0, t_yobs.data_ptr(), t_noise.data_ptr(),
                                           t_aux.data_ptr(), logL.data_ptr(), mis.data_ptr(), ws.data_ptr(), need, None))
        torch.cuda.synchronize()
        return logL.cpu().numpy(), mis.cpu().numpy()
    big_l, big_m = run(0, B)
    for lo, hi in ((0, 9000), (9000, 9017), (31000, 40000)):
        l, m = run(lo, hi)
        assert np.array_equal(l, big_l[lo:hi]) and np.array_equal(m, big_m[lo:hi]), (n, lo, hi)


@pytest.mark.gpu
def test_gpu_likelihood_in_two_stages_equals_one_call(lib):
    """bh_likelihood_stage: the dense Gaussian products first (an evaluation plan runs them behind the
    receiver-function kernel on its side stream, beside the dispersion searches), the rest afterwards -- the same
    bits as bh_likelihood_batch; without a workspace the stages cannot be split."""
    import torch
    from bayhunter_amd import _lib
    B, n1, n2 = 777, 21, 201
    rs = np.random.RandomState(5)
    out = rs.normal(size=(B, n1 + n2))
    yobs = rs.normal(size=n1 + n2)
    Rinv = rs.normal(size=(n2, n2)) / n2
    noise = np.column_stack([np.zeros(B), rs.uniform(0.5, 2.0, B), np.full(B, 0.9), rs.uniform(0.5, 2.0, B)])
    dev = torch.device('cuda')
    t_out, t_yobs, t_noise, t_aux = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (out, yobs, noise, Rinv.ravel()))
    desc = (_lib.LikeTarget * 2)(_lib.LikeTarget(n1, 0, _lib.COV_EXP, 0, 0.0), _lib.LikeTarget(n2, n1, _lib.COV_GAUSS, 0, 1.5))
    need = lib.bh_likelihood_workspace_bytes(B, 2, desc)
    res = []
    for stages in ((3,), (1, 2)):
        ws = torch.full((need // 8,), float('nan'), dtype=torch.float64, device=dev)
        logL = torch.zeros(B, dtype=torch.float64, device=dev)
        mis = torch.zeros((B, 3), dtype=torch.float64, device=dev)
        for st in stages:
            _lib.check(lib.bh_likelihood_stage(st, B, 2, desc, t_out.data_ptr(), n1 + n2, None, 0, t_yobs.data_ptr(),
                                               t_noise.data_ptr(), t_aux.data_ptr(), logL.data_ptr(), mis.data_ptr(),
                                               ws.data_ptr(), need, None))
        torch.cuda.synchronize()
        res.append((logL.cpu().numpy(), mis.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.isfinite(res[0][0]).all()
    logL = torch.zeros(B, dtype=torch.float64, device=dev)
    mis = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    for bad in (0, 4):
        assert lib.bh_likelihood_stage(bad, B, 2, desc, t_out.data_ptr(), n1 + n2, None, 0, t_yobs.data_ptr(), t_noise.data_ptr(),
                                       t_aux.data_ptr(), logL.data_ptr(), mis.data_ptr(), None, 0, None) != 0
    assert lib.bh_likelihood_stage(1, B, 2, desc, t_out.data_ptr(), n1 + n2, None, 0, t_yobs.data_ptr(), t_noise.data_ptr(),
                                   t_aux.data_ptr(), logL.data_ptr(), mis.data_ptr(), None, 0, None) != 0
