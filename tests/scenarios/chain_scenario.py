"""Shared by tests/test_chains.py, tests/golden/make_golden_chains.py and the GPU chain tests:
the inversion set-ups the chain pool is compared on, and a CPU evaluator built on the oracle
(test infrastructure only)."""
import os

import numpy as np

CASES = {
    # the tutorial's joint inversion: free vp/vs, Gaussian-correlated RF noise, crossing iiter = -1000, 0
    'tutorial': dict(seed=7, burnin=1100, main=400,
                     priors=dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=None,
                                 rfnoise_corr=0.9, swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05),
                                 swdnoise_sigma=(1e-5, 0.05)),
                     initparams=dict(propdist=(0.015, 0.015, 0.015, 0.005, 0.005), acceptance=(40, 55),
                                     thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, station='test',
                                     savepath='results', maxmodels=50000)),
    # fixed vp/vs, Moho estimate, mantle rule, velocity-zone limits, free (exponential) RF correlation
    'constrained': dict(seed=21, burnin=300, main=200,
                        priors=dict(vpvs=1.73, layers=(2, 8), vs=(2, 5), z=(0, 60), mohoest=(38, 4),
                                    mantle=(4.3, 1.8), rfnoise_corr=(0.5, 0.95), swdnoise_corr=0.,
                                    rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.05)),
                        initparams=dict(propdist=(0.02, 0.5, 0.05, 0.005, 0.005), acceptance=(40, 50),
                                        thickmin=0.5, lvz=0.1, hvz=0.45, rcond=None, station='test',
                                        savepath='results', maxmodels=50000)),
    # everything about the noise fixed: no hyper-parameter moves, widths never adapted
    'fixednoise': dict(seed=5, burnin=150, main=100,
                       priors=dict(vpvs=(1.5, 2.0), layers=(1, 12), vs=(2, 5), z=(0, 60), mohoest=None,
                                   rfnoise_corr=0.0, swdnoise_corr=0., rfnoise_sigma=0.01,
                                   swdnoise_sigma=0.012),
                       initparams=dict(propdist=(0.03, 0.8, 0.1, 0.005, 0.01), acceptance=(30, 60),
                                       thickmin=0.1, lvz=None, hvz=None, rcond=None, station='test',
                                       savepath='results', maxmodels=50000)),
    # all six data sets of the tutorial station: exponential covariance for the four dispersion
    # targets (fixed correlation) and for the two receiver functions (free correlation)
    'sixtargets': dict(seed=33, burnin=260, main=140, refs=('rdispph', 'rdispgr', 'ldispph', 'ldispgr', 'prf', 'srf'),
                       priors=dict(vpvs=(1.5, 2.0), layers=(1, 10), vs=(2, 5), z=(0, 60), mohoest=None,
                                   rfnoise_corr=(0.8, 0.98), swdnoise_corr=0.25,
                                   rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.1)),
                       initparams=dict(propdist=(0.02, 0.4, 0.05, 0.005, 0.01), acceptance=(40, 70),
                                       thickmin=0.2, lvz=None, hvz=None, rcond=None, station='test',
                                       savepath='results', maxmodels=50000)),
}


REFS = {'rdispph': ('swd', 2, 0), 'rdispgr': ('swd', 2, 1), 'ldispph': ('swd', 1, 0), 'ldispgr': ('swd', 1, 1),
        'prf': ('rf', 0), 'srf': ('rf', 1)}
TWO = ('rdispph', 'prf')


class OraclePlugin(object):
    """The forward-plugin contract on top of the CPU oracle.  kind: 'swd' (Rayleigh phase), 'rf'
    (P receiver function) or a target ref of REFS."""

    def __init__(self, oracle, x, kind):
        self.oracle, self.obsx = oracle, x
        spec = REFS.get(kind, ('swd', 2, 0) if kind == 'swd' else ('rf', 0))
        self.kind, self.args = spec[0], spec[1:]

    def __getstate__(self):
        return dict(obsx=self.obsx, kind=self.kind, args=self.args, oracle=None)   # a module does not pickle

    def run_model(self, h, vp, vs, rho, **kw):
        if self.kind == 'swd':
            y, err = self.oracle.swd(h, vp, vs, rho, self.obsx, self.args[0], self.args[1])
            return (self.obsx, y) if err == 0 else (np.nan, np.nan)
        return self.obsx, self.oracle.rf_model(h, vp, vs, rho, nout=self.obsx.size, waveno=self.args[0])


def target_classes(T):
    return {'rdispph': T.RayleighDispersionPhase, 'rdispgr': T.RayleighDispersionGroup,
            'ldispph': T.LoveDispersionPhase, 'ldispgr': T.LoveDispersionGroup,
            'prf': T.PReceiverFunction, 'srf': T.SReceiverFunction}


def joint_target(data_dir, plugins=None, refs=TWO, oracle=None):
    """JointTarget on the tutorial's observed data.  plugins(xsw, xrf) -> two plugins (the two-target
    set-ups), or oracle: an OraclePlugin per ref; neither: the package's own GPU plugins."""
    from bayhunter_amd import targets as T
    cls = target_classes(T)
    out = []
    for ref in refs:
        d = np.loadtxt(os.path.join(data_dir, 'st3_%s.dat' % ref))
        out.append(cls[ref](d[:, 0], d[:, 1]))
    if plugins is not None:
        p1, p2 = plugins(out[0].obsdata.x, out[1].obsdata.x)
        out[0].update_plugin(p1)
        out[1].update_plugin(p2)
    elif oracle is not None:
        for ref, t in zip(refs, out):
            t.update_plugin(OraclePlugin(oracle, t.obsdata.x, ref))
    return T.JointTarget(out)


def oracle_evaluator(joint):
    """(packed, nlay, noise) -> (logL, misfits) one model at a time through JointTarget.evaluate."""
    def run(packed, nlay, noise):
        B = packed.shape[0]
        logL, misfits = np.zeros(B), np.zeros((B, joint.ntargets + 1))
        for b in range(B):
            n = int(nlay[b])
            h, vp, vs, rho = (packed[b, k, :n].copy() for k in range(4))
            joint.evaluate(h=h, vp=vp, vs=vs, noise=noise[b], rho=rho)
            logL[b], misfits[b] = joint.proposallikelihood, joint.proposalmisfits
        return logL, misfits
    return run


def make_pool(oracle, data_dir, case, seeds, groups=None, evaluator=None, lookahead=None):
    from bayhunter_amd.chains import ChainPool
    refs = case.get('refs', TWO)
    if evaluator is None:
        joint = joint_target(data_dir, refs=refs, oracle=oracle)
        evaluator = oracle_evaluator(joint)
    else:
        joint = joint_target(data_dir, refs=refs)
        evaluator = evaluator(joint)
    ip = dict(case['initparams'], iter_burnin=case['burnin'], iter_main=case['main'])
    return ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=seeds, evaluator=evaluator,
                     groups=groups, lookahead=lookahead)
