"""MEASUREMENT TOOLING, outside the scope table of SURVEY section 8 (the reference's command-line / config
layer is not rebuilt): a convenience for running whole inversions on the GPU box.

Run an inversion the way the reference's tutorial/tutorialhunt.py does -- BayHunter config.ini,
observed data files -- on the lock-step chain pool, and write the reference's result files.

    python tools/hunt.py tutorial/config.ini --target rdispph=obs/st3_rdispph.dat \\
                         --target prf=obs/st3_prf.dat [--nchains 4096] [--gauss 1.0 --p 6.4]

The .ini format is the reference's (src/utils.py:33-68: every value is a Python literal, tuples
without parentheses; the sections [modelpriors] and [initparams]); data files hold x and y columns
(and optionally yerr).  Afterwards the reference's own
`PlotFromStorage('<savepath>/data/<station>_config.pkl')` works on the output unchanged.
"""
import argparse
import configparser
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_params(initfile):
    """-> (modelpriors, initparams) of a BayHunter config.ini."""
    cp = configparser.ConfigParser()
    with open(initfile) as f:
        cp.read_file(f)
    out = []
    for sec in ('modelpriors', 'initparams'):
        d = {}
        for key, val in cp[sec].items():
            try:
                d[key] = eval(val, {'__builtins__': {}}, {'None': None})     # 1e-5, 0.05 -> tuple
            except Exception:
                d[key] = val.strip().strip('\'"')                              # station = test
        out.append(d)
    return out


def build_targets(specs, gauss=None, p=None):
    from bayhunter_amd import targets as T
    cls = {'rdispph': T.RayleighDispersionPhase, 'rdispgr': T.RayleighDispersionGroup,
           'ldispph': T.LoveDispersionPhase, 'ldispgr': T.LoveDispersionGroup,
           'prf': T.PReceiverFunction, 'srf': T.SReceiverFunction}
    tl = []
    for spec in specs:
        ref, path = spec.split('=', 1)
        d = np.loadtxt(path)
        t = cls[ref](d[:, 0], d[:, 1], yerr=d[:, 2] if d.shape[1] > 2 else None)
        if ref in ('prf', 'srf'):
            kw = {k: v for k, v in (('gauss', gauss), ('p', p)) if v is not None}
            if kw:
                t.moddata.plugin.set_modelparams(**kw)
        tl.append(t)
    return T.JointTarget(tl)


def main(argv=None, evaluator=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('config')
    ap.add_argument('--target', action='append', required=True, metavar='REF=FILE')
    ap.add_argument('--nchains', type=int, default=None, help='overrides initparams nchains')
    ap.add_argument('--seed', type=int, default=None, help='random_seed of the optimizer (per-chain seeds follow)')
    ap.add_argument('--gauss', type=float, default=None)
    ap.add_argument('--p', type=float, default=None, help='ray parameter in s/deg')
    ap.add_argument('--savepath', default=None)
    ap.add_argument('--full-storage', action='store_true',
                    help='one sample row per iteration instead of iterations*max(acceptance)/100')
    args = ap.parse_args(argv)
    from bayhunter_amd.chains import ChainPool
    priors, initparams = load_params(args.config)
    if args.savepath:
        initparams['savepath'] = args.savepath
    joint = build_targets(args.target, args.gauss, args.p)
    iters = int(initparams['iter_burnin']) + int(initparams['iter_main'])
    pool = ChainPool(joint, initparams=initparams, modelpriors=priors, random_seed=args.seed, nchains=args.nchains,
                     evaluator=evaluator, nmodels=iters + 1 if args.full_storage else None)
    t0 = time.perf_counter()
    pool.run(progress=(max(1, iters // 10), lambda q: print('iteration %d  %.1f s' % (q.iteration, time.perf_counter() - t0),
                                                            flush=True)))
    dt = time.perf_counter() - t0
    n = pool.save()
    print('%d chains x %d iterations in %.1f s (%.3g chain iterations/s); %d files under %s'
          % (pool.nchains, iters, dt, pool.nchains * iters / dt, n, os.path.join(pool.initparams['savepath'], 'data')))
    return pool


if __name__ == '__main__':
    main()
