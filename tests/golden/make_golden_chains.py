"""Golden chains: the REFERENCE's own, unmodified sampler (src/SingleChain.py: SingleChain.run_chain,
with src/Targets.py and src/Models.py, loaded file-wise by tests/scenarios/reference_chain.py) run in
the development container on the tutorial's observed data, its forward plugins backed by the
reference's native solvers (oracle/_ref, built from /root/reference by oracle/Makefile).

    python tests/golden/make_golden_chains.py        (needs /root/reference and `make -C oracle ref`)

For every set-up in tests/scenarios/chain_scenario.py::CASES and three seeds the file holds what the
chain keeps: the accepted models, likelihoods, misfits, noise parameters, vp/vs (float32, as the
reference stores them) and the iteration at which each was accepted.
Output: chains_golden.npz (data only).

For the 'tutorial' set-up with maxmodels = 150 the file chain_files_golden.npz additionally holds the
ten result files the reference chain itself writes (SingleChain.save_finalmodels, src/SingleChain.py:
646-690: c000_p{1,2}{models,likes,misfits,noise,vpvs}.npy -- residence-time weighting, thinning), one
array per file under the file's name, so that the writer of the chain pool (ChainPool.save) can be
compared with them where the reference tree does not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def main():
    import pyoracle
    import reference_chain as rc
    from chain_scenario import CASES, OraclePlugin
    backend = 'ref' if pyoracle.have_ref() else 'port'
    print('forward solvers:', backend)

    class oracle(object):             # the module's functions, pinned to one back end
        swd = staticmethod(lambda *a, **k: pyoracle.swd(*a, backend=backend, **k))
        rf_model = staticmethod(lambda *a, **k: pyoracle.rf_model(*a, backend=backend, **k))
    data = os.path.join(HERE, 'tutorial_observed')
    out = {}
    for name, case in sorted(CASES.items()):
        seeds = [case['seed'], case['seed'] + 1, case['seed'] + 2]
        out['%s/seeds' % name] = np.array(seeds)
        for seed in seeds:
            res = rc.run_chain(None, refs=case.get('refs', ('rdispph', 'prf')), plugin_for=lambda ref, x: OraclePlugin(oracle, x, ref),
                               seed=seed, burnin=case['burnin'], main=case['main'], data_dir=data,
                               priors=case['priors'], initparams=case['initparams'])
            print(name, seed, 'accepted', res['n'], 'propdist', res['propdist'])
            for k in ('models', 'likes', 'misfits', 'noise', 'vpvs', 'iter', 'n', 'propdist', 'accepted', 'proposed'):
                out['%s/%d/%s' % (name, seed, k)] = np.asarray(res[k])
    np.savez_compressed(os.path.join(HERE, 'chains_golden.npz'), **out)
    print('wrote chains_golden.npz', os.path.getsize(os.path.join(HERE, 'chains_golden.npz')), 'bytes')
    # the reference chain's own result files for one set-up
    import tempfile
    case = CASES['tutorial']
    files = {'maxmodels': np.array(150), 'seed': np.array(case['seed'])}
    with tempfile.TemporaryDirectory() as td:
        rc.run_chain(None, refs=case.get('refs', ('rdispph', 'prf')), plugin_for=lambda ref, x: OraclePlugin(oracle, x, ref),
                     seed=case['seed'], burnin=case['burnin'], main=case['main'], data_dir=data,
                     priors=case['priors'], initparams=dict(case['initparams'], maxmodels=150), savepath=td)
        names = sorted(f for f in os.listdir(os.path.join(td, 'data')) if f.endswith('.npy'))
        assert len(names) == 10, names
        for f in names:
            files[f[:-4]] = np.load(os.path.join(td, 'data', f))
    np.savez_compressed(os.path.join(HERE, 'chain_files_golden.npz'), **files)
    print('wrote chain_files_golden.npz', sorted(k for k in files if k.startswith('c')))


if __name__ == '__main__':
    main()
