"""Drive the REFERENCE's own, unmodified sampler -- src/SingleChain.py: SingleChain.run_chain(), with
src/Targets.py and src/Models.py -- through bayhunter_amd's plugin surface.  Development container
only (needs /root/reference); imported by tests/test_reference_chain.py.

The three reference modules are loaded file-wise.  What they import from the `BayHunter` package but
cannot be loaded here (zmq/configobj/matplotlib are not installed) is provided by a stand-in package
exposing exactly the names SingleChain needs: the reference's own Model/ModelMatrix, and
`utils.get_path/load_params` returning the dictionaries of src/defaults/defaults.ini.  NumPy aliases
removed in NumPy 2 (`np.float`, `np.int`, `np.product`) are restored.
"""
import configparser
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/src'


def available():
    return os.path.isdir(REF)


def _load(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _defaults():
    cp = configparser.ConfigParser()
    cp.read(os.path.join(REF, 'defaults', 'defaults.ini'))
    out = []
    for sec in ('modelpriors', 'initparams'):
        d = {}
        for k, v in cp[sec].items():
            d[k] = v.strip("'") if k in ('station', 'savepath') else eval(v)   # like utils.string_decode
        out.append(d)
    return out


def load_reference():
    """-> (Targets module, SingleChain class) of the reference."""
    for alias, real in (('float', float), ('int', int)):
        if not hasattr(np, alias):
            setattr(np, alias, real)
    if not hasattr(np, 'product'):
        np.product = np.prod
    for name in ('matplotlib', 'matplotlib.pyplot'):
        sys.modules.setdefault(name, types.ModuleType(name))
    models = _load('ref_models', 'Models.py')
    pkg = types.ModuleType('BayHunter')
    utils = types.ModuleType('BayHunter.utils')
    utils.get_path = lambda name: name
    utils.load_params = lambda path: _defaults()
    pkg.Model, pkg.ModelMatrix, pkg.utils = models.Model, models.ModelMatrix, utils
    rfm, swm = types.ModuleType('BayHunter.rfmini_modrf'), types.ModuleType('BayHunter.surf96_modsw')
    rfm.RFminiModRF = lambda obsx, ref: None          # replaced through update_plugin below
    swm.SurfDisp = lambda obsx, ref: None
    sys.modules.update({'BayHunter': pkg, 'BayHunter.utils': utils, 'BayHunter.rfmini_modrf': rfm,
                        'BayHunter.surf96_modsw': swm})
    targets = _load('ref_targets', 'Targets.py')
    chain = _load('ref_singlechain', 'SingleChain.py')
    return targets, chain.SingleChain


def load_plot_from_storage():
    """-> the reference's PlotFromStorage class (src/Plotting.py, unmodified, loaded file-wise).
    matplotlib is not installed: empty stand-in modules satisfy the imports (nothing is plotted);
    `utils.read_config` is the reference's two-line pickle reader (src/utils.py:156-164)."""
    import pickle
    T, _ = load_reference()
    for name in ('matplotlib', 'matplotlib.cm', 'matplotlib.pyplot', 'matplotlib.colors'):
        sys.modules.setdefault(name, types.ModuleType(name))
    pkg = sys.modules['BayHunter']

    def read_config(configfile):
        with open(configfile, 'rb') as f:
            return pickle.load(f)
    pkg.utils.read_config = read_config
    pkg.Targets = T
    return _load('ref_plotting', 'Plotting.py').PlotFromStorage


TUTORIAL_PRIORS = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=None,
                       rfnoise_corr=0.9, swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05),
                       swdnoise_sigma=(1e-5, 0.05))
TUTORIAL_INITPARAMS = dict(nchains=1, propdist=(0.015, 0.015, 0.015, 0.005, 0.005), acceptance=(40, 45),
                           thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, station='test',
                           savepath='results', maxmodels=50000)


def run_chain(make_plugins, seed=7, burnin=120, main=60, data_dir=None, priors=None, initparams=None,
              savepath=None, refs=('rdispph', 'prf'), plugin_for=None):
    """Build the tutorial's joint target (Rayleigh phase + P-RF), install the plugins returned by
    make_plugins(xsw, xrf) with the reference's update_plugin hook, run one reference chain.
    With `savepath` the chain also writes its result files (SingleChain.save_finalmodels)."""
    T, SingleChain = load_reference()
    cls = {'rdispph': T.RayleighDispersionPhase, 'rdispgr': T.RayleighDispersionGroup,
           'ldispph': T.LoveDispersionPhase, 'ldispgr': T.LoveDispersionGroup,
           'prf': T.PReceiverFunction, 'srf': T.SReceiverFunction}
    tl = []
    for ref in refs:
        d = np.loadtxt(os.path.join(data_dir, 'st3_%s.dat' % ref))
        tl.append(cls[ref](d[:, 0], d[:, 1]))
    if plugin_for is not None:                     # plugin_for(ref, x) -> plugin, any number of targets
        for ref, t in zip(refs, tl):
            t.update_plugin(plugin_for(ref, t.obsdata.x))
    else:
        p1, p2 = make_plugins(tl[0].obsdata.x, tl[1].obsdata.x)
        tl[0].update_plugin(p1)
        tl[1].update_plugin(p2)
    joint = T.JointTarget(targets=tl)
    pr = dict(TUTORIAL_PRIORS)
    pr.update(priors or {})
    ip = dict(TUTORIAL_INITPARAMS, iter_burnin=burnin, iter_main=main)
    ip.update(initparams or {})
    if savepath is not None:
        ip['savepath'] = savepath
        os.makedirs(os.path.join(savepath, 'data'), exist_ok=True)
    nmodels = int((burnin + main) * max(ip['acceptance']) / 100.)
    maxlayers = int(pr['layers'][1]) + 1
    f32 = np.float32
    shared = dict(sharedmodels=np.full(nmodels * maxlayers * 2, np.nan, dtype=f32),
                  sharedmisfits=np.full(nmodels * (len(refs) + 1), np.nan, dtype=f32),
                  sharedlikes=np.full(nmodels, np.nan, dtype=f32),
                  sharednoise=np.full(nmodels * 2 * len(refs), np.nan, dtype=f32),
                  sharedvpvs=np.full(nmodels, np.nan, dtype=f32))
    chain = SingleChain(targets=joint, chainidx=0, initparams=ip, modelpriors=pr,
                        random_seed=seed, **shared)
    if savepath is None:
        chain.save_finalmodels = lambda *a, **k: None          # no files
    chain.run_chain()
    return dict(models=np.array(chain.chainmodels), likes=np.array(chain.chainlikes),
                misfits=np.array(chain.chainmisfits), noise=np.array(chain.chainnoise),
                vpvs=np.array(chain.chainvpvs), iter=np.array(chain.chainiter), n=chain.n,
                propdist=np.array(chain.propdist), accepted=np.array(chain.accepted),
                proposed=np.array(chain.proposed))
