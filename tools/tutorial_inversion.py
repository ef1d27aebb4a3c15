"""GPU box: the reference's tutorial inversion (tutorial/tutorialhunt.py: Rayleigh phase dispersion
+ P receiver function of station st3, priors and initparams of tutorial/config.ini) run with a
lock-step chain pool, and a summary of what the chains found next to the model the data were
computed from (tutorial/create_testdata.py: h = 5, 23, 8 km, vs = 2.7, 3.6, 3.8, 4.4 km/s,
vp/vs 1.73; noise added to the data).
usage: python tools/tutorial_inversion.py [nchains] [iter_burnin] [iter_main] [lookahead]
(tutorialhunt.py itself: 5 chains, 2048*32 + 2048*16 iterations; lookahead: proposals per chain and call, default the pool's)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bayhunter_amd import targets as T
    from bayhunter_amd.chains import ChainPool
    from bayhunter_amd.models import Model
    nchains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    burnin = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    main_it = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    lookahead = int(sys.argv[4]) if len(sys.argv) > 4 else None
    if 'BH_SWD_KERNEL' in os.environ:                      # pin a kernel form (experiments)
        from bayhunter_amd import _lib as _l
        _l.set_swd_kernel(os.environ['BH_SWD_KERNEL'])
    d = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    sw, rf = np.loadtxt(os.path.join(d, 'st3_rdispph.dat')), np.loadtxt(os.path.join(d, 'st3_prf.dat'))
    joint = T.JointTarget([T.RayleighDispersionPhase(sw[:, 0], sw[:, 1]), T.PReceiverFunction(rf[:, 0], rf[:, 1])])
    joint.targets[1].moddata.plugin.set_modelparams(gauss=1.0, water=0.01, p=6.4)
    # tutorial/config.ini
    priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=(38, 4), rfnoise_corr=0.9,
                  swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.05))
    ip = dict(iter_burnin=burnin, iter_main=main_it, propdist=(0.015, 0.015, 0.015, 0.005, 0.005),
              acceptance=(40, 45), thickmin=0.1, lvz=None, hvz=None, rcond=1e-5, station='st3', maxmodels=50000)
    # the reference sizes its sample arrays by max(acceptance) = 45 % of the iterations; a chain that
    # runs hotter overflows them (IndexError there).  Short runs do: give every chain full storage.
    from bayhunter_amd import _lib
    import hashlib
    pool = ChainPool(joint, initparams=ip, modelpriors=priors, random_seed=1, nchains=nchains,
                     nmodels=burnin + main_it + 1, lookahead=lookahead)
    t0 = time.perf_counter()
    try:
        pool.run(progress=(2000, lambda p: print('iteration', p.iteration, 'evaluated', p.evaluated,
                                                 '%.1f s' % (time.perf_counter() - t0), flush=True)))
    except Exception as e:
        print('stopped:', e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nacc, propdist, accepted, proposed = pool.counters()
    last = np.maximum(nacc - 1, 0)
    idx = np.arange(pool.nchains)
    likes = pool.likes[idx, last]
    mis = pool.misfits[idx, last]
    models = pool.models[idx, last]
    nlay = np.sum(~np.isnan(models), axis=1) // 2
    best = int(np.argmax(likes))
    vp, vs, h = Model.get_vp_vs_h(models[best].astype(np.float64), float(pool.vpvs[best, last[best]]), None)
    out = dict(nchains=nchains, iterations=burnin + main_it, seconds=round(dt, 2),
               chain_iterations_per_s=round(nchains * (pool.iteration + burnin) / dt),
               models_evaluated=int(pool.evaluated), reached_iteration=int(pool.iteration),
               accept_rate_percent=float(100. * nacc.mean() / max(1, pool.iteration + burnin)),
               median_final_logL=float(np.median(likes)), best_final_logL=float(likes[best]),
               median_final_rms=dict(rdispph=float(np.median(mis[:, 0])), prf=float(np.median(mis[:, 1]))),
               layers_median=float(np.median(nlay - 1)), propdist_median=[float(x) for x in np.median(propdist, axis=0)],
               best_chain=dict(h=[round(float(x), 2) for x in h], vs=[round(float(x), 3) for x in vs],
                               vpvs=round(float(pool.vpvs[best, last[best]]), 3),
                               noise=[float(x) for x in pool.noise[best, last[best]]]),
               true_model=dict(h=[5, 23, 8, 0], vs=[2.7, 3.6, 3.8, 4.4], vpvs=1.73),
               host_seconds={k: round(v, 2) for k, v in pool.seconds.items()},
               lookahead=pool.lookahead, device_calls=pool.advance()[0],
               chains_sha256=hashlib.sha256(b''.join(np.ascontiguousarray(a).tobytes() for a in
                                                     (pool.models, pool.likes, pool.iter, pool.noise, pool.vpvs))).hexdigest()[:16],
               library=_lib.source_hash())
    print(json.dumps(out))


if __name__ == '__main__':
    main()
