"""GPU-box diagnostic: nuclei -> layers -> forward -> likelihood at bench scale, per-stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayhunter_amd import targets as T
from bayhunter_amd.models import layers_from_voronoi

B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 10
rs = np.random.RandomState(0)
VSN = np.sort(rs.uniform(2, 5, (B, L)), axis=1); ZV = np.sort(rs.uniform(0, 60, (B, L)), axis=1)
nl = np.full(B, L, dtype=np.int32); vpvs = np.full(B, 1.73)
per, trf = np.linspace(1, 41, 21), np.linspace(-5, 35, 201)
pri = dict(layers=(1, 20), vs=(2, 5), z=(0, 60))
for covs, name in (([True, True], [0.0, 0.98]), ([True, False], [0.0, 0.6])):
    t1 = T.RayleighDispersionPhase(per, rs.normal(3.5, .2, 21)); t2 = T.PReceiverFunction(trf, rs.normal(0, .05, 201))
    joint = T.JointTarget([t1, t2]); joint.set_target_covariance(covs, name, rcond=1e-5)
    noise = torch.from_numpy(np.stack([np.zeros(B), rs.uniform(.01, .05, B), np.full(B, name[1]), rs.uniform(.005, .02, B)], 1)).cuda()
    dV, dZ, dn, dv = (torch.from_numpy(x).cuda() for x in (VSN, ZV, nl, vpvs))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for rep in range(3):
        ev[0].record()
        models, valid = layers_from_voronoi(dV, dZ, dn, dv, pri, 0.1)
        ev[1].record()
        out, err = (joint._batch or joint._build_batch())['eng'].run(models)
        ev[2].record()
        logL, mis = joint.evaluate_batch(models, noise=noise)
        ev[3].record()
        torch.cuda.synchronize()
    fwd = ev[1].elapsed_time(ev[2]); tot = ev[2].elapsed_time(ev[3])
    print('B=%d rf-cov=%s: voronoi %.3f ms | forward %.2f ms | forward+likelihood %.2f ms -> likelihood ~%.2f ms | %.3e models/s end to end'
          % (B, 'gauss' if covs[1] else 'exp', ev[0].elapsed_time(ev[1]), fwd, tot, tot - fwd, B / ((ev[0].elapsed_time(ev[1]) + tot) * 1e-3)))
