"""GPU box: time of the fused likelihood (gauss_q_kernel + like_kernel) on resident modelled data, for the tutorial's
two targets (21 dispersion points, exponential covariance; 201 receiver-function points, dense Gaussian covariance).
    python tools/like_bench.py [B ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import torch  # noqa: E402
from chain_scenario import joint_target  # noqa: E402

joint = joint_target(os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed'))
joint.set_target_covariance([True, True], [0.0, 0.9], 1e-5)
bt = joint._batch or joint._build_batch()
eng = bt['eng']
res = []
for B in [int(a) for a in sys.argv[1:]] or [64, 2048, 16384, 131072]:
    rs = np.random.RandomState(B)
    out = torch.from_numpy(rs.normal(0, 0.02, (B, eng.row)) + np.concatenate([t.obsdata.y for t in joint.targets])).cuda()
    err = torch.zeros((B, 1), dtype=torch.int32, device='cuda')
    noise = torch.from_numpy(np.tile([0.0, 0.012, 0.9, 0.01], (B, 1))).cuda()
    for _ in range(3):
        logL, mis = joint.likelihood_batch(out, err, noise)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            logL, mis = joint.likelihood_batch(out, err, noise)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    res.append('B=%d %.4f ms (logL[0] %.9g)' % (B, min(ts), float(logL[0])))
print('likelihood_batch, best of 5 x 20: ' + '   '.join(res))
