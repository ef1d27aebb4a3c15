"""GPU box: where does the host time of one small chain-pool iteration go?  (cProfile over the pool's own
launch / land calls; the GPU work of such a pool is ~1 ms per iteration, so every 100 us of host time
between two launches shows.)

    python tools/submit_probe.py [nchains] [iterations]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import torch  # noqa: E402
from chain_scenario import CASES, joint_target  # noqa: E402
from bayhunter_amd.chains import ChainPool, GpuEvaluator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
case = CASES['tutorial']
joint = joint_target(os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed'))
ip = dict(case['initparams'], iter_burnin=100000, iter_main=10, acceptance=(40, 100))
pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=np.arange(n) % 1000,
                 evaluator=GpuEvaluator(joint), groups=1)
g = pool.groups[0]
for _ in range(20):
    pool._launch(g)
    pool._land(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    pool._launch(g)
    pool._land(g)
dt = time.perf_counter() - t0
print('%d chains: %.1f us per iteration (%.0f chain iterations/s)' % (n, dt / N * 1e6, n * N / dt))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    pool._launch(g)
    pool._land(g)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative')
st.print_stats(28)
