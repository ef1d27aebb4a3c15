"""GPU box: do pools and plans give back what they take?  N pools (two plans each) made, run and closed in a row;
prints the process's resident set and the device's free memory every few pools.
    python tools/soak_lifecycle.py [npools] [nchains]"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import torch  # noqa: E402
from chain_scenario import CASES, joint_target  # noqa: E402
from bayhunter_amd.chains import ChainPool  # noqa: E402

npools = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nchains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
case = CASES['tutorial']
data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
ip = dict(case['initparams'], iter_burnin=30, iter_main=10, acceptance=(40, 100))
t0 = time.perf_counter()
first = None
for k in range(npools):
    with ChainPool(joint_target(data), initparams=ip, modelpriors=case['priors'], seeds=np.arange(nchains) % 1000) as pool:
        pool.run()
        digest = float(np.nansum(pool.likes[:, :3]))
    first = digest if first is None else first
    assert digest == first
    if k % 10 == 9 or k == 0:
        free, total = torch.cuda.mem_get_info()
        print('pool %3d  rss %d MB  device free %d MB  threads %d  %.1f s' % (
            k + 1, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, free >> 20, len(os.listdir('/proc/self/task')),
            time.perf_counter() - t0), flush=True)
print('done: %d pools of %d chains, every pool the same inversion' % (npools, nchains))
