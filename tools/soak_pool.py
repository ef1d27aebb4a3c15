import os, sys, time, numpy as np, torch, resource
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests/scenarios')
from chain_scenario import CASES, joint_target
from bayhunter_amd.chains import ChainPool, GpuEvaluator
case = CASES['tutorial']
joint = joint_target(ROOT + '/tests/golden/tutorial_observed')
ip = dict(case['initparams'], iter_burnin=3000, iter_main=1000, acceptance=(40, 45))
pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=np.arange(4096) % 1000, nmodels=4001)
t0 = time.perf_counter()
def prog(p):
    print('it', p.iteration, 'dev MB', torch.cuda.memory_allocated() >> 20, 'reserved MB', torch.cuda.memory_reserved() >> 20,
          'rss MB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, '%.1f s' % (time.perf_counter() - t0), flush=True)
pool.run(progress=(500, prog))
n, pd, acc, pro = pool.counters()
print('done', time.perf_counter() - t0, 's; accepted mean', n.mean(), 'rate %', 100 * n.mean() / 4000, 'propdist median', np.median(pd, axis=0))
