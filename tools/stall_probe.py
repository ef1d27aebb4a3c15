import os, sys, time, numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests/scenarios')
from chain_scenario import CASES, joint_target
from bayhunter_amd.chains import ChainPool
import bayhunter_amd.chains as ch
case=CASES['tutorial']; data=ROOT+'/tests/golden/tutorial_observed'
n=4096
def mk(it):
    return ChainPool(joint_target(data), initparams=dict(case['initparams'], iter_burnin=it, iter_main=it//2, acceptance=(40,100)), modelpriors=case['priors'], seeds=np.arange(n)%1000, nmodels=(12 if it<10 else None))
mk(6).run()
pool=mk(200)
ev=[]
ol, od = pool._launch, pool._land
def launch(g):
    t=time.perf_counter(); ol(g); ev.append(('launch', time.perf_counter()-t, len(ev)))
def land(g):
    t=time.perf_counter(); od(g); ev.append(('land', time.perf_counter()-t, len(ev)))
pool._launch=launch; pool._land=land
t0=time.perf_counter(); pool.run(); print('total', time.perf_counter()-t0, pool.seconds)
ev.sort(key=lambda e:-e[1])
print(ev[:8]); print('n events', len(ev))
