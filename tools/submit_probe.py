"""GPU box: where does the host time of one small chain-pool submission go?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))
import torch
from chain_scenario import CASES, joint_target
from bayhunter_amd.chains import ChainPool, GpuEvaluator
from bayhunter_amd.engine import DeviceModels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
case = CASES['tutorial']
joint = joint_target(os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed'))
ip = dict(case['initparams'], iter_burnin=5000, iter_main=10, acceptance=(40, 100))
ev = GpuEvaluator(joint)
pool = ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=np.arange(n) % 1000, evaluator=ev, groups=1)
g = pool.groups[0]
for _ in range(5):
    pool._launch(g); pool._land(g)
torch.cuda.synchronize()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 200
for _ in range(N):
    t0 = time.perf_counter(); cnt = g.propose(); tick('propose', t0)
    bufs, outs, stream = ev._pin[g.packed.ctypes.data]
    B = cnt
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        dp = bufs[0][:B].to(ev.device, non_blocking=True); dn = bufs[1][:B].to(ev.device, non_blocking=True)
        dz = bufs[2][:B].to(ev.device, non_blocking=True)
        tick('h2d', t0); t0 = time.perf_counter()
        bt = joint._batch or joint._build_batch()
        m = bt['eng'].reorder(dp, dn, ragged=True)
        tick('reorder', t0); t0 = time.perf_counter()
        out, err = bt['eng'].run(m)
        tick('forward launches', t0); t0 = time.perf_counter()
        logL, misfits = joint.evaluate_batch(m, noise=dz)      # (forward again + likelihood: total cost of the call)
        tick('evaluate_batch (forward + like)', t0); t0 = time.perf_counter()
        outs[0][:B].copy_(logL, non_blocking=True); outs[1][:B].copy_(misfits, non_blocking=True)
        e = torch.cuda.Event(); e.record(stream)
        tick('d2h + event', t0); t0 = time.perf_counter()
    e.synchronize(); tick('wait', t0)
    t0 = time.perf_counter(); g.accept(outs[0].numpy()[:B], outs[1].numpy()[:B]); tick('accept', t0)
for k, v in T.items():
    print('%-34s %8.1f us' % (k, v / N * 1e6))
