"""GPU box: throughput of the lock-step chain pool on the tutorial inversion (Rayleigh phase + P-RF,
free vp/vs and noise) -- chain iterations per second end to end (host proposals + device forward
and likelihood + host acceptance), for several pool sizes.
usage: python tools/chain_bench.py [nchains ...]   -> one JSON line per pool size
env: CHAIN_BENCH_ITERS, CHAIN_BENCH_GROUPS, CHAIN_BENCH_LOOKAHEAD (proposals per chain and call; default: the pool's own
choice), BH_SWD_KERNEL (pin a kernel form)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))


def main():
    from chain_scenario import CASES, joint_target
    from bayhunter_amd.chains import ChainPool, GpuEvaluator
    import torch
    sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 16384]
    if 'BH_SWD_KERNEL' in os.environ:                      # pin a kernel form (experiments)
        from bayhunter_amd import _lib
        _lib.set_swd_kernel(os.environ['BH_SWD_KERNEL'])
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    case = CASES['tutorial']
    for n in sizes:
        iters = int(os.environ.get('CHAIN_BENCH_ITERS', 120 if n <= 4096 else 60))
        # a short pool of the same size first: first-use costs (kernel forms loaded on their first launch, helper
        # threads, pinned buffers) are not chain iterations
        with ChainPool(joint_target(data), initparams=dict(case['initparams'], iter_burnin=6, iter_main=2, acceptance=(40, 100)),
                       modelpriors=case['priors'], seeds=np.arange(n) % 1000, nmodels=9) as warm:
            warm.run()
        time.sleep(0.25)          # numpy's OpenBLAS workers spin ~0.1 s after the set-up's factorisation (bench.py: BLAS_SETTLE_S)
        joint = joint_target(data)
        ip = dict(case['initparams'], iter_burnin=iters, iter_main=iters // 2, acceptance=(40, 100))
        with ChainPool(joint, initparams=ip, modelpriors=case['priors'], seeds=np.arange(n) % 1000,
                       evaluator=GpuEvaluator(joint),
                       groups=int(os.environ['CHAIN_BENCH_GROUPS']) if 'CHAIN_BENCH_GROUPS' in os.environ else None,
                       lookahead=int(os.environ['CHAIN_BENCH_LOOKAHEAD']) if 'CHAIN_BENCH_LOOKAHEAD' in os.environ else None) as pool:
            t0 = time.perf_counter()
            pool.run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        total = n * (iters + iters // 2)
        acc = pool.counters()[0]
        calls, advanced, rows = pool.advance()
        print(json.dumps(dict(nchains=n, iterations=iters + iters // 2, seconds=round(dt, 3),
                              chain_iterations_per_s=round(total / dt), models_evaluated=int(pool.evaluated),
                              mean_accepted=float(acc.mean()), groups=len(pool.groups), lookahead=pool.lookahead,
                              calls=calls, iterations_per_call=round(advanced / max(calls, 1) / (n / len(pool.groups)), 2),
                              models_per_call=round(rows / max(calls, 1), 1),
                              seconds_in={k: round(v, 3) for k, v in pool.seconds.items()})), flush=True)


if __name__ == '__main__':
    main()
