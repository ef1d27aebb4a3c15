"""GPU-box diagnostic: lane vs team SWD kernel over the number of searches (crossover)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models

def run(B, L, refs, P, mode, reps=3):
    H, VP, VS, RHO, nl = draw_models(B, L, seed=1)
    eng = ForwardEngine(swd=[SwdSpec(r, np.linspace(1, 41, P)) for r in refs])
    d = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.alloc_out(B)
    _lib.set_swd_kernel(mode)
    eng.run(d, out=out, err=err); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    _lib.set_swd_kernel('auto')
    return (time.perf_counter() - t0) / reps * 1e3

MODES = ('lane', 'team', 'team32', 'team16', 'team8', 'team128', 'team256', 'team512', 'auto')
QUICK = os.environ.get('SWEEP_QUICK')
for L, refs, P in (((2, 31), ['rdispph'], 21), (3, ['rdispph'], 21), (5, ['rdispph'], 21), (10, ['rdispph'], 21), (15, ['rdispph'], 21),
                   (10, ['rdispph', 'rdispgr', 'ldispph', 'ldispgr'], 40)):
    for B in (64, 256, 512, 1024, 2048, 4096, 6144, 8192, 12288, 16384, 20480, 24576, 32768, 65536, 131072):
        if len(refs) == 4 and B > 16384: continue
        if QUICK and (B not in (4096, 8192, 16384, 32768) or L in (5, 15, 3)): continue
        ts = [run(B, L, refs, P, m) if not (m in ('team128', 'team256', 'team512') and B * len(refs) > 8192) else float('inf') for m in MODES]
        best = MODES[int(np.argmin(ts[:-1]))] + ('  auto within %.0f %% of best' % (100 * (ts[-1] / min(ts[:-1]) - 1)))
        print('L=%s targets=%d P=%d B=%6d  ' % (('%2d' % L) if isinstance(L, int) else 'ragged', len(refs), P, B) +
              '  '.join('%s %8.2f ms' % (m, t) for m, t in zip(MODES, ts)) + '  -> ' + best)
        sys.stdout.flush()
