"""GPU-box diagnostic: bench-seed models (joint10 workload), lane kernel vs the oracle on a large
sample -- how many dispersion values are bit-identical, max RF difference."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
from bayhunter_amd.synthetic import draw_models
from oracle import pyoracle as po

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
per = np.linspace(1, 41, 21)
H, VP, VS, RHO, nl = draw_models(B, 10, seed=6000, sorted_vs=True)     # bench.py's seed for rank 0
eng = ForwardEngine(swd=[SwdSpec('rdispph', per)], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
_lib.set_swd_kernel('lane')
out, err = eng.run(H, VP, VS, RHO, nl)
out, err = out.cpu().numpy(), err.cpu().numpy()
threads = min(len(os.sched_getaffinity(0)), 16)
t0 = time.time()
want, werr, nc = po.swd_batch(H, VP, VS, RHO, nl, per, 2, 0, nthreads=threads)
wrf = po.rf_batch(H, VP, VS, RHO, nl, nthreads=threads)
d = np.abs(out[:, :21] - want)
print('joint10 bench models, B=%d (oracle took %.0f s on %d threads)' % (B, time.time() - t0, threads))
print('rdispph: %d values, %d differ (max %.3e), err flags equal: %s, N_dltar/model %.1f'
      % (d.size, int((d != 0).sum()), d.max(), np.array_equal(err[:, 0], werr), nc / B))
print('prf: max |diff| %.3e at amplitude %.3f' % (np.abs(out[:, 21:] - wrf).max(), np.abs(wrf).max()))
