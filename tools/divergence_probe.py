"""GPU box: how much of swd_kernel's time is lane divergence?  The same batch size once with random
models (every lane its own search) and once with one model repeated (all lanes of a wave do exactly
the same thing: no divergent branches, no waiting for the slowest lane)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
per = np.linspace(1, 41, 21)
H, VP, VS, RHO, nl = draw_models(B, 10, seed=1)
eng = ForwardEngine(swd=[SwdSpec('rdispph', per)])
_lib.set_swd_kernel('lane')
eng.sort_ragged = False        # the order of the input rows IS the processing order here
def run(tag, *arrs):
    d = eng.upload(*arrs)
    out, err = eng.alloc_out(B)
    eng.run(d, out=out, err=err); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    print('%-28s %8.2f ms' % (tag, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
run('random models', H, VP, VS, RHO, nl)
for k in (0, 1, 2):
    rep = lambda a: np.repeat(a[k:k + 1], B, axis=0)
    run('model %d repeated' % k, rep(H), rep(VP), rep(VS), rep(RHO), np.repeat(nl[k:k + 1], B))
# same models, but every wave holds 64 copies of one model (divergence only between waves)
idx = np.repeat(np.arange(B // 64), 64)
run('64 copies per wave', H[idx], VP[idx], VS[idx], RHO[idx], nl[idx])
# do lanes with similar models diverge less?  order the batch by a scalar (or two) of the model
Z = np.cumsum(H, axis=1)
tt = (H[:, :-1] / VS[:, :-1]).sum(axis=1)
def vs_at(depth):
    i = np.minimum((Z[:, :-1] < depth).sum(axis=1), nl - 1)
    return VS[np.arange(B), i]
def bins(x, n):
    return np.digitize(x, np.quantile(x, np.linspace(0, 1, n + 1)[1:-1]))
def csur(T):
    # crude surrogate of the Rayleigh phase velocity at period T: vs averaged with a depth kernel
    zmid = Z - H / 2.0
    zmid[:, -1] = Z[:, -2] + 10.0
    hh = H.copy(); hh[:, -1] = 0.35 * 3.5 * T
    w = np.exp(-zmid / (0.35 * 3.5 * T)) * hh
    return 0.92 * (VS * w).sum(axis=1) / w.sum(axis=1)
c1, c20, c41 = csur(1.0), csur(20.0), csur(41.0)
span = c41 - c1                    # ~ number of grid steps of all scans together: a search-length predictor
vsmin = np.where(VS > 0, VS, 1e9).min(axis=1)
span2 = c41 - 0.79 * vsmin        # + the first period's scan from cc = 0.855 c_R(slowest layer)
keys = {
    'S travel time (ascending)': (tt,),
    'span in 8 bins (longest first), then travel time': (tt, -bins(span, 8)),
    'span in 16 bins (longest first), then travel time': (tt, -bins(span, 16)),
    'span in 32 bins (longest first), then travel time': (tt, -bins(span, 32)),
    'span2 in 8 bins (longest first), then travel time': (tt, -bins(span2, 8)),
    'span2 in 16 bins (longest first), then travel time': (tt, -bins(span2, 16)),
    'span2 in 32 bins (longest first), then travel time': (tt, -bins(span2, 32)),
    'span2 descending': (-span2,),
    'span2 in 16 bins (longest first), then c(1)': (c1, -bins(span2, 16)),
}
for tag, ks in keys.items():
    o = np.lexsort(ks)
    run('sorted: ' + tag, H[o], VP[o], VS[o], RHO[o], nl[o])
