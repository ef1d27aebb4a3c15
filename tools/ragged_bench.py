"""GPU-box diagnostic: lane kernel on ragged (2..31 layers) vs uniform 16-layer batches, unsorted and
sorted by layer count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models

def run(H, VP, VS, RHO, nl, reps=3, sort=False):
    eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, 21))])
    eng.sort_ragged = sort
    d = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.alloc_out(H.shape[0])
    _lib.set_swd_kernel('lane')
    eng.run(d, out=out, err=err); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
H, VP, VS, RHO, nl = draw_models(B, (2, 31), seed=5, Lmax=31)
print('ragged 2..31, B=%d: unsorted %.2f ms' % (B, run(H, VP, VS, RHO, nl)))
o = np.argsort(nl, kind='stable')
print('ragged 2..31, B=%d: sorted ascending on host %.2f ms' % (B, run(H[o], VP[o], VS[o], RHO[o], nl[o])))
print('ragged 2..31, B=%d: engine sort_ragged (descending, incl. un-permute) %.2f ms' % (B, run(H, VP, VS, RHO, nl, sort=True)))
H2, VP2, VS2, RHO2, nl2 = draw_models(B, 16, seed=6)
print('uniform 16 layers, B=%d: %.2f ms (mean ragged layer count %.1f)' % (B, run(H2, VP2, VS2, RHO2, nl2), nl.mean()))
