"""Child of tools/rf_probe.py / rocprofv3: times rf_kernel with the library given on the command line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayhunter_amd import _lib  # noqa: E402
_lib.LIB_PATH = sys.argv[1]
import torch  # noqa: E402
from bayhunter_amd.engine import ForwardEngine, RfSpec  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402

for B, L in ((64, 15), (524288, 10)) if len(sys.argv) < 4 else ((int(sys.argv[3]), 10),):
    H, VP, VS, RHO, nl = draw_models(min(B, 4096), L, seed=6000)
    rep = B // H.shape[0]
    H, VP, VS, RHO, nl = (np.tile(a, (rep, 1)) if a.ndim == 2 else np.tile(a, rep) for a in (H, VP, VS, RHO, nl))
    eng = ForwardEngine(rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    d = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.alloc_out(B)
    eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 20 if B < 1000 else 5
    ev[0].record()
    for _ in range(n):
        eng.run(d, out=out, err=err)
    ev[1].record()
    torch.cuda.synchronize()
    print('%-9s B=%6d L=%2d  %.3f ms' % (sys.argv[2], B, L, ev[0].elapsed_time(ev[1]) / n))
