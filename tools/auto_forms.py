"""GPU-box diagnostic: ms per call and the kernel form of each target as bh_swd_batch chooses them.
Run with BH_SWD_NO_MIXED=1 for the one-form-per-call choice.

    python tools/auto_forms.py LAYERS PERIODS MODELS ref[,ref...] [more LAYERS PERIODS MODELS refs ...]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bayhunter_amd import _lib  # noqa: E402
from bayhunter_amd.engine import ForwardEngine, SwdSpec  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402

a = sys.argv[1:]
tag = 'one form ' if os.environ.get('BH_SWD_NO_MIXED') else 'per target'
for k in range(0, len(a), 4):
    L, P, B, refs = int(a[k]), int(a[k + 1]), int(a[k + 2]), a[k + 3].split(',')
    H, VP, VS, RHO, nl = draw_models(B, L, seed=3000, sorted_vs=True)
    eng = ForwardEngine(swd=[SwdSpec(r, np.linspace(1, 41, P)) for r in refs])
    d = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.alloc_out(B)
    for _ in range(2):
        eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            eng.run(d, out=out, err=err)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    forms = (C.c_int * len(refs))()
    _lib.check(_lib.load().bh_swd_last_forms(forms, len(refs)))
    print('%s L=%d P=%d B=%d %-40s forms %-22s %.2f ms per call' % (tag, L, P, B, '+'.join(refs), list(forms), min(ts)), flush=True)
