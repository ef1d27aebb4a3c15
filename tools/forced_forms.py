"""GPU-box diagnostic: ms per call with the kernel form of each target forced (bh_swd_set_forms).

    python tools/forced_forms.py LAYERS PERIODS MODELS ref,ref,... form,form,... [form,form,... ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bayhunter_amd import _lib  # noqa: E402
from bayhunter_amd.engine import ForwardEngine, SwdSpec  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402

L, P, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
refs = sys.argv[4].split(',')
H, VP, VS, RHO, nl = draw_models(B, L, seed=3000, sorted_vs=True)
eng = ForwardEngine(swd=[SwdSpec(r, np.linspace(1, 41, P)) for r in refs])
d = eng.upload(H, VP, VS, RHO, nl)
out, err = eng.alloc_out(B)
for plan in sys.argv[5:]:
    _lib.set_swd_forms(None if plan == 'auto' else plan.split(','))
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
    print('L=%d P=%d B=%d %s  %-36s %.2f ms per call' % (L, P, B, '+'.join(refs), plan, min(ts)), flush=True)
_lib.set_swd_forms(None)
