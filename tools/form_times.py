"""GPU-box diagnostic: ms per call of chosen kernel forms on one batch shape.

    python tools/form_times.py LAYERS PERIODS MODELS form [form ...]
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

L = tuple(int(x) for x in sys.argv[1].split('-')) if '-' in sys.argv[1] else int(sys.argv[1])   # 10, or 2-31 (ragged)
P, B = int(sys.argv[2]), int(sys.argv[3])
H, VP, VS, RHO, nl = draw_models(B, L, seed=2000, sorted_vs=True)
eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, P))])
d = eng.upload(H, VP, VS, RHO, nl)
import bayhunter_amd.engine as _e  # noqa: E402
_e.ORDER_MIN = int(os.environ.get('FORM_TIMES_ORDER_MIN', _e.ORDER_MIN))   # A/B of the processing order
d = eng.reorder(d.packed, d.nlay)
out, err = eng.alloc_out(B)
res = []
for form in sys.argv[4:]:
    _lib.set_swd_kernel(form)
    for _ in range(3):
        eng.run(d, out=out, err=err)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            eng.run(d, out=out, err=err)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    res.append('%s %.3f' % (form, min(ts)))
_lib.set_swd_kernel('auto')
print('L=%s P=%d B=%d ms per call (best of 5 x 20): ' % (L, P, B) + '  '.join(res))
