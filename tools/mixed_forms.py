"""GPU-box experiment: dispersion targets of one batch launched as separate kernels of different forms on
concurrent streams (what a per-target choice of the kernel form inside bh_swd_batch would do).

    python tools/mixed_forms.py LAYERS PERIODS MODELS "ref:form,ref:form|ref:form ..." [more plans]
A plan is a list of launches separated by '|'; a launch is a list of ref:form sharing one kernel form.
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
H, VP, VS, RHO, nl = draw_models(B, L, seed=3000, sorted_vs=True)
per = np.linspace(1, 41, P)
for plan in sys.argv[4:]:
    launches = []
    for part in plan.split('|'):
        items = [x.split(':') for x in part.split(',')]
        eng = ForwardEngine(swd=[SwdSpec(r, per) for r, _ in items])
        d = eng.upload(H, VP, VS, RHO, nl)
        out, err = eng.alloc_out(B)
        launches.append((eng, d, out, err, items[0][1], torch.cuda.Stream()))
    torch.cuda.synchronize()

    def step():
        for eng, d, out, err, form, st in launches:
            _lib.set_swd_kernel(form)
            eng.run(d, out=out, err=err, stream=st)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    _lib.set_swd_kernel('auto')
    print('L=%d P=%d B=%d  %-70s %.2f ms per step' % (L, P, B, plan, min(ts)), flush=True)
