"""GPU box: does the processing order (engine.reorder) help or hurt a kernel form?  ms per call, natural order vs ordered.
usage: order_effect.py LAYERS PERIODS MODELS ref,ref,... form [form ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayhunter_amd import _lib
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models
L = sys.argv[1]
L = tuple(int(x) for x in L.split('-')) if '-' in L else int(L)
P, B = int(sys.argv[2]), int(sys.argv[3])
refs = sys.argv[4].split(',')
H, VP, VS, RHO, nl = draw_models(B, L, seed=3000, sorted_vs=True)
eng = ForwardEngine(swd=[SwdSpec(r, np.linspace(1, 41, P)) for r in refs])
d = eng.upload(H, VP, VS, RHO, nl)
out, err = eng.alloc_out(B)
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    return min(ts)
for form in sys.argv[5:]:
    _lib.set_swd_kernel(form)
    a = timeit(lambda: eng.run(d, out=out, err=err))
    b = timeit(lambda: eng.run(eng.reorder(d.packed, d.nlay, depth=d.depth, mean_depth=d.mean_depth), out=out, err=err))
    o = eng.reorder(d.packed, d.nlay, depth=d.depth, mean_depth=d.mean_depth)
    c = timeit(lambda: eng.run(o, out=out, err=err))
    print('L=%s P=%d B=%d %s %-8s natural %.2f ms  ordered (order computed per call) %.2f ms  ordered (order given) %.2f ms' % (sys.argv[1], P, B, '+'.join(refs), form, a, b, c), flush=True)
_lib.set_swd_kernel('auto')
# where the extra time of "order computed per call" sits: the order's own kernels, alone; then events around both parts
for form in sys.argv[5:6]:
    _lib.set_swd_kernel(form)
    ro = timeit(lambda: eng.reorder(d.packed, d.nlay, depth=d.depth, mean_depth=d.mean_depth))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(6)]
    for i in range(6):
        ev[i][0].record()
        o = eng.reorder(d.packed, d.nlay, depth=d.depth, mean_depth=d.mean_depth)
        ev[i][1].record()
        eng.run(o, out=out, err=err)
        ev[i][2].record()
    torch.cuda.synchronize()
    print('%s: order alone %.3f ms per call; inside the loop: order %s ms, kernels %s ms' % (
        form, ro, ' '.join('%.2f' % e[0].elapsed_time(e[1]) for e in ev), ' '.join('%.2f' % e[1].elapsed_time(e[2]) for e in ev)))
_lib.set_swd_kernel('auto')
