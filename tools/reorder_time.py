import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models
B = 524288
H, VP, VS, RHO, nl = draw_models(B, 10, seed=1)
eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, 21))])
d = eng.upload(H, VP, VS, RHO, nl)
for _ in range(3): eng.reorder(d.packed, d.nlay)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m = eng.reorder(d.packed, d.nlay)
torch.cuda.synchronize(); print('reorder %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
o = m.order.cpu().numpy(); print(np.array_equal(np.sort(o), np.arange(B)))
