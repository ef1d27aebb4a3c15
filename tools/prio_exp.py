import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
from bayhunter_amd.synthetic import draw_models
B = 524288
H, VP, VS, RHO, nl = draw_models(B, 10, seed=6000)
per = np.linspace(1, 41, 21)
eng = ForwardEngine(swd=[SwdSpec('rdispph', per)], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
dm = eng.upload(H, VP, VS, RHO, nl)
out, err = eng.alloc_out(B)
def bench(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.run(eng.reorder(dm.packed, dm.nlay), out=out, err=err)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            eng.run(eng.reorder(dm.packed, dm.nlay), out=out, err=err)
        torch.cuda.synchronize()
        print(label, 'ms/step %.2f' % ((time.perf_counter() - t0) / 8 * 1e3), flush=True)
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
bench(torch.cuda.current_stream(), 'default stream      ')
bench(torch.cuda.Stream(priority=-1), 'high-priority stream')
bench(torch.cuda.Stream(priority=0), 'plain stream        ')
eng.overlap = False
bench(torch.cuda.current_stream(), 'no overlap          ')
