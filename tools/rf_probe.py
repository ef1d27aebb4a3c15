"""GPU-box diagnostic: what do the phases of rf_kernel cost?  Builds variants of the library into
gpurun_out/ (never the shipped one) and times rf_kernel for the bench batch and a 64-model batch:
  base        the shipped kernel
  nopad       per-model LDS block a multiple of 8 KiB (round 1)
  nobitrev    without the bit-reversal pass (wrong numbers; its cost)
  nofft       without bit reversal and butterflies
  nop3        without the frequency tasks (phase 3)
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [('base', []), ('nopad', ['-DBH_RF_NO_PAD']), ('nobitrev', ['-DBH_RF_EXP=1']), ('nofft', ['-DBH_RF_EXP=2']),
            ('nop3', ['-DBH_RF_EXP=3'])]

CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
from bayhunter_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import torch
from bayhunter_amd.engine import ForwardEngine, RfSpec
from bayhunter_amd.synthetic import draw_models
for B, L in ((64, 15), (524288, 10)):
    H, VP, VS, RHO, nl = draw_models(min(B, 4096), L, seed=6000)
    rep = B // H.shape[0]
    H, VP, VS, RHO, nl = (np.tile(a, (rep, 1)) if a.ndim == 2 else np.tile(a, rep) for a in (H, VP, VS, RHO, nl))
    eng = ForwardEngine(rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    d = eng.upload(H, VP, VS, RHO, nl)
    out, err = eng.alloc_out(B)
    eng.run(d, out=out, err=err); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 20 if B < 1000 else 5
    ev[0].record()
    for _ in range(n):
        eng.run(d, out=out, err=err)
    ev[1].record(); torch.cuda.synchronize()
    print('%%-9s B=%%6d L=%%2d  %%.3f ms' %% (sys.argv[2], B, L, ev[0].elapsed_time(ev[1]) / n))
""" % ROOT


def main():
    from bayhunter_amd import _lib
    for name, flags in VARIANTS:
        so = os.path.join(ROOT, 'gpurun_out', 'librf_%s.so' % name)
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(['/opt/rocm/bin/hipcc'] + _lib.HIPCC_FLAGS + flags + _lib.SOURCES + ['-o', so], cwd=_lib.CSRC, check=True)
        subprocess.run([sys.executable, '-c', CHILD, so, name], check=True)
        sys.stdout.flush()


if __name__ == '__main__':
    main()
