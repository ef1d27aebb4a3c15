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

CHILD = os.path.join(ROOT, 'tools', 'rf_probe_child.py')


def main():
    from bayhunter_amd import _lib
    for name, flags in VARIANTS:
        so = os.path.join(ROOT, 'gpurun_out', 'librf_%s.so' % name)
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(['/opt/rocm/bin/hipcc'] + _lib.HIPCC_FLAGS + flags + _lib.SOURCES + ['-o', so], cwd=_lib.CSRC, check=True)
        subprocess.run([sys.executable, CHILD, so, name], check=True)
        sys.stdout.flush()


if __name__ == '__main__':
    main()
