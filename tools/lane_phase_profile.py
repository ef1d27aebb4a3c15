"""GPU-box diagnostic: how do the lane-cycles of swd_kernel split between the driver (events, task fetch),
the period equation and the control code?

Builds a -DBH_LANE_PROFILE copy of the library into gpurun_out/ (never the shipped one) and runs the
bench's joint10 models through the throughput kernel.

    python tools/lane_phase_profile.py [models] > gpurun_out/lane_phase_profile.txt
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayhunter_amd import _lib  # noqa: E402


def main(B=131072):
    so = os.path.join(ROOT, 'gpurun_out', 'libbayhunter_amd_laneprof.so')
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc'] + _lib.HIPCC_FLAGS + ['-DBH_LANE_PROFILE'] + _lib.SOURCES + ['-o', so],
                   cwd=_lib.CSRC, check=True)
    _lib.LIB_PATH = so
    import torch
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    from bayhunter_amd.synthetic import draw_models
    lib = _lib.load()
    lib.bh_debug_lane_profile.argtypes = [C.c_void_p, C.c_int]
    print('# %s, %d ten-layer models, Rayleigh phase, 21 periods' % (torch.cuda.get_device_name(0), B))
    H, VP, VS, RHO, nl = draw_models(B, 10, seed=6000, sorted_vs=True)
    eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, 21))])
    dm = eng.upload(H, VP, VS, RHO, nl)
    _lib.set_swd_kernel('lane')
    eng.run(dm)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    lib.bh_debug_lane_profile(buf, 1)
    eng.run(dm)
    torch.cuda.synchronize()
    lib.bh_debug_lane_profile(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    tot = v[:3].sum()
    for name, x in zip(('driver', 'period equation', 'control'), v[:3]):
        print('%-16s %5.1f %% of the lane-cycles, %7.0f cycles per evaluation' % (name, 100 * x / tot, x / v[3]))
    print('   of the control code, Neville steps: %.1f %% of the kernel (wave-cycles in swd_neville x 64 / lane-cycles)'
          % (100 * 64 * v[4] / tot))
    print('evaluations per search %.1f' % (v[3] / B))
    _lib.set_swd_kernel('auto')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 131072)
