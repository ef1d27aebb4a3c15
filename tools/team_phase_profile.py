"""GPU-box diagnostic: where does a round of the 64-lane team kernel spend its cycles?

Builds a -DBH_TEAM_PROFILE copy of the library into gpurun_out/ (never the shipped one), runs
BASELINE-shaped batches through the 64-lane team kernel and prints shader-clock cycles per phase
(driver | plan | assemble | chain | consume), rounds per search and the wall time per call.

    python tools/team_phase_profile.py > gpurun_out/team_phase_profile.txt
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayhunter_amd import _lib  # noqa: E402


def profile_library():
    """The -DBH_TEAM_PROFILE build, keyed by the source hash: tools/_prof/ (git-ignored, travels to the GPU
    box: `python tools/team_phase_profile.py --build-only` here saves a minute of box time) or gpurun_out/."""
    extra = os.environ.get('BH_EXTRA_HIPCC_FLAGS', '').split()
    name = 'libbayhunter_amd_teamprof_%s.so' % _lib.source_hash()
    pre = os.path.join(ROOT, 'tools', '_prof', name)
    if os.path.exists(pre):
        return pre
    d = os.path.join(ROOT, 'tools', '_prof') if '--build-only' in sys.argv else os.path.join(ROOT, 'gpurun_out')
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, name)
    if not os.path.exists(so):
        subprocess.run(['/opt/rocm/bin/hipcc'] + _lib.HIPCC_FLAGS + extra + ['-DBH_TEAM_PROFILE'] + _lib.SOURCES + ['-o', so],
                       cwd=_lib.CSRC, check=True)
    return so


def main():
    so = profile_library()
    if '--build-only' in sys.argv:
        print(so)
        return
    _lib.LIB_PATH = so
    import torch
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    from bayhunter_amd.synthetic import draw_models
    lib = _lib.load()
    lib.bh_debug_team_profile.argtypes = [C.c_void_p, C.c_int]
    names = ['driver', 'plan:trial', 'assemble', 'chain', 'find+rest', 'fastfwd', 'control', 'driver2']
    extra = {13: 'plan:round', 14: 'tree nodes', 15: 'tree walk'}
    print('# %s' % torch.cuda.get_device_name(0))
    for L, P, B, mode in ((5, 20, 1024, 'team'), (5, 20, 64, 'team'), (15, 21, 64, 'team256'), (15, 21, 64, 'team512')):
        H, VP, VS, RHO, nl = draw_models(B, L, seed=100 + L)
        eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1, 41, P))])
        dm = eng.upload(H, VP, VS, RHO, nl)
        _lib.set_swd_kernel(mode)
        eng.run(dm)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 20)()
        lib.bh_debug_team_profile(buf, 1)
        t0 = time.perf_counter()
        for _ in range(5):
            eng.run(dm)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        lib.bh_debug_team_profile(buf, 1)
        v = np.array(list(buf), dtype=np.float64)
        nsearch, rounds = v[19], v[18]
        print('%s L=%d P=%d B=%d: %.3f ms per call, %.1f rounds per search (%.1f per period)'
              % (mode, L, P, B, ms, rounds / nsearch, rounds / nsearch / P))
        tot = v[:8].sum() + sum(v[i] for i in extra)
        for i, n in list(enumerate(names)) + sorted(extra.items()):
            print('   %-10s %8.0f cycles per round  (%4.1f %%)' % (n, v[i] / rounds, 100 * v[i] / tot))
        print('   total     %8.0f cycles per round, %.0f cycles per search; consumed %.2f of %.2f trials per round'
              % (tot / rounds, tot / nsearch, v[8] / rounds, v[9] / rounds))
        print('   per round: %.2f fast-forward steps, %.2f control calls, %.2f tree nodes taken without one'
              % (v[10] / rounds, v[11] / rounds, v[12] / rounds))
    _lib.set_swd_kernel('auto')


if __name__ == '__main__':
    main()
