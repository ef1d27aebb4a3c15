"""CPU: rounds per period of the wide-team search plan (swd_team.h) on the bench shapes, from the host replay
(tests/hostsim: the device's plan / consume code compiled with g++, glibc math).  The testbed for changes
of the speculation policy -- no GPU needed; the replay also checks the plan's invariants and that the values
are the reference's.

    python tools/replay_rounds.py [nmodels] [extra g++ flags, e.g. -DSWD_TEAMW_MIDROOM=16]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

SHAPES = [  # name, layers, periods, lanes, igr
    ('cfg2  5 layers x 64 lanes', 5, 20, 64, 0),
    ('cfg2  5 layers x 128 lanes', 5, 20, 128, 0),
    ('cfg2  5 layers x 256 lanes', 5, 20, 256, 0),
    ('pool 10 layers x 64 lanes', 10, 21, 64, 0),
    ('cfg5 ragged x 128 lanes', (2, 31), 21, 128, 0),
    ('cfg4 15 layers x 256 lanes', 15, 21, 256, 0),
    ('cfg4 15 layers x 512 lanes', 15, 21, 512, 0),
    ('cfg3 rdispgr 10 layers x 128 lanes', 10, 40, 128, 1),
    # narrow teams, one trial per lane (swd_tpl_body): slots per round = lanes
    ('10 layers x 8 lanes', 10, 21, 8, 0),
    ('10 layers x 16 lanes', 10, 21, 16, 0),
    ('10 layers x 32 lanes', 10, 21, 32, 0),
    ('cfg5 ragged x 16 lanes', (2, 31), 21, 16, 0),
    ('cfg3 rdispgr 10 layers x 8 lanes', 10, 40, 8, 1),
    ('cfg3 ldispph 10 layers x 8 lanes', 10, 40, 8, 0, 1),
]
ST = ['A (entry)', 'B (scan)', 'TOP', 'MID']


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 48
    extra = [a for a in sys.argv[1:] if a.startswith('-')]
    import conftest
    so = os.path.join(ROOT, 'gpurun_out', 'libhostsim_replay.so')
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-DBH_HOSTSIM_GLIBC_MATH'] + extra +
                   ['-o', so, os.path.join(ROOT, 'tests', 'hostsim', 'hostsim.cpp')], check=True)
    raw = C.CDLL(so)
    hs = conftest._wrap_hostsim(raw)
    from bayhunter_amd.synthetic import draw_models
    for shape in SHAPES:
        name, L, P, lanes, igr = shape[:5]
        iwave = 2 if len(shape) < 6 else shape[5]
        H, VP, VS, RHO, nl = draw_models(n, L, seed=100 + (L if isinstance(L, int) else 99), sorted_vs=True)
        per = np.linspace(1, 41, P)
        hist = np.zeros(4 * 3 * 16, dtype=np.int64)
        raw.hs_teamw_set_histogram(hist.ctypes.data_as(C.POINTER(C.c_long)))
        calls = spec = rounds = 0
        for b in range(n):
            k = nl[b]
            _, e, nc, ns, nr = hs.swd_team(H[b, :k], VP[b, :k], VS[b, :k], RHO[b, :k], per, iwave, igr, 1, 0, lanes, wide=True)
            assert e >= 0, e
            calls += nc; spec += ns; rounds += nr
        raw.hs_teamw_set_histogram(None)
        h = hist.reshape(4, 3, 16)
        print('%-36s rounds/period %.2f   consumed/round %.2f of %.2f evaluated   evaluations/period %.1f'
              % (name, rounds / (n * P), calls / rounds, spec / rounds, calls / (n * P)))
        for st in range(4):
            for nev in range(3):
                c = h[st, nev]
                if c.sum():
                    mean = (c * np.arange(16)).sum() / c.sum()
                    print('      planned in %-10s nev=%d: %5.2f rounds/period, mean consumed %.1f   %s'
                          % (ST[st], nev, c.sum() / (n * P), mean, ' '.join('%d' % x for x in c)))


if __name__ == '__main__':
    main()
