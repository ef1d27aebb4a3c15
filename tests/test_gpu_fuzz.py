"""GPU tier (-m gpu): short runs of the random campaigns under tests/scenarios/ (the long runs are
recorded in profiles/r02_kernel_fuzz.txt and profiles/r02_rf_fuzz.txt).

kernel_fuzz: random dispersion configurations (batch, depth, LVZ, water, irregular periods, wave types,
modes, flat/spherical, several targets) through all eight kernel forms -- bit-identical to the throughput
kernel; err flags, zero fill and velocity-increasing fundamental-mode values equal to the oracle's.
rf_fuzz: random receiver-function configurations (depth, LVZ, Gauss factor, slowness, every transform
length, sampling rate, P/SV, rotation velocity) against the oracle: NaN patterns identical, <= 1e-10."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))


def test_dispersion_kernel_forms_random_configurations(lib, oracle, capsys):
    import kernel_fuzz
    assert kernel_fuzz.main(seconds=20.0, seed=20261004) == 0
    assert 'all forms bit-identical' in capsys.readouterr().out


def test_receiver_function_random_configurations(lib, oracle, capsys):
    import rf_fuzz
    assert rf_fuzz.main(seconds=12.0, seed=20261004) == 0
    assert 'NaN patterns identical' in capsys.readouterr().out


def test_lvz_worst_cases_on_the_gpu(lib, oracle):
    """The models behind the campaign's largest deviations (tests/golden/lvz_worst_cases.npz): the GPU returns
    what the CPU replay of the device program returned when the fixture was made, every value is inside the
    bound with the conditioning term, and the worst value of each case is one of the values the REFERENCE
    itself returns there when its libm is accurate to 1 ulp (committed runs of the reference binary under
    tests/scenarios/ulp_noise_libm.c)."""
    import numpy as np
    from bayhunter_amd.engine import ForwardEngine, SwdSpec
    from tolerances import TOL_PHASE_2BRACKETS, group_bound
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'lvz_worst_cases.npz'))
    names = {(2, 0): 'rdispph', (2, 1): 'rdispgr', (1, 0): 'ldispph', (1, 1): 'ldispgr'}
    for i in range(int(z['ncases'])):
        p = 'c%d_' % i
        eng = ForwardEngine(swd=[SwdSpec(names[(int(z[p + 'iwave']), int(z[p + 'igr']))], z[p + 'per'],
                                         mode=int(z[p + 'mode']), flsph=int(z[p + 'fl']))])
        out, err = eng.run(*[z[p + k] for k in ('H', 'VP', 'VS', 'RHO', 'nl')])
        out, err = out.cpu().numpy(), err.cpu().numpy()
        ref, noise, c = z[p + 'ref_fma'], z[p + 'ref_noise'], z[p + 'ref_phase']
        assert np.array_equal(err[:, 0], z[p + 'ref_err']), p
        nz = (ref != 0) & (z[p + 'ref_err'] == 0)[:, None]
        rel = np.zeros_like(ref)
        rel[nz] = np.abs(out[nz] - ref[nz]) / np.abs(ref[nz])
        bound = group_bound(ref, c) if int(z[p + 'igr']) else np.full_like(ref, TOL_PHASE_2BRACKETS)
        assert np.all(rel <= bound), (p, rel.max())
        assert np.array_equal(out, z[p + 'device_replay']), p           # the replay IS the device program
        b, k = np.unravel_index(np.argmax(rel), rel.shape)
        assert out[b, k] in set(noise[:, b, k].tolist()), (p, out[b, k])
