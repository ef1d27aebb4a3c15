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
