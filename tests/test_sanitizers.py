"""Sanitizers on the CPU build of the chain pool's host code (csrc/chains.cpp; GPU sanitizers are not
available on the pool): AddressSanitizer + UBSan for memory and arithmetic errors, ThreadSanitizer
for the helper-thread protocol (spinning workers, job hand-over, pools of different size stepped
alternately).  The driver's checksum must not depend on the number of threads."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = [os.path.join(ROOT, 'tests', 'sanitize', 'chains_driver.cpp'),
       os.path.join(ROOT, 'bayhunter_amd', 'csrc', 'chains.cpp')]


def _run(tmp_path, flags, threads, env=None, order='descending', lookahead=False):
    exe = str(tmp_path / ('driver_' + '_'.join(f.strip('-=').replace(',', '_') for f in flags)))
    c = subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-pthread', '-ffp-contract=off'] + flags + SRC + ['-o', exe],
                       capture_output=True, text=True)
    if c.returncode != 0 and flags and ('cannot find' in c.stderr or 'sanitizer' in c.stderr.lower()):
        pytest.skip('sanitizer runtime not installed: ' + c.stderr[-300:])
    assert c.returncode == 0, c.stderr[-4000:]
    e = dict(os.environ, BH_CHAIN_SPIN_US='200')
    e.update(env or {})
    r = subprocess.run([exe, str(threads), order] + (['lookahead'] if lookahead else []), capture_output=True, text=True,
                       timeout=600, env=e)
    if flags and ('unexpected memory mapping' in r.stderr or 'Shadow memory range interleaves' in r.stderr
                  or 'ReserveShadowMemoryRange failed' in r.stderr):
        pytest.skip('the sanitizer runtime cannot start in this environment: ' + r.stderr[-300:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'runtime error' not in r.stderr and 'WARNING: ThreadSanitizer' not in r.stderr, r.stderr[-4000:]
    return [l for l in r.stdout.splitlines() if l.startswith('checksum')][0]


def test_chain_pool_host_code_under_asan_ubsan_and_tsan(tmp_path):
    plain1 = _run(tmp_path, [], 1)
    asan = _run(tmp_path, ['-fsanitize=address,undefined', '-fno-omit-frame-pointer'], 8,
                env={'ASAN_OPTIONS': 'detect_leaks=1', 'UBSAN_OPTIONS': 'print_stacktrace=1'})
    tsan = _run(tmp_path, ['-fsanitize=thread'], 8, env={'TSAN_OPTIONS': 'halt_on_error=1'})
    assert plain1 == asan == tsan
    # the look-ahead trees (2, 5 and 16 proposals per chain and call): same samples, clean under both
    assert _run(tmp_path, ['-fsanitize=address,undefined', '-fno-omit-frame-pointer'], 8, lookahead=True,
                env={'ASAN_OPTIONS': 'detect_leaks=1', 'UBSAN_OPTIONS': 'print_stacktrace=1'}) == plain1
    assert _run(tmp_path, ['-fsanitize=thread'], 8, env={'TSAN_OPTIONS': 'halt_on_error=1'}, lookahead=True) == plain1
    # smallest pool first: helper threads are created while earlier jobs have already been published
    # (the order smoke() -> a larger ChainPool, or tools/chain_bench.py with growing pools, produces)
    for _ in range(3):
        assert _run(tmp_path, ['-fsanitize=thread'], 8, env={'TSAN_OPTIONS': 'halt_on_error=1'},
                    order='ascending') == plain1


def test_device_solver_cores_under_asan_ubsan(tmp_path):
    """swd_core.h / swd_team.h / rf_core.h compiled for the host (tests/hostsim) over 220 pseudo-random
    models incl. 1 and 100 layers, water layer, LVZ, modes 1-3, 60 periods, four FFT lengths."""
    src = os.path.join(ROOT, 'tests', 'sanitize', 'cores_driver.cpp')
    exe = str(tmp_path / 'cores')
    c = subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-ffp-contract=off', '-fsanitize=address,undefined',
                        '-fno-omit-frame-pointer', src, '-o', exe], capture_output=True, text=True)
    if c.returncode != 0 and ('cannot find' in c.stderr or 'sanitizer' in c.stderr.lower()):
        pytest.skip('sanitizer runtime not installed: ' + c.stderr[-300:])
    assert c.returncode == 0, c.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, UBSAN_OPTIONS='print_stacktrace=1'))
    if 'Shadow memory range interleaves' in r.stderr or 'ReserveShadowMemoryRange failed' in r.stderr:
        pytest.skip('the sanitizer runtime cannot start in this environment: ' + r.stderr[-300:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'cores ok' in r.stdout and 'runtime error' not in r.stderr, r.stderr[-4000:]
