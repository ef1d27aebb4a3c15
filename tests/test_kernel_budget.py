"""Register budgets the design depends on, checked on the cross-compiled kernels (no GPU).

A joint step is 61 ms instead of 65 because one rf_kernel workgroup is resident beside swd_kernel's two waves
per SIMD: 2 x 192 + 128 VGPRs is all a SIMD has.  Six more registers in swd_kernel -- it
happened once, through an innocent-looking kernel argument -- and the two kernels run one after the other again
without any test failing."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _alloc(n):
    return -(-n // 8) * 8          # VGPRs are allocated in blocks of 8


def test_swd_and_rf_kernels_fit_one_simd_together():
    from kernel_resources import kernel_resources
    r = kernel_resources()
    assert set(r) >= {'swd_kernel', 'rf_kernel<false>', 'rf_kernel<true>', 'swd_team_kernel', 'swd_team8_kernel',
                      'swd_team512_kernel'}, sorted(r)
    assert all(v['scratch'] == 0 for v in r.values()), r          # nothing spills to memory
    assert _alloc(r['swd_kernel']['vgpr']) <= 192 and _alloc(r['rf_kernel<false>']['vgpr']) <= 128, r
    assert 2 * _alloc(r['swd_kernel']['vgpr']) + _alloc(r['rf_kernel<false>']['vgpr']) <= 512
    # a multi-target call that is latency-bound on the lane kernel moves its heaviest target to 128-lane teams beside it
    # (capi.hip: plan_forms; BASELINE cfg3: 22.7 -> 17 ms): one lane-kernel wave and two team waves share a SIMD.
    # (Ten more registers in the wide teams -- a Neville table in register lanes, round 4 -- and cfg3 took 28.7 ms.)
    assert _alloc(r['swd_kernel']['vgpr']) + 2 * _alloc(r['swd_team128_kernel']['vgpr']) <= 512, r
    # the narrow teams (8 .. 32 lanes per search, a lane's whole period equation like swd_kernel) keep two waves per SIMD
    # -- built for three they take 168 VGPRs + 96-112 B of scratch: cfg3 12.7 -> 12.0 ms but cfg5 4.4 -> 5.5 ms, its 2 048
    # waves no longer one full layer of the chip (profiles/r04_ab_narrow_waves.txt) --, the teams of 128 .. 512 lanes three,
    # the one-wave team four (one trial per lane in 127 VGPRs)
    for k, v in r.items():
        if k.startswith('swd_team'):
            narrow = k in ('swd_team8_kernel', 'swd_team16_kernel', 'swd_team32_kernel')
            assert _alloc(v['vgpr']) <= (256 if narrow else 168), (k, v)
    assert _alloc(r['swd_team_kernel']['vgpr']) <= 128, r
