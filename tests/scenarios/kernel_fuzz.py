"""GPU-box campaign: random dispersion configurations through every kernel form.

All kernel forms must return the throughput kernel's bits (values, err flags), and the throughput
kernel is checked against the oracle: err flags and zero fill equal, velocity-increasing fundamental-mode
flat-earth values without a water layer EXACT, everything else (low-velocity zones, water layers, higher
modes, earth flattening) within the
derived bounds of tests/tolerances.py -- phase 2.2e-6, group max(2.5e-4, 4.2e-4 |U/c|) -- and the run
fails on the first value outside them.  Configurations are drawn at random: batch size, depth (uniform or
ragged, 1..40 layers), low-velocity zones, water layer, irregular period lists (1..60 periods), wave
type, phase / group, modes 1..3, flat / spherical, several targets per launch.

    python tests/scenarios/kernel_fuzz.py [seconds] [seed]  > gpurun_out/kernel_fuzz.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayhunter_amd import _lib  # noqa: E402
from bayhunter_amd.engine import ForwardEngine, SwdSpec  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tolerances import TOL_PHASE_2BRACKETS, group_bound  # noqa: E402

REFS = {'rdispph': (2, 0), 'rdispgr': (2, 1), 'ldispph': (1, 0), 'ldispgr': (1, 1)}
FORMS = ('team', 'team128', 'team256', 'team512', 'team32', 'team16', 'team8')


def draw_config(rs):
    """One random configuration; consumes `rs` in a fixed order, so configuration i of a seed is the same
    whoever draws it (tests/scenarios/lvz_worst_cases.py re-derives the models behind a campaign line)."""
    B = int(rs.choice([1, 3, 17, 64, 200, 700]))
    lo = int(rs.randint(1, 20))
    L = lo if rs.rand() < 0.5 else (lo, int(lo + rs.randint(0, 21)))
    srt = rs.rand() < 0.5
    deep = (L if isinstance(L, int) else L[1]) > 25
    H, VP, VS, RHO, nl = draw_models(B, L, seed=int(rs.randint(1 << 30)), sorted_vs=srt,
                                     **(dict(zmax=200.0, thickmin=0.05) if deep else {}))
    water = False
    if rs.rand() < 0.2:                                   # water on top of some models
        w = (rs.rand(B) < 0.5) & (nl > 2)
        water = bool(w.any())
        H[w, 0] = rs.uniform(0.3, 4.0, size=int(w.sum()))
        VP[w, 0], VS[w, 0], RHO[w, 0] = 1.5, 0.0, 1.03
    nper = int(rs.choice([1, 2, 5, 13, 21, 40, 60]))
    per = np.sort(rs.uniform(0.8, 60.0, size=nper)) if rs.rand() < 0.5 else np.linspace(1, 41, nper)
    refs = list(rs.choice(sorted(REFS), size=int(rs.randint(1, 4)), replace=False))
    mode = int(rs.choice([1, 1, 1, 2, 3]))
    fl = int(rs.rand() < 0.25)
    tag = 'B=%d L=%s %s per=%d %s mode=%d fl=%d' % (B, L, 'sorted' if srt else 'lvz', nper, '+'.join(refs), mode, fl)
    return dict(B=B, L=L, srt=srt, H=H, VP=VP, VS=VS, RHO=RHO, nl=nl, per=per, nper=nper, refs=refs, mode=mode,
                fl=fl, tag=tag + (' water' if water else ''), water=water)


def main(seconds=300.0, seed=1):
    rs = np.random.RandomState(seed)
    threads = min(len(os.sched_getaffinity(0)), 16)
    t_end = time.time() + seconds
    ncfg = nsearch = nmixed = 0
    worst = dict(phase=0.0, group=0.0)
    worst_plain = dict(phase=0.0, group=0.0)          # fundamental mode, flat earth only
    where = dict(phase='', group='')
    nbound = dict(phase=0, group=0)                    # values checked against the derived LVZ bounds
    margin = dict(phase=0.0, group=0.0)                # largest deviation / bound seen
    while time.time() < t_end:
        cfg = draw_config(rs)
        B, L, srt, H, VP, VS, RHO, nl, per, nper, refs, mode, fl = (cfg[k] for k in (
            'B', 'L', 'srt', 'H', 'VP', 'VS', 'RHO', 'nl', 'per', 'nper', 'refs', 'mode', 'fl'))
        eng = ForwardEngine(swd=[SwdSpec(r, per, mode=mode, flsph=fl) for r in refs])
        res = {}
        for form in ('lane',) + FORMS:
            _lib.set_swd_kernel(form)
            try:
                out, err = eng.run(H, VP, VS, RHO, nl)
                res[form] = (out.cpu().numpy(), err.cpu().numpy())
            finally:
                _lib.set_swd_kernel('auto')
        if len(refs) > 1:        # the targets of one call on two or three different forms, concurrent streams
            pick = list(rs.choice(('lane',) + FORMS, size=int(rs.randint(2, 4)), replace=False))
            assign = [pick[i % len(pick)] for i in rs.permutation(len(refs))]
            _lib.set_swd_forms(assign)
            try:
                out, err = eng.run(H, VP, VS, RHO, nl)
                res['mixed'] = (out.cpu().numpy(), err.cpu().numpy())
            finally:
                _lib.set_swd_forms(None)
            nmixed += 1
        tag = cfg['tag']
        for form in FORMS + (('mixed',) if 'mixed' in res else ()):
            if not (np.array_equal(res['lane'][0], res[form][0], equal_nan=True) and
                    np.array_equal(res['lane'][1], res[form][1])):
                bad = np.argwhere(res['lane'][0] != res[form][0])
                print('MISMATCH %s vs lane: %s  first at %s' % (form, tag, bad[:3].tolist()), flush=True)
                np.savez(os.path.join(ROOT, 'gpurun_out', 'fuzz_fail_%d.npz' % ncfg), H=H, VP=VP, VS=VS, RHO=RHO, nl=nl,
                         per=per, refs=np.array(refs), mode=mode, fl=fl, lane=res['lane'][0], other=res[form][0])
                return 1
        out, err = res['lane']
        for t, r in enumerate(refs):
            iw, ig = REFS[r]
            want, werr, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, mode, fl, nthreads=threads)
            if not np.array_equal(err[:, t], werr):
                print('ERR FLAGS differ from the oracle: %s target %s' % (tag, r), flush=True)
                return 1
            ok = werr == 0
            got, ref = out[ok][:, t * nper:(t + 1) * nper], want[ok]
            nz = ref != 0                                       # (higher modes: zero-filled beyond cut-off)
            if not np.array_equal(got == 0, ref == 0):
                print('ZERO FILL differs from the oracle: %s target %s' % (tag, r), flush=True)
                return 1
            if nz.any():
                relv = np.abs(got - ref) / np.where(nz, np.abs(ref), 1.0)
                relv[~nz] = 0.0
                rel = float(relv.max())
                key = 'group' if ig else 'phase'
                if rel > worst[key]:
                    worst[key], where[key] = rel, tag + ' target ' + r
                if mode == 1 and not fl:
                    worst_plain[key] = max(worst_plain[key], rel)
                # velocity increasing with depth, no water layer on top (whose tail of the period equation is
                # as ill-conditioned as a low-velocity zone: 4.4e-5 seen in a group velocity): exact
                if srt and mode == 1 and not fl and not cfg['water']:
                    if rel > 0.0:
                        print('VALUE differs from the oracle on monotone models: %s target %s rel %.3e' % (tag, r, rel), flush=True)
                        return 1
                    continue
                # Everything else is asserted against the derived bound (tests/tolerances.py): two searches
                # stop anywhere inside their own 1e-6 brackets; a group velocity amplifies that by U/(c h),
                # unbounded where the finite difference of surfdisp96.f:306 nearly vanishes (U >> c) --
                # tests/golden/lvz_worst_cases.npz shows the reference itself moving that far there.
                if ig:
                    cph, _, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, 0, mode, fl, nthreads=threads)
                    lim = group_bound(ref, cph[ok])
                else:
                    lim = np.full_like(ref, TOL_PHASE_2BRACKETS)
                if fl:                                          # sphere: powf/log differ in the last bit too
                    lim = lim * 4
                over = relv > lim
                if over.any():
                    b, k = np.argwhere(over)[0]
                    print('VALUE outside the derived bound: %s target %s: rel %.3e > %.3e (model %d of the solved ones, '
                          'period %d: %.9g vs %.9g)' % (tag, r, relv[b, k], lim[b, k], b, k, got[b, k], ref[b, k]), flush=True)
                    return 1
                nbound[key] += int(nz.sum())
                margin[key] = max(margin[key], float((relv / lim).max()))
        ncfg += 1
        nsearch += B * len(refs)
        if ncfg % 20 == 0:
            print('%d configurations, %d searches per form, worst rel. deviation from the oracle: phase %.2e group %.2e'
                  % (ncfg, nsearch, worst['phase'], worst['group']), flush=True)
    print('DONE: %d configurations, %d searches x %d forms, all forms bit-identical to the throughput kernel; '
          'worst relative deviation from the oracle: phase %.2e, group %.2e' % (ncfg, nsearch, len(FORMS) + 1,
                                                                                 worst['phase'], worst['group']))
    print('   %d multi-target configurations also with their targets on two or three different forms in one call '
          '(bh_swd_set_forms; launches on concurrent streams): bit-identical' % nmixed)
    print('   fundamental mode, flat earth only: phase %.2e, group %.2e' % (worst_plain['phase'], worst_plain['group']))
    print('   non-monotone / higher-mode / spherical values asserted against the derived bounds (phase 2.2e-6; group '
          'max(2.5e-4, 4.2e-4 |U/c|)): %d phase, %d group; largest deviation / bound: %.2f, %.2f'
          % (nbound['phase'], nbound['group'], margin['phase'], margin['group']))
    print('   worst phase: %s' % where['phase'])
    print('   worst group: %s' % where['group'])
    return 0


if __name__ == '__main__':
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
