"""How much does the REFERENCE differ from ITSELF when only the libm underneath it changes?

glibc ships several builds of sin/cos/exp (x86-64: `__sin_fma`, `__sin_avx`/`__sin_sse2`, ...)
and picks one per process by CPU feature (ifunc).  `GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2`
masks the FMA variants, i.e. it makes this CPU behave like one without FMA.  The reference's
dispersion code is untouched; only the last bit of some sin/cos/exp results moves -- which is
exactly what separates the device math (bh_math.h) from glibc.

Usage (development container; `oracle/_ref` is used when present, else the C restatement):

    python tests/scenarios/libm_selfdiff.py [models_per_set] > profiles/r02_libm_selfdiff.txt

The child mode (`--child tunables out.npz B`) computes one table of results.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REFS = [('rdispph', 2, 0), ('rdispgr', 2, 1), ('ldispph', 1, 0), ('ldispgr', 1, 1)]
SETS = [(L, srt) for L in (5, 10, 15, (2, 31)) for srt in (True, False)]
MASK = 'glibc.cpu.hwcaps=-FMA,-AVX2,-FMA4'


def set_seed(L, srt):
    return 4242 + (sum(L) if isinstance(L, tuple) else L) * 2 + int(srt)


def compute(B, backend, nthreads=1):
    from bayhunter_amd.synthetic import draw_models
    from oracle import pyoracle as po
    per = np.linspace(1, 41, 21)
    d = {}
    for L, srt in SETS:
        H, VP, VS, RHO, nl = draw_models(B, L, seed=set_seed(L, srt), sorted_vs=srt)
        tag = 'L%s_%s' % (L, 'sorted' if srt else 'lvz')
        for name, iw, ig in REFS:
            out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, backend=backend, nthreads=nthreads)
            d[tag + '_' + name] = out
            d[tag + '_' + name + '_err'] = err
        d[tag + '_prf'] = po.rf_batch(H, VP, VS, RHO, nl, backend=backend, nthreads=nthreads)
    return d


def probe_variant():
    """Which libm variant does this process run?  sin(x) for an argument where the FMA and the
    non-FMA builds of glibc 2.35 round differently is not portable knowledge; report the ifunc
    decision indirectly through /proc/cpuinfo + the tunable."""
    flags = open('/proc/cpuinfo').read()
    has = ' fma ' in flags and ' avx2 ' in flags
    masked = '-FMA' in os.environ.get('GLIBC_TUNABLES', '')
    return 'fma' if (has and not masked) else 'no-fma'


def run_child(tunables, B, backend):
    env = dict(os.environ)
    if tunables:
        env['GLIBC_TUNABLES'] = tunables
    else:
        env.pop('GLIBC_TUNABLES', None)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'r.npz')
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', path, str(B), backend],
                       check=True, env=env)
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


def compare(a, b):
    """Rows of (set, target, err_equal, identical fraction, max abs, max rel)."""
    rows = []
    for L, srt in SETS:
        tag = 'L%s_%s' % (L, 'sorted' if srt else 'lvz')
        for name, _, _ in REFS:
            ea, eb = a[tag + '_' + name + '_err'], b[tag + '_' + name + '_err']
            ok = (ea == 0) & (eb == 0)
            x, y = a[tag + '_' + name][ok], b[tag + '_' + name][ok]
            dd = np.abs(x - y)
            rel = dd / np.maximum(np.abs(x), 1e-9)
            rows.append((tag, name, bool(np.array_equal(ea, eb)), float((dd == 0).mean()), float(dd.max()),
                         float(rel.max())))
        x, y = a[tag + '_prf'], b[tag + '_prf']
        fin = np.isfinite(x) & np.isfinite(y)
        dd = np.abs(x[fin] - y[fin])
        rows.append((tag, 'prf', bool(np.array_equal(np.isfinite(x), np.isfinite(y))), float((dd == 0).mean()),
                     float(dd.max()), float(dd.max() / max(1.0, np.abs(x[fin]).max()))))
    return rows


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        path, B, backend = sys.argv[2], int(sys.argv[3]), sys.argv[4]
        d = compute(B, backend, nthreads=len(os.sched_getaffinity(0)))
        d['variant'] = np.array(probe_variant())
        np.savez(path, **d)
        return
    from oracle import pyoracle as po
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    backend = 'ref' if po.have_ref() else 'port'
    a = run_child('', B, backend)
    b = run_child(MASK, B, backend)
    print('# reference (%s) against itself: glibc libm variant %s vs %s (GLIBC_TUNABLES=%s)'
          % ('oracle/_ref: flang/g++ build of the reference sources' if backend == 'ref' else 'C restatement',
             a['variant'], b['variant'], MASK))
    print('# %d models per set, 21 periods; values where both runs solved' % B)
    print('%-20s %-8s %-9s %-10s %-10s %-10s' % ('set', 'target', 'err_equal', 'identical', 'max_abs', 'max_rel'))
    for r in compare(a, b):
        print('%-20s %-8s %-9s %-10.5f %-10.3e %-10.3e' % r)


if __name__ == '__main__':
    main()
