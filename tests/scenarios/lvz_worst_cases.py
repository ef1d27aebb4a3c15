"""The worst low-velocity-zone deviations of the random campaign, traced to the reference's own behaviour.

profiles/r02_kernel_fuzz.txt reports, against the oracle, up to 1.99e-6 (phase) and 1.51e-2 / 4.5e-3
(group velocity) on fundamental- and higher-mode LVZ models -- far above the bounds the fixed test sets
assert.  This script

  1. re-derives the configurations behind those lines from the campaign's seeds
     (kernel_fuzz.draw_config: configuration i of a seed is deterministic) and keeps the models with the
     largest deviations, found with the CPU replay of the device program (tests/hostsim, device math);
  2. runs the REFERENCE's native code (oracle/_ref) on them three ways: glibc's FMA libm, glibc's non-FMA
     libm (GLIBC_TUNABLES, as tests/scenarios/libm_selfdiff.py), and under tests/scenarios/
     ulp_noise_libm.c -- glibc's sin/cos/exp moved by <= 1 ulp, 24 noise seeds;
  3. writes tests/golden/lvz_worst_cases.npz: inputs + all of those outputs + the device replay + the
     phase velocities that enter the conditioning term U/c.

Development container only (needs /root/reference's build in oracle/_ref for `backend=ref`; falls back to
the C restatement, which is bit-identical to it under every libm variant tried).

    python tests/scenarios/lvz_worst_cases.py            # regenerate the fixture
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)

REFS = {'rdispph': (2, 0), 'rdispgr': (2, 1), 'ldispph': (1, 0), 'ldispgr': (1, 1)}
# (campaign seed, configuration index, its tag in profiles/r02_kernel_fuzz.txt, target, models kept)
CASES = [
    (7, 2525, 'B=700 L=3 lvz per=60 rdispgr+ldispph+ldispgr mode=1 fl=0', 'rdispgr', 3),
    (21, 640, 'B=700 L=9 lvz per=1 rdispph+rdispgr mode=2 fl=0', 'rdispgr', 1),
    (7, 10358, 'B=700 L=(4, 24) lvz per=40 rdispgr+rdispph+ldispgr mode=1 fl=0', 'rdispph', 3),
    (7, 10358, 'B=700 L=(4, 24) lvz per=40 rdispgr+rdispph+ldispgr mode=1 fl=0', 'rdispgr', 3),
    (41, 1298, 'B=700 L=(18, 22) lvz per=60 ldispph+rdispgr+rdispph mode=1 fl=0', 'rdispph', 3),
]
NOISE_SEEDS = 24
MASK = 'glibc.cpu.hwcaps=-FMA,-AVX2,-FMA4'
FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'lvz_worst_cases.npz')


def build_noise_lib():
    so = os.path.join(HERE, 'libulpnoise.so')
    src = os.path.join(HERE, 'ulp_noise_libm.c')
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.run(['gcc', '-O2', '-fPIC', '-shared', '-o', so, src, '-ldl', '-lm'], check=True)
    return so


def variant_env(variant):
    """Environment of a child that runs the oracle under a libm variant: 'fma' (this CPU's default),
    'nofma' (GLIBC_TUNABLES masks the FMA builds of sin/cos/exp) or 'noise<seed>' (the <= 1 ulp shim)."""
    env = dict(os.environ)
    env.pop('GLIBC_TUNABLES', None)
    env.pop('LD_PRELOAD', None)
    if variant == 'nofma':
        env['GLIBC_TUNABLES'] = MASK
    elif variant.startswith('noise'):
        env['LD_PRELOAD'] = build_noise_lib()
        env['BH_ULP_NOISE_SEED'] = variant[5:]
    return env


def solve_children(fixture_path, backend, variants):
    """{variant: {case key: values}}: one child process per libm variant, each solving every case of the
    fixture (group/phase target + the phase velocity) with `backend`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for v in variants:
            path = os.path.join(td, v + '.npz')
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', fixture_path, backend, path],
                           check=True, env=variant_env(v))
            with np.load(path) as z:
                out[v] = {k: z[k] for k in z.files}
    return out


def child(fixture_path, backend, out_path):
    from oracle import pyoracle as po
    z = np.load(fixture_path)
    res = {}
    for i in range(int(z['ncases'])):
        p = 'c%d_' % i
        iw, ig = int(z[p + 'iwave']), int(z[p + 'igr'])
        a = [z[p + k] for k in ('H', 'VP', 'VS', 'RHO', 'nl', 'per')]
        res[p + 'val'], res[p + 'err'], _ = po.swd_batch(*a, iw, ig, int(z[p + 'mode']), int(z[p + 'fl']), backend=backend)
        res[p + 'phase'], _, _ = po.swd_batch(*a, iw, 0, int(z[p + 'mode']), int(z[p + 'fl']), backend=backend)
    np.savez(out_path, **res)


def device_replay(z, i):
    """The device program with the device's math, replayed on the CPU (tests/hostsim)."""
    import conftest
    fma = ['-mfma'] if ' fma ' in open('/proc/cpuinfo').read() else []
    hs = conftest._wrap_hostsim(conftest._build_hostsim('libhostsim_devmath.so', fma))
    p = 'c%d_' % i
    H, VP, VS, RHO, nl, per = (z[p + k] for k in ('H', 'VP', 'VS', 'RHO', 'nl', 'per'))
    out = np.zeros((H.shape[0], per.size))
    for b in range(H.shape[0]):
        n = int(nl[b])
        out[b] = hs.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, int(z[p + 'iwave']), int(z[p + 'igr']),
                        int(z[p + 'mode']), int(z[p + 'fl']))[0]
    return out


def main():
    from kernel_fuzz import draw_config
    from oracle import pyoracle as po
    import conftest
    backend = 'ref' if po.have_ref() else 'port'
    fma = ['-mfma'] if ' fma ' in open('/proc/cpuinfo').read() else []
    hs = conftest._wrap_hostsim(conftest._build_hostsim('libhostsim_devmath.so', fma))
    cfgs = {}
    for seed in sorted(set(c[0] for c in CASES)):
        rs = np.random.RandomState(seed)
        need = {c[1]: c[2] for c in CASES if c[0] == seed}
        for i in range(max(need) + 1):
            cfg = draw_config(rs)
            if i in need:
                assert cfg['tag'].replace(' water', '') == need[i], (seed, i, cfg['tag'])
                cfgs[(seed, i)] = cfg
    fx = dict(ncases=len(CASES))
    for ci, (seed, idx, tag, ref, keep) in enumerate(CASES):
        cfg = cfgs[(seed, idx)]
        iw, ig = REFS[ref]
        H, VP, VS, RHO, nl, per = (cfg[k] for k in ('H', 'VP', 'VS', 'RHO', 'nl', 'per'))
        want, werr, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, cfg['mode'], cfg['fl'], nthreads=8, backend=backend)
        got = np.zeros_like(want)
        for b in range(H.shape[0]):
            n = int(nl[b])
            got[b] = hs.swd(H[b, :n], VP[b, :n], VS[b, :n], RHO[b, :n], per, iw, ig, cfg['mode'], cfg['fl'])[0]
        nz = (want != 0) & (werr == 0)[:, None]
        rel = np.zeros_like(want)
        rel[nz] = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
        sel = np.argsort(rel.max(axis=1))[::-1][:keep]
        p = 'c%d_' % ci
        fx.update({p + 'H': H[sel], p + 'VP': VP[sel], p + 'VS': VS[sel], p + 'RHO': RHO[sel], p + 'nl': nl[sel],
                   p + 'per': per, p + 'iwave': iw, p + 'igr': ig, p + 'mode': cfg['mode'], p + 'fl': cfg['fl'],
                   p + 'campaign': np.array('seed %d, configuration %d: %s, target %s, models %s'
                                            % (seed, idx, tag, ref, sel.tolist())),
                   p + 'device_replay': got[sel]})
        print('%s: worst rel. deviation of the device replay %.3e (model %d)' % (fx[p + 'campaign'], rel.max(), sel[0]))
    np.savez(FIXTURE, **fx)
    variants = ['fma', 'nofma'] + ['noise%d' % s for s in range(NOISE_SEEDS)]
    runs = solve_children(FIXTURE, backend, variants)
    for ci in range(len(CASES)):
        p = 'c%d_' % ci
        fx[p + 'ref_fma'], fx[p + 'ref_nofma'] = runs['fma'][p + 'val'], runs['nofma'][p + 'val']
        fx[p + 'ref_phase'] = runs['fma'][p + 'phase']
        fx[p + 'ref_err'] = runs['fma'][p + 'err']
        fx[p + 'ref_noise'] = np.stack([runs['noise%d' % s][p + 'val'] for s in range(NOISE_SEEDS)])
    fx['backend'] = np.array(backend)
    np.savez(FIXTURE, **fx)
    report(np.load(FIXTURE))


def report(z):
    print('# reference (%s) on the worst models of the campaign: glibc FMA libm | non-FMA libm | %d runs under '
          'sin/cos/exp moved by <= 1 ulp' % (z['backend'], z['c0_ref_noise'].shape[0]))
    for i in range(int(z['ncases'])):
        p = 'c%d_' % i
        ref, dev, noise, c = z[p + 'ref_fma'], z[p + 'device_replay'], z[p + 'ref_noise'], z[p + 'ref_phase']
        nz = ref != 0
        rel = np.zeros_like(ref)
        rel[nz] = np.abs(dev[nz] - ref[nz]) / np.abs(ref[nz])
        b, k = np.unravel_index(np.argmax(rel), rel.shape)
        spread = (noise[:, b, k].max() - noise[:, b, k].min()) / abs(ref[b, k])
        member = dev[b, k] in set(noise[:, b, k].tolist())
        print('%s\n   worst value: model %d period %d (T = %.3f s): reference %.9g, device replay %.9g (rel %.3e); '
              'phase velocity %.6g -> U/c = %.1f;\n   reference under <= 1 ulp noise: %.9g .. %.9g (spread %.3e, %d distinct '
              'values), device value among them: %s; non-FMA libm: %.9g'
              % (z[p + 'campaign'], b, k, z[p + 'per'][k], ref[b, k], dev[b, k], rel[b, k], c[b, k], ref[b, k] / c[b, k],
                 noise[:, b, k].min(), noise[:, b, k].max(), spread, len(set(noise[:, b, k].tolist())), member,
                 z[p + 'ref_nofma'][b, k]))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2], sys.argv[3], sys.argv[4])
    elif len(sys.argv) > 1 and sys.argv[1] == '--report':
        report(np.load(FIXTURE))
    else:
        main()
