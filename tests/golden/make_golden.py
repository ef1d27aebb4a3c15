"""Generate the golden vectors under tests/golden/ from the REFERENCE's own native code.

Run in the development container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

The reference sources are compiled where they lie (oracle/Makefile) and driven through ctypes
(oracle/pyoracle.py, backend="ref"); only inputs and full-precision outputs are stored.
tutorial_observed/*.dat are the data files shipped with the reference's tutorial
(tutorial/observed/, written by tutorial/create_testdata.py with 4 decimals).

Files
  swd_rf_random.npz   seeded random model sets (sorted-Vs and low-velocity-zone variants,
                      L in {2,5,10,15,31}), 21 periods linspace(1,41,21): the four dispersion
                      targets (+ err flags) and the P receiver function (201 samples, 5 Hz)
  swd_variants.npz    modes 1..3, flsph 0/1, period counts 20/40/60 on a 10-layer set
  rf_variants.npz     P/SV x gauss x slowness x nsamp x nsv on a 10-layer set
  tutorial_full.npz   the tutorial model (st3) at full precision: 4 SWD targets, prf, srf
  swd_water.npz       models under a water layer (vs[0] = 0 -> llw = 2, surfdisp96.f:134-135,
                      :850-867), 5 and 10 layers, sorted and low-velocity-zone variants, the four
                      dispersion targets (`python make_golden.py water` writes only this file)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayhunter_amd.synthetic import draw_models, tutorial_model  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REFS = [('rdispph', 2, 0), ('rdispgr', 2, 1), ('ldispph', 1, 0), ('ldispgr', 1, 1)]


def water_models(B, L, seed, sorted_vs):
    """draw_models with the top layer turned into 0.3-4 km of water (vp 1.5, vs 0, rho 1.03)."""
    H, VP, VS, RHO, nl = draw_models(B, L, seed=seed, sorted_vs=sorted_vs)
    rs = np.random.RandomState(seed + 1)
    H[:, 0] = rs.uniform(0.3, 4.0, size=B)
    VP[:, 0], VS[:, 0], RHO[:, 0] = 1.5, 0.0, 1.03
    return H, VP, VS, RHO, nl


def make_water():
    per = np.linspace(1, 41, 21)
    d = {'periods': per}
    k = 0
    for L in (5, 10):
        for srt in (True, False):
            H, VP, VS, RHO, nl = water_models(24, L, 9400 + k, srt)
            tag = 'L%d_%s' % (L, 'sorted' if srt else 'lvz')
            d[tag + '_model'] = np.stack([H, VP, VS, RHO])
            for name, iw, ig in REFS:
                out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, backend='ref')
                d[tag + '_' + name] = out
                d[tag + '_' + name + '_err'] = err
            k += 1
    np.savez_compressed(os.path.join(OUT, 'swd_water.npz'), **d)


def main():
    assert po.have_ref(), "oracle/_ref missing: run `make -C oracle ref` where /root/reference exists"
    make_water()
    if sys.argv[1:] == ['water']:
        return
    per = np.linspace(1, 41, 21)
    d = {}
    k = 0
    for L in (2, 5, 10, 15, 31):
        for srt in (True, False):
            H, VP, VS, RHO, nl = draw_models(24, L, seed=9000 + k, sorted_vs=srt)
            tag = 'L%d_%s' % (L, 'sorted' if srt else 'lvz')
            d[tag + '_model'] = np.stack([H, VP, VS, RHO])
            for name, iw, ig in REFS:
                out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, backend='ref')
                d[tag + '_' + name] = out
                d[tag + '_' + name + '_err'] = err
            d[tag + '_prf'] = po.rf_batch(H, VP, VS, RHO, nl, backend='ref')
            k += 1
    # ragged set
    H, VP, VS, RHO, nl = draw_models(48, (2, 31), seed=9100, sorted_vs=True)
    d['ragged_model'] = np.stack([H, VP, VS, RHO])
    d['ragged_nlay'] = nl
    for name, iw, ig in REFS:
        out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, backend='ref')
        d['ragged_' + name] = out
        d['ragged_' + name + '_err'] = err
    d['ragged_prf'] = po.rf_batch(H, VP, VS, RHO, nl, backend='ref')
    d['periods'] = per
    np.savez_compressed(os.path.join(OUT, 'swd_rf_random.npz'), **d)

    d = {}
    H, VP, VS, RHO, nl = draw_models(16, 10, seed=9200, sorted_vs=True)
    d['model'] = np.stack([H, VP, VS, RHO])
    for name, iw, ig in REFS:
        for mode in (1, 2, 3):
            for fl in (0, 1):
                out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, mode, fl, backend='ref')
                d['%s_m%d_f%d' % (name, mode, fl)] = out
                d['%s_m%d_f%d_err' % (name, mode, fl)] = err
        for P in (20, 40, 60):
            out, err, _ = po.swd_batch(H, VP, VS, RHO, nl, np.linspace(1, 41, P), iw, ig, backend='ref')
            d['%s_P%d' % (name, P)] = out
            d['%s_P%d_err' % (name, P)] = err
    np.savez_compressed(os.path.join(OUT, 'swd_variants.npz'), **d)

    d = {}
    H, VP, VS, RHO, nl = draw_models(8, 10, seed=9300, sorted_vs=True)
    d['model'] = np.stack([H, VP, VS, RHO])
    for wn in (0, 1):
        for gauss in (1.0, 2.5):
            for p in (4.0, 6.4, 8.0):
                for nsamp in (256, 512, 1024):
                    for nsv in (None, 3.0):
                        key = 'w%d_g%.1f_p%.1f_n%d_%s' % (wn, gauss, p, nsamp, 'nsv' if nsv else 'top')
                        d[key] = po.rf_batch(H, VP, VS, RHO, nl, p, gauss, nsamp, 5.0, 5.0, nsv, wn,
                                             nout=nsamp // 2, backend='ref')
    np.savez_compressed(os.path.join(OUT, 'rf_variants.npz'), **d)

    d = {}
    h, vp, vs, rho = tutorial_model()
    for name, iw, ig in REFS:
        d[name], err = po.swd(h, vp, vs, rho, per, iw, ig, backend='ref')
        assert err == 0
    d['prf'] = po.rf_model(h, vp, vs, rho, waveno=0, nout=201, backend='ref')
    d['srf'] = po.rf_model(h, vp, vs, rho, waveno=1, nout=201, backend='ref')
    np.savez_compressed(os.path.join(OUT, 'tutorial_full.npz'), **d)
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
