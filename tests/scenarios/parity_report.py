"""GPU-box diagnostic: distribution of |HIP - oracle| per target over many seeded models.
Prints one line per (set, target): fraction bit-identical, max / p99.9 abs difference, err-flag
agreement.  Used to state the parity tolerances of tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec  # noqa: E402
from bayhunter_amd.synthetic import draw_models  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REFS = [('rdispph', 2, 0), ('rdispgr', 2, 1), ('ldispph', 1, 0), ('ldispgr', 1, 1)]


def main(B=2048, kernel='auto'):
    from bayhunter_amd import _lib
    _lib.set_swd_kernel(kernel)
    print('# swd kernel mode: %s, %d models per set' % (kernel, B))
    per = np.linspace(1, 41, 21)
    eng = ForwardEngine(swd=[SwdSpec(r[0], per) for r in REFS], rf=[RfSpec('prf', np.linspace(-5, 35, 201))])
    threads = len(os.sched_getaffinity(0))
    for L in (5, 10, 15, (2, 31)):
        for srt in (True, False):
            H, VP, VS, RHO, nl = draw_models(B, L, seed=31337 + (hash(str(L)) % 100) + int(srt), sorted_vs=srt)
            out, err = eng.run(H, VP, VS, RHO, nl)
            out, err = out.cpu().numpy(), err.cpu().numpy()
            tag = 'L=%s %s' % (L, 'sorted' if srt else 'lvz')
            for t, (name, iw, ig) in enumerate(REFS):
                want, werr, _ = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=threads)
                got = out[:, eng.slices[t]]
                ok = (werr == 0) & (err[:, t] == 0)
                d = np.abs(got[ok] - want[ok])
                rel = d / np.maximum(want[ok], 1e-9)
                rms = np.sqrt(np.mean((got[ok] - want[ok]) ** 2, axis=1))
                print('%-18s %-8s err_equal=%s n_err=%4d identical=%.5f max=%.3e maxrel=%.3e p999=%.3e max_rms=%.3e'
                      % (tag, name, np.array_equal(werr, err[:, t]), int(werr.sum()), float((d == 0).mean()),
                         d.max(), rel.max(), np.quantile(d, 0.999), rms.max()))
            wrf = po.rf_batch(H, VP, VS, RHO, nl, nthreads=threads)
            d = np.abs(out[:, eng.slices[4]] - wrf)
            print('%-18s %-8s max=%.3e  scale=%.3e' % (tag, 'prf', np.nanmax(d), np.nanmax(np.abs(wrf))))
            sys.stdout.flush()


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2048, sys.argv[2] if len(sys.argv) > 2 else 'auto')
