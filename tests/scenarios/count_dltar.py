"""How bench.py's N_DLTAR table was made: period-equation evaluations per model of the reference
path (the oracle's counter) and mean layer count on 128 benchmark-seed models of every workload.
    python tests/scenarios/count_dltar.py
tests/test_capi_host.py checks the committed table against this count for the default workload."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def count(wl):
    import bench
    from oracle import pyoracle as po
    from bayhunter_amd.synthetic import draw_models
    H, VP, VS, RHO, nl = draw_models(128, wl['L'], seed=1000 * wl['cfg'], sorted_vs=True)
    per = np.linspace(1, 41, wl['P'])
    counts = {}
    for ref in wl['refs']:
        iw, ig = bench.REF_TAGS[ref]
        _, _, nc = po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, nthreads=4)
        counts[ref] = nc / 128.0
    return counts, float(np.mean(nl))


if __name__ == '__main__':
    import bench
    for name, wl in bench.WORKLOADS.items():
        print(repr(name) + ':', count(wl))
