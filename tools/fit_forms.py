"""Fits the launcher's table (capi.hip, bh_swd_batch: latency and saturation rate per kernel form and
depth regime) to a tools/team_sweep.py log.

    python tools/fit_forms.py gpurun_out/team_sweep.txt

lat = time of the smallest batch (chip mostly idle); thr = searches per ms, the median of B / t over
the batches that take more than 1.5 lat (every SIMD busy).  Prints the C initialisers.
"""
import re
import sys

import numpy as np

REGIME = {'3': 0, '5': 1, '10': 2, '15': 3, 'ragged': 4}
FORMS = [('lane', 0), ('team8', 8), ('team16', 16), ('team32', 32), ('team', 64), ('team128', 128),
         ('team256', 256), ('team512', 512)]
data = {}
for line in open(sys.argv[1]):
    m = re.match(r'L=\s*(\S+) targets=1 P=(\d+) B=\s*(\d+)\s+(.*?)\s+-> ', line)
    if not m:
        continue
    L, _, B, rest = m.groups()
    for name, t in re.findall(r'(\w+)\s+([\d.]+|inf) ms', rest):
        if t != 'inf':
            data.setdefault((name, REGIME[L]), []).append((int(B), float(t)))
for name, width in FORMS:
    lat, thr = [], []
    for r in range(5):
        pts = sorted(data[(name, r)])
        l0 = pts[0][1]
        sat = [b / t for b, t in pts if t > 1.5 * l0]
        lat.append(l0)
        thr.append(float(np.median(sat)) if sat else pts[-1][0] / pts[-1][1])
    print('        {%d, {%s}, {%s}},' % (width, ', '.join('%.3g' % v for v in lat), ', '.join('%.0f' % v for v in thr)))
