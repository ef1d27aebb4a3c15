#!/bin/bash
# GPU-box helper: batch-size sweep of bench.py (no CPU baseline), one JSON line per batch size.
set -e
mkdir -p gpurun_out
: > gpurun_out/sweep.jsonl
for B in "$@"; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $B >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/sweep.jsonl'):
    r=json.loads(l)
    print(r['config']['models_per_gpu'], '%.3e evals/s'%r['value'], 'step %.2f ms'%r['ms_per_step'], 'swd %.2f ms rf %.2f ms'%(r['kernels_ms']['swd_kernel'], r['kernels_ms']['rf_kernel']), 'fp64 frac %.3f'%r['roofline']['frac'])
PY
