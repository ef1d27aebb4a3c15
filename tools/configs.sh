#!/bin/bash
# GPU-box helper: one bench line per BASELINE.json config (parity-test cases, recorded for DESIGN.md)
mkdir -p gpurun_out; : > gpurun_out/configs.jsonl
for W in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload $W "$@" >> gpurun_out/configs.jsonl 2>> gpurun_out/configs.err
done
python - <<'PY'
import json
for l in open('gpurun_out/configs.jsonl'):
    r=json.loads(l)
    c=r.get('cpu_baseline') or {}
    print(r['config']['workload'][:70], '| %.3e evals/s'%r['value'], '| step %.2f ms'%r['ms_per_step'], '| swd %.2f rf %.2f'%(r['kernels_ms']['swd_kernel'], r['kernels_ms']['rf_kernel']), '| cpu %.0f/s on %s cores'%(c.get('value',0), c.get('cores')))
PY
