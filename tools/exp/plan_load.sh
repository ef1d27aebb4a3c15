#!/bin/bash
# (record of a round-4 experiment: the switch BH_PLAN_LOAD it sets lived in capi.hip for that measurement only and is not in the tree)
cd "$(dirname "$0")/../.."
for v in 2 1; do
  export BH_PLAN_LOAD=$v
  echo "== load of two groups priced as $v"
  timeout -k 10 200 python tools/tutorial_inversion.py 1024 6000 3000 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('inv %5d chains %6.2f s  %8d it/s  calls %6d' % (d['nchains'], d['seconds'], d['chain_iterations_per_s'], d['device_calls']))"
  timeout -k 10 200 python tools/chain_bench.py 512 1024 2048 4096 8192 16384 32768 65536 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('pool %6d  %8d it/s' % (d['nchains'], d['chain_iterations_per_s']))"
done
