#!/bin/bash
# (record of a round-4 experiment: the switch BH_EVAL_LOAD_RF it sets lived in evalplan.hip for that measurement only and is not in the tree)
cd "$(dirname "$0")/../.."
for v in 1 2 3; do
  export BH_EVAL_LOAD_RF=$v
  echo "== load factor $v"
  for a in "5 20000 10000" "16 10000 5000" "64 6000 3000" "256 3000 1500"; do
    timeout -k 10 200 python tools/tutorial_inversion.py $a | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('inv %5d chains %6.2f s  %8d it/s  calls %6d  wait/call %.3f ms' % (d['nchains'], d['seconds'], d['chain_iterations_per_s'], d['device_calls'], 1e3*d['host_seconds']['wait']/d['device_calls']))"
  done
  timeout -k 10 200 python tools/chain_bench.py 1024 4096 16384 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('pool %6d  %8d it/s' % (d['nchains'], d['chain_iterations_per_s']))"
done
