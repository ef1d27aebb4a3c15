#!/bin/bash
# GPU box: one against two chain groups for small pools with the default look-ahead
cd "$(dirname "$0")/../.."
export CHAIN_BENCH_ITERS=1200
for n in ${POOLS:-64 128 256 384 512 1024}; do
  for g in ${GROUPS_LIST:-1 2}; do
    echo -n "chains $n groups $g  "
    CHAIN_BENCH_GROUPS=$g timeout -k 10 200 python tools/chain_bench.py $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%8d it/s  lookahead %2d  calls %5d  it/call %.2f  host %s' % (d['chain_iterations_per_s'], d['lookahead'], d['calls'], d['iterations_per_call'], d['seconds_in']))"
  done
done
