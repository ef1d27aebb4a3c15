#!/bin/bash
# GPU box: chain iterations/s of small pools against the look-ahead (proposals per chain and call)
# usage: tools/exp/lookahead_sweep.sh > gpurun_out/lookahead_sweep.txt
set -e
cd "$(dirname "$0")/../.."
for n in 5 16 64 256 1024 4096; do
  for la in 1 2 4 8 16 32 64; do
    if [ $((n * la)) -gt 16384 ]; then continue; fi
    it=600; [ $n -ge 1024 ] && it=240
    echo -n "chains $n lookahead $la  "
    CHAIN_BENCH_ITERS=$it CHAIN_BENCH_LOOKAHEAD=$la timeout -k 10 300 python tools/chain_bench.py $n | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%9d it/s  %6.2f s  groups %d  calls %5d  it/call %5.2f  models/call %7.1f  evaluated %8d  host %s' % (d['chain_iterations_per_s'], d['seconds'], d['groups'], d['calls'], d['iterations_per_call'], d['models_per_call'], d['models_evaluated'], d['seconds_in']))"
  done
done
