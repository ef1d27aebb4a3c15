#!/bin/bash
# GPU box: look-ahead sweep with the default number of chain groups (two from 32 chains on)
cd "$(dirname "$0")/../.."
export CHAIN_BENCH_ITERS=1200
run() { n=$1; shift; for la in "$@"; do echo -n "chains $n lookahead $la  "; CHAIN_BENCH_LOOKAHEAD=$la timeout -k 10 200 python tools/chain_bench.py $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%8d it/s  groups %d  calls %5d  it/call %.2f  models/call %.0f' % (d['chain_iterations_per_s'], d['groups'], d['calls'], d['iterations_per_call'], d['models_per_call']))"; done; }
run 64 11 16 22 32 45
run 256 5 8 11 16
run 1024 3 4 5 8
run 4096 1 2 3 4
