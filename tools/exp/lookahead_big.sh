#!/bin/bash
cd "$(dirname "$0")/../.."
run() { n=$1; it=$2; shift; shift; for la in "$@"; do echo -n "chains $n lookahead $la  "; CHAIN_BENCH_ITERS=$it CHAIN_BENCH_LOOKAHEAD=$la timeout -k 10 200 python tools/chain_bench.py $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%8d it/s  groups %d  calls %5d  it/call %.2f  models/call %.0f  host %s' % (d['chain_iterations_per_s'], d['groups'], d['calls'], d['iterations_per_call'], d['models_per_call'], d['seconds_in']))"; done; }
run 4096 600 4 6 8
run 16384 200 1 2 3 4
run 65536 100 1 2
run 2048 800 3 4 6 8
