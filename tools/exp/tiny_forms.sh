#!/bin/bash
# GPU box: a 5-chain tutorial inversion (30 000 iterations) with the kernel form pinned
cd "$(dirname "$0")/../.."
for k in auto team512 team256 team128 team team32; do
  for la in 57 32; do
  echo -n "$k lookahead $la  "
  if [ $k = auto ]; then unset BH_SWD_KERNEL; else export BH_SWD_KERNEL=$k; fi
  timeout -k 10 200 python tools/tutorial_inversion.py 5 20000 10000 $la | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%6.2f s  %7d it/s  calls %6d  evaluated %8d  wait/call %.3f ms  sha %s' % (d['seconds'], d['chain_iterations_per_s'], d['device_calls'], d['models_evaluated'], 1e3*d['host_seconds']['wait']/d['device_calls'], d['chains_sha256']))"
  done
done
