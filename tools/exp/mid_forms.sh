#!/bin/bash
# GPU box: the 1 024-chain tutorial inversion (9 000 iterations) with the kernel form pinned and two look-aheads
cd "$(dirname "$0")/../.."
for k in auto team team32 team128 team16; do
  for la in 4 5; do
  echo -n "$k lookahead $la  "
  if [ $k = auto ]; then unset BH_SWD_KERNEL; else export BH_SWD_KERNEL=$k; fi
  timeout -k 10 200 python tools/tutorial_inversion.py 1024 6000 3000 $la | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%6.2f s  %7d it/s  calls %6d  evaluated %8d  wait %.2f s  sha %s' % (d['seconds'], d['chain_iterations_per_s'], d['device_calls'], d['models_evaluated'], d['host_seconds']['wait'], d['chains_sha256']))"
  done
done
