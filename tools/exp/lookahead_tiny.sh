#!/bin/bash
# (record of a round-4 experiment: the switch BH_GROUPS it sets lived in tools/tutorial_inversion.py for that measurement only and is not in the tree)
# GPU box: the reference's 5-chain tutorial run against look-ahead and chain groups
set -e
cd "$(dirname "$0")/../.."
for g in 1 2; do
  for la in 32 64 128 256 512; do
    echo -n "groups $g lookahead $la  "
    BH_GROUPS=$g timeout -k 10 200 python tools/tutorial_inversion.py 5 65536 32768 $la | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%6.2f s  %7d it/s  calls %6d  evaluated %8d  host %s  sha %s' % (d['seconds'], d['chain_iterations_per_s'], d['device_calls'], d['models_evaluated'], d['host_seconds'], d['chains_sha256']))"
  done
done
for n in 16 64; do for la in 32 64 128; do
    echo -n "chains $n lookahead $la  "
    timeout -k 10 200 python tools/tutorial_inversion.py $n 20000 10000 $la | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%6.2f s  %7d it/s  calls %6d  evaluated %8d  host %s  sha %s' % (d['seconds'], d['chain_iterations_per_s'], d['device_calls'], d['models_evaluated'], d['host_seconds'], d['chains_sha256']))"
done; done
