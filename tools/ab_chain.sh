#!/bin/bash
# Same-box A/B of library builds on the chain pool (tools/chain_bench.py) -- companion of tools/ab_build.sh.
#   tools/ab_chain.sh <out-file> "<name>|<extra hipcc flags>|<pool sizes>" ...
set -e
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for v in "$@"; do
    IFS='|' read -r name flags sizes <<< "$v"
    echo "== $name: flags [$flags]" | tee -a "$out"
    BH_EXTRA_HIPCC_FLAGS="$flags" python -c "from bayhunter_amd import _lib; _lib.build(force=True)"
    BH_EXTRA_HIPCC_FLAGS="$flags" python tools/chain_bench.py ${sizes:-4096} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$name', d['nchains'], 'chains:', d['chain_iterations_per_s'], 'iterations/s', d['seconds_in'])
" | tee -a "$out"
done
python -c "from bayhunter_amd import _lib; _lib.build(force=True)"
