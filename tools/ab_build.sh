#!/bin/bash
# Same-box A/B of library builds (box-to-box variance is ~5 %, so variants must share a box).
#   tools/ab_build.sh <out-file> <variant> [<variant> ...]
# A variant is "name|extra hipcc flags|ENV=val ENV=val|workload workload ...".  Each variant is built in
# place (BH_EXTRA_HIPCC_FLAGS), bench.py runs once per workload, and one line per run goes to <out-file>.
# The last step rebuilds the default library.
set -e
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for v in "$@"; do
    IFS='|' read -r name flags envs loads <<< "$v"
    echo "== $name: flags [$flags] env [$envs]" | tee -a "$out"
    BH_EXTRA_HIPCC_FLAGS="$flags" python -c "from bayhunter_amd import _lib; _lib.build(force=True)"
    for w in ${loads:-joint10}; do
        env BH_EXTRA_HIPCC_FLAGS="$flags" $envs python bench.py --workload "$w" --steps 5 --warmup 2 --no-cpu-baseline --no-chain-pool --no-configs \
            | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('$name', '$w', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], json.dumps(d.get('kernels_ms')))
" | tee -a "$out"
    done
done
python -c "from bayhunter_amd import _lib; _lib.build(force=True)"
