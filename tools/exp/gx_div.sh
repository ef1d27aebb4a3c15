#!/bin/bash
# (record of a round-4 experiment: the switch BH_NARROW_GX_DIV it sets lived in kernels.hip for that measurement only and is not in the tree)
cd "$(dirname "$0")/../.."
for v in 1 2 3 4; do
  export BH_NARROW_GX_DIV=$v
  echo "== persistent waves of a narrow-team launch / $v"
  python tools/exp/order_effect.py 2-31 21 8192 rdispph team16 team32 team8 2>&1 | grep "^L="
  python tools/exp/order_effect.py 2-31 21 4096 rdispph team16 team32 2>&1 | grep "^L="
  python tools/exp/order_effect.py 10 21 8192 rdispph team16 team8 2>&1 | grep "^L="
  python tools/exp/order_effect.py 10 40 8192 rdispph,rdispgr,ldispph,ldispgr team8 team16 2>&1 | grep "^L="
done
