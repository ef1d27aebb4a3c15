#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of a chain-pool run (which kernels an inversion spends its
# device time in).  usage: tools/profile_chains.sh [nchains]  -> gpurun_out/prof_chains/
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_chains
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHAIN_BENCH_ITERS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/chain_bench.py ${1:-16384} > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
