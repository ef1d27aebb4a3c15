#!/bin/bash
# GPU-box helper: rocprofv3 kernel trace + PMC passes of the default bench (no CPU baseline).
# usage: tools/profile.sh <tag> [bench.py arguments, e.g. --workload cfg4]   -> gpurun_out/prof_<tag>/...
#        PROFILE_CMD="python3 $GRAFT_REPO_ROOT/tools/chain_bench.py 4096" tools/profile.sh <tag>   (any other program)
# then on the build host: python tools/summarize_profile.py <tag>   -> profiles/<tag>_{kernel_stats.csv,traffic.json,rocprofv3_summary.md}
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --serial: rf_kernel runs behind swd_kernel on one stream, so rocprof's per-kernel durations are kernel
# times (with the product's side stream rf_kernel overlaps the tail of swd_kernel: 19-67 ms per dispatch)
if [ -n "$PROFILE_CMD" ]; then
  BENCH="$PROFILE_CMD"
  echo "${PROFILE_CMD//$R\//}" > $OUT/config.txt
else
  BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-chain-pool --no-configs --serial ${@:2}"
  echo "bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-chain-pool --no-configs --serial ${@:2}" > $OUT/config.txt
fi
(cd $R && python3 -c "from bayhunter_amd import _lib; print(_lib.loaded_hash())") > $OUT/src_hash.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/pmc_lds.log 2>&1 || true
find $OUT -name "*.csv" | head -40
