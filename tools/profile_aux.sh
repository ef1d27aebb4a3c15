#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the auxiliary kernels (voronoi, likelihood, MFMA
# Gaussian product, team kernel) on the full device pipeline and a 1024-model team run.
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_aux; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipeline -- python3 $R/tools/pipeline_bench.py 131072 > $OUT/pipeline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/team -- python3 /tmp/team_run.py > $OUT/team.log 2>&1 || true
