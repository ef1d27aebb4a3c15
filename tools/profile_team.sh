#!/bin/bash
# GPU-box helper: PMC of the team kernel on a 1024-model, 10-layer, 21-period run
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_team; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/team_run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from bayhunter_amd.engine import ForwardEngine, SwdSpec
from bayhunter_amd.synthetic import draw_models
H,VP,VS,RHO,nl = draw_models(1024, 10, seed=1)
eng = ForwardEngine(swd=[SwdSpec('rdispph', np.linspace(1,41,21))])
d = eng.upload(H,VP,VS,RHO,nl)
for _ in range(4): eng.run(d)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/team_run.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- python3 /tmp/team_run.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- python3 /tmp/team_run.py > $OUT/pmc_lds.log 2>&1 || true
