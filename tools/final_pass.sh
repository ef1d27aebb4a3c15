#!/bin/bash
# GPU-box helper: the end-of-round pass -- GPU tests, rocprof + PMC passes, phase profile, chain pool, random
# campaigns, the bench line and the two-rank rehearsal.  Outputs under gpurun_out/ (copy the summaries to profiles/).
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1
python tools/team_phase_profile.py > gpurun_out/r03_team_phase_profile.txt 2>&1
python tools/chain_bench.py 64 256 1024 4096 16384 65536 > gpurun_out/r03_chain_pool.jsonl 2>gpurun_out/chain_bench.err
python tests/scenarios/rf_fuzz.py 60 79 > gpurun_out/r03_rf_fuzz_79.txt 2>&1
python tests/scenarios/rf_fuzz.py 60 86 > gpurun_out/r03_rf_fuzz_86.txt 2>&1
python tests/scenarios/kernel_fuzz.py 150 97 > gpurun_out/r03_kernel_fuzz_97.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2>gpurun_out/r03_bench.err
BH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03_bench_n2_rehearsal.json 2>gpurun_out/r03_bench_n2.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
