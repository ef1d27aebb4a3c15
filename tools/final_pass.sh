#!/bin/bash
# GPU-box helper: the end-of-round pass, in pieces that each fit one gpurun call (<= 20 min):
#   tools/final_pass.sh tests      GPU tests, smoke
#   tools/final_pass.sh profiles   rocprofv3 kernel trace + PMC passes: headline, cfg2, cfg3, cfg4, cfg5, 4096-chain pool
#   tools/final_pass.sh bench      bench line, two-rank rehearsal, one-rank RCCL line, chain pool sizes, phase profile
#   tools/final_pass.sh campaigns  random campaigns (kernel_fuzz, rf_fuzz)
# Outputs under gpurun_out/; on the build host `python tools/summarize_profile.py <tag>` turns each profile into
# profiles/<tag>_{kernel_stats.csv,traffic.json,rocprofv3_summary.md} (stamped with the library's source hash).
set -e
cd $GRAFT_REPO_ROOT
R=${ROUND:-r04}
case "$1" in
tests)
    python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
    python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
    ;;
profiles)
    tools/profile.sh $R > gpurun_out/profile_$R.log 2>&1
    for w in cfg2 cfg3 cfg4 cfg5; do tools/profile.sh ${R}_$w --workload $w > gpurun_out/profile_${R}_$w.log 2>&1; done
    PROFILE_CMD="python3 $GRAFT_REPO_ROOT/tools/chain_bench.py 4096" tools/profile.sh ${R}_pool4096 > gpurun_out/profile_${R}_pool.log 2>&1
    ;;
bench)
    python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2>gpurun_out/${R}_bench.err
    BH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-configs > gpurun_out/${R}_bench_n2_rehearsal.json 2>gpurun_out/${R}_bench_n2.err
    BH_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-configs > gpurun_out/${R}_bench_rccl1.json 2>gpurun_out/${R}_bench_rccl1.err
    python tools/chain_bench.py 64 256 1024 4096 16384 65536 > gpurun_out/${R}_chain_pool.jsonl 2>gpurun_out/chain_bench.err
    python tools/host_phase_probe.py 256 1024 4096 16384 > gpurun_out/${R}_host_phase_probe.jsonl 2>/dev/null
    python tools/team_phase_profile.py > gpurun_out/${R}_team_phase_profile.txt 2>&1
    ( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
    ;;
campaigns)
    python tests/scenarios/kernel_fuzz.py 150 104 > gpurun_out/${R}_kernel_fuzz.txt 2>&1
    python tests/scenarios/rf_fuzz.py 60 91 > gpurun_out/${R}_rf_fuzz.txt 2>&1
    ;;
*)
    echo "usage: tools/final_pass.sh tests|profiles|bench|campaigns"; exit 2 ;;
esac
