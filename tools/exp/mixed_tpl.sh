#!/bin/bash
# GPU box: cfg3-like calls (four dispersion targets) with per-target kernel forms forced
set -e
cd "$(dirname "$0")/../.."
R=rdispph,rdispgr,ldispph,ldispgr
for B in 8192 4096 2048 16384; do
python tools/forced_forms.py 10 40 $B $R auto team8,team8,team8,team8 team16,team16,team16,team16 team8,team8,lane,team8 team8,team8,lane,lane \
   team16,team8,lane,team8 team16,team16,lane,lane team8,team8,lane,team16 lane,team8,lane,lane team16,team16,lane,team16 team16,team8,lane,lane lane,team128,lane,lane 2>&1 | grep -v amdgpu.ids
done
python tools/forced_forms.py 5 40 8192 $R auto team8,team8,team8,team8 team8,team8,lane,team8 team8,team8,lane,lane lane,team8,lane,lane 2>&1 | grep -v amdgpu.ids
python tools/forced_forms.py 15 21 8192 rdispph,ldispph auto team16,team16 team16,lane team8,lane team8,team8 2>&1 | grep -v amdgpu.ids
