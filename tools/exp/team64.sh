#!/bin/bash
# GPU box: the one-wave team on a few shapes (ms per call)
cd "$(dirname "$0")/../.."
for s in "3 21 1024" "5 21 256" "5 21 1024" "5 21 2048" "5 21 4096" "10 21 256" "10 21 1024" "10 21 2048" "15 21 1024" "5 20 1024"; do
  python tools/forced_forms.py $s rdispph team 2>&1 | grep -v amdgpu.ids
done
