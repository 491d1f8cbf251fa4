#!/bin/bash
# gpurun helper: a subset of the GPU tests plus the quick timings of every leg.  Usage (from the repo root, on the dev box):
#   gpurun --timeout 1500 -- 'bash tools/gpu_check.sh [pytest files...]'
cd $GRAFT_REPO_ROOT
files=${@:-tests/test_matcher_gpu.py tests/test_spa_gpu.py tests/test_mapper_gpu.py}
timeout 1200 python -m pytest $files -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -4
python tools/seq_latency.py 20 resident
timeout 300 python tools/quick_spa.py 2>&1 | tail -1 | cut -c1-60
timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-100
timeout 300 python tools/replay.py --scans 3000 2>/dev/null | cut -c1-200
