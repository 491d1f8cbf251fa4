#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_mapper_gpu.py tests/test_dropin_mapper_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -4
timeout 300 python tools/replay.py --scans 3000 2>/dev/null | cut -c1-900
timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-330
