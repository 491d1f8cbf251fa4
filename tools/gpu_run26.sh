#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_mapper_gpu.py tests/test_dropin_mapper_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -6
timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-100
timeout 300 python tools/prof_legs.py loop 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print({k:b[k] for k in ('loop_batch_ms','loop_batch_ms_host_scans','loop_gpu_ms')})"
timeout 300 python tools/replay.py --scans 3000 2>/dev/null | cut -c1-300
