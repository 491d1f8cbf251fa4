#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_mapper_gpu.py tests/test_loops_gpu.py tests/test_dropin_mapper_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -4
timeout 600 python tools/replay.py --scans 50000 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','scans_per_s','accepted','alive')}); print({k:round(v) for k,v in d['stats'].items()})"
