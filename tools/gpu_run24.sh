#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_comm_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -30 | cut -c1-250
done
