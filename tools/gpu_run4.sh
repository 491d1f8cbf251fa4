#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mapper_gpu.py tests/test_loops_gpu.py tests/test_lifelong_gpu.py -m gpu -q -rs -s > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids" $out/pytest.log | tail -40
