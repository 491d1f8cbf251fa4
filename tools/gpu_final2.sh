#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2fin3; mkdir -p $out
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -3
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-160 $out/bench.json
bash tools/prof_bench.sh r2fin3_prof > /dev/null 2>&1
find gpurun_out/r2fin3_prof -name "*.db" -delete
ls gpurun_out/r2fin3_prof | head
