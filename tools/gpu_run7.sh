#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_shard_gpu.py -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids" $out/pytest.log | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-solver --no-loop > $out/bench_dual.json 2> $out/bench.err
cut -c1-1200 $out/bench_dual.json
