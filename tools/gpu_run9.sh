#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2j
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_mapper_gpu.py -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids" $out/pytest.log | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/loop_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py loop > $out/loop.json 2> $out/loop.err
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
cut -c1-400 $out/loop.json
