#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mapper_gpu.py tests/test_lifelong_gpu.py -m gpu -q -s > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids" $out/pytest.log | tail -12
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; cut -c1-300 $out/bench.json
bash tools/prof_bench.sh r2g_prof > /dev/null 2>&1
ls gpurun_out/r2g_prof
