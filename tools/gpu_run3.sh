#!/bin/bash
# round-2 GPU session 2: GPU FindValidPoints / active set / tile kernel -- matcher parity + loop leg trace
out=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_spa_sharded_gpu.py -m gpu -x -q -rs > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
KH_MATCH_TIMING=2 timeout 300 rocprofv3 --kernel-trace --stats -d $out/loop_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py loop > $out/loop.json 2> $out/loop.err
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
cat $out/loop.json | cut -c1-600
grep "kh raster\|kh match" $out/loop.err | tail -12
