#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py::test_config2_loop_batch_256_pairs -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids" $out/pytest.log | tail -5
cd /tmp && export TMPDIR=/tmp
for tb in 2048; do
KH_TILE_BLOCKS=$tb timeout 300 rocprofv3 --kernel-trace --stats -d $out/loop_trace_$tb -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py loop > $out/loop_$tb.json 2> $out/loop_$tb.err
echo "tile blocks $tb: $(cut -c1-90 $out/loop_$tb.json)"
grep -E "k_raster_tile|k_raster_bin|k_raster_fill" $out/loop_trace_$tb/t_kernel_stats.csv | cut -c1-110
done
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
