#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2x; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spa_gpu.py tests/test_baseline_shapes_gpu.py::test_config3_spa_10k_nodes_30k_edges -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -6
KH_SPA_TIMING=1 timeout 300 python tools/quick_spa.py 2> $out/timing.err | tail -2 | cut -c1-420
grep "k_factor" $out/timing.err | tail -14 > $out/last_factor.txt
cat $out/last_factor.txt | cut -c1-400
timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python tools/quick_spa.py > /dev/null 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/level_times.py $f
