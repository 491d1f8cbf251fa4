#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z4; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spa_gpu.py tests/test_baseline_shapes_gpu.py::test_config3_spa_10k_nodes_30k_edges -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -5
timeout 300 python tools/quick_spa.py 2>&1 | tail -1 | cut -c1-130
KH_SPA_SPLIT=0 timeout 300 python tools/quick_spa.py 2>&1 | tail -1 | cut -c1-130
KH_SPA_TIMING=1 timeout 300 python tools/quick_spa.py 2> $out/timing.err | tail -1 | cut -c1-100
grep "k_factor" $out/timing.err | tail -14 | sed -n 4,8p | cut -c1-300
KH_SPA_BACKWARD=1 timeout 300 python tools/quick_spa.py 2>&1 | tail -1 | cut -c1-130
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python tools/quick_spa.py > /dev/null 2> $out/trace.err
cut -c1-150 $(find $out/trace -name "*kernel_stats.csv" | head -1) | head -7
