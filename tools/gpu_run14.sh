#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r2s; mkdir -p $out
for ea in 128; do
KH_SPA_EXTEND_ADD=$ea timeout 300 rocprofv3 --kernel-trace -d $out/t$ea -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py solver > /dev/null 2> $out/err$ea
echo "extend-add limit $ea"; python $GRAFT_REPO_ROOT/tools/level_times.py $out/t$ea/t_kernel_trace.csv
done
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_spa_gpu.py tests/test_posegraph_gpu.py tests/test_baseline_shapes_gpu.py::test_config3_spa_10k_nodes_30k_edges tests/test_spa_sharded_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu\|^\[W9\|Gloo" | tail -3
timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-330
