#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_spa_gpu.py tests/test_posegraph_gpu.py tests/test_baseline_shapes_gpu.py::test_config3_spa_10k_nodes_30k_edges tests/test_comm_gpu.py -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -8
timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-700
KH_SPA_GROUP=1 timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-200
