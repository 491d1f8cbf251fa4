#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout 600 python -m pytest tests/test_spa_gpu.py tests/test_posegraph_gpu.py tests/test_baseline_shapes_gpu.py::test_config3_spa_10k_nodes_30k_edges tests/test_dropin_mapper_gpu.py::test_reference_mapper_runs_on_the_gpu_solver_plugin -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -3
timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-1100
KH_SPA_EXTEND_ADD=0 timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-120
KH_SPA_EXTEND_ADD=1000 timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-120
