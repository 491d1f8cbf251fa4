#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
KH_SPA_TIMING=1 timeout 300 python tools/prof_legs.py solver > gpurun_out/r2o/g8.json 2> gpurun_out/r2o/g8.err
KH_SPA_GROUP=1 KH_SPA_TIMING=1 timeout 300 python tools/prof_legs.py solver > gpurun_out/r2o/g1.json 2> gpurun_out/r2o/g1.err
grep "k_factor" gpurun_out/r2o/g8.err | sed -n 15,28p | cut -c1-330
echo ---
grep "k_factor" gpurun_out/r2o/g1.err | sed -n 15,28p | cut -c1-330
timeout 300 python tools/prof_legs.py solver 2>/dev/null | cut -c1-420
timeout 300 python -m pytest tests/test_spa_gpu.py -m gpu -x -q 2>&1 | tail -2
