#!/bin/bash
# one GPU call: solver parity tests, timings of the numeric kernels, per-launch profile of the last LM iteration
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spa_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/spa_tests.log
cat gpurun_out/spa_tests.log
rm -f gpurun_out/spa_timing.log
for mode in ${MODES:-3}; do
  echo "== KH_SPA_FACTOR=$mode" >> gpurun_out/spa_timing.log
  KH_SPA_FACTOR=$mode timeout 300 python tools/quick_spa.py 10000 30000 $([ $mode = 3 ] && echo --check) >> gpurun_out/spa_timing.log 2>&1
done
grep -o "compute [0-9.]* ms\|factor_gpu_ms': [0-9.]*\|backward_gpu_ms': [0-9.]*\|symbolic_ms': [0-9.]*\|max diff.*\|== .*" gpurun_out/spa_timing.log | paste -sd' ' | sed 's/== /\n== /g'
KH_SPA_TIMING=1 timeout 300 python tools/quick_spa.py 10000 30000 2>&1 | grep k_potrf | tail -14 > gpurun_out/spa_potrf_stamps.txt
cat gpurun_out/spa_potrf_stamps.txt
bash tools/gpu_spa_prof.sh > /dev/null 2>&1
grep -A40 "^sums" gpurun_out/spa_levels.txt
