#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_full_tests.sh r2fin
bash tools/gpu_final.sh r2fin 2>&1 | tail -12
python tools/quick_latency.py 2>/dev/null | tail -8
timeout 600 python tools/replay.py --scans 50000 > gpurun_out/r2fin/replay_50k.json 2> gpurun_out/r2fin/replay_50k.err
tail -c 1500 gpurun_out/r2fin/replay_50k.json | cut -c1-1500
