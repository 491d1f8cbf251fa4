#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2fin2; mkdir -p $out
bash tools/gpu_full_tests.sh r2fin2
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout 600 python tools/replay.py --scans 50000 > $out/replay_50k.json 2> $out/replay_50k.err
tail -n 1 $out/replay_50k.json | cut -c1-200
