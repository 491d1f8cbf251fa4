#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 32 64 128; do
echo "threads $t"; KH_HOST_THREADS=$t KH_MATCH_TIMING=2 timeout 300 python tools/prof_legs.py loop 2> gpurun_out/loop_timing.err | cut -c1-80
grep "kh raster\|kh match" gpurun_out/loop_timing.err | tail -24 | head -7 | cut -c1-150
done
