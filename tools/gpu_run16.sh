#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2w; mkdir -p $out
export TMPDIR=/tmp
KH_SPA_TIMING=1 timeout 300 python tools/quick_spa.py 2> $out/timing.err | tail -3
grep "k_factor\]" $out/timing.err | tail -14 > $out/last_factor.txt
cat $out/last_factor.txt | cut -c1-400
timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python tools/quick_spa.py > /dev/null 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/level_times.py $f
timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-330
