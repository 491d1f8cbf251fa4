#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2loop; mkdir -p $out; rm -rf $out/trace
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python tools/prof_legs.py loop > $out/loop.json 2> $out/loop.err
cut -c1-120 $(find $out/trace -name "*kernel_stats.csv" | head -1) | head -22
