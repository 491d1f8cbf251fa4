#!/bin/bash
# a mapper-sized solve (2000 nodes, 2250 edges) under a kernel trace: are its launches back to back, or is the host's enqueueing what it takes?
out=$GRAFT_REPO_ROOT/gpurun_out/small_spa
mkdir -p $out; cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/quick_spa.py 2000 2250 2>&1 | tail -3 | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_spa.py 2000 2250 > $out/trace.out 2> $out/trace.err
python $GRAFT_REPO_ROOT/tools/level_times3.py $out/trace/t_kernel_trace.csv > $out/levels_timeline.txt 2>&1
tail -30 $out/levels_timeline.txt
find $out -name "*kernel_trace.csv" -size +1M -delete
