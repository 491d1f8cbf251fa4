#!/bin/bash
# kernel trace of three solves of the 10k / 30k graph on the GPU box + the per-launch timeline of the last LM iteration:
# tools/spa_trace.sh <tag> [ENV=...]   -> gpurun_out/<tag>_levels.txt, gpurun_out/<tag>_kernel_stats.csv
tag=${1:-spa}; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${tag}_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_spa.py 10000 30000 > $out/${tag}_run.txt 2>&1
python $GRAFT_REPO_ROOT/tools/level_times3.py /tmp/${tag}_trace/t_kernel_trace.csv > $out/${tag}_levels.txt
cp /tmp/${tag}_trace/t_kernel_stats.csv $out/${tag}_kernel_stats.csv
grep -o "'solve_ms': [0-9.]*" $out/${tag}_run.txt
