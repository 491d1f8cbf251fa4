#!/bin/bash
# rocprofv3 kernel trace of the solver on the 10k/30k graph (level pipeline), reduced to per-launch durations of the last LM iteration
mkdir -p gpurun_out/prof_spa
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_spa -o spa -- python tools/quick_spa.py 10000 30000 > gpurun_out/prof_spa/run.log 2>&1
f=$(find gpurun_out/prof_spa -name "*kernel_trace.csv" | head -1)
python tools/level_times3.py $f > gpurun_out/spa_levels.txt 2>&1
s=$(find gpurun_out/prof_spa -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/spa_kernel_stats.csv
rm -rf gpurun_out/prof_spa
tail -80 gpurun_out/spa_levels.txt
