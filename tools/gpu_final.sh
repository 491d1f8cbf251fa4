#!/bin/bash
# final measurements of a round: bench line, rocprofv3 kernel stats + PMC passes of the matcher leg, per-leg kernel stats
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; cut -c1-200 $out/bench.json
bash tools/prof_bench.sh ${tag}_prof > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/solver_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py solver > $out/solver.json 2> $out/solver.err
timeout 300 rocprofv3 --kernel-trace --stats -d $out/loop_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py loop > $out/loop.json 2> $out/loop.err
python $GRAFT_REPO_ROOT/tools/level_times.py $out/solver_trace/t_kernel_trace.csv
find $out $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
ls $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof | head -20
