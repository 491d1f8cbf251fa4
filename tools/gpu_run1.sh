#!/bin/bash
# round-2 GPU session 1: tests at the BASELINE shapes, the bench line, per-leg kernel traces
out=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
KH_SPA_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/solver_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py solver > $out/solver.json 2> $out/solver.err
KH_MATCH_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/loop_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_legs.py loop > $out/loop.json 2> $out/loop.err
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
ls -R $out | head -40
