#!/bin/bash
# the fused MatchScan path on the GPU box: parity tests, regression of the neighbours, latency, phase timing, kernel trace
out=$GRAFT_REPO_ROOT/gpurun_out/seq
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_seq_gpu.py -x -q > $out/test_seq.log 2>&1; echo "test_seq rc=$?" | tee -a $out/summary.txt
tail -25 $out/test_seq.log
if [ "$1" != "quick" ]; then
timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_mapper_gpu.py tests/test_dropin_mapper_gpu.py tests/test_raster_corner_cases_gpu.py tests/test_group_gpu.py -x -q > $out/test_neighbours.log 2>&1; echo "neighbours rc=$?" | tee -a $out/summary.txt
tail -8 $out/test_neighbours.log
fi
timeout 300 python tools/seq_latency.py > $out/latency.txt 2>&1; cat $out/latency.txt
for p in S L K; do KH_SEQ_TIMING=1 timeout 100 python tools/seq_latency.py --loop 130 $p 2>&1 | tail -3; done | tee $out/phases.txt
cd /tmp && export TMPDIR=/tmp
for p in S L; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace_$p -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py --loop 200 $p > $out/trace_$p.log 2>&1
  f=$(find $out/trace_$p -name "*kernel_stats.csv" | head -1); echo "== $p fused"; [ -n "$f" ] && cut -d, -f1-4 $f | head -16
done
find $out -name "*kernel_trace.csv" -size +2M -delete
