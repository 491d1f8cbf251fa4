#!/bin/bash
# End-of-round run on the GPU box: the whole GPU suite, then EVERY profile family of the round re-collected on the round's final
# code -- the bench's matcher leg (kernel trace + PMC passes), the loop-closure batch (trace + HBM traffic), ONE MatchScan (traces
# of the fused path, presets S and L, and of the general path), the solver (trace + PMC passes + level timeline) -- then the default
# bench (which reads the PMC summaries just collected) and the driver's form of it (--steps 20).
#   tools/final_run.sh <tag> [round, default r6]     outputs: gpurun_out/<tag>/<round>_*  (copy them into profiles/)
# Exits non-zero and lists what is missing when a leg did not produce its file.
tag=${1:-final}
R=${2:-r6}
out=gpurun_out/$tag
mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $out/gputest.txt
cat $out/gputest.txt
# ---- matcher leg of the bench
tools/prof_bench.sh $tag > /dev/null 2>&1
python tools/pmc_traffic.py $out k_score_lds profiles/${R}_k_score_pmc.json 51.2 > /dev/null
cp profiles/${R}_k_score_pmc.json $out/
grep "k_score_lds" $out/pmc_FETCH_SIZE/p_counter_collection.csv > $out/${R}_k_score_pmc_FETCH_SIZE.csv
grep "k_score_lds" $out/pmc_WRITE_SIZE/p_counter_collection.csv > $out/${R}_k_score_pmc_WRITE_SIZE.csv
cp $out/trace/t_kernel_stats.csv $out/${R}_bench_kernel_stats.csv
# ---- loop-closure batch: kernel trace + HBM traffic of the rasteriser
KH_PROF_LOOP_LIGHT=1 tools/prof_loop.sh ${tag}_loop > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_loop profiles/${R}_loop_pmc.json "${R}: loop-closure batch, first-point rasteriser for both presets" > /dev/null
cp profiles/${R}_loop_pmc.json $out/; cp gpurun_out/${tag}_loop/trace/t_kernel_stats.csv $out/${R}_loop_kernel_stats.csv
# ---- ONE MatchScan: kernel trace of the fused path (presets S, L) and of the general path (S)
(cd /tmp && export TMPDIR=/tmp
 for p in S L; do timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/seq_$p -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py --loop 200 $p > /dev/null 2>&1
   cp $GRAFT_REPO_ROOT/$out/seq_$p/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/${R}_seq_${p}_kernel_stats.csv; done
 timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/seq_Sg -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py --loop 200 S general > /dev/null 2>&1
 cp $GRAFT_REPO_ROOT/$out/seq_Sg/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/${R}_seq_S_general_kernel_stats.csv)
python tools/seq_latency.py > $out/${R}_seq_latency.txt 2>&1
KH_SEQ_TIMING=1 python tools/seq_latency.py --loop 130 S 2>&1 | tail -2 > $out/${R}_seq_phases.txt
# ---- solver: trace, PMC passes, level timeline
tools/prof_spa.sh ${tag}_spa > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_spa profiles/${R}_spa_pmc.json "${R}: solver, three solves of the 10k / 30k graph (tools/quick_spa.py)" > /dev/null
cp profiles/${R}_spa_pmc.json $out/; cp gpurun_out/${tag}_spa/levels_timeline.txt $out/${R}_spa_levels_timeline.txt
cp gpurun_out/${tag}_spa/trace/t_kernel_stats.csv $out/${R}_spa_kernel_stats.csv
# ---- the bench itself: default form (details -> the round's line) and the driver's form
python bench.py --details $out/${R}_bench_line.json > $out/${R}_bench_line_compact.json 2> $out/bench.err
python bench.py --steps 20 --no-replay-50k --details '' > $out/${R}_bench_line_steps20_compact.json 2> $out/bench20.err
find $out gpurun_out/${tag}_loop gpurun_out/${tag}_spa -name "*kernel_trace.csv" -size +1M -delete
find $out gpurun_out/${tag}_loop gpurun_out/${tag}_spa -name "p_counter_collection.csv" -size +4M -delete
missing=0
for f in ${R}_k_score_pmc.json ${R}_bench_kernel_stats.csv ${R}_loop_pmc.json ${R}_loop_kernel_stats.csv ${R}_seq_S_kernel_stats.csv ${R}_seq_L_kernel_stats.csv \
         ${R}_seq_S_general_kernel_stats.csv ${R}_seq_latency.txt ${R}_seq_phases.txt ${R}_spa_pmc.json ${R}_spa_levels_timeline.txt ${R}_spa_kernel_stats.csv \
         ${R}_bench_line.json ${R}_bench_line_compact.json ${R}_bench_line_steps20_compact.json; do
  if [ ! -s $out/$f ]; then echo "MISSING: $out/$f"; missing=1; fi
done
wc -c $out/${R}_bench_line_compact.json
cut -c1-1500 $out/${R}_bench_line_compact.json
exit $missing
