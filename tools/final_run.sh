#!/bin/bash
# end-of-round run on the GPU box: the whole GPU suite, the profile set of the bench's matcher leg (kernel trace + PMC passes), the
# loop-closure batch's (trace + FETCH_SIZE / WRITE_SIZE), traces of ONE MatchScan, then the default bench (which reads the PMC
# summaries just collected) and the driver's form of it (--steps 20).  Outputs under gpurun_out/$1 (copy the r5_* files into profiles/).
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $out/gputest.txt
cat $out/gputest.txt
tools/prof_bench.sh $tag > /dev/null 2>&1
python tools/pmc_traffic.py $out k_score_lds profiles/r5_k_score_pmc.json 51.2 > /dev/null
cp profiles/r5_k_score_pmc.json $out/
grep "k_score_lds" $out/pmc_FETCH_SIZE/p_counter_collection.csv > $out/r5_k_score_pmc_FETCH_SIZE.csv
grep "k_score_lds" $out/pmc_WRITE_SIZE/p_counter_collection.csv > $out/r5_k_score_pmc_WRITE_SIZE.csv
cp $out/trace/t_kernel_stats.csv $out/r5_bench_kernel_stats.csv
# loop-closure batch: kernel trace + HBM traffic of the rasteriser
KH_PROF_LOOP_LIGHT=1 tools/prof_loop.sh ${tag}_loop > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${tag}_loop profiles/r5_loop_pmc.json "round 5: first-point rasteriser for both presets" > /dev/null
cp profiles/r5_loop_pmc.json $out/; cp gpurun_out/${tag}_loop/trace/t_kernel_stats.csv $out/r5_loop_kernel_stats.csv
# ONE MatchScan: kernel trace of the fused path (presets S, L) and of the general path (S)
(cd /tmp && export TMPDIR=/tmp
 for p in S L; do timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/seq_$p -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py --loop 200 $p > /dev/null 2>&1
   cp $GRAFT_REPO_ROOT/$out/seq_$p/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/r5_seq_${p}_kernel_stats.csv; done
 timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/seq_Sg -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py --loop 200 S general > /dev/null 2>&1
 cp $GRAFT_REPO_ROOT/$out/seq_Sg/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/r5_seq_S_general_kernel_stats.csv)
python tools/seq_latency.py > $out/r5_seq_latency.txt 2>&1
KH_SEQ_TIMING=1 python tools/seq_latency.py --loop 130 S 2>&1 | tail -2 > $out/r5_seq_phases.txt
python bench.py --details $out/r5_bench_line.json > $out/r5_bench_line_compact.json 2> $out/bench.err
python bench.py --steps 20 --details '' > $out/r5_bench_line_steps20_compact.json 2> $out/bench20.err
find $out gpurun_out/${tag}_loop -name "*kernel_trace.csv" -size +1M -delete
find $out gpurun_out/${tag}_loop -name "p_counter_collection.csv" -size +4M -delete
wc -c $out/r5_bench_line_compact.json
cut -c1-1200 $out/r5_bench_line_compact.json
