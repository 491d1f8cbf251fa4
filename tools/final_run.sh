#!/bin/bash
# end-of-round run on the GPU box: the whole GPU suite, the profile set of the bench's matcher leg, then the default bench
# (which reads the PMC summary just collected).  Outputs under gpurun_out/$1 (copy the r5_* files into profiles/).
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
python bench.py --details $out/r5_bench_line.json > $out/r5_bench_line_compact.json 2> $out/bench.err
python bench.py --steps 20 --details '' > $out/r5_bench_line_steps20_compact.json 2> $out/bench20.err
wc -c $out/r5_bench_line_compact.json
cut -c1-900 $out/r5_bench_line_compact.json
