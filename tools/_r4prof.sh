tools/prof_bench.sh r4prof > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/r4prof k_score_lds gpurun_out/r4prof/r4_k_score_pmc.json 51.2 > /dev/null
grep -h "k_score_lds" gpurun_out/r4prof/pmc_FETCH_SIZE/p_counter_collection.csv | head -3 > /dev/null
python bench.py --details gpurun_out/r4prof/r4_bench_line.json > gpurun_out/r4prof/r4_bench_line_compact.json 2> gpurun_out/r4prof/bench.err
wc -c gpurun_out/r4prof/r4_bench_line_compact.json; cut -c1-700 gpurun_out/r4prof/r4_bench_line_compact.json
cp gpurun_out/r4prof/trace/t_kernel_stats.csv gpurun_out/r4prof/r4_bench_kernel_stats.csv
head -4 gpurun_out/r4prof/r4_bench_kernel_stats.csv | cut -c1-160
