#!/bin/bash
# Profiles the solver on the 10k / 30k graph (tools/quick_spa.py: three solves) on the GPU box: kernel-trace stats, then PMC
# counters in separate passes.  Outputs under gpurun_out/$1; the number of numeric factorisations of the run goes to
# gpurun_out/$1/factorizations.txt (from the solve summaries of the traced run).
tag=${1:-prof_spa}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/tools/quick_spa.py 10000 30000"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $cmd > $out/trace.out 2> $out/trace.err
grep -o "'factorizations': [0-9]*" $out/trace.out | awk '{s += $2} END {print s}' > $out/factorizations.txt
grep -o "'levels': [0-9]*" $out/trace.out | head -1 | awk '{print $2}' > $out/levels.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_$name -o p --output-format csv -- $cmd > /dev/null 2> $out/pmc_$name.err
done
python $GRAFT_REPO_ROOT/tools/level_times3.py $out/trace/t_kernel_trace.csv > $out/levels_timeline.txt 2>&1
cat $out/factorizations.txt $out/levels.txt
