#!/bin/bash
# Profiles `bench.py` (matcher leg only) on the GPU box: kernel-trace stats, then PMC counters in separate passes (gpurun refuses
# --pmc combined with API traces; FETCH_SIZE and WRITE_SIZE cannot share a pass).  Outputs under gpurun_out/$1.
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-solver --no-loop --no-variants --details ''"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $cmd > $out/bench_trace.json 2> $out/trace.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_$name -o p --output-format csv -- $cmd > /dev/null 2> $out/pmc_$name.err
done
ls -R $out | head -50
