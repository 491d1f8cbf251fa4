#!/bin/bash
# Profiles the loop-closure batch (tools/loop_pieces.py, one piece: the kernels of the two stages do not overlap) on the GPU
# box: kernel-trace stats, then PMC counters in separate passes.  Outputs under gpurun_out/$1.
tag=${1:-prof_loop}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PIECES=1,1
cmd="python $GRAFT_REPO_ROOT/tools/loop_pieces.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $cmd > $out/trace.out 2> $out/trace.err
sets=(FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_ATOMIC_sum TCC_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum")
# KH_PROF_LOOP_LIGHT=1: only the two HBM-traffic passes
[ -n "$KH_PROF_LOOP_LIGHT" ] && sets=(FETCH_SIZE WRITE_SIZE)
for c in "${sets[@]}"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_$name -o p --output-format csv -- $cmd > /dev/null 2> $out/pmc_$name.err
done
ls -R $out | head -40
