#!/bin/bash
# kseq_tile's batch grid (KH_TILE_BLOCKS workgroups per job) against the kernel's time in a trace of the loop-closure batch
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/tilesweep; cd /tmp && export TMPDIR=/tmp
export PIECES=1,1
for tb in ${TBS:-2048 256 128 64 32 16}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/tilesweep/tb$tb
  KH_TILE_BLOCKS=$tb timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/loop_pieces.py > $out.out 2>&1
  echo "KH_TILE_BLOCKS=$tb $(tail -1 $out.out) $(grep -h 'kseq_tile\|kseq_prep_batch' $out/t_kernel_stats.csv | cut -d, -f1,2,4 | tr '\n' ' ' | sed 's/kh::RasterJob const\*[^"]*//g')"
  find $out -name "*kernel_trace.csv" -delete
done
