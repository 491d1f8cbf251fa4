#!/bin/bash
# kseq_prep_batch's block size (KH_PREP_THREADS) against its time in a trace of the loop-closure batch
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prepsweep; cd /tmp && export TMPDIR=/tmp
export PIECES=1,1
for th in ${THS:-256 512 1024}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prepsweep/th$th
  KH_PREP_THREADS=$th timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/loop_pieces.py > $out.out 2>&1
  python - $out/t_kernel_stats.csv $th <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'prep_batch' in r['Name'] or 'kseq_tile' in r['Name']:
        print('threads', sys.argv[2], r['Name'][:28], r['Calls'], '%.1f us' % (float(r['AverageNs']) / 1e3))
PY
  grep "pieces:" $out.out | tail -1
  find $out -name "*kernel_trace.csv" -delete
done
