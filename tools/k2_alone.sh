#!/bin/bash
# K2' / K3' / K4 durations without one another beside them: the headline leg unchunked (KH_PIPELINE=0) under a kernel trace
# tools/k2_alone.sh <tag> [lib]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/k2alone_$1
[ -n "$2" ] && export KH_LIBRARY=$GRAFT_REPO_ROOT/$2
KH_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-solver --no-loop --no-cpu-baseline --no-variants --no-replay-50k --steps 6 --warmup 2 --details '' > $out.out 2>&1
python - $out/t_kernel_stats.csv $1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print(sys.argv[2], r['Name'][:60].ljust(60), r['Calls'].rjust(5), '%.1f us' % (float(r['AverageNs']) / 1e3))
PY
find $out -name "*kernel_trace.csv" -delete
