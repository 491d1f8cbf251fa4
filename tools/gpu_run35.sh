#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 32 64 32 64; do
echo "KH_HOST_THREADS=$t" $(KH_HOST_THREADS=$t timeout 300 python bench.py --no-solver --no-loop --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))") $(KH_HOST_THREADS=$t timeout 300 python tools/prof_legs.py loop 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['loop_batch_ms'],2))")
done
