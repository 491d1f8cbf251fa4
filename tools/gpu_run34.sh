#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 64 128; do
echo "KH_HOST_THREADS=$t"
KH_HOST_THREADS=$t timeout 600 python tools/replay.py --scans 50000 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','scans_per_s')}); print({k:round(v) for k,v in d['stats'].items() if k.endswith('_ms')})"
KH_HOST_THREADS=$t timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-90
done
