#!/bin/bash
# the headline under explicit chunk splits (KH_CHUNK_SIZES): tools/chunk_sweep.sh <out> <lib or -> "<sizes>" ...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
lib=$1; shift
mkdir -p $out
[ "$lib" != "-" ] && export KH_LIBRARY=$GRAFT_REPO_ROOT/$lib
for sz in "$@"; do
  if [ "$sz" = "-" ]; then unset KH_CHUNK_SIZES; else export KH_CHUNK_SIZES=$sz; fi
  timeout 300 python bench.py --no-solver --no-loop --no-cpu-baseline --no-variants --no-replay-50k --steps 40 --warmup 5 --verbose --details '' > $out/c_$sz.json 2> $out/c_$sz.err
  python - $out/c_$sz.json "$lib $sz" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("[%s]" % sys.argv[2], "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "K3' %.3f ms" % r.get("avg_launch_ms", 0), "side", {k: round(v, 3) for k, v in r.get("side_kernels_ms_per_launch", {}).items()})
PY
done
