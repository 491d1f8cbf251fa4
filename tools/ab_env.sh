#!/bin/bash
# Same-box A/B of the headline under environment settings: tools/ab_env.sh <out dir> "<ENV=..>" "<ENV=..>" ...  ("-" = none); two rounds, alternating
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
for round in 1 2; do
  k=0
  for e in "$@"; do
    k=$((k+1)); [ "$e" = "-" ] && e=""
    env $e timeout 600 python bench.py --no-solver --no-loop --no-cpu-baseline ${AB_FLAGS:---no-variants} --steps 50 --warmup 5 --verbose --details '' > $out/v${k}_$round.json 2> $out/v${k}_$round.err
    python - $out/v${k}_$round.json "$e" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("[%s]" % sys.argv[2], "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "K3' %.3f ms" % r.get("avg_launch_ms", 0), "side", r.get("side_kernels_ms_per_launch"),
      "no_skip", d.get("value_no_skipping"), "dense", d.get("value_dense_world"))
PY
  done
done
