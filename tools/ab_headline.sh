#!/bin/bash
# Same-box A/B of the headline (config-2 CorrelateScan batches): tools/ab_headline.sh <out dir> <lib A> [<lib B> ...]
# ("-" = the in-tree library).  Each library runs the headline leg only, twice, alternating.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so); [ "$lib" = "-" ] && name=tree
    if [ "$lib" = "-" ]; then unset KH_LIBRARY; else export KH_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
    timeout 600 python bench.py --no-solver --no-loop --no-cpu-baseline ${AB_FLAGS:---no-variants} --steps 50 --warmup 5 --verbose --details '' > $out/${name}_$round.json 2> $out/${name}_$round.err
    python - $out/${name}_$round.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[2], "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "K3' %.3f ms" % r.get("avg_launch_ms", 0), "side", r.get("side_kernels_ms_per_launch"),
      "reads/launch %.3g" % r.get("window_reads_per_launch", 0), "no_skip", d.get("value_no_skipping"), "dense", d.get("value_dense_world"))
PY
  done
done
