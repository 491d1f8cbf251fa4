#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z3; mkdir -p $out
bash tools/gpu_full_tests.sh r2z3
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.load(open("gpurun_out/r2z3/bench.json"))
for k in ("value","ms_per_step","solve_ms","solve_ms_cached_analysis","solve_symbolic_ms","solve_factor_gpu_ms","solve_backward_gpu_ms","loop_batch_ms","loop_pairs_per_s","replay_scans_per_s"):
    print(k, b.get(k))
print(b.get("solve_rooflines"))
PY
