#!/bin/bash
# functional run of the N = 2 path of bench.py on a single-GPU box (ranks share the device, gloo for the process group)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2n2; mkdir -p $out
KH_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 1 > $out/bench_n2.json 2> $out/bench_n2.err
echo "rc=$?"
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/r2n2/bench_n2.json"))
    for k in ("value","n_gpus","solve_ms","solve_ms_edge_sharded","solve_collective","strong_scaling"):
        print(k, b.get(k))
    print([k for k in b if "error" in k])
except Exception as e:
    print("no json", e)
PY
tail -5 $out/bench_n2.err | cut -c1-300
