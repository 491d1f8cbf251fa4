#!/bin/bash
# quick A/B of the LDS-staged scoring path on the GPU box: parity tests of the path, then the headline leg of bench.py with the
# default build and every build under variants/ (tools/build_variant.sh).  Outputs under gpurun_out/$1.
tag=${1:-ldsq}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout 600 python -m pytest tests/test_matcher_gpu.py -x -q -m gpu -k "lds" 2>&1 | tail -3
export KH_LDS_SCORE=1
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-solver --no-loop"
$cmd > $out/bench_default.json 2> $out/bench_default.err


grep "kh lds" $out/bench_*.err | head -6
for v in $GRAFT_REPO_ROOT/variants/*.so; do
  [ -e "$v" ] || continue
  n=$(basename $v .so)
  KH_BENCH_NO_CHECK=1 KH_LIBRARY=$v timeout 200 $cmd > $out/bench_$n.json 2> $out/bench_$n.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(os.path.basename(f), "matches/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "k3 ms", round(r["avg_launch_ms"], 4))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
if [ -n "$2" ]; then
  cd /tmp && export TMPDIR=/tmp
  cmd="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-solver --no-loop"
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $cmd > $out/bench_trace.json 2> $out/trace.err
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
    name=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $c -d $out/pmc_$name -o p --output-format csv -- $cmd > /dev/null 2> $out/pmc_$name.err
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $out $out/summary.json > /dev/null 2>&1
  python - <<PY
import json
d = json.load(open("$out/summary.json"))["kernels"]
for k in d:
    if "lds" in k: print(k, json.dumps({a: round(b) if isinstance(b, float) else b for a, b in d[k].items() if not a.startswith("launches")}))
PY
fi
