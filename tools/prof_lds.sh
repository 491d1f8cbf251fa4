#!/bin/bash
# LDS-staged scoring path on the GPU box: variants A/B (KH_LIBRARY builds from tools/build_variant.sh), kernel trace, LDS counters.
# Outputs under gpurun_out/$1.
tag=${1:-lds}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export KH_LDS_SCORE=1
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-solver --no-loop"
for v in $GRAFT_REPO_ROOT/variants/*.so; do
  n=$(basename $v .so)
  KH_BENCH_NO_CHECK=1 KH_LIBRARY=$v timeout 200 $cmd > $out/bench_$n.json 2> $out/bench_$n.err
done
$cmd > $out/bench_default.json 2> $out/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $cmd > $out/bench_trace.json 2> $out/trace.err
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*" | sort -u > $out/counters.txt
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_$name -o p --output-format csv -- $cmd > /dev/null 2> $out/pmc_$name.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(os.path.basename(f), round(d["value"]), round(d["ms_per_step"], 3), round(r["avg_launch_ms"], 4))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
