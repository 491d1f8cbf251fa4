#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2y; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spa_gpu.py -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -3
KH_SPA_TIMING=1 timeout 300 python tools/quick_spa.py 2> $out/timing.err | tail -1 | cut -c1-120
grep "k_factor" $out/timing.err | tail -14 | sed -n 4,7p | cut -c1-300
timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python tools/quick_spa.py > /dev/null 2> $out/trace.err
python tools/level_times.py $(find $out/trace -name "*kernel_trace.csv" | head -1)
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_[A-Z]*" | sort -u > $out/counters.txt
cat $out/counters.txt | tr '\n' ' '
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  d=$out/pmc_$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c -d $d -o p --output-format csv -- python tools/quick_spa.py > /dev/null 2> $d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot):
    if "factor" in k[0] or "backward" in k[0] or "extend" in k[0]:
        print(k, "calls", n[k], "total %.4g" % tot[k], "per call %.4g" % (tot[k] / n[k]))
PY
done
