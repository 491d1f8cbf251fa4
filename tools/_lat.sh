python tools/seq_latency.py 20 resident
KH_MATCH_TIMING=1 python tools/seq_latency.py 20 resident 2>&1 | grep -v "^S Match" | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4lat -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/seq_latency.py 20 resident > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r4lat/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: find the last k_find_valid and print from there
idx = max(i for i, r in enumerate(rows) if "k_find_valid" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s dur %6.1f us  gap %6.1f  t=%7.1f  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][-60:], (e - s) / 1e3, (s - prev_end) / 1e3, (s - t0) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"]))
    prev_end = e
PY
