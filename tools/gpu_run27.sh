#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2seq; mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python tools/seq_latency.py 20 resident > /dev/null 2> $out/trace.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2seq/trace/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"].split("(")[0].replace("kh::","")[:34],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,int(r["Start_Timestamp"]),int(r["End_Timestamp"]),int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"]))) for r in rows]
ic=[i for i,s in enumerate(seq) if s[0].startswith("k_raster_clear")]
a,b=ic[-2],ic[-1]
print("one call: %.1f us from clear to next clear" % ((seq[b][2]-seq[a][2])/1e3))
for i in range(a,b):
    n,d,st,en,g=seq[i]
    print(f"{n:36s} wgs {g:5d} dur {d:6.1f} gap_before {(st-seq[i-1][3])/1e3:6.1f}")
PY
