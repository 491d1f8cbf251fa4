"""Counts the fill kernels (hipMemsetAsync -> __amd_rocclr_fillBufferAligned) in the STEADY STATE of a rocprofv3 kernel trace of
tools/loop_pieces.py: everything from the first rasterisation of the last five batches on (two k_raster_scan launches per
batch).  usage: python tools/steady_state_fills.py gpurun_out/r3loop/trace/t_kernel_trace.csv"""
import csv
import sys

rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
scans = [i for i, r in enumerate(rows) if "k_raster_scan" in r[1]]
first = scans[-10]
tail = rows[first:]
print(f"kernel launches in the run: {len(rows)}, of them fillBufferAligned: {sum('fillBufferAligned' in r[1] for r in rows)} (slot set-up)")
print(f"kernel launches in the last five batches: {len(tail)}, of them fillBufferAligned: {sum('fillBufferAligned' in r[1] for r in tail)}")
