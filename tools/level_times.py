"""Per-level kernel durations of the solver from a rocprofv3 --kernel-trace CSV: python tools/level_times.py <t_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("kh::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
# last factorisation: the last run of k_factor launches before the last k_backward run
idx = [i for i, s in enumerate(seq) if s[0] in ("k_factor", "k_factor2")]
last = idx[-14:] if idx else []
print("k_factor  (us, workgroups):", [(round(seq[i][1]), seq[i][2]) for i in last], "sum", round(sum(seq[i][1] for i in last)))
ea = [i for i in range(last[0], last[-1]) if seq[i][0] == "k_extend_add"]
print("k_extend_add (us):", [round(seq[i][1]) for i in ea], "sum", round(sum(seq[i][1] for i in ea)))
bw = [i for i, s in enumerate(seq) if s[0] == "k_backward"][-14:]
print("k_backward (us):", [round(seq[i][1]) for i in bw], "sum", round(sum(seq[i][1] for i in bw)))
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(last[0], bw[-1])]
print("gaps between launches from the first factor level to the last backward level (us): sum", round(sum(gaps)), "max", round(max(gaps)))
