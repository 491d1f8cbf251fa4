"""Per-launch kernel durations of the solver's LAST factorisation + backward sweep from a rocprofv3 --kernel-trace CSV:
python tools/level_times3.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0].replace("kh::", "").replace("void ", "")
    return n
seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
        int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1))) for r in rows]
# last k_assemble marks the start of the last factorisation
starts = [i for i, s in enumerate(seq) if s[0] == "k_assemble"]
if not starts:
    sys.exit("no k_assemble in the trace")
i0 = starts[-1]
end = len(seq)
tot = {}
t_first = seq[i0][2]
print("last LM iteration, launch by launch (us, grid):")
for i in range(i0, end):
    name, us, st, en, gx, gy = seq[i]
    gap = (st - seq[i - 1][3]) / 1e3 if i > i0 else 0.0
    tot.setdefault(name, [0, 0.0])
    tot[name][0] += 1; tot[name][1] += us
    print(f"  {name:28s} {us:8.1f}  grid {gx}x{gy}  gap {gap:6.1f}  t={(st - t_first) / 1e3:8.1f}")
print("sums:")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} calls {n:3d}  {us:8.1f} us")
print("span %.1f us" % ((seq[end - 1][3] - t_first) / 1e3))
