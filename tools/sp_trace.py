"""per-level durations and gaps of one sparse-preconditioner application from a rocprofv3 kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# find the last complete run of consecutive sp kernels
idx = [i for i, n in enumerate(names) if "k_sp_" in n]
runs, cur = [], []
for i in idx:
    if cur and i != cur[-1] + 1:
        runs.append(cur); cur = []
    cur.append(i)
if cur: runs.append(cur)
runs = [r for r in runs if len(r) >= 5]
run = runs[len(runs) // 2]
t_end_prev = None
tot = 0
for i in run:
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t_end_prev) / 1e3 if t_end_prev else 0.0
    nm = r["Kernel_Name"].split("k_sp_")[1][:14]
    print("%-16s grid %7s  dur %6.2f us  gap %5.2f us" % (nm, r.get("Grid_Size", "?"), (e - s) / 1e3, gap))
    t_end_prev = e
print("application: %.2f us over %d launches" % ((int(rows[run[-1]]["End_Timestamp"]) - int(rows[run[0]]["Start_Timestamp"])) / 1e3, len(run)))
