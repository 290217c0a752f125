"""Dev tool: per-launch durations of the kernels of the last full vision frame from a rocprofv3 kernel trace CSV."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void eqf::", "")[:28] for r in rows]
# last k_update_prep
idx = max(i for i, n in enumerate(names) if "k_update_prep" in n)
prev = max(i for i, n in enumerate(names[:idx]) if "k_update_prep" in n)
t0 = int(rows[prev]["Start_Timestamp"])
last_end = None
for i in range(prev - 2, idx):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - last_end) / 1e3 if last_end else 0.0
    print(f"{names[i]:30s} start {(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:6.2f} us  gap {gap:5.2f}")
    last_end = e
