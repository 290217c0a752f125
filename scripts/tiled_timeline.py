#!/usr/bin/env python
"""Kernel timeline of ONE update of the partitioned filter from a rocprofv3 --kernel-trace CSV: start (us from the update's first
launch), duration, queue, kernel.   python scripts/tiled_timeline.py <kernel_trace.csv> [max_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_tl_prep" in n][-1]
t0 = int(rows[idx]["Start_Timestamp"])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for r in rows[idx: idx + limit]:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("eqf::", "").replace("void ", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short))
