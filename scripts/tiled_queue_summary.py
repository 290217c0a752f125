#!/usr/bin/env python
"""Per hardware queue (= stream) of ONE update of the partitioned filter, from a rocprofv3 --kernel-trace CSV: span, busy time, and the kernel time
by kernel -- which stream carries what, and where a stream sits idle.   python scripts/tiled_queue_summary.py <kernel_trace.csv> [which update from the end = 1]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
preps = [i for i, r in enumerate(rows) if "k_tl_prep" in r["Kernel_Name"]]
fins = [i for i, r in enumerate(rows) if "k_tl_finish" in r["Kernel_Name"]]
i0 = preps[-back]
i1 = next(i for i in fins if i > i0)
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["End_Timestamp"])
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
print(f"update: {(t1 - t0) / 1e6:.2f} ms from k_tl_prep to the end of k_tl_finish, {len(sel)} kernels")
byq = collections.defaultdict(list)
for r in sel:
    byq[r.get("Queue_Id", "?")].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    s, e = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: from {(s - t0) / 1e6:6.2f} to {(e - t0) / 1e6:6.2f} ms, {len(rs)} kernels, busy {busy / 1e6:6.2f} ms, idle inside its span {(e - s - busy) / 1e6:6.2f} ms")
    by = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].replace("eqf::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:44]
        by[n][0] += 1
        by[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"      {d / 1e6:7.2f} ms  {c:5d} x {d / c / 1e3:8.1f} us  {n}")
