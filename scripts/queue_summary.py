#!/usr/bin/env python
"""Per hardware queue of a rocprofv3 --kernel-trace CSV: span, busy time, idle time, time by kernel name, the largest idle gaps (and what ran
before / after them).  For the partitioned filter's update: which stream is the critical one and what it waits for.
   python scripts/queue_summary.py kernel_trace.csv [first_kernel_substring] [last_kernel_substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else None
last = sys.argv[3] if len(sys.argv) > 3 else None
i0 = max((i for i, r in enumerate(rows) if first and first in r["Kernel_Name"]), default=0) if first else 0
# the LAST window [first .. last]
if first:
    idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    i0 = idx[-1]
i1 = len(rows) - 1
if last:
    idx = [i for i, r in enumerate(rows) if last in r["Kernel_Name"] and i > i0]
    i1 = idx[0] if idx else i1
win = rows[i0: i1 + 1]
t0 = int(win[0]["Start_Timestamp"])
print(f"window: {len(win)} kernels, {(int(win[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms from {win[0]['Kernel_Name'][:40]} to {win[-1]['Kernel_Name'][:40]}")
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e6
    print(f"\nqueue {q}: {len(rs)} kernels, first at {(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.2f} ms, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms")
    names = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        n = r["Kernel_Name"].split("(")[0][-40:]
        names[n][0] += 1
        names[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"    {n:42s} {c:5d} x  {t:8.2f} ms")
    gaps = []
    for a, b in zip(rs, rs[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        gaps.append((g, a["Kernel_Name"].split("(")[0][-28:], b["Kernel_Name"].split("(")[0][-28:], (int(a["End_Timestamp"]) - t0) / 1e6))
    gaps.sort(reverse=True)
    tot = sum(g for g, *_ in gaps if g > 0)
    print(f"    gaps: {tot / 1e3:.2f} ms in total; the largest:")
    for g, a, b, at in gaps[:8]:
        print(f"      {g:8.1f} us at {at:7.2f} ms  after {a}  before {b}")
