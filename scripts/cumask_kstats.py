#!/usr/bin/env python
"""Per-rank kernel time of a scripts/tiled_cumask.py run traced with `rocprofv3 --kernel-trace --stats --output-format csv -d DIR`:
one *_kernel_stats.csv per process.  Prints, per kernel class, calls and total ms of every process (rank order is not known to the
profiler: processes are listed by their k_tile_gemm_tn time), and the mean / max over processes.
    python scripts/cumask_kstats.py DIR [updates=3]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
procs = []
for f in files:
    tot = {}
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("eqf::", "")
        c, t = tot.get(name, (0, 0.0))
        tot[name] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]) / 1e6)
    if any(k.startswith("k_tile") for k in tot):
        procs.append(tot)
procs.sort(key=lambda t: -t.get("k_tile_gemm_tn", (0, 0))[1])
names = sorted({k for p in procs for k in p}, key=lambda k: -sum(p.get(k, (0, 0))[1] for p in procs))
print(f"# {len(procs)} processes with tile kernels; ms per update ({updates} updates in the run), per process, sorted by their product time")
for k in names[:12]:
    v = [p.get(k, (0, 0.0)) for p in procs]
    ms = [t / updates for _, t in v]
    print(f"{k[:38]:38s} calls/upd {v[0][0] / updates:7.1f}  " + " ".join(f"{m:7.2f}" for m in ms) + f"   mean {sum(ms) / len(ms):7.2f}  max {max(ms):7.2f}")
allms = [sum(t for _, t in p.values()) / updates for p in procs]
print(f"{'ALL KERNELS':38s} {'':17s}  " + " ".join(f"{m:7.2f}" for m in allms) + f"   mean {sum(allms) / len(allms):7.2f}  max {max(allms):7.2f}")
