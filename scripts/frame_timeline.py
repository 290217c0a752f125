"""Kernel timeline of one frame from a rocprofv3 --kernel-trace csv: start offset, duration and the gap to the previous kernel.
usage: frame_timeline.py <kernel_trace.csv> [frame index]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void eqf::", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_burst_build")]
fi = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0, i1 = starts[fi], starts[fi + 1]
t0 = rows[i0][0]
prev_end = rows[i0 - 1][1]
tot_k = tot_g = 0.0
for s, e, name in rows[i0:i1]:
    gap = (s - prev_end) / 1e3
    print("%8.2f us  +%6.2f gap  %7.2f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name[:60]))
    tot_k += (e - s) / 1e3
    tot_g += gap
    prev_end = e
print("frame: %.2f us = kernels %.2f + gaps %.2f" % ((rows[i1][0] - t0) / 1e3, tot_k, tot_g))
