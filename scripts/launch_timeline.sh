#!/bin/bash
# Runs ON THE GPU BOX: per-launch durations (rocprofv3 --kernel-trace) and HBM traffic (two --pmc passes: FETCH_SIZE, WRITE_SIZE,
# KiB per launch) of the LAST frame of a short bench run, in dispatch order.
# Usage: scripts/launch_timeline.sh <filters per GPU> [landmarks]
set -u
B=${1:-64}
N=${2:-200}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt1 /tmp/lt2 /tmp/lt3
ARGS="--filters-per-gpu $B --landmarks $N --steps 66 --warmup 22 --pmc-child"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt1 -o b -- python $ROOT/bench.py $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/lt2 -o b -- python $ROOT/bench.py $ARGS > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/lt3 -o b -- python $ROOT/bench.py $ARGS > /dev/null 2>&1
python - "$B" "$N" <<'PY'
import csv, glob, sys
B, N = sys.argv[1], sys.argv[2]
def last_frame(rows, key):
    rows.sort(key=key)
    idx = [i for i, r in enumerate(rows) if "k_update_prep" in r["Kernel_Name"]]
    i0 = idx[-1]
    j = i0 - 1
    while j > 0 and "k_burst" in rows[j]["Kernel_Name"]:
        j -= 1
    k = i0 + 1
    while k < len(rows) and "k_burst" not in rows[k]["Kernel_Name"] and "k_update_prep" not in rows[k]["Kernel_Name"]:
        k += 1
    return rows[j + 1:k]
t = last_frame([r for r in csv.DictReader(open(glob.glob('/tmp/lt1/**/*kernel_trace.csv', recursive=True)[0]))], lambda r: int(r["Start_Timestamp"]))
f = last_frame([r for r in csv.DictReader(open(glob.glob('/tmp/lt2/**/*counter_collection.csv', recursive=True)[0]))], lambda r: int(r["Dispatch_Id"]))
w = last_frame([r for r in csv.DictReader(open(glob.glob('/tmp/lt3/**/*counter_collection.csv', recursive=True)[0]))], lambda r: int(r["Dispatch_Id"]))
print(f"# last frame of `bench.py --filters-per-gpu {B} --landmarks {N}` on MI355X: launches in dispatch order")
print("# duration: rocprofv3 --kernel-trace; HBM read / write: --pmc FETCH_SIZE / WRITE_SIZE (separate passes, raw counter, MiB per launch)")
print(f"{'kernel':44s} {'grid':>8s} {'us':>9s} {'read MiB':>9s} {'write MiB':>9s} {'TB/s':>6s}")
for a, b, c in zip(t, f, w):
    us = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1000
    rd, wr = float(b["Counter_Value"]) / 1024, float(c["Counter_Value"]) / 1024
    grid = a.get("Grid_Size_X", a.get("Grid_Size", "?"))
    name = a["Kernel_Name"].replace("void eqf::", "").split("(")[0][:44]
    print(f"{name:44s} {grid:>8s} {us:9.1f} {rd:9.1f} {wr:9.1f} {(rd + wr) * 1.048576 / us:6.2f}")
PY
