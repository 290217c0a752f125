#!/bin/bash
# PMC counters of the GEMM kernels in scripts/gemm_bench.py (ours and the vendor yardstick), one rocprofv3 --pmc pass per counter, medians per
# kernel next to the kernel-trace durations:   bash scripts/gemm_pmc.sh ["COUNTER ..."]
# default: matrix-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and the clock (GUI_ACTIVE / 8 / duration)
ROOT=$(cd "$(dirname "$0")/.." && pwd); PY=${PYTHON:-python}
COUNTERS=${1:-"SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"}
cd /tmp && export TMPDIR=/tmp
for c in $COUNTERS; do
  rm -rf /tmp/gpmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/gpmc_$c -o b -- $PY $ROOT/scripts/gemm_bench.py > /dev/null 2>&1
done
$PY - $COUNTERS <<'P'
import csv, glob, statistics, collections, sys
out = collections.defaultdict(dict)
for c in sys.argv[1:]:
    f = glob.glob(f"/tmp/gpmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no data for", c)
        continue
    rows = list(csv.DictReader(open(f[0])))
    kt = glob.glob(f"/tmp/gpmc_{c}/**/*kernel_trace.csv", recursive=True)
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    for r in rows:
        name = r["Kernel_Name"][:48]
        if "gemm" not in name.lower() and "Cijk" not in name:
            continue
        key = (name, r.get("Grid_Size", ""))
        out[key].setdefault(c, []).append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            out[key].setdefault("us", []).append(dur[r["Dispatch_Id"]])
for (name, grid), d in out.items():
    m = {k: statistics.median(v) for k, v in d.items()}
    us = m.pop("us", float("nan"))
    line = f"{name:48s} grid {grid:>8s} {us:8.1f} us "
    line += " ".join(f"{k}={v:.4g}" for k, v in m.items())
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        gui = m["GRBM_GUI_ACTIVE"]
        line += f"  clock {gui / 8 / us / 1e3:.2f} GHz  MFMA-busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024):.3f}"
    print(line)
P
