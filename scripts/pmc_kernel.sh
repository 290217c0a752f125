#!/bin/bash
# per-kernel means of a few PMC counters, one rocprofv3 pass each: bash scripts/pmc_kernel.sh "<counters>" <kernel substring> <bench.py flags...>
ROOT=$(cd "$(dirname "$0")/.." && pwd); PY=${PYTHON:-python}
counters=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for c in $counters; do
  rm -rf /tmp/pmck
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmck -o b -- $PY $ROOT/bench.py "$@" --steps 44 --warmup 22 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --pmc-child > /dev/null 2>&1
  f=$(find /tmp/pmck -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && $PY - "$f" "$pat" "$c" <<'P'
import csv,sys,statistics
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
print(sys.argv[3].ljust(28), "calls", len(v), "mean %.4g" % (statistics.mean(v) if v else float('nan')))
P
done
