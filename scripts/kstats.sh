#!/bin/bash
# rocprofv3 kernel summary of a bench.py run: bash scripts/kstats.sh <bench.py flags...>   (top kernels: name, calls, avg us)
ROOT=$(cd "$(dirname "$0")/.." && pwd); PY=${PYTHON:-python}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- $PY $ROOT/bench.py --steps 220 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state "$@" > /dev/null 2>&1
f=$(find /tmp/kst -name '*kernel_stats.csv' | head -1)
$PY - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), '%9.1f us' % (float(r['AverageNs'])/1e3), r['Percentage'])
P
