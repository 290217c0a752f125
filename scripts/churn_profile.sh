#!/bin/bash
# (GPU box) rocprofv3 kernel summary of every mode of scripts/churn_bench.py: which launches a frame consists of with the gate / with churn
ROOT=$(cd "$(dirname "$0")/.." && pwd); PY=${PYTHON:-python}
cd /tmp && export TMPDIR=/tmp
for mode in fixed fixed+outlier-gate churn churn+outlier-gate; do
  rm -rf /tmp/chp
  CHURN_MODE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/chp -o k -- $PY $ROOT/scripts/churn_bench.py 2>/dev/null | grep "steps/s"
  f=$(find /tmp/chp -name '*kernel_stats.csv' | head -1)
  $PY - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print('   ', r['Name'].split('(')[0][:60].ljust(60), r['Calls'].rjust(6), '%9.1f us avg' % (float(r['AverageNs'])/1e3), '%8.2f ms' % (float(r['TotalDurationNs'])/1e6))
print('    all kernels %.2f ms' % (tot/1e6))
P
done
