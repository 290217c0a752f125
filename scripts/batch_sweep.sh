#!/bin/bash
# steps/s and the slowest kernels per batch size (N = 200): bash scripts/batch_sweep.sh "4 8 16 64" [extra bench.py flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd); PY=${PYTHON:-python}
for B in ${1:-2 4 6 8 12 16 64}; do
  timeout 600 $PY $ROOT/bench.py --filters-per-gpu $B --steps 440 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state ${@:2} 2>/dev/null | $PY -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
done
