#!/bin/bash
# Runs ON THE GPU BOX: the burst leaving the update's operands (eqf_debug_option "cs_in_burst") on and off, per batch size
# -> gpurun_out/cs_eval.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/cs_eval.txt
mkdir -p $ROOT/gpurun_out
CASES=${1:-"2:200 8:200 16:200 64:200 1:1000"}
( echo "# bench.py --filters-per-gpu B --landmarks N: steps/s, kernel classes (avg us per launch); cs_in_burst = 1 (default) / 0"
  for C in $CASES; do
    B=${C%%:*}; N=${C##*:}
    ST=880; WU=110
    [ "$B" -ge 64 ] && ST=440
    [ "$N" -ge 1000 ] && ST=220
    for V in 1 0; do
      timeout 300 python $ROOT/bench.py --filters-per-gpu $B --landmarks $N --steps $ST --warmup $WU --debug-option cs_in_burst=$V --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B N=$N cs_in_burst=$V', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
    done
  done ) > $OUT 2>&1
cat $OUT
