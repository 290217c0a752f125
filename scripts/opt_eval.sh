#!/bin/bash
# Runs ON THE GPU BOX: one eqf_debug_option at several values per batch size -> gpurun_out/opt_eval.txt
# Usage: scripts/opt_eval.sh NAME "V1 V2 .." "B:N B:N .."
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/opt_eval.txt
mkdir -p $ROOT/gpurun_out
NAME=$1; VALS=$2; CASES=${3:-"8:200"}
( echo "# bench.py --filters-per-gpu B --landmarks N --debug-option $NAME=V: steps/s, kernel classes (avg us per launch)"
  for C in $CASES; do
    B=${C%%:*}; N=${C##*:}
    ST=880; WU=110
    [ "$B" -ge 64 ] && ST=440
    [ "$N" -ge 1000 ] && ST=220
    for V in $VALS; do
      timeout 300 python $ROOT/bench.py --filters-per-gpu $B --landmarks $N --steps $ST --warmup $WU --debug-option $NAME=$V --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B N=$N $NAME=$V', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
    done
  done ) > $OUT 2>&1
cat $OUT
