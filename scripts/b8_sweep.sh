#!/bin/bash
# Runs ON THE GPU BOX: the 8-filters-per-GPU operating point under the launch-shape switches -> gpurun_out/b8_sweep.txt
# Usage: scripts/b8_sweep.sh [B] ["ENV1=..;ENV2=.. ..." variants separated by spaces]
set -u
B=${1:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/b${B}_sweep.txt
mkdir -p $ROOT/gpurun_out
VARIANTS=${2:-"default EQF_BURST_ROWS=1 EQF_BURST_ROWS=4 EQF_RES_FOLD_PREP=3 EQF_CHOL_RESIDENT=0"}
( echo "# bench.py --filters-per-gpu $B --steps 880 --warmup 110 (N = 200): steps/s, kernel classes (avg us per launch)"
  for V in $VARIANTS; do
    if [ "$V" = default ]; then envs="X=1"; else envs=$(echo "$V" | tr ';' ' '); fi
    env $envs timeout 300 python $ROOT/bench.py --filters-per-gpu $B --steps 880 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B $V', round(d['value']), 'steps/s  err', d['device_error_flag'], 'cover', d['profile_coverage']['kernel_time_over_wall'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
  done ) > $OUT 2>&1
cat $OUT
