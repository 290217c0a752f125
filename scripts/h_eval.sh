#!/bin/bash
# Runs ON THE GPU BOX: quick evaluation of a change to k_chol_resident -- the bitwise pins, then the update kernel at B = 1, 2, 4, 8, 16, 64, N = 1000,
# then the row-head stamps of the instrumented build -> gpurun_out/h_eval_<tag>.txt
# Usage: scripts/h_eval.sh <tag> ["ENV=.. ENV=.." extra environment for the bench runs]
TAG=${1:-x}
EXTRA=${2:-X=1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/h_eval_$TAG.txt
cd $ROOT
( python -m pytest tests/test_gpu_parity.py -x -q -k "resident or layouts or prep_roles or handoff or stream_parity or golden" 2>&1 | tail -4
  python -m pytest tests/test_gpu_configs.py -x -q -k "withheld or batch_of_16 or mid_size or cfg4" 2>&1 | tail -4
  for B in 1 2 4 8 16 64; do
    env $EXTRA python bench.py --filters-per-gpu $B --steps 880 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', round(d['value']), 'steps/s  err', d['device_error_flag'], 'cover', d['profile_coverage']['kernel_time_over_wall'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
  done
  env $EXTRA python bench.py --landmarks 1000 --steps 220 --warmup 55 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1000', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:5]])"
  if [ -f build_variants/libeqf_stamps.so ]; then
    EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_stamps.so python scripts/res_stamps.py 200 1 2>/dev/null | grep -v "^factor64\|^   wave"
  fi ) > $OUT 2>&1
cat $OUT
