#!/bin/bash
# Runs ON THE GPU BOX: the same bench points for several builds of the library (build_variants/libeqf_<name>.so; "product" = the in-tree one)
# Usage: scripts/variants_eval.sh <tag> "<name> <name> ..." "<B B ...>" -> gpurun_out/variants_<tag>.txt
TAG=${1:-x}; NAMES=${2:-product}; BS=${3:-"1 8"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/variants_$TAG.txt
cd $ROOT
( for NM in $NAMES; do
    if [ $NM = product ]; then LIBENV="X=1"; else LIBENV="EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_$NM.so"; fi
    for B in $BS; do
      for rep in 1 2; do
      env $LIBENV python bench.py --filters-per-gpu $B --steps 880 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NM B=$B', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:4]])"
      done
    done
  done ) > $OUT 2>&1
cat $OUT
