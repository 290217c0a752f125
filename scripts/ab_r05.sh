#!/bin/bash
# Same box, same run: this round's library against round 5's (build_variants/libeqf_r05.so, built from `git show 02bf264:eqf_vio_amd/csrc/...`)
# -- boxes differ by 2-3 % on the update kernels, so cross-round comparisons of profiles/ files cannot show a 2 % change.  steps/s, best of 3.
#   scripts/ab_r05.sh > profiles/r06_ab_against_r05.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { # run <label> <env> <bench args>
  local label=$1 envs=$2; shift 2
  best=0
  for i in 1 2 3; do
    v=$(env $envs python $ROOT/bench.py "$@" --no-roofline --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-steady-state --no-batch8 --no-n1000 --no-tiled --no-churn 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    best=$(python -c "print(max($best, $v))")
  done
  echo "$label  $best steps/s"
}
for spec in "1 200 2200 220" "2 200 880 110" "4 200 880 110" "8 200 880 110" "16 200 440 110" "64 200 440 110" "1 1000 220 55"; do
  set -- $spec
  run "B=$1 N=$2 round 6" "X=1" --filters-per-gpu $1 --landmarks $2 --steps $3 --warmup $4
  run "B=$1 N=$2 round 5" "EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_r05.so" --filters-per-gpu $1 --landmarks $2 --steps $3 --warmup $4
done
