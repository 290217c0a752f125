#!/bin/bash
# (round 6, GPU box) the launch-shape knobs that have a name (eqf_debug_option) at 3 .. 64 filters of N = 200, best of three each: are the size
# heuristics still where the optimum is behind this round's launches?  (They are, within 1 %: profiles/r06_knob_sweep.txt.  The one that was
# not -- the prep roles inside the update launch at 5 and 6 filters -- has no name: EQF_RES_FOLD_PREP, profiles/r06_fold_batch.txt.)
run() { # run <label> <bench args...>
  local label=$1; shift
  best=0
  for i in 1 2 3; do
    v=$(python bench.py "$@" --no-roofline --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-steady-state --no-batch8 --no-n1000 --no-tiled --no-churn 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    best=$(python -c "print(max($best, $v))")
  done
  echo "$label  $best steps/s"
}
for cs in 0 1 2; do run "B=8 cs_in_burst=$cs" --filters-per-gpu 8 --steps 880 --warmup 110 --debug-option cs_in_burst=$cs; done
for cs in 0 1 2; do run "B=12 cs_in_burst=$cs" --filters-per-gpu 12 --steps 440 --warmup 110 --debug-option cs_in_burst=$cs; done
for r in 2 4; do run "B=12 burst_rows=$r" --filters-per-gpu 12 --steps 440 --warmup 110 --debug-option burst_rows=$r; done
for r in 2 4; do run "B=16 burst_rows=$r" --filters-per-gpu 16 --steps 440 --warmup 110 --debug-option burst_rows=$r; done
for cs in 0 1; do run "B=64 cs_in_burst=$cs" --filters-per-gpu 64 --steps 440 --warmup 110 --debug-option cs_in_burst=$cs; done
for lm in 8 16; do run "B=3 burst_lm=$lm" --filters-per-gpu 3 --steps 880 --warmup 110 --debug-option burst_lm=$lm; done
for r in 1 2; do run "B=3 burst_rows=$r" --filters-per-gpu 3 --steps 880 --warmup 110 --debug-option burst_rows=$r; done
for r in 1 2; do run "B=4 burst_rows=$r" --filters-per-gpu 4 --steps 880 --warmup 110 --debug-option burst_rows=$r; done
