#!/bin/bash
# steps/s and the burst's kernel time per (filters, builder landmarks, rows per wave, two-steps-ahead) -- round 6: the 8-landmark builder and the
# two-row block kernel's second register set.   scripts/burst_shapes.sh "4 6 8 10 12" > profiles/r06_burst_shapes.txt
for B in ${1:-4 8 12}; do
  for opt in "burst_lm=0 burst_rows=0 ring_ahead2=1" "burst_lm=16 burst_rows=2 ring_ahead2=0" "burst_lm=16 burst_rows=2 ring_ahead2=1" "burst_lm=8 burst_rows=2 ring_ahead2=0" "burst_lm=8 burst_rows=2 ring_ahead2=1" "burst_lm=4 burst_rows=2 ring_ahead2=1" "burst_lm=8 burst_rows=1 ring_ahead2=1" "burst_lm=8 burst_rows=4 ring_ahead2=1"; do
    args=""; for o in $opt; do args="$args --debug-option $o"; done
    python bench.py --filters-per-gpu $B --steps 880 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-steady-state --no-batch8 --no-n1000 --no-tiled --no-churn $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={r['kernel']:r for r in d['kernels']}
b=k['k_imu_burst']
print('B=%2d  %-48s %8.0f steps/s  burst %6.1f us  update %6.1f us  shape lm=%d rows=%d fused=%d' % ($B, '$opt', d['value'], b['avg_us'], k.get('k_chol_resident',{}).get('avg_us',0), b['launch_shape']['builder_landmarks'], b['launch_shape']['rows_per_wave'], b['launch_shape']['fused']))"
  done
done
