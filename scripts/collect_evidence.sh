#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects the round's measurement evidence into gpurun_out/evidence/.
# Usage: scripts/collect_evidence.sh <round tag, e.g. r01>
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python

stats() {  # stats <name> <bench args...>: rocprofv3 kernel-trace summary of one bench invocation
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o b -- $PY $ROOT/bench.py "$@" --no-cpu-baseline --no-traffic --no-roofline --no-batch64 --no-parity --no-steady-state --no-n1000 --no-batch8 --no-tiled --no-churn > /dev/null 2>&1
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${name}_kernel_stats.csv"
}
# (pmc children stop right after the main timed job: no extra legs)
pmc() {  # pmc <name> <counter> <bench args...>: per-kernel mean of one PMC counter (FETCH/WRITE_SIZE: KiB), its own pass
  local name=$1 counter=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc $counter --kernel-trace --output-format csv -d /tmp/pmc_$name -o b -- $PY $ROOT/bench.py "$@" --pmc-child > /dev/null 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && $PY - "$f" "$OUT/${TAG}_${name}_pmc_${counter}_summary.csv" <<'PYEOF'
import csv, statistics, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Calls", "Mean", "Median", "Max"])  # FETCH_SIZE / WRITE_SIZE: KiB per launch; cycle counters: summed over the chip
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(statistics.mean(v), 2), round(statistics.median(v), 2), round(max(v), 2)])
PYEOF
}

# 1. the headline line (configs[1]: one filter, N = 200), with roofline, PMC traffic and the CPU baseline
timeout 1200 $PY $ROOT/bench.py > "$OUT/${TAG}_bench_N200.json" 2> "$OUT/${TAG}_bench_N200.stderr"
# 1' the driver's own invocation shape: 20 timed steps after 5 of warm-up
timeout 1200 $PY $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_bench_driver_shape.json" 2>/dev/null
stats bench_N200
pmc bench_N200 FETCH_SIZE --steps 220 --warmup 110
pmc bench_N200 WRITE_SIZE --steps 220 --warmup 110
# 1a. MFMA-busy next to GPU-active cycles at N = 200 (separate passes)
pmc bench_N200 SQ_VALU_MFMA_BUSY_CYCLES --steps 220 --warmup 110
pmc bench_N200 GRBM_GUI_ACTIVE --steps 220 --warmup 110
# 1b. the same workload with the per-column launches instead of the resident update kernel, and with one launch per IMU call
EQF_CHOL_RESIDENT=0 timeout 900 $PY $ROOT/bench.py --no-cpu-baseline --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N200_launches.json" 2>/dev/null
# 1c. the same with the prep launch in front of the update launch (round 3's shape) instead of the prep roles inside it
EQF_RES_FOLD_PREP=0 timeout 900 $PY $ROOT/bench.py --no-cpu-baseline --no-batch64 --no-parity --no-traffic --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N200_preplaunch.json" 2>/dev/null
EQF_RES_FOLD_PREP=0 stats bench_N200_preplaunch
EQF_CHOL_RESIDENT=0 stats bench_N200_launches
EQF_IMU_BURST=0 timeout 900 $PY $ROOT/bench.py --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N200_noburst.json" 2>/dev/null
# 2. a batch of 64 filters on one GPU (cfg 4's filters, all on one device)
timeout 900 $PY $ROOT/bench.py --filters-per-gpu 64 --steps 440 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_batch64.json" 2>/dev/null
stats bench_N200_batch64 --filters-per-gpu 64 --steps 220 --warmup 110
timeout 900 $PY $ROOT/bench.py --filters-per-gpu 8 --steps 880 --warmup 110 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_batch8.json" 2>/dev/null
# 2b. small batches: the resident update kernel on an interleaved grid larger than the chip (default) against the per-column launches
( echo "# bench.py --filters-per-gpu B --steps 440 --warmup 110 (N = 200): steps/s, update kernels (avg us per launch)"
  for B in 2 4 6 8 12 16; do for O in default 0; do
    if [ $O = default ]; then envs="X=1"; else envs="EQF_CHOL_RESIDENT=0"; fi
    env $envs timeout 600 $PY $ROOT/bench.py --filters-per-gpu $B --steps 440 --warmup 110 --no-batch64 --no-cpu-baseline --no-traffic --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 2>/dev/null | $PY -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B EQF_CHOL_RESIDENT=$O', round(d['value']), 'steps/s  err', d['device_error_flag'], [(k['kernel'], k['avg_us']) for k in d['kernels'][:4]])"
  done; done ) > "$OUT/${TAG}_batch_sweep.txt" 2>&1
# 3. N = 1000: structured kernel and the dense MFMA Riccati backend (cfg 3)
timeout 900 $PY $ROOT/bench.py --landmarks 1000 --steps 220 --warmup 55 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N1000.json" 2>/dev/null
timeout 900 $PY $ROOT/bench.py --landmarks 1000 --steps 110 --warmup 22 --dense-propagate --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N1000_dense.json" 2>/dev/null
# 3b. MFMA busy cycles next to the GPU-active cycles for the N = 1000 runs (structured path and dense Riccati), separate passes
pmc bench_N1000 SQ_VALU_MFMA_BUSY_CYCLES --landmarks 1000 --steps 44 --warmup 11
pmc bench_N1000 GRBM_GUI_ACTIVE --landmarks 1000 --steps 44 --warmup 11
pmc bench_N1000_dense SQ_VALU_MFMA_BUSY_CYCLES --landmarks 1000 --steps 22 --warmup 11 --dense-propagate
pmc bench_N1000_dense GRBM_GUI_ACTIVE --landmarks 1000 --steps 22 --warmup 11 --dense-propagate
# 4. N = 4000 (Sigma = 1.15 GB)
timeout 900 $PY $ROOT/bench.py --landmarks 4000 --steps 22 --warmup 11 --no-cpu-baseline --no-traffic --no-batch64 --no-parity --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N4000.json" 2>/dev/null
# 4b. MFMA busy next to GPU-active cycles for the 64-filter batch and for N = 4000 (separate passes)
pmc bench_batch64 SQ_VALU_MFMA_BUSY_CYCLES --filters-per-gpu 64 --steps 44 --warmup 22
pmc bench_batch64 GRBM_GUI_ACTIVE --filters-per-gpu 64 --steps 44 --warmup 22
pmc bench_N4000 SQ_VALU_MFMA_BUSY_CYCLES --landmarks 4000 --steps 22 --warmup 11
pmc bench_N4000 GRBM_GUI_ACTIVE --landmarks 4000 --steps 22 --warmup 11
# 4c. BASELINE configs[4] on one GPU: the 2-D partitioned filter (1 x 1 grid) at N = 4000 next to the monolithic path; kernel summary,
# timeline of one update, the fp64 GEMM on its own
timeout 900 $PY $ROOT/scripts/tiled_bench.py 4000 250 3 > "$OUT/${TAG}_bench_tiled_N4000.json" 2>/dev/null
rm -rf /tmp/prof_tiled
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tiled -o b -- $PY $ROOT/scripts/tiled_bench.py 4000 250 1 > /dev/null 2>&1
f=$(find /tmp/prof_tiled -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_tiled_N4000_kernel_stats.csv"
f=$(find /tmp/prof_tiled -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && $PY $ROOT/scripts/tiled_timeline.py "$f" 400 > "$OUT/${TAG}_tiled_N4000_timeline.txt"
timeout 300 $PY $ROOT/scripts/gemm_bench.py > "$OUT/${TAG}_gemm_tn.txt" 2>/dev/null
# 4d. per-launch timeline of a frame at 8 and 64 filters per GPU
timeout 900 bash $ROOT/scripts/launch_timeline.sh 8 > "$OUT/${TAG}_timeline_batch8.txt" 2>/dev/null
timeout 900 bash $ROOT/scripts/launch_timeline.sh 64 > "$OUT/${TAG}_timeline_batch64.txt" 2>/dev/null
# 5. long-run parity against the C++ oracle
( echo "# scripts/dev_compare.py 200 10.0 on MI355X: HIP path (fp64, per-call C ABI) vs oracle/eqf_oracle.cpp, same synthetic stream"
  echo "# (2000 IMU + 200 vision events, N = 200, template settings); relS = |Sigma_gpu - Sigma_ref|_F / |Sigma_ref|_F after the event"
  cd $ROOT && timeout 1200 $PY scripts/dev_compare.py 200 10.0 | grep -E "vision|worst|eqf_vio" | awk 'NR%20==1 || /worst/' ) > "$OUT/${TAG}_parity_N200_10s.txt" 2>&1
# 6. parity at the large sizes against the structured oracle, with the kernels that are the default at those sizes THIS round
# (k_chol_resident in its two-per-CU build; the N = 4000 oracle needs minutes of one core per frame)
( echo "# scripts/dev_compare.py N seconds 0 structured on MI355X: HIP path (fp64, per-call C ABI, IMU bursts, default kernels of this round:"
  echo "# ONE update launch, k_chol_resident<double, PIPEH, OCC2>) vs oracle/eqf_oracle.cpp structured backend, same synthetic stream"
  for spec in "1000 0.3" "2000 0.16"; do  # (N = 4000: tests/golden/large_N4000.npz, asserted by the GPU tests)
    set -- $spec
    echo "# N = $1, $2 s"
    cd $ROOT && timeout 1500 $PY scripts/dev_compare.py $1 $2 0 structured | grep -E "vision|worst"
  done ) > "$OUT/${TAG}_parity_large_N.txt" 2>&1
# 7. the chain of ONE filter of N = 200 from the inside: wall-clock stamps of every row head (instrumented build, if it was shipped)
if [ -f $ROOT/build_variants/libeqf_stamps.so ]; then
  EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_stamps.so timeout 300 $PY $ROOT/scripts/res_stamps.py 200 1 > "$OUT/${TAG}_res_stamps_N200.txt" 2>&1
fi
# 7b. an IMU burst from the inside: builder workgroup 0 and block workgroup 9 of the fused launch on one clock (instrumented build, if shipped),
#     and the bench line with the burst as two launches
if [ -f $ROOT/build_variants/libeqf_bstamps.so ]; then
  { echo "## k_burst_fused (one launch)"; EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_bstamps.so timeout 300 $PY $ROOT/scripts/burst_fused_stamps.py;
    echo; echo "## EQF_BURST_FUSED=0: k_burst_build, then k_burst_riccati_ring (block workgroup stamps: math done | handed on | barrier passed)";
    EQF_BURST_FUSED=0 EQF_VIO_AMD_LIB=$ROOT/build_variants/libeqf_bstamps.so timeout 300 $PY $ROOT/scripts/burst_fused_stamps.py; } > "$OUT/${TAG}_burst_stamps_N200.txt" 2>&1
fi
EQF_BURST_FUSED=0 timeout 900 $PY $ROOT/bench.py --no-cpu-baseline --no-batch64 --no-parity --no-traffic --no-tiled --no-churn --no-steady-state --no-n1000 --no-batch8 > "$OUT/${TAG}_bench_N200_burst_two_launches.json" 2>/dev/null
# 7c. landmark churn and the outlier gate on the per-call API: which launches a frame consists of in each mode
timeout 600 bash $ROOT/scripts/churn_profile.sh > "$OUT/${TAG}_churn_profile.txt" 2>&1
# 8. the partitioned filter's host loop from plain C++: host cost of an update, frame time (eqf_example_tiled)
( for spec in "4000 6 250" "1000 9 125" "200 15 64"; do $ROOT/eqf_vio_amd/cpp/eqf_example_tiled $spec host | tail -1; done ) > "$OUT/${TAG}_tiled_host_cost.txt" 2>&1
ls -la "$OUT"
