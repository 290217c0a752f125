#!/bin/bash
# SQ counters of the IMU burst's two kernels at a batch size: what the block kernel's waves do with their cycles.
#   scripts/pmc_burst.sh 8 > profiles/r06_pmc_burst_B8.txt
B=${1:-8}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
for k in k_burst_riccati_ring k_burst_build; do
  echo "== $k, $B filters of N=200 (per launch means; SQ counters are summed over the chip's SEs as rocprofv3 reports them)"
  bash $ROOT/scripts/pmc_kernel.sh "$C" $k --filters-per-gpu $B "$@"
done
