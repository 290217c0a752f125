#!/usr/bin/env python
"""eqf_tile_gemm_tn (C += alpha A^T B, fp64 MFMA) at the shapes of the N = 4000 partitioned update: TFLOP/s by torch.cuda events.
   python scripts/gemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eqf_vio_amd import tiled  # noqa: E402

be = tiled.HipBackend({}, capacity=8)
dev = be.device
shapes = [(12000, 12000, 750, None, "downdate / E trailing, full"), (12000, 12000, 500, None, "downdate (k = 2 bl)"),
          (8000, 20018, 500, None, "S trailing incl. right-hand sides"), (12000, 12000, 750, (750, 750, 0, 1, 0, 0, 1, 0), "E trailing, upper mask"),
          (6000, 3000, 750, None, "a 2 x 4 rank's share of the downdate"), (18, 12018, 500, None, "reductions")]
for (m, n, k, mask, what) in shapes:
    A = torch.randn(k, m, dtype=torch.float64, device=dev)
    B = torch.randn(k, n, dtype=torch.float64, device=dev)
    C = torch.zeros(m, n, dtype=torch.float64, device=dev)
    be.gemm_tn(C, A, B, -1.0, mask)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps):
        be.gemm_tn(C, A, B, -1.0, mask)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    fl = 2.0 * m * n * k * (0.5 if mask else 1.0)
    print(f"{m:6d} x {n:6d} x {k:4d}  {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s ({fl / ms / 1e9 / 78.6 * 100:5.1f} % of the fp64 MFMA peak)  {what}")
    if mask is None and m >= 1000 and os.environ.get("EQF_GEMM_YARDSTICK", "1") != "0":
        # yardstick only (never on the product path): the vendor library's DGEMM on the same operands, C = C - A^T B
        C.addmm_(A.t(), B, alpha=-1.0)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            C.addmm_(A.t(), B, alpha=-1.0)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print(f"{'':27s}{ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s ({fl / ms / 1e9 / 78.6 * 100:5.1f} %)  yardstick: torch.addmm (rocBLAS / hipBLASLt DGEMM), same operands")
