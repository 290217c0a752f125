#!/usr/bin/env python
"""Does a high-priority stream get the CUs first?  (round 6; GPU box.)  Stream A keeps the chip full with the update's product (eqf_tile_gemm_tn,
12000 x 12000 x 750, back to back); stream B runs ONE such product, at normal and at high priority; B's duration alone and next to A.
    python scripts/stream_priority_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eqf_vio_amd import tiled  # noqa: E402

be = tiled.HipBackend({}, capacity=8)
dev = be.device
m, k = 12000, 750
A1, A2 = (torch.randn(k, m, dtype=torch.float64, device=dev) for _ in range(2))
C1, C2 = (torch.zeros(m, m, dtype=torch.float64, device=dev) for _ in range(2))
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)


def b_time(prio_b, prio_a, with_a):
    sa = torch.cuda.Stream(device=dev, priority=prio_a)
    sb = torch.cuda.Stream(device=dev, priority=prio_b)
    with torch.cuda.stream(sb):
        be.gemm_tn(C2, A2, A2, -1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_a:
        with torch.cuda.stream(sa):
            for _ in range(12):
                be.gemm_tn(C1, A1, A1, -1.0)
    import time
    time.sleep(0.01)  # (A is well under way: twelve products take ~45 ms)
    with torch.cuda.stream(sb):
        e0.record(sb)
        be.gemm_tn(C2, A2, A2, -1.0)
        e1.record(sb)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


print(f"B alone:                                   {b_time(0, 0, False):7.3f} ms")
for pb, pa, what in ((0, 0, "B normal, A normal"), (-1, 0, "B high,   A normal"), (-1, 1, "B high,   A low"), (0, 1, "B normal, A low")):
    try:
        ts = [b_time(pb, pa, True) for _ in range(3)]
        print(f"B next to A ({what}):          " + "  ".join(f"{t:7.3f}" for t in ts) + " ms")
    except Exception as e:
        print(what, "failed:", e)
