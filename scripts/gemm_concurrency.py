#!/usr/bin/env python
"""Do two of the update's products running side by side on two streams reach the rate of one alone?  (round 6; GPU box.)  eqf_tile_gemm_tn at the
S-chain's and the E-chain's trailing shapes: each alone, then both at once on two streams, aggregate TFLOP/s.   python scripts/gemm_concurrency.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eqf_vio_amd import tiled  # noqa: E402

be = tiled.HipBackend({}, capacity=8)
dev = be.device


def mk(m, n, k):
    return (torch.zeros(m, n, dtype=torch.float64, device=dev), torch.randn(k, m, dtype=torch.float64, device=dev), torch.randn(k, n, dtype=torch.float64, device=dev))


shapes = {"S trailing 8000 x 20018 x 500": mk(8000, 20018, 500), "E trailing 12000 x 12000 x 750": mk(12000, 12000, 750)}
flops = {k: 2.0 * v[0].shape[0] * v[0].shape[1] * v[1].shape[0] for k, v in shapes.items()}
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
reps = 6


def run(jobs):
    """jobs: list of (stream, name); every job `reps` products on its stream; wall time from one start event to the last end"""
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    ends = []
    a.record()
    for st, name in jobs:
        st.wait_event(a)
    for _ in range(reps):
        for st, name in jobs:
            with torch.cuda.stream(st):
                C, A, B = shapes[name]
                be.gemm_tn(C, A, B, -1.0)
    for st, name in jobs:
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        ends.append(e)
    torch.cuda.synchronize()
    return max(a.elapsed_time(e) for e in ends)


names = list(shapes)
for n in names:
    run([(s1, n)])
    ms = run([(s1, n)])
    print(f"alone      {n:34s} {ms / reps:7.3f} ms per product  {flops[n] * reps / ms / 1e9:6.1f} TFLOP/s")
run([(s1, names[0]), (s2, names[1])])
ms = run([(s1, names[0]), (s2, names[1])])
tot = (flops[names[0]] + flops[names[1]]) * reps
print(f"both at once on two streams: {ms / reps:7.3f} ms per pair  {tot / ms / 1e9:6.1f} TFLOP/s aggregate")
ms = run([(s1, names[0]), (s2, names[0])])
print(f"two of the first at once:    {ms / reps:7.3f} ms per pair  {2 * flops[names[0]] * reps / ms / 1e9:6.1f} TFLOP/s aggregate")
