#!/usr/bin/env python
"""The cfg 5 leg of bench.py on its own (one N-landmark filter, Sigma 2-D block-partitioned, 1 x 1 grid on one GPU): for rocprofv3
--kernel-trace --stats runs and for sweeps over the block size.   python scripts/tiled_bench.py [N] [block] [frames] [bench.py flags, e.g. --no-i8-downdate]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import bench  # noqa: E402

if __name__ == "__main__":
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    bl = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    sys.argv = [sys.argv[0], "--tiled-landmarks", str(N), "--tiled-block", str(bl), "--tiled-frames", str(frames)] + sys.argv[4:]  # (+ bench.py flags)
    args = bench.parse()
    out = bench.tiled_leg(args, None, 0, 1, 0)
    print(json.dumps(out))
