"""Kernel times of the 'next' rows: python tools/next_rows_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi  # noqa: E402
import bench  # noqa: E402

ffi.init(0)
n = 1 << 18
pts, sc = ark_bp.msm_bench_inputs(0, n)
for _ in range(3):
    t = time.perf_counter()
    out = ffi.hadamard_points(0, pts, sc)
    wall = time.perf_counter() - t
    print("hadamard_points 2^18: kernel ms", ffi.last_op_ms(), "wall ms", wall * 1e3, flush=True)
