#!/usr/bin/env python3
"""Dev probe: accumulate-phase time of a 2^k-point MSM over several scalar seeds and window widths."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi
k = int(sys.argv[1]); ffi.init(0)
n = 1 << k
pts = bench.synth_points(ffi, 0, n, seed=5)
d_pts = ffi.points_to_device(0, pts, fmt=0)
for c in [int(v) for v in sys.argv[2:]]:
    ffi.set_window_bits(c)
    row = []
    for seed in range(8):
        sc = bench.synth_scalars(n, 100 + seed)
        d_sc = ffi.scalars_to_device(sc)
        for _ in range(3):
            ffi.msm_device(0, d_pts, d_sc, n)
        t = ffi.last_msm_timings()
        row.append((round(t[3], 3), round(t[7], 3)))
        d_sc.free()
    print("c", c, row, flush=True)
