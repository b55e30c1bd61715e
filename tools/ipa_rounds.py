#!/usr/bin/env python3
"""Per-round timing of the device-resident IPA session (dev tool)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi, ark_bp

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ffi.init(0)
if len(sys.argv) > 2:
    ffi.set_window_bits(int(sys.argv[2]))
n = 1 << k
pts = bench.synth_points(ffi, 0, 2 * n + 1, seed=5)
A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
a, b = bench.synth_scalars(n, 6), bench.synth_scalars(n, 7)
for rep in range(2):
    t0 = time.perf_counter(); s = ffi.IpaSession(0, A, B, q, a, b); print('begin %.3f ms' % (1e3*(time.perf_counter()-t0)))
    rnd = 100
    while len(s) > 1:
        m = len(s)
        rnd += 1
        x = bench.synth_scalars(1, rnd)[0]
        xi = ark_bp.scalar_inverse(0, x)
        t0 = time.perf_counter(); s.round_lr(); t1 = time.perf_counter(); tm = ffi.last_msm_timings(); s.round_fold(x, xi); t2 = time.perf_counter()
        if rep == 1:
            print(f"len {m:8d}  lr {1e3*(t1-t0):8.3f} ms  fold {1e3*(t2-t1):8.3f} ms   msm phases {[round(v,3) for v in tm]}")
    s.free()
