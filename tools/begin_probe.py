#!/usr/bin/env python3
"""Dev probe: cost of opening / closing an IPA session at N = 2^k (host vectors and device-resident generators)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi, ark_bp
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ffi.init(0)
n = 1 << k
pts = bench.synth_points(ffi, 0, 2 * n + 1, seed=5)
A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
a, b = bench.synth_scalars(n, 6), bench.synth_scalars(n, 7)
for rep in range(4):
    t0 = time.perf_counter(); s = ffi.IpaSession(0, A, B, q, a, b); t1 = time.perf_counter(); s.free(); t2 = time.perf_counter()
    print(f"host vectors: begin {1e3*(t1-t0):.3f} ms  free {1e3*(t2-t1):.3f} ms")
import ctypes
lib = ffi.load()
t0 = time.perf_counter()
for _ in range(5):
    p = ctypes.c_void_p(); lib.bpgpu_dev_alloc(ctypes.c_size_t(335 << 20), ctypes.byref(p)); lib.bpgpu_dev_free(p)
print(f"dev_alloc+free of 335 MB: {1e3*(time.perf_counter()-t0)/5:.3f} ms")
