#!/usr/bin/env python3
"""dev: wall time and GPU phases of bpgpu_msm with host buffers (the PCIe-inclusive path)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ffi.init(0)
n = 1 << k
pts = bench.synth_points(ffi, 0, n, seed=5)
sc = bench.synth_scalars(n, 6)
ffi.msm(0, pts, sc)
for _ in range(6):
    t = time.perf_counter()
    ffi.msm(0, pts, sc)
    dt = time.perf_counter() - t
    print(f"2^{k} host-to-host {dt*1e3:.3f} ms  phases {[round(v, 3) for v in ffi.last_msm_timings()]}")
