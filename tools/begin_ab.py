"""A/B of bpgpu_ipa_begin from host vectors (dev tool): alternate an environment switch between sessions."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi
ffi.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
switch = sys.argv[2] if len(sys.argv) > 2 else "BPGPU_CONV_SAME_STREAM"
n = 1 << k
pts = bench.synth_points(ffi, 0, 2 * n + 1, seed=5)
A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
a, b = bench.synth_scalars(n, 6), bench.synth_scalars(n, 7)
res = {0: [], 1: []}
for rep in range(16):
    on = rep & 1
    if on: os.environ[switch] = "1"
    else: os.environ.pop(switch, None)
    t0 = time.perf_counter(); s = ffi.IpaSession(0, A, B, q, a, b); dt = 1e3 * (time.perf_counter() - t0)
    s.free()
    if rep >= 2: res[on].append(dt)
for on in (0, 1):
    v = sorted(res[on])
    print(f"{switch}={on}: begin min {v[0]:.3f}  median {v[len(v)//2]:.3f}  max {v[-1]:.3f} ms   {[round(x,2) for x in v]}")
