#!/usr/bin/env python3
"""dev: Bp::prove time at 2^20 (generators resident) against the freeze length"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi, ark_bp
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ffi.init(0)
n = 1 << k
pts = bench.synth_points(ffi, 0, 2 * n + 1, seed=5)
A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
a, b = bench.synth_scalars(n, 6), bench.synth_scalars(n, 7)
for fl in [1 << 14, 1 << 14]:
    ffi.set_freeze_len(fl)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        s = ffi.IpaSession(0, A, B, q, a, b)
        rnd = 100
        while len(s) > 1:
            rnd += 1
            x = bench.synth_scalars(1, rnd)[0]
            xi = ark_bp.scalar_inverse(0, x)
            s.round_lr(); s.round_fold(x, xi)
        s.free()
        best = min(best, time.perf_counter() - t0)
    print(f"freeze_len 2^{fl.bit_length()-1}: session {best*1e3:.2f} ms (host generators in, challenges synthetic)")
