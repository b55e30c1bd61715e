#!/usr/bin/env python3
"""Times Bp::prove (C++ host side over the C ABI) at N = 2^k on synthetic inputs; dev tool."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi, ark_bp

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ffi.init(0)
if len(sys.argv) > 2:
    ffi.set_freeze_len(1 << int(sys.argv[2]))
if len(sys.argv) > 3:
    ffi.set_window_bits(int(sys.argv[3]))
n = 1 << k
t0 = time.perf_counter()
pts = bench.synth_points(ffi, 0, 2 * n + 1, seed=5)
print("inputs", time.perf_counter() - t0, flush=True)
A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
a, b = bench.synth_scalars(n, 6), bench.synth_scalars(n, 7)
seed = bytes(range(32))
for rep in range(4):
    t0 = time.perf_counter()
    pf = ark_bp.bp_prove(0, A, B, q, a, b, seed)
    print(f"N=2^{k} prove wall {1e3*(time.perf_counter()-t0):.1f} ms, inner {pf['prove_ms']:.1f} ms, rounds {len(pf['ls'])}", flush=True)
if k <= 16:
    res = ark_bp.commit(0, A, B, q, a, b)
    ark_bp.bp_verify(0, A, B, q, res, pf, seed)
    print("verified")
