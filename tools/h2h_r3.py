"""Host-to-host MSM (examples/msm_bench.rs:45-48 window) and Bp::prove from host vectors, for a given number of
staging threads (BPGPU_STAGING_THREADS): python tools/h2h_r3.py [log2n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi  # noqa: E402
import bench  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ffi.init(0)
pts, sc = ark_bp.msm_bench_inputs(0, 1 << lg)
for _ in range(3):
    ffi.msm(0, pts, sc)
ts = []
for _ in range(9):
    t = time.perf_counter()
    ffi.msm(0, pts, sc)
    ts.append(time.perf_counter() - t)
print("threads", os.environ.get("BPGPU_STAGING_THREADS"), "h2h ms median", sorted(ts)[4] * 1e3, "min", min(ts) * 1e3, flush=True)
aff = np.ascontiguousarray(pts[:, :8])
ts = []
for _ in range(7):
    t = time.perf_counter()
    ffi.msm_affine(0, aff, sc)
    ts.append(time.perf_counter() - t)
print("  affine input: median", sorted(ts)[3] * 1e3, flush=True)
if len(sys.argv) > 2:
    n = 1 << lg
    g = bench.synth_points(ffi, 0, 2 * n + 1, seed=77)
    a, b = bench.synth_scalars(n, 78), bench.synth_scalars(n, 79)
    ms = [ark_bp.bp_prove(0, g[:n], g[n:2 * n], g[2 * n:], a, b, bytes(range(32)))["prove_ms"] for _ in range(5)]
    print("  prove ms", sorted(ms), flush=True)
