"""bpgpu_fold_points_device (two arbitrary multipliers, a6) at several sizes: kernel time (dev tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ffi  # noqa: E402
import bench  # noqa: E402
ffi.init(0)
curve = 0
base = bench.synth_points(ffi, curve, 1 << 16, seed=3)
s_lo, s_hi = bench.synth_scalars(2, 77)
for lg in (12, 14, 16, 18, 19):
    n = 1 << lg
    reps = -(-2 * n // len(base))
    d_g = ffi.points_to_device(curve, np.concatenate([base] * reps)[:2 * n])
    d_o = ffi.DeviceBuffer(n * 64)
    best = 1e9
    for _ in range(3):
        ffi.fold_points_device(curve, d_g, n, s_lo, s_hi, d_o)
        best = min(best, ffi.last_op_ms())
    print(f"n_out 2^{lg}: {best:8.3f} ms  ops {ffi.last_fold_ops()}  digest {int(np.bitwise_xor.reduce(d_o.download(np.uint64))):016x}", flush=True)
    d_g.free(); d_o.free()
