"""k-ary generator fold (bpgpu_fold_points_kary_device) at several (ck, k): kernel time of the whole fold and the group
operations it executed.  BPGPU_NO_FOLD_MULTI=1 in a build that reads it selects the one-multiplier-at-a-time schedule.
    python tools/kary_fold_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ffi  # noqa: E402
import bench  # noqa: E402

ffi.init(0)
curve = 0
base = bench.synth_points(ffi, curve, 1 << 16, seed=3)
x = bench.synth_scalars(1, 31)[0]
for ck, k in [(1 << 18, 4), (1 << 19, 4), (1 << 18, 2), (1 << 14, 8), (1 << 15, 8), (1 << 16, 8), (1 << 16, 4), (1 << 16, 3), (1 << 16, 16), (1 << 12, 8), (1 << 18, 8), (1 << 20, 8), (1 << 20, 3)]:
    y = ffi.powers(curve, x, 0, k)
    reps = -(-ck * k // len(base))
    d_g = ffi.points_to_device(curve, np.concatenate([base] * reps)[:ck * k])
    d_o = ffi.DeviceBuffer(ck * 64)
    best = 1e9
    for _ in range(3):
        ffi.fold_points_kary_device(curve, d_g, ck, k, y, d_o)
        best = min(best, ffi.last_op_ms())
    outputs, dbls, adds, kind = ffi.last_fold_ops()
    digest = int(np.bitwise_xor.reduce(d_o.download(np.uint64)))
    print(f"ck 2^{ck.bit_length() - 1} k {k:2d}: {best:8.3f} ms   kind {kind}  doublings/output {dbls / max(outputs, 1):6.1f}  "
          f"additions/output {adds / max(outputs, 1):6.1f}  xor of the output words {digest:016x}", flush=True)
    d_g.free()
    d_o.free()
