#!/usr/bin/env python3
"""MSM at the largest sizes the header admits (dev tool; tests/test_gpu_maxsize.py is the test form).

Points and scalars are grown ON THE DEVICE from a small seeded set (bpgpu_fold_points_device /
bpgpu_fold_scalars_device write new random-looking combinations behind the ones they read), so a 2^28-point
input costs seconds, not the minutes host sampling would.  Check: one call over all n points against the sum of
the 2^24-point sub-ranges (a size the parity suite pins against the oracles)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi


class View:
    """A sub-range of a DeviceBuffer (anything with .ptr works for the ffi wrappers)."""

    def __init__(self, base, off):
        self.ptr = base.ptr + off


def grow_inputs(curve, n, seed=11, seed_len=4096):
    lib = ffi.load()
    d_pts = ffi.DeviceBuffer(n * 64)
    d_sc = ffi.DeviceBuffer(n * 32)
    m0 = min(seed_len, n)
    p0 = bench.synth_points(ffi, curve, m0, seed)
    ffi._check(lib.bpgpu_points_to_device(curve, ffi._ptr(np.ascontiguousarray(p0)), m0, 0, d_pts.ptr))
    d_sc.upload(bench.synth_scalars(m0, seed + 1))
    have, step = m0, 0
    while have < n:
        m = min(have // 2, n - have)
        a, b = bench.synth_scalars(2, 1000 + step)
        ffi.fold_points_device(curve, d_pts, m, a, b, View(d_pts, have * 64))
        ffi.fold_scalars_device(curve, d_sc, m, a, b, View(d_sc, have * 32))
        have += m
        step += 1
    return d_pts, d_sc


def main():
    ffi.init(0)
    curve = 0
    for lg in [int(a) for a in sys.argv[1:]] or [26]:
        n = 1 << lg
        t0 = time.perf_counter()
        d_pts, d_sc = grow_inputs(curve, n)
        t1 = time.perf_counter()
        full = ffi.msm_device(curve, d_pts, d_sc, n)
        t2 = time.perf_counter()
        full2 = ffi.msm_device(curve, d_pts, d_sc, n)
        t3 = time.perf_counter()
        phases = [round(v, 2) for v in ffi.last_msm_timings()]
        piece = min(n, 1 << 24)
        parts = [ffi.msm_device(curve, View(d_pts, k * piece * 64), View(d_sc, k * piece * 32), piece) for k in range(n // piece)]
        total = ffi.points_sum(curve, np.stack(parts))
        ok = bool((full == total).all() and (full2 == total).all())
        print(f"2^{lg}: inputs {t1 - t0:.1f} s, msm {1e3 * (t2 - t1):.1f} ms / {1e3 * (t3 - t2):.1f} ms "
              f"({n / (t3 - t2) / 1e6:.0f} M points/s), phases {phases}, "
              f"equals the sum of {n // piece} x 2^{min(lg, 24)}: {ok}", flush=True)
        d_pts.free()
        d_sc.free()
        if not ok:
            sys.exit(1)


if __name__ == "__main__":
    main()
