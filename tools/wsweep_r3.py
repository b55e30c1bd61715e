"""Window sweep at small / middle sizes: python tools/wsweep_r3.py 14 16 18"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi  # noqa: E402
ffi.init(0)
for a in sys.argv[1:]:
    lg = int(a); n = 1 << lg
    pts, sc = ark_bp.msm_bench_inputs(0, n)
    d_p, d_s = ffi.points_to_device(0, pts), ffi.scalars_to_device(sc)
    for c in range(max(8, lg - 5), min(20, lg + 1) + 1):
        ffi.set_window_bits(c)
        for _ in range(3): ffi.msm_device(0, d_p, d_s, n)
        ph = np.zeros(8); reps = 15
        t0 = time.perf_counter()
        for _ in range(reps):
            ffi.msm_device(0, d_p, d_s, n); ph += np.array(ffi.last_msm_timings())
        dt = (time.perf_counter() - t0) / reps; ph /= reps
        print(f"2^{lg} c={c}: {dt*1e3:.4f} ms sort {ph[0]+ph[1]+ph[2]:.4f} accum {ph[3]:.4f} reduce {ph[5]:.4f} host {ph[6]:.4f}", flush=True)
    ffi.set_window_bits(0)
    d_p.free(); d_s.free()
