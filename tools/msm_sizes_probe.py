"""MSM time and phases at several sizes, and Bp::prove, under the current environment knobs:
python tools/msm_sizes_probe.py 16 18 20 [ipa]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi  # noqa: E402
import bench  # noqa: E402

ffi.init(0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BPGPU_"))
for a in sys.argv[1:]:
    if a == "ipa":
        n = 1 << 20
        g = bench.synth_points(ffi, 0, 2 * n + 1, seed=77)
        aa, bb = bench.synth_scalars(n, 78), bench.synth_scalars(n, 79)
        gens = ark_bp.DeviceGens(0, g[:n], g[n:2 * n], g[2 * n:])
        ms = sorted(ark_bp.bp_prove_cached(gens, aa, bb, bytes(range(32)))["prove_ms"] for _ in range(5))
        print(f"[{tag}] prove 2^20 resident ms {ms[0]:.3f} median {ms[2]:.3f}", flush=True)
        gens.free()
        continue
    lg = int(a)
    n = 1 << lg
    pts, sc = ark_bp.msm_bench_inputs(0, n)
    d_p, d_s = ffi.points_to_device(0, pts), ffi.scalars_to_device(sc)
    for _ in range(3):
        ffi.msm_device(0, d_p, d_s, n)
    ph = np.zeros(8)
    reps = 20 if lg <= 20 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ffi.msm_device(0, d_p, d_s, n)
        ph += np.array(ffi.last_msm_timings())
    dt = (time.perf_counter() - t0) / reps
    ph /= reps
    print(f"[{tag}] 2^{lg}: {dt * 1e3:.4f} ms  sort {ph[0] + ph[1] + ph[2]:.4f} accum {ph[3]:.4f} reduce {ph[5]:.4f} host {ph[6]:.4f}", flush=True)
    d_p.free()
    d_s.free()
