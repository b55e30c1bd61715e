import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ffi, ark_bp
ffi.init(0)
curve, k, nk = 0, 8, 1 << 17
pts, sc = ark_bp.msm_bench_inputs(curve, 1 << 18)
A, B, q = pts[:nk], pts[nk:2*nk], bench.synth_points(ffi, curve, 3, seed=41)[2:]
a, b = sc[:nk], sc[nk:2*nk]
x = bench.synth_scalars(1, 31)[0]
y = ffi.powers(curve, x, 0, k)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("cross terms      %.3f ms" % t(lambda: ffi.kary_cross_terms(curve, A, B, q, a, b, k)), " kernel/op ms", ffi.last_op_ms())
print("fold_points_kary %.3f ms" % t(lambda: ffi.fold_points_kary(curve, A, k, y)), " op ms", ffi.last_op_ms())
print("fold_scalars_kary %.3f ms" % t(lambda: ffi.fold_scalars_kary(curve, a, k, y)))
print("commit           %.3f ms" % t(lambda: ark_bp.commit(curve, A, B, q, a, b)))
res = ark_bp.commit(curve, A, B, q, a, b)
seed = b"x".ljust(32, b"\0")
print("prove_step       %.3f ms" % t(lambda: ark_bp.kary_prove_step(curve, k, A, B, q, res, a, b, seed)))
for c in (12, 13, 14, 15, 16):
    ffi.set_window_bits(c)
    print("cross terms, window %d: %.3f ms" % (c, t(lambda: ffi.kary_cross_terms(curve, A, B, q, a, b, k))))
ffi.set_window_bits(0)
for kk, n2 in ((4, 1 << 17), (16, 1 << 17), (8, 1 << 15), (8, 1 << 18)):
    for c in (0, 12, 13, 14, 15, 16):
        ffi.set_window_bits(c)
        print("k %2d n 2^%d window %2d: %.3f ms" % (kk, n2.bit_length() - 1, c, t(lambda: ffi.kary_cross_terms(curve, pts[:n2], pts[-n2:], q, sc[:n2], sc[-n2:], kk))))
ffi.set_window_bits(0)
