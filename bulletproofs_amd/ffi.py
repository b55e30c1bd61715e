"""ctypes binding of the C ABI in include/bpgpu.h (libbpgpu.so, built in-tree by `make`).

Plumbing only: numpy uint64 arrays in ark-ff memory layout go in, numpy arrays come out.
There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int, c_size_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbpgpu.so")

PALLAS, VESTA = 0, 1

# every symbol include/bpgpu.h declares: name -> (restype, argtypes)
_U64P = POINTER(c_uint64)
SIGNATURES = {
    "bpgpu_init": (c_int, [c_int]),
    "bpgpu_shutdown": (None, []),
    "bpgpu_last_error": (ctypes.c_char_p, []),
    "bpgpu_set_window_bits": (c_int, [c_int]),
    "bpgpu_set_freeze_len": (c_int, [c_size_t]),
    "bpgpu_set_pair_fold": (c_int, [c_int]),
    "bpgpu_set_ipa_table": (c_int, [c_size_t]),
    "bpgpu_reset_tuning": (None, []),
    "bpgpu_get_tuning": (c_int, [POINTER(c_size_t)]),
    "bpgpu_msm": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_msm_affine": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_ip": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_fold_scalars": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bpgpu_fold_points": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bpgpu_dev_alloc": (c_int, [c_size_t, POINTER(c_void_p)]),
    "bpgpu_dev_free": (c_int, [c_void_p]),
    "bpgpu_dev_upload": (c_int, [c_void_p, c_void_p, c_size_t]),
    "bpgpu_dev_download": (c_int, [c_void_p, c_void_p, c_size_t]),
    "bpgpu_points_to_device": (c_int, [c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "bpgpu_points_from_device": (c_int, [c_int, c_void_p, c_size_t, c_void_p]),
    "bpgpu_msm_device": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_ip_device": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_fold_scalars_device": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bpgpu_fold_points_device": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bpgpu_fold_scalars_kary": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "bpgpu_fold_points_kary": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "bpgpu_fold_scalars_kary_device": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "bpgpu_fold_points_kary_device": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "bpgpu_kary_cross_terms": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p,
                                       c_void_p]),
    "bpgpu_hadamard": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_hadamard_points": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_powers": (c_int, [c_int, c_void_p, c_uint64, c_size_t, c_void_p]),
    "bpgpu_mat_vec_mult": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "bpgpu_vec_mat_mult": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "bpgpu_curve_lift_x": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "bpgpu_ipa_begin": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_void_p)]),
    "bpgpu_ipa_begin_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_void_p)]),
    "bpgpu_ipa_verify_msm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_verify_msm_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_len": (c_size_t, [c_void_p]),
    "bpgpu_ipa_round_lr": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_round_fold": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_finish": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bpgpu_ipa_free": (None, [c_void_p]),
    "bpgpu_points_sum": (c_int, [c_int, c_void_p, c_size_t, c_void_p]),
    "bpgpu_ip_device_noalloc": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_sharded_init": (c_int, [c_int, c_void_p]),
    "bpgpu_sharded_count": (c_int, []),
    "bpgpu_msm_sharded": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_msm_sharded_affine": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bpgpu_sharded_gens_load": (c_int, [c_int, c_void_p, c_size_t, c_int, POINTER(c_void_p)]),
    "bpgpu_sharded_gens_msm": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpgpu_sharded_gens_free": (None, [c_void_p]),
    "bpgpu_last_msm_timings": (c_int, [POINTER(c_float)]),
    "bpgpu_last_msm_entries": (c_int, [POINTER(c_uint64)]),
    "bpgpu_last_op_ms": (c_int, [POINTER(c_float)]),
    "bpgpu_last_fold_ops": (c_int, [POINTER(c_uint64)]),
    "bpgpu_measure_mad_peak": (c_int, [POINTER(c_double)]),
}


class BpGpuError(RuntimeError):
    pass


_lib = None


def load():
    """Load libbpgpu.so and declare its prototypes.  Needs no GPU (bpgpu_init does)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BpGpuError(f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise BpGpuError(f"bpgpu error {rc}: {load().bpgpu_last_error().decode()}")


def _ptr(a):
    return None if a is None else a.ctypes.data


def _u64(a, words=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if words is not None:
        a = a.reshape(-1, words)
    return a


def init(device: int = 0):
    _check(load().bpgpu_init(device))


def shutdown():
    load().bpgpu_shutdown()


def set_window_bits(c: int):
    _check(load().bpgpu_set_window_bits(c))


def set_pair_fold(enable: bool):
    _check(load().bpgpu_set_pair_fold(1 if enable else 0))


def set_ipa_table(min_len: int):
    """0 = never build the short rounds' fixed-base table, 1 = default threshold, n = build from n frozen generators."""
    _check(load().bpgpu_set_ipa_table(int(min_len)))


def set_freeze_len(n: int):
    _check(load().bpgpu_set_freeze_len(n))


def reset_tuning():
    """All four process-wide knobs back to the shipped defaults."""
    load().bpgpu_reset_tuning()


def get_tuning():
    """{window_bits, freeze_len, pair_fold, ipa_table_min} as the library holds them now."""
    out = (c_size_t * 4)()
    _check(load().bpgpu_get_tuning(out))
    return {"window_bits": int(out[0]), "freeze_len": int(out[1]), "pair_fold": bool(out[2]), "ipa_table_min": int(out[3])}


def msm(curve, points_xyz, scalars):
    """points_xyz: (n, 12) u64 ark jacobian; scalars: (n, 4) u64 ark form -> (9,) u64 xy_inf"""
    p, s = _u64(points_xyz, 12), _u64(scalars, 4)
    n = min(len(p), len(s))  # the reference zips (msm.rs:94-97)
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_msm(curve, _ptr(p), _ptr(s), n, _ptr(out)))
    return out


def msm_affine(curve, points_xy, scalars):
    p, s = _u64(points_xy, 8), _u64(scalars, 4)
    n = min(len(p), len(s))
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_msm_affine(curve, _ptr(p), _ptr(s), n, _ptr(out)))
    return out


def ip(curve, a, b):
    a, b = _u64(a, 4), _u64(b, 4)
    if len(a) != len(b):
        raise AssertionError("ip: length mismatch")  # reference asserts (util/mod.rs:18)
    out = np.zeros(4, dtype=np.uint64)
    _check(load().bpgpu_ip(curve, _ptr(a), _ptr(b), len(a), _ptr(out)))
    return out


def fold_scalars(curve, v, s_lo, s_hi):
    v = _u64(v, 4)
    assert len(v) % 2 == 0
    n = len(v) // 2
    out = np.zeros((n, 4), dtype=np.uint64)
    _check(load().bpgpu_fold_scalars(curve, _ptr(v), n, _ptr(_u64(s_lo)), _ptr(_u64(s_hi)), _ptr(out)))
    return out


def fold_points(curve, g_xyz, s_lo, s_hi):
    g = _u64(g_xyz, 12)
    assert len(g) % 2 == 0
    n = len(g) // 2
    out = np.zeros((n, 12), dtype=np.uint64)
    _check(load().bpgpu_fold_points(curve, _ptr(g), n, _ptr(_u64(s_lo)), _ptr(_u64(s_hi)), _ptr(out)))
    return out


def fold_scalars_kary(curve, v, k, y):
    """sum_vecs(chunks.zip(y).map(scale_vec)): out[i] = sum_j y[j] v[j*ck + i]."""
    v, y = _u64(v, 4), _u64(y, 4)
    assert len(y) == k and (k == 0 or len(v) % k == 0)
    ck = len(v) // k if k else 0
    out = np.zeros((ck, 4), dtype=np.uint64)
    _check(load().bpgpu_fold_scalars_kary(curve, _ptr(v), ck, k, _ptr(y), _ptr(out)))
    return out


def fold_points_kary(curve, g_xyz, k, y):
    g, y = _u64(g_xyz, 12), _u64(y, 4)
    assert len(y) == k and (k == 0 or len(g) % k == 0)
    ck = len(g) // k if k else 0
    out = np.zeros((ck, 12), dtype=np.uint64)
    _check(load().bpgpu_fold_points_kary(curve, _ptr(g), ck, k, _ptr(y), _ptr(out)))
    return out


def fold_scalars_kary_device(curve, d_v, ck, k, y, d_out):
    _check(load().bpgpu_fold_scalars_kary_device(curve, d_v.ptr, ck, k, _ptr(_u64(y, 4)), d_out.ptr))


def fold_points_kary_device(curve, d_g, ck, k, y, d_out):
    _check(load().bpgpu_fold_points_kary_device(curve, d_g.ptr, ck, k, _ptr(_u64(y, 4)), d_out.ptr))


def kary_cross_terms(curve, a_gens_xyz, b_gens_xyz, ip_gen_xyz, a, b, k):
    """(pos, neg): the k - 1 positive and negative cross terms of a k-ary round over already padded vectors
    (ipa_to_bp_unroll.rs:189-215 == bp_rec_kary.rs:52-76), as xy_inf rows."""
    A, B, q = _u64(a_gens_xyz, 12), _u64(b_gens_xyz, 12), _u64(ip_gen_xyz, 12)
    a, b = _u64(a, 4), _u64(b, 4)
    assert len(A) == len(B) == len(a) == len(b) and k >= 1 and len(a) % k == 0
    ck = len(a) // k
    pos, neg = np.zeros((max(k - 1, 0), 9), np.uint64), np.zeros((max(k - 1, 0), 9), np.uint64)
    _check(load().bpgpu_kary_cross_terms(curve, _ptr(A), _ptr(B), _ptr(q), _ptr(a), _ptr(b), ck, k, _ptr(pos), _ptr(neg)))
    return pos, neg


def hadamard(curve, a, b):
    a, b = _u64(a, 4), _u64(b, 4)
    if len(a) != len(b):
        raise AssertionError("hadamard: length mismatch")  # reference asserts (util/mod.rs:82)
    out = np.zeros_like(a)
    _check(load().bpgpu_hadamard(curve, _ptr(a), _ptr(b), len(a), _ptr(out)))
    return out


def hadamard_points(curve, bases_xyz, scalars):
    g, s = _u64(bases_xyz, 12), _u64(scalars, 4)
    if len(g) != len(s):
        raise AssertionError("hadamard_gp: length mismatch")  # reference asserts (util/mod.rs:72)
    out = np.zeros_like(g)
    _check(load().bpgpu_hadamard_points(curve, _ptr(g), _ptr(s), len(g), _ptr(out)))
    return out


def powers(curve, f, start, end):
    """powers(f, start..end), util/mod.rs:57-64."""
    n = max(0, end - start)
    out = np.zeros((n, 4), dtype=np.uint64)
    _check(load().bpgpu_powers(curve, _ptr(_u64(f)), start, n, _ptr(out)))
    return out


def mat_vec_mult(curve, row_ptr, col_idx, vals, vec):
    """mat_vec_mult (src/relations/com_r1cs.rs:43-57) on a matrix in compressed rows."""
    row_ptr, col_idx = np.ascontiguousarray(row_ptr, np.uint64), np.ascontiguousarray(col_idx, np.uint64)
    vals, vec = _u64(vals, 4), _u64(vec, 4)
    n_rows = len(row_ptr) - 1
    out = np.zeros((n_rows, 4), dtype=np.uint64)
    _check(load().bpgpu_mat_vec_mult(curve, _ptr(row_ptr), _ptr(col_idx), _ptr(vals), n_rows, _ptr(vec), len(vec), _ptr(out)))
    return out


def vec_mat_mult(curve, vec, row_ptr, col_idx, vals, output_size):
    """vec_mat_mult (src/relations/com_r1cs.rs:67-83) on a matrix in compressed rows."""
    row_ptr, col_idx = np.ascontiguousarray(row_ptr, np.uint64), np.ascontiguousarray(col_idx, np.uint64)
    vals, vec = _u64(vals, 4), _u64(vec, 4)
    n_rows = len(row_ptr) - 1
    out = np.zeros((output_size, 4), dtype=np.uint64)
    _check(load().bpgpu_vec_mat_mult(curve, _ptr(vec), len(vec), _ptr(row_ptr), _ptr(col_idx), _ptr(vals), n_rows,
                                     output_size, _ptr(out)))
    return out


def curve_lift_x(curve, xs, greatest):
    """Candidates (x ark form, greatest) of G::rand -> (ys ark form, ok bytes)."""
    xs = _u64(xs, 4)
    g = np.ascontiguousarray(greatest, dtype=np.uint8)
    ys, ok = np.zeros_like(xs), np.zeros(len(xs), dtype=np.uint8)
    _check(load().bpgpu_curve_lift_x(curve, _ptr(xs), _ptr(g), len(xs), _ptr(ys), _ptr(ok)))
    return ys, ok


def ipa_verify_msm(curve, a_gens_xyz, b_gens_xyz, ip_gen_xyz, ls_xyz, rs_xyz, challenges, final_a, final_b, d_gens=None):
    """Right-hand side of Bp::verify (bp_iter.rs:160-176) -> xy_inf.  d_gens = (d_a, d_b, d_q, n) uses
    generators already in HBM."""
    ls, rs, ch = _u64(ls_xyz, 12), _u64(rs_xyz, 12), _u64(challenges, 4)
    k = len(ls)
    assert len(rs) == k and len(ch) == k
    out = np.zeros(9, dtype=np.uint64)
    if d_gens is not None:
        d_a, d_b, d_q, n = d_gens
        _check(load().bpgpu_ipa_verify_msm_device(curve, d_a.ptr, d_b.ptr, d_q.ptr, n, _ptr(ls), _ptr(rs), k, _ptr(ch),
                                                  _ptr(_u64(final_a)), _ptr(_u64(final_b)), _ptr(out)))
    else:
        A, B, q = _u64(a_gens_xyz, 12), _u64(b_gens_xyz, 12), _u64(ip_gen_xyz, 12)
        assert len(A) == len(B)
        _check(load().bpgpu_ipa_verify_msm(curve, _ptr(A), _ptr(B), _ptr(q), len(A), _ptr(ls), _ptr(rs), k, _ptr(ch),
                                           _ptr(_u64(final_a)), _ptr(_u64(final_b)), _ptr(out)))
    return out


def points_sum(curve, xy_inf):
    x = _u64(xy_inf, 9)
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_points_sum(curve, _ptr(x), len(x), _ptr(out)))
    return out


class DeviceBuffer:
    """hipMalloc'd bytes owned through the C ABI."""

    def __init__(self, nbytes: int):
        self.nbytes = nbytes
        p = c_void_p()
        _check(load().bpgpu_dev_alloc(nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(load().bpgpu_dev_upload(self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype=np.uint64, count=None):
        out = np.zeros(self.nbytes // np.dtype(dtype).itemsize if count is None else count, dtype=dtype)
        _check(load().bpgpu_dev_download(_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _check(load().bpgpu_dev_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def points_to_device(curve, points, fmt=0):
    p = _u64(points, 12 if fmt == 0 else 8)
    buf = DeviceBuffer(max(1, len(p)) * 64)
    _check(load().bpgpu_points_to_device(curve, _ptr(p), len(p), fmt, buf.ptr))
    return buf


def points_from_device(curve, buf, n):
    out = np.zeros((n, 12), dtype=np.uint64)
    _check(load().bpgpu_points_from_device(curve, buf.ptr, n, _ptr(out)))
    return out


def scalars_to_device(scalars):
    s = _u64(scalars, 4)
    return DeviceBuffer(max(1, len(s)) * 32).upload(s)


def msm_device(curve, d_points, d_scalars, n):
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_msm_device(curve, d_points.ptr, d_scalars.ptr, n, _ptr(out)))
    return out


def ip_device(curve, d_a, d_b, n):
    out = np.zeros(4, dtype=np.uint64)
    _check(load().bpgpu_ip_device(curve, d_a.ptr, d_b.ptr, n, _ptr(out)))
    return out


def ip_device_noalloc(curve, d_a, d_b, n, d_out):
    """bpgpu_ip_device without any allocation: the result lands in the caller's device buffer d_out (32 B)."""
    _check(load().bpgpu_ip_device_noalloc(curve, d_a.ptr, d_b.ptr, n, d_out.ptr))


def sharded_init(devices):
    """One context per entry of `devices` (HIP device numbers; repeats allowed: rehearsal on one GPU)."""
    d = np.ascontiguousarray(devices, dtype=np.int32)
    _check(load().bpgpu_sharded_init(len(d), _ptr(d)))


def sharded_count():
    return load().bpgpu_sharded_count()


def msm_sharded(curve, points_xyz, scalars):
    """bpgpu_msm over the contexts of sharded_init: contiguous ranges, one per context, summed on the host."""
    p, s = _u64(points_xyz, 12), _u64(scalars, 4)
    n = min(len(p), len(s))
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_msm_sharded(curve, _ptr(p), _ptr(s), n, _ptr(out)))
    return out


def msm_sharded_affine(curve, points_xy, scalars):
    p, s = _u64(points_xy, 8), _u64(scalars, 4)
    n = min(len(p), len(s))
    out = np.zeros(9, dtype=np.uint64)
    _check(load().bpgpu_msm_sharded_affine(curve, _ptr(p), _ptr(s), n, _ptr(out)))
    return out


class ShardedGens:
    """Generators resident on every context of sharded_init (each holds its contiguous range)."""

    def __init__(self, curve, points, fmt=0):
        p = _u64(points, 12 if fmt == 0 else 8)
        h = c_void_p()
        _check(load().bpgpu_sharded_gens_load(curve, _ptr(p), len(p), fmt, ctypes.byref(h)))
        self.h, self.n = h.value, len(p)

    def msm(self, scalars):
        s = _u64(scalars, 4)
        assert len(s) == self.n
        out = np.zeros(9, dtype=np.uint64)
        _check(load().bpgpu_sharded_gens_msm(self.h, _ptr(s), _ptr(out)))
        return out

    def free(self):
        if self.h:
            load().bpgpu_sharded_gens_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def fold_scalars_device(curve, d_v, n_out, s_lo, s_hi, d_out=None):
    d_out = d_out or d_v
    _check(load().bpgpu_fold_scalars_device(curve, d_v.ptr, n_out, _ptr(_u64(s_lo)), _ptr(_u64(s_hi)), d_out.ptr))


def fold_points_device(curve, d_g, n_out, s_lo, s_hi, d_out=None):
    d_out = d_out or d_g
    _check(load().bpgpu_fold_points_device(curve, d_g.ptr, n_out, _ptr(_u64(s_lo)), _ptr(_u64(s_hi)), d_out.ptr))


def auto_window_bits(n: int) -> int:
    """Window width the library picks for an n-point MSM (mirror of choose_window in bpgpu.hip;
    reporting only -- the library does not take it from here)."""
    lg = max(0, int(n).bit_length() - 1)
    if 16 <= lg <= 24:
        return (15, 15, 16, 16, 17, 17, 17, 19, 20)[lg - 16]   # measured optimum (choose_window's table)
    if lg == 14:
        return 13
    if 5 <= lg <= 15:
        return lg
    return min(20, max(2, lg - 4))


def last_msm_timings():
    out = (c_float * 8)()
    _check(load().bpgpu_last_msm_timings(out))
    return list(out)


def last_msm_entries():
    """Non-zero digits (= bucket-accumulation additions) of the last MSM call."""
    v = c_uint64()
    _check(load().bpgpu_last_msm_entries(ctypes.byref(v)))
    return int(v.value)


def last_op_ms():
    v = c_float()
    _check(load().bpgpu_last_op_ms(ctypes.byref(v)))
    return v.value


def last_fold_ops():
    """(outputs, doublings, mixed additions, kernel kind) of the last generator-fold launch."""
    out = (c_uint64 * 4)()
    _check(load().bpgpu_last_fold_ops(out))
    return [int(v) for v in out]


def measure_mad_peak():
    v = c_double()
    _check(load().bpgpu_measure_mad_peak(ctypes.byref(v)))
    return v.value


class IpaSession:
    """Device-resident prover state for Bp::prove (reference src/proofs/bp_iter.rs:49-128)."""

    def __init__(self, curve, a_gens_xyz, b_gens_xyz, ip_gen_xyz, a, b):
        A, B = _u64(a_gens_xyz, 12), _u64(b_gens_xyz, 12)
        q = _u64(ip_gen_xyz, 12)
        a, b = _u64(a, 4), _u64(b, 4)
        n = len(a)
        assert len(A) == len(B) == len(b) == n
        h = c_void_p()
        _check(load().bpgpu_ipa_begin(curve, _ptr(A), _ptr(B), _ptr(q), _ptr(a), _ptr(b), n, ctypes.byref(h)))
        self.h = h.value
        self.curve = curve

    @classmethod
    def from_device(cls, curve, d_a_gens, d_b_gens, d_ip_gen, a, b):
        """Generators already in HBM (DeviceBuffers of dev affine points, see points_to_device)."""
        a, b = _u64(a, 4), _u64(b, 4)
        n = len(a)
        assert len(b) == n and d_a_gens.nbytes >= 64 * n and d_b_gens.nbytes >= 64 * n
        h = c_void_p()
        _check(load().bpgpu_ipa_begin_device(curve, d_a_gens.ptr, d_b_gens.ptr, d_ip_gen.ptr, _ptr(a), _ptr(b), n,
                                             ctypes.byref(h)))
        self = cls.__new__(cls)
        self.h = h.value
        self.curve = curve
        return self

    def __len__(self):
        return load().bpgpu_ipa_len(self.h)

    def round_lr(self):
        l, r = np.zeros(9, dtype=np.uint64), np.zeros(9, dtype=np.uint64)
        _check(load().bpgpu_ipa_round_lr(self.h, _ptr(l), _ptr(r)))
        return l, r

    def round_fold(self, x, x_inv):
        _check(load().bpgpu_ipa_round_fold(self.h, _ptr(_u64(x)), _ptr(_u64(x_inv))))

    def finish(self):
        fa, fb = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        _check(load().bpgpu_ipa_finish(self.h, _ptr(fa), _ptr(fb)))
        return fa, fb

    def export(self):
        n = len(self)
        A, B = np.zeros((n, 12), dtype=np.uint64), np.zeros((n, 12), dtype=np.uint64)
        a, b = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
        _check(load().bpgpu_ipa_export(self.h, _ptr(A), _ptr(B), _ptr(a), _ptr(b)))
        return A, B, a, b

    def free(self):
        if self.h:
            load().bpgpu_ipa_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
