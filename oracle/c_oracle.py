"""ctypes loader for Oracle B (oracle/pasta_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy uint64 in ark-ff memory layout, the same bytes the C ABI of the product takes.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "pasta_oracle.c")
LIB = os.path.join(_HERE, "_build", "libpasta_oracle.so")

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-std=c11", "-shared", "-fPIC", "-pthread", SRC, "-o", LIB])
    return LIB


def load(threads=1):
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    _lib.po_init(int(threads))
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _u64(a, w):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, w)


def msm_pippenger(curve, points_xyz, scalars, threads=1, raw=False):
    lib = load(threads)
    p, s = _u64(points_xyz, 12), _u64(scalars, 4)
    n = min(len(p), len(s))
    o_raw, o_norm = np.zeros(12, np.uint64), np.zeros(12, np.uint64)
    lib.po_msm_pippenger(curve, _p(p), _p(s), ctypes.c_size_t(n), _p(o_raw), _p(o_norm))
    return (o_raw, o_norm) if raw else o_norm


def msm_naive(curve, points_xyz, scalars):
    lib = load()
    p, s = _u64(points_xyz, 12), _u64(scalars, 4)
    n = min(len(p), len(s))
    o = np.zeros(12, np.uint64)
    lib.po_msm_naive(curve, _p(p), _p(s), ctypes.c_size_t(n), None, _p(o))
    return o


def ip(curve, a, b):
    lib = load()
    a, b = _u64(a, 4), _u64(b, 4)
    assert len(a) == len(b)
    o = np.zeros(4, np.uint64)
    lib.po_ip(curve, _p(a), _p(b), ctypes.c_size_t(len(a)), _p(o))
    return o


def fold_scalars(curve, v, s_lo, s_hi, threads=1):
    lib = load(threads)
    v = _u64(v, 4)
    n = len(v) // 2
    o = np.zeros((n, 4), np.uint64)
    lib.po_fold_scalars(curve, _p(v), ctypes.c_size_t(n), _p(_u64(s_lo, 4)), _p(_u64(s_hi, 4)), _p(o))
    return o


def fold_points(curve, g_xyz, s_lo, s_hi, threads=1, normalise=True):
    lib = load(threads)
    g = _u64(g_xyz, 12)
    n = len(g) // 2
    o = np.zeros((n, 12), np.uint64)
    lib.po_fold_points(curve, _p(g), ctypes.c_size_t(n), _p(_u64(s_lo, 4)), _p(_u64(s_hi, 4)), _p(o), int(normalise))
    return o


def normalise(curve, pts_xyz):
    lib = load()
    p = _u64(pts_xyz, 12)
    o = np.zeros_like(p)
    lib.po_normalise(curve, _p(p), ctypes.c_size_t(len(p)), _p(o))
    return o


def scalar_inverse(curve, x):
    lib = load()
    o = np.zeros(4, np.uint64)
    lib.po_scalar_inverse(curve, _p(_u64(x, 4)), _p(o))
    return o


def on_curve(curve, pts_xyz):
    lib = load()
    p = _u64(pts_xyz, 12)
    return all(lib.po_on_curve(curve, ctypes.c_void_p(p[i].ctypes.data)) for i in range(len(p)))


def msm_bench_inputs(curve, n):
    """examples/msm_bench.rs:40-44 (faithful recipe; sequential)."""
    lib = load()
    pts, sc = np.zeros((n, 12), np.uint64), np.zeros((n, 4), np.uint64)
    lib.po_msm_bench_inputs(curve, ctypes.c_size_t(n), _p(pts), _p(sc))
    return pts, sc


def ipa_inputs(curve, n):
    """src/lib.rs:111-118 sampling order -> a_gens, b_gens, ip_gen, a, b, fs_seed."""
    lib = load()
    A, B, q = np.zeros((n, 12), np.uint64), np.zeros((n, 12), np.uint64), np.zeros((1, 12), np.uint64)
    a, b = np.zeros((n, 4), np.uint64), np.zeros((n, 4), np.uint64)
    seed = np.zeros(32, np.uint8)
    lib.po_ipa_inputs(curve, ctypes.c_size_t(n), _p(A), _p(B), _p(q), _p(a), _p(b), _p(seed))
    return A, B, q, a, b, seed.tobytes()


def gen_points(curve, seed, n, threads=1):
    lib = load(threads)
    pts = np.zeros((n, 12), np.uint64)
    lib.po_gen_points(curve, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(pts))
    return pts


def gen_scalars(curve, seed, n):
    lib = load()
    sc = np.zeros((n, 4), np.uint64)
    lib.po_gen_scalars(curve, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(sc))
    return sc


def bp_prove(curve, a_gens, b_gens, ip_gen, a, b, fs_seed, threads=1):
    """Inputs must already be padded to a power of two.  Returns ls, rs (normalised jacobian),
    challenges, final_a, final_b."""
    lib = load(threads)
    A, B, q = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12)
    a, b = _u64(a, 4), _u64(b, 4)
    n = len(a)
    k = max(0, n.bit_length() - 1)
    ls, rs = np.zeros((k, 12), np.uint64), np.zeros((k, 12), np.uint64)
    ch = np.zeros((k, 4), np.uint64)
    fa, fb = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    seed = np.frombuffer(fs_seed, dtype=np.uint8).copy()
    rc = lib.po_bp_prove(curve, _p(A), _p(B), _p(q), _p(a), _p(b), ctypes.c_size_t(n), _p(seed), _p(ls), _p(rs), _p(ch),
                         _p(fa), _p(fb))
    assert rc == 0
    return ls, rs, ch, fa, fb


def fs_challenge(curve, fs_seed, absorbed96: bytes):
    lib = load()
    o = np.zeros(4, np.uint64)
    buf = np.frombuffer(absorbed96, dtype=np.uint8).copy() if absorbed96 else np.zeros(1, np.uint8)
    seed = np.frombuffer(fs_seed, dtype=np.uint8).copy()
    lib.po_fs_challenge(curve, _p(seed), _p(buf), ctypes.c_size_t(len(absorbed96) // 96), _p(o))
    return o


# ---- vector utilities of the k-ary protocols (src/util/mod.rs:25-134) -------------------------
def fold_scalars_kary(curve, v, k, y):
    lib = load()
    v, y = _u64(v, 4), _u64(y, 4)
    ck = len(v) // k if k else 0
    o = np.zeros((ck, 4), np.uint64)
    lib.po_fold_scalars_kary(curve, _p(v), ctypes.c_size_t(ck), ctypes.c_size_t(k), _p(y), _p(o))
    return o


def fold_points_kary(curve, g_xyz, k, y, threads=1):
    lib = load(threads)
    g, y = _u64(g_xyz, 12), _u64(y, 4)
    ck = len(g) // k if k else 0
    o = np.zeros((ck, 12), np.uint64)
    lib.po_fold_points_kary(curve, _p(g), ctypes.c_size_t(ck), ctypes.c_size_t(k), _p(y), _p(o))
    return o


def hadamard(curve, a, b):
    lib = load()
    a, b = _u64(a, 4), _u64(b, 4)
    assert len(a) == len(b)
    o = np.zeros_like(a)
    lib.po_hadamard(curve, _p(a), _p(b), ctypes.c_size_t(len(a)), _p(o))
    return o


def hadamard_points(curve, bases_xyz, scalars, threads=1):
    lib = load(threads)
    g, s = _u64(bases_xyz, 12), _u64(scalars, 4)
    assert len(g) == len(s)
    o = np.zeros_like(g)
    lib.po_hadamard_points(curve, _p(g), _p(s), ctypes.c_size_t(len(g)), _p(o))
    return o


def powers(curve, f, start, n):
    lib = load()
    o = np.zeros((n, 4), np.uint64)
    lib.po_powers(curve, _p(_u64(f, 4)), ctypes.c_uint64(start), ctypes.c_size_t(n), _p(o))
    return o


def scalar_from_u64(curve, v):
    lib = load()
    o = np.zeros(4, np.uint64)
    lib.po_scalar_from_u64(curve, ctypes.c_uint64(v), _p(o))
    return o


def _pad(fn, v, k):
    v = _u64(v, 4)
    fn.restype = ctypes.c_size_t
    n = fn(_p(v), ctypes.c_size_t(len(v)), ctypes.c_size_t(k), None)
    o = np.zeros((n, 4), np.uint64)
    got = fn(_p(v), ctypes.c_size_t(len(v)), ctypes.c_size_t(k), _p(o))
    assert got == n
    return o


def zero_pad_to_multiple(v, k):
    return _pad(load().po_zero_pad_to_multiple, v, k)


def zero_intersperse_to_multiple(v, k):
    return _pad(load().po_zero_intersperse_to_multiple, v, k)


# ---- sparse products of the Com-R1CS compiler (src/relations/com_r1cs.rs:43-83) ----------------
def csr(mat):
    """list of rows of (value 4 x u64, column) -> row_ptr, col_idx, vals (numpy uint64)."""
    row_ptr = np.zeros(len(mat) + 1, np.uint64)
    cols, vals = [], []
    for r, row in enumerate(mat):
        row_ptr[r + 1] = row_ptr[r] + np.uint64(len(row))
        for v, c_ in row:
            vals.append(np.asarray(v, dtype=np.uint64))
            cols.append(c_)
    return (row_ptr, np.asarray(cols, dtype=np.uint64),
            np.stack(vals).astype(np.uint64) if vals else np.zeros((0, 4), np.uint64))


def mat_vec_mult(curve, row_ptr, col_idx, vals, vec):
    lib = load()
    n_rows = len(row_ptr) - 1
    vec, vals = _u64(vec, 4), _u64(vals, 4)
    o = np.zeros((n_rows, 4), np.uint64)
    rc = lib.po_mat_vec_mult(curve, _p(row_ptr), _p(col_idx), _p(vals), ctypes.c_size_t(n_rows), _p(vec),
                             ctypes.c_size_t(len(vec)), _p(o))
    assert rc == 0, "the reference asserts here"
    return o


def vec_mat_mult(curve, vec, row_ptr, col_idx, vals, output_size):
    lib = load()
    n_rows = len(row_ptr) - 1
    vec, vals = _u64(vec, 4), _u64(vals, 4)
    o = np.zeros((output_size, 4), np.uint64)
    rc = lib.po_vec_mat_mult(curve, _p(vec), ctypes.c_size_t(len(vec)), _p(row_ptr), _p(col_idx), _p(vals),
                             ctypes.c_size_t(n_rows), ctypes.c_size_t(output_size), _p(o))
    assert rc == 0, "the reference asserts here"
    return o
