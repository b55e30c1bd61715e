"""ctypes view of libarkbp.so -- the C++ host side (bulletproofs_amd/host/ark_bp.hpp) that mirrors
the reference's util::msm / Bp / Bp2aryStep / FiatShamirRng above the bpgpu C ABI.  Plumbing only."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_int, c_size_t, c_void_p

import numpy as np

from . import ffi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libarkbp.so")
_lib = None


class Panic(RuntimeError):
    """The reference would have panicked here (assert!/unwrap on this path)."""


def load():
    global _lib
    if _lib is None:
        ffi.load()
        if not os.path.exists(LIB_PATH):
            raise ffi.BpGpuError(f"{LIB_PATH} is missing: build it with `make`")
        lib = ctypes.CDLL(LIB_PATH)
        lib.arkbp_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else c_void_p(a.ctypes.data)


def _u64(a, w):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, w)


def _check(rc):
    if rc < 0:
        raise Panic(load().arkbp_last_error().decode())
    return rc


def msm(curve, bases, scalars):
    b, s = _u64(bases, 12), _u64(scalars, 4)
    out = np.zeros(12, np.uint64)
    _check(load().arkbp_msm(curve, _p(b), c_size_t(len(b)), _p(s), c_size_t(len(s)), _p(out)))
    return out


def bos_coster_msm(curve, bases, scalars):
    """util::msm::bos_coster_msm (src/util/msm.rs:34-81): serial heap algorithm, runs on the host (no GPU needed)."""
    b, s = _u64(bases, 12), _u64(scalars, 4)
    out = np.zeros(12, np.uint64)
    _check(load().arkbp_bos_coster_msm(curve, _p(b), c_size_t(len(b)), _p(s), c_size_t(len(s)), _p(out)))
    return out


def ip(curve, a, b):
    a, b = _u64(a, 4), _u64(b, 4)
    out = np.zeros(4, np.uint64)
    _check(load().arkbp_ip(curve, _p(a), c_size_t(len(a)), _p(b), c_size_t(len(b)), _p(out)))
    return out


def commit(curve, a_gens, b_gens, ip_gen, a, b):
    A, B, q, a, b = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(a, 4), _u64(b, 4)
    out = np.zeros(12, np.uint64)
    _check(load().arkbp_commit(curve, _p(A), _p(B), _p(q), _p(a), _p(b), c_size_t(len(a)), _p(out)))
    return out


def bp_prove(curve, a_gens, b_gens, ip_gen, a, b, fs_seed: bytes):
    """Bp::prove -> dict(ls, rs, challenges, final_a, final_b, prove_ms)"""
    A, B, q, a, b = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(a, 4), _u64(b, 4)
    n = len(a)
    kmax = max(1, (max(n, 1) - 1).bit_length())
    ls, rs = np.zeros((kmax, 12), np.uint64), np.zeros((kmax, 12), np.uint64)
    ch = np.zeros((kmax, 4), np.uint64)
    fa, fb = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    ms = c_double()
    k = _check(load().arkbp_bp_prove(curve, _p(A), _p(B), _p(q), _p(a), _p(b), c_size_t(n), _p(seed), _p(ls), _p(rs),
                                     _p(ch), _p(fa), _p(fb), ctypes.byref(ms)))
    return dict(ls=ls[:k], rs=rs[:k], challenges=ch[:k], final_a=fa, final_b=fb, prove_ms=ms.value)


class DeviceGens:
    """IpaGens uploaded once (host mirror: ark_bp.hpp DeviceGens); pass to bp_prove_cached."""

    def __init__(self, curve, a_gens, b_gens, ip_gen):
        A, B, q = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12)
        assert len(A) == len(B)
        self.curve, self.n = curve, len(A)
        h = ctypes.c_void_p()
        _check(load().arkbp_gens_upload(curve, _p(A), _p(B), _p(q), c_size_t(self.n), ctypes.byref(h)))
        self.h = h.value

    def free(self):
        if self.h:
            load().arkbp_gens_free(ctypes.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def bp_prove_cached(gens: DeviceGens, a, b, fs_seed: bytes):
    """Bp::prove with the generators already in HBM -> same dict as bp_prove."""
    a, b = _u64(a, 4), _u64(b, 4)
    n = len(a)
    kmax = max(1, (max(n, 1) - 1).bit_length())
    ls, rs = np.zeros((kmax, 12), np.uint64), np.zeros((kmax, 12), np.uint64)
    ch = np.zeros((kmax, 4), np.uint64)
    fa, fb = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    ms = c_double()
    k = _check(load().arkbp_bp_prove_cached(gens.curve, ctypes.c_void_p(gens.h), _p(a), _p(b), c_size_t(n), _p(seed),
                                            _p(ls), _p(rs), _p(ch), _p(fa), _p(fb), ctypes.byref(ms)))
    return dict(ls=ls[:k], rs=rs[:k], challenges=ch[:k], final_a=fa, final_b=fb, prove_ms=ms.value)


def bp_verify(curve, a_gens, b_gens, ip_gen, result, proof, fs_seed: bytes):
    """Bp::verify: returns the verify time in ms, raises Panic where the reference's assert_eq! fails."""
    A, B, q, res = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12)
    ls, rs = _u64(proof["ls"], 12), _u64(proof["rs"], 12)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    ms = c_double()
    _check(load().arkbp_bp_verify(curve, _p(A), _p(B), _p(q), _p(res), c_size_t(len(A)), _p(ls), _p(rs), c_size_t(len(ls)),
                                  _p(_u64(proof["final_a"], 4)), _p(_u64(proof["final_b"], 4)), _p(seed), ctypes.byref(ms)))
    return ms.value


def bp_verify_cached(gens: DeviceGens, result, proof, fs_seed: bytes):
    """Bp::verify with the generators already in HBM; returns the verify time in ms."""
    res = _u64(result, 12)
    ls, rs = _u64(proof["ls"], 12), _u64(proof["rs"], 12)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    ms = c_double()
    _check(load().arkbp_bp_verify_cached(gens.curve, ctypes.c_void_p(gens.h), _p(res), c_size_t(gens.n), _p(ls), _p(rs),
                                         c_size_t(len(ls)), _p(_u64(proof["final_a"], 4)), _p(_u64(proof["final_b"], 4)),
                                         _p(seed), ctypes.byref(ms)))
    return ms.value


def step_prove(curve, a_gens, b_gens, ip_gen, result, a, b, fs_seed: bytes):
    """Bp2aryStep::prove -> ((l, r), next instance dict, next witness dict)"""
    A, B, q, res, a, b = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12), _u64(a, 4), _u64(b, 4)
    n = len(a)
    m = max(1, (1 << max(0, (n - 1).bit_length())) // 2)
    l, r, res2 = np.zeros(12, np.uint64), np.zeros(12, np.uint64), np.zeros(12, np.uint64)
    A2, B2 = np.zeros((m, 12), np.uint64), np.zeros((m, 12), np.uint64)
    a2, b2 = np.zeros((m, 4), np.uint64), np.zeros((m, 4), np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    k = _check(load().arkbp_2ary_prove(curve, _p(A), _p(B), _p(q), _p(res), _p(a), _p(b), c_size_t(n), _p(seed), _p(l), _p(r),
                                       _p(A2), _p(B2), _p(a2), _p(b2), _p(res2)))
    assert k == m
    return (l, r), dict(a_gens=A2, b_gens=B2, ip_gen=q, result=res2), dict(a=a2, b=b2)


def step_verify(curve, a_gens, b_gens, ip_gen, result, l, r, fs_seed: bytes):
    A, B, q, res = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12)
    n = len(A)
    m = max(1, (1 << max(0, (n - 1).bit_length())) // 2)
    A2, B2, res2 = np.zeros((m, 12), np.uint64), np.zeros((m, 12), np.uint64), np.zeros(12, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    k = _check(load().arkbp_2ary_verify(curve, _p(A), _p(B), _p(q), _p(res), c_size_t(n), _p(_u64(l, 12)), _p(_u64(r, 12)),
                                        _p(seed), _p(A2), _p(B2), _p(res2)))
    assert k == m
    return dict(a_gens=A2, b_gens=B2, ip_gen=q, result=res2)


def fs_challenge(curve, fs_seed: bytes, points_xyz):
    pts = _u64(points_xyz, 12)
    out = np.zeros(4, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    _check(load().arkbp_fs_challenge(curve, _p(seed), _p(pts), c_size_t(len(pts)), _p(out)))
    return out


def scalar_inverse(curve, x):
    out = np.zeros(4, np.uint64)
    _check(load().arkbp_scalar_inverse(curve, _p(_u64(x, 4)), _p(out)))
    return out


def ipa_check(curve, a_gens, b_gens, ip_gen, result, a, b):
    """IpaRelation::check (ipa.rs:134-146): raises Panic where the reference's assert fails."""
    A, B, q, res = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12)
    a, b = _u64(a, 4), _u64(b, 4)
    _check(load().arkbp_ipa_check(curve, _p(A), _p(B), _p(q), _p(res), _p(a), _p(b), c_size_t(len(a))))


def kary_prove_step(curve, k, a_gens, b_gens, ip_gen, result, a, b, fs_seed: bytes):
    """KaryBp prove_step (bp_rec_kary.rs:27-143) -> dict(pos_cross, neg_cross, a_gens, b_gens, a, b, result)."""
    A, B, q, res = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12)
    a, b = _u64(a, 4), _u64(b, 4)
    n = len(a)
    ck = (n - 1) // k + 1
    pos, neg = np.zeros((max(k - 1, 1), 12), np.uint64), np.zeros((max(k - 1, 1), 12), np.uint64)
    A2, B2 = np.zeros((ck, 12), np.uint64), np.zeros((ck, 12), np.uint64)
    a2, b2 = np.zeros((ck, 4), np.uint64), np.zeros((ck, 4), np.uint64)
    r2 = np.zeros(12, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    m = _check(load().arkbp_kary_prove_step(curve, c_size_t(k), _p(A), _p(B), _p(q), _p(res), _p(a), _p(b), c_size_t(n),
                                            _p(seed), _p(pos), _p(neg), _p(A2), _p(B2), _p(a2), _p(b2), _p(r2)))
    assert m == ck
    return dict(pos_cross=pos[:k - 1], neg_cross=neg[:k - 1], a_gens=A2, b_gens=B2, a=a2, b=b2, result=r2)


def kary_verify_step(curve, k, a_gens, b_gens, ip_gen, result, pos_cross, neg_cross, fs_seed: bytes):
    A, B, q, res = _u64(a_gens, 12), _u64(b_gens, 12), _u64(ip_gen, 12), _u64(result, 12)
    pos, neg = _u64(pos_cross, 12), _u64(neg_cross, 12)
    n = len(A)
    ck = (n - 1) // k + 1
    A2, B2 = np.zeros((ck, 12), np.uint64), np.zeros((ck, 12), np.uint64)
    r2 = np.zeros(12, np.uint64)
    seed = np.frombuffer(fs_seed, np.uint8).copy()
    m = _check(load().arkbp_kary_verify_step(curve, c_size_t(k), _p(A), _p(B), _p(q), _p(res), c_size_t(n),
                                             _p(pos) if len(pos) else None, _p(neg) if len(neg) else None, _p(seed),
                                             _p(A2), _p(B2), _p(r2)))
    assert m == ck
    return dict(a_gens=A2, b_gens=B2, result=r2)


def pad(mode, v, k):
    """util::zero_pad_to_multiple (mode 0) / zero_intersperse_to_multiple (mode 1) on scalars; pure host code."""
    v = _u64(v, 4)
    out = np.zeros((len(v) + k + 1, 4), np.uint64)
    lib = load()
    lib.arkbp_pad.restype = ctypes.c_long
    m = _check(lib.arkbp_pad(mode, _p(v), c_size_t(len(v)), c_size_t(k), _p(out)))
    return out[:m]


def msm_bench_inputs(curve, n):
    """examples/msm_bench.rs:40-44 over test_rng(): n points (square roots taken on the device), n scalars."""
    pts, sc = np.zeros((n, 12), np.uint64), np.zeros((n, 4), np.uint64)
    _check(load().arkbp_sample(curve, 0, c_size_t(n), _p(pts), None, None, _p(sc), None, None))
    return pts, sc


def ipa_inputs(curve, n):
    """src/lib.rs:111-118: IpaInstance::sample_from_length draws, then the Fiat-Shamir seed."""
    A, B, q = np.zeros((n, 12), np.uint64), np.zeros((n, 12), np.uint64), np.zeros((1, 12), np.uint64)
    a, b, seed = np.zeros((n, 4), np.uint64), np.zeros((n, 4), np.uint64), np.zeros(32, np.uint8)
    _check(load().arkbp_sample(curve, 1, c_size_t(n), _p(A), _p(B), _p(q), _p(a), _p(b), _p(seed)))
    return A, B, q, a, b, seed.tobytes()


def mat_vec_mult(curve, row_ptr, col_idx, vals, vec):
    """util::mat_vec_mult (src/relations/com_r1cs.rs:43-57) through the C++ mirror."""
    row_ptr, col_idx = np.ascontiguousarray(row_ptr, np.uint64), np.ascontiguousarray(col_idx, np.uint64)
    vals, vec = _u64(vals, 4), _u64(vec, 4)
    out = np.zeros((len(row_ptr) - 1, 4), np.uint64)
    _check(load().arkbp_mat_mult(curve, 0, _p(row_ptr), _p(col_idx), _p(vals), c_size_t(len(row_ptr) - 1), _p(vec),
                                 c_size_t(len(vec)), c_size_t(0), _p(out)))
    return out


def vec_mat_mult(curve, vec, row_ptr, col_idx, vals, output_size):
    """util::vec_mat_mult (src/relations/com_r1cs.rs:67-83) through the C++ mirror."""
    row_ptr, col_idx = np.ascontiguousarray(row_ptr, np.uint64), np.ascontiguousarray(col_idx, np.uint64)
    vals, vec = _u64(vals, 4), _u64(vec, 4)
    out = np.zeros((output_size, 4), np.uint64)
    _check(load().arkbp_mat_mult(curve, 1, _p(row_ptr), _p(col_idx), _p(vals), c_size_t(len(row_ptr) - 1), _p(vec),
                                 c_size_t(len(vec)), c_size_t(output_size), _p(out)))
    return out
