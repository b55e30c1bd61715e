"""MSM at the largest sizes include/bpgpu.h admits (BPGPU_MSM_MAX_N = 2^28 points per call): the sort keeps entry
positions in 32 bits and a 2^28-point MSM has 3.5e9 (point, window) pairs, so this is where an index that wraps or
goes negative would show.

The inputs are grown ON THE DEVICE from a small seeded set (bpgpu_fold_points_device / bpgpu_fold_scalars_device write
new combinations behind the ones they read): 2^28 points cost seconds instead of the minutes host sampling would, and
no 25 GB host array crosses PCIe.  Size-independent check: one call over all n points against the sum of the calls
over its 2^24-point sub-ranges -- 2^24 is pinned by tests/test_gpu_parity.py (eight 2^21 shards, an oracle slice) --
plus a ragged tail.  Bit-exact."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

PIECE = 1 << 24


class View:
    """A sub-range of a DeviceBuffer (the ffi wrappers only read .ptr)."""

    def __init__(self, base, off):
        self.ptr = base.ptr + off


@pytest.fixture(scope="module")
def ffi():
    from bulletproofs_amd import ffi as f
    f.init(0)
    return f


def grow_inputs(ffi, curve, n, seed, seed_len=4096):
    d_pts = ffi.DeviceBuffer(n * 64)
    d_sc = ffi.DeviceBuffer(n * 32)
    m0 = min(seed_len, n)
    p0 = np.ascontiguousarray(bench.synth_points(ffi, curve, m0, seed))
    ffi._check(ffi.load().bpgpu_points_to_device(curve, ffi._ptr(p0), m0, 0, d_pts.ptr))
    d_sc.upload(bench.synth_scalars(m0, seed + 1))
    have, step = m0, 0
    while have < n:
        m = min(have // 2, n - have)
        a, b = bench.synth_scalars(2, 1000 * seed + step)
        ffi.fold_points_device(curve, d_pts, m, a, b, View(d_pts, have * 64))    # reads [0, 2m), writes [have, have + m)
        ffi.fold_scalars_device(curve, d_sc, m, a, b, View(d_sc, have * 32))
        have += m
        step += 1
    return d_pts, d_sc


def sum_of_pieces(ffi, curve, d_pts, d_sc, n):
    parts = []
    for lo in range(0, n, PIECE):
        cnt = min(PIECE, n - lo)
        parts.append(ffi.msm_device(curve, View(d_pts, lo * 64), View(d_sc, lo * 32), cnt))
    return ffi.points_sum(curve, np.stack(parts))


@pytest.mark.parametrize("curve,n", [(0, 1 << 28), (1, (1 << 26) + 12345)], ids=["pallas-2^28", "vesta-2^26+12345"])
def test_msm_at_the_size_limit(ffi, curve, n):
    d_pts, d_sc = grow_inputs(ffi, curve, n, seed=11 + curve)
    try:
        full = ffi.msm_device(curve, d_pts, d_sc, n)
        assert full[8] == 0  # a random combination of 2^28 points is not the identity
        entries = ffi.last_msm_entries()
        windows = -(-256 // ffi.auto_window_bits(n))
        assert n * (windows - 2) < entries <= n * windows  # every non-zero digit was sorted: 3.4e9 of them at 2^28
        want = sum_of_pieces(ffi, curve, d_pts, d_sc, n)
        assert (full == want).all()
    finally:
        d_pts.free()
        d_sc.free()


def test_ipa_at_the_size_limit(ffi):
    """Bp::prove / Bp::verify at BPGPU_IPA_MAX_N = 2^26 elements per vector, through the host-vector entry points (two-phase
    generator upload, first round inside bpgpu_ipa_begin, batches of two MSMs over 2^26 + 1 slots).  The check is the
    reference's own (src/proofs/bp_iter.rs:130-178): the verifier's single MSM over the ORIGINAL generators accepts the
    proof and rejects a tampered one."""
    from bulletproofs_amd import ark_bp
    curve, n = 0, 1 << 26
    d_pts, d_sc = grow_inputs(ffi, curve, 2 * n + 1, seed=21)
    try:
        pts = ffi.points_from_device(curve, d_pts, 2 * n + 1)
        sc = d_sc.download(np.uint64, count=2 * n * 4).reshape(2 * n, 4)
    finally:
        d_pts.free()
        d_sc.free()
    A, B, q, a, b = pts[:n], pts[n:2 * n], pts[2 * n:], sc[:n], sc[n:]
    result = ark_bp.commit(curve, A, B, q, a, b)
    seed = b"ipa at the size limit".ljust(32, b"\0")
    proof = ark_bp.bp_prove(curve, A, B, q, a, b, seed)
    assert len(proof["ls"]) == len(proof["rs"]) == 26
    ark_bp.bp_verify(curve, A, B, q, result, proof, seed)
    bad = dict(proof, final_b=proof["final_b"].copy())
    bad["final_b"][1] ^= np.uint64(4)
    with pytest.raises(ark_bp.Panic):
        ark_bp.bp_verify(curve, A, B, q, result, bad, seed)


def test_one_point_beyond_the_limit_is_refused(ffi):
    # (the IPA limit is checked the same way, without a GPU, in tests/test_abi.py)
    d = ffi.DeviceBuffer(64)
    try:
        with pytest.raises(ffi.BpGpuError):
            ffi.msm_device(0, d, d, (1 << 28) + 1)
    finally:
        d.free()
