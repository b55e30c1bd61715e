"""GPU parity for the last "next" row: the sparse matrix - vector products of the Com-R1CS
compiler (mat_vec_mult / vec_mat_mult, src/relations/com_r1cs.rs:43-83, called at
src/reductions/com_r1cs_to_ipa.rs:63-64,93-99) and the seed-faithful point sampler (ark-ec's G::rand as
examples/msm_bench.rs:41 and src/relations/ipa.rs:28-36 use it), through the C ABI and the C++ mirror
against the oracles.  Bit-exact (integer work)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pasta_model as pm
from tests import helpers as H

pytestmark = pytest.mark.gpu
CURVES = [pm.PALLAS, pm.VESTA]


@pytest.fixture(scope="module")
def ffi():
    from bulletproofs_amd import ffi as f
    f.init(0)
    return f


@pytest.fixture(scope="module")
def bp():
    from bulletproofs_amd import ark_bp, ffi as f
    f.init(0)
    return ark_bp


def random_matrix(c, n_rows, n_cols, seed, long_row=0):
    """R1CS-like: most rows hold 0-4 entries, repeated columns allowed, optionally one long row."""
    rng = np.random.default_rng(seed)
    vals = H.rand_scalars(c, 4 * n_rows + long_row + 8, seed + 1)
    mat, k = [], 0
    for r in range(n_rows):
        ln = long_row if (long_row and r == n_rows // 2) else int(rng.integers(0, 5))
        row = []
        for _ in range(ln):
            v = [0, 1, c.r - 1][k % 3] if k % 17 == 0 else vals[k]
            row.append((v, int(rng.integers(0, n_cols))))
            k += 1
        mat.append(row)
    return mat


def to_csr(c, mat):
    return co.csr([[(H.scalar_abi(c, v), col) for v, col in row] for row in mat])


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n_rows,n_cols,long_row", [(1, 1, 0), (7, 3, 0), (300, 200, 0), (2000, 1500, 1000), (64, 5000, 70)])
def test_mat_vec_and_vec_mat(ffi, bp, c, n_rows, n_cols, long_row):
    cid = c.curve_id
    mat = random_matrix(c, n_rows, n_cols, 40 + n_rows, long_row)
    z = H.rand_scalars(c, n_cols, 50 + n_rows)
    y = H.rand_scalars(c, n_rows, 60 + n_rows)
    row_ptr, col_idx, vals = to_csr(c, mat)
    want = pm.mat_vec_mult(c, mat, z)
    assert H.scalars_from_abi(c, co.mat_vec_mult(cid, row_ptr, col_idx, vals, H.scalars_abi(c, z))) == want
    assert H.scalars_from_abi(c, ffi.mat_vec_mult(cid, row_ptr, col_idx, vals, H.scalars_abi(c, z))) == want
    assert H.scalars_from_abi(c, bp.mat_vec_mult(cid, row_ptr, col_idx, vals, H.scalars_abi(c, z))) == want
    for out_size in (n_cols, n_cols + 5):
        want_t = pm.vec_mat_mult(c, y, mat, out_size)
        assert H.scalars_from_abi(c, co.vec_mat_mult(cid, H.scalars_abi(c, y), row_ptr, col_idx, vals, out_size)) == want_t
        assert H.scalars_from_abi(c, ffi.vec_mat_mult(cid, H.scalars_abi(c, y), row_ptr, col_idx, vals, out_size)) == want_t
    assert H.scalars_from_abi(c, bp.vec_mat_mult(cid, H.scalars_abi(c, y), row_ptr, col_idx, vals, n_cols)) == want_t[:n_cols]


def test_mat_products_edge_cases_and_asserts(ffi, bp):
    c = pm.PALLAS
    z = H.scalars_abi(c, H.rand_scalars(c, 4, 71))
    empty = (np.zeros(1, np.uint64), np.zeros(0, np.uint64), np.zeros((0, 4), np.uint64))
    assert ffi.mat_vec_mult(0, *empty, z).shape == (0, 4)                                  # no rows
    assert not ffi.vec_mat_mult(0, np.zeros((0, 4), np.uint64), *empty, 3).any()           # zeros of the output size
    rows_only = (np.zeros(4, np.uint64), np.zeros(0, np.uint64), np.zeros((0, 4), np.uint64))
    assert not ffi.mat_vec_mult(0, *rows_only, z).any()                                    # empty rows give zero
    mat = [[(3, 0), (5, 3)], [], [(7, 3)]]
    row_ptr, col_idx, vals = to_csr(c, mat)
    with pytest.raises(ffi.BpGpuError):     # assert!(c < vec.len()), com_r1cs.rs:51
        ffi.mat_vec_mult(0, row_ptr, col_idx, vals, z[:3])
    with pytest.raises(bp.Panic):
        bp.mat_vec_mult(0, row_ptr, col_idx, vals, z[:3])
    y = H.scalars_abi(c, H.rand_scalars(c, 3, 72))
    with pytest.raises(ffi.BpGpuError):     # assert!(mat_cols(mat) <= output_size), :72
        ffi.vec_mat_mult(0, y, row_ptr, col_idx, vals, 3)
    with pytest.raises(ffi.BpGpuError):     # assert_eq!(vec.len(), mat.len()), :74
        ffi.vec_mat_mult(0, y[:2], row_ptr, col_idx, vals, 4)
    with pytest.raises(bp.Panic):
        bp.vec_mat_mult(0, y[:2], row_ptr, col_idx, vals, 4)
    bad = row_ptr.copy()
    bad[1], bad[2] = 2, 1
    with pytest.raises(ffi.BpGpuError):     # malformed compressed rows never reach a kernel
        ffi.mat_vec_mult(0, bad, col_idx, vals, z)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_curve_lift_x_matches_model(ffi, c):
    """Per-candidate half of G::rand: squares and non-squares, both values of `greatest`."""
    rng = H.seeded_rng(7700 + c.curve_id)
    xs = [c.from_mont_q(pm.rand_field_mont(rng, c.q)) for _ in range(96)] + [0, 1, c.q - 1]
    gr = [rng.gen_bool() for _ in xs]
    x_abi = np.frombuffer(b"".join(pm.mont_limbs_bytes(c.to_mont_q(x)) for x in xs), np.uint64).reshape(-1, 4)
    ys, ok = ffi.curve_lift_x(c.curve_id, x_abi, np.array(gr, np.uint8))
    n_ok = 0
    for i, x in enumerate(xs):
        want = pm.lift_x(c, x, gr[i])
        if want is None:
            assert ok[i] == 0 and not ys[i].any()
        else:
            n_ok += 1
            assert ok[i] == 1 and c.from_mont_q(int.from_bytes(ys[i].tobytes(), "little")) == want
            assert pm.is_on_curve(c, (x, want))
    assert 20 < n_ok < len(xs) - 20
    ys0, ok0 = ffi.curve_lift_x(c.curve_id, x_abi[:0], np.zeros(0, np.uint8))
    assert len(ys0) == 0 and len(ok0) == 0


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_batched_sampling_reproduces_the_sequential_stream(bp, c):
    """test_rng() recipes with the square roots on the device are byte-identical to the oracle's sequential
    G::rand loop, including the generator state afterwards (the scalars / the Fiat-Shamir seed follow)."""
    cid = c.curve_id
    for n in (0, 1, 2, 100, 3000):
        pts, sc = bp.msm_bench_inputs(cid, n)
        wp, ws = co.msm_bench_inputs(cid, n)
        assert (pts == wp).all() and (sc == ws).all(), n
    for n in (1, 17, 500):
        got = bp.ipa_inputs(cid, n)
        want = co.ipa_inputs(cid, n)
        for g, w in zip(got[:5], want[:5]):
            assert (g == w).all(), n
        assert got[5] == want[5]
