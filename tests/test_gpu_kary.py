"""GPU parity for the vector utilities of the k-ary protocols: the generalised
fold sum_vecs(chunks.zip(y).map(scale_vec)) of bp_rec_kary.rs:100-112 / ipa_to_bp_unroll.rs:248-260,
hadamard, hadamard_gp and powers (src/util/mod.rs:25-85), through the C ABI against the oracles.
Bit-exact (integer work)."""
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


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_powers_and_hadamard(ffi, c):
    cid = c.curve_id
    x = H.rand_scalars(c, 1, 900)[0]
    for f, start, end in [(x, 0, 1), (x, 0, 8), (x, 1, 12), (x, 5, 5), (x, 3, 1000), (0, 0, 4), (1, 0, 9), (c.r - 1, 0, 5),
                          (x, (1 << 40) + 3, (1 << 40) + 30)]:
        got = H.scalars_from_abi(c, ffi.powers(cid, H.scalar_abi(c, f), start, end))
        assert got == pm.powers(c, f, start, end), (start, end)
    # neg_powers = powers of the inverse (util/mod.rs:66-68)
    xi = pow(x, -1, c.r)
    assert H.scalars_from_abi(c, ffi.powers(cid, H.scalar_abi(c, xi), 0, 6)) == [pow(x, -i, c.r) for i in range(6)]
    for n in [0, 1, 3, 257, 5000]:
        a, b = H.rand_scalars(c, n, 901 + n), H.rand_scalars(c, n, 902 + n)
        if n >= 3:
            a[0], b[1], a[2], b[2] = 0, 1, c.r - 1, c.r - 1
        got = H.scalars_from_abi(c, ffi.hadamard(cid, H.scalars_abi(c, a), H.scalars_abi(c, b)))
        assert got == pm.hadamard(c, a, b), n
    with pytest.raises(AssertionError):   # assert_eq!(a.len(), b.len())
        ffi.hadamard(cid, H.scalars_abi(c, [1, 2]), H.scalars_abi(c, [1]))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fold_scalars_kary(ffi, c):
    cid = c.curve_id
    x = H.rand_scalars(c, 1, 910)[0]
    for k, ck in [(0, 0), (1, 5), (2, 4), (3, 7), (8, 1), (8, 129), (12, 1000), (100, 3), (1024, 2)]:
        v = H.rand_scalars(c, k * ck, 911 + k)
        if k * ck >= 3:
            v[0], v[1], v[-1] = 0, c.r - 1, c.r - 1
        y = pm.powers(c, x, 0, k)
        want = pm.fold_scalars_kary(c, v, k, y)
        got = H.scalars_from_abi(c, ffi.fold_scalars_kary(cid, H.scalars_abi(c, v), k, H.scalars_abi(c, y)))
        assert got == want, (k, ck)
    # arbitrary multipliers; k = 1 is scale_vec, y = (1, .., 1) is sum_vecs; 2-ary equals fold_scalars
    v = H.rand_scalars(c, 24, 920)
    for y in [H.rand_scalars(c, 4, 921), [0, 1, c.r - 1, 7], [1, 1, 1, 1]]:
        got = H.scalars_from_abi(c, ffi.fold_scalars_kary(cid, H.scalars_abi(c, v), 4, H.scalars_abi(c, y)))
        assert got == pm.fold_scalars_kary(c, v, 4, y)
    xi = pow(x, -1, c.r)
    two = ffi.fold_scalars_kary(cid, H.scalars_abi(c, v), 2, H.scalars_abi(c, [x, xi]))
    assert (two == ffi.fold_scalars(cid, H.scalars_abi(c, v), H.scalar_abi(c, x), H.scalar_abi(c, xi))).all()
    with pytest.raises(Exception):
        ffi.fold_scalars_kary(cid, H.scalars_abi(c, [1] * 1025), 1025, H.scalars_abi(c, [1] * 1025))
    # device-resident form, in place
    k, ck = 5, 300
    v = H.rand_scalars(c, k * ck, 930)
    y = pm.powers(c, xi, 0, k)
    d = ffi.scalars_to_device(H.scalars_abi(c, v))
    ffi.fold_scalars_kary_device(cid, d, ck, k, H.scalars_abi(c, y), d)
    got = d.download(np.uint64, ck * 4).reshape(-1, 4)
    d.free()
    assert H.scalars_from_abi(c, got) == pm.fold_scalars_kary(c, v, k, y)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fold_points_kary_small(ffi, c):
    """Against the big-integer model: powers (chunk 0 added as it is), arbitrary multipliers, k = 2 (k_fold_monic),
    identity points from zero padding, repeated points (doubling inside the fold)."""
    cid = c.curve_id
    x = H.rand_scalars(c, 1, 940)[0]
    for k, ck in [(1, 3), (2, 2), (3, 3), (4, 2), (7, 1)]:
        g = H.rand_points(c, k * ck, 941 + k)
        if k * ck >= 4:
            g[1] = None
            g[-1] = None
            g[2] = g[0]
        for y in (pm.powers(c, x, 0, k), H.rand_scalars(c, k, 942 + k), [1] * k, [0] * k):
            want = [pm.jac_to_affine(c, p) for p in pm.fold_points_kary(c, [pm.to_jac(p) for p in g], k, y)]
            got = H.jac_from_abi(c, ffi.fold_points_kary(cid, H.jac_abi(c, g), k, H.scalars_abi(c, y)))
            assert got == want, (k, ck, y[:2])
    # edge multipliers on the shared chain: -1, 0, 1, the cube roots of unity (GLV halves (0, +-1)), 2, -2; every chunk the
    # same point (the chain adds a point to its own multiple) and an all-identity chunk
    lam = next(pow(gg, (c.r - 1) // 3, c.r) for gg in range(2, 50) if pow(gg, (c.r - 1) // 3, c.r) != 1)
    y = [c.r - 1, 0, 1, lam, lam * lam % c.r, 2, c.r - 2, c.r - lam]
    k, ck = len(y), 3
    g = H.rand_points(c, k * ck, 977)
    for j in range(1, k):
        g[j * ck] = g[0]
    g[2 * ck:3 * ck] = [None] * ck
    for yy in (y, [1] + y[1:]):
        want = [pm.jac_to_affine(c, p) for p in pm.fold_points_kary(c, [pm.to_jac(p) for p in g], k, yy)]
        got = H.jac_from_abi(c, ffi.fold_points_kary(cid, H.jac_abi(c, g), k, H.scalars_abi(c, yy)))
        assert got == want
    # sum_vecs of nothing: identities
    assert ffi.fold_points_kary(cid, np.zeros((0, 12), np.uint64), 0, np.zeros((0, 4), np.uint64)).shape == (0, 12)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fold_points_kary_and_hadamard_gp_vs_c_oracle(ffi, c):
    cid = c.curve_id
    pts, sc = co.msm_bench_inputs(cid, 8 * 300)
    x = sc[0]
    xi = co.scalar_inverse(cid, x)
    # (k = 12: two launches of the shared-chain fold, 7 + 4 multipliers; k = 17 with arbitrary y: 7 + 7 + 3)
    for k, ck in [(8, 300), (3, 257), (12, 200), (17, 140)]:
        g = pts[:k * ck].copy()
        g[5] = co.normalise(cid, np.zeros((1, 12), np.uint64))[0]      # identity (zero_pad_to_multiple)
        for y in (co.powers(cid, xi, 0, k), sc[10:10 + k]):
            want = co.fold_points_kary(cid, g, k, y, threads=8)
            got = ffi.fold_points_kary(cid, g, k, y)
            assert (got == want).all(), (k, ck)
    # 2-ary equals the prover's generator fold (bp_iter.rs:97-104)
    g = pts[:512]
    assert (ffi.fold_points_kary(cid, g, 2, np.stack([xi, x])) == ffi.fold_points(cid, g, xi, x)).all()
    # device-resident form
    k, ck = 4, 130
    d_g = ffi.points_to_device(cid, pts[:k * ck])
    d_o = ffi.DeviceBuffer(ck * 64)
    y = co.powers(cid, x, 0, k)
    ffi.fold_points_kary_device(cid, d_g, ck, k, y, d_o)
    assert (ffi.points_from_device(cid, d_o, ck) == co.fold_points_kary(cid, pts[:k * ck], k, y, threads=8)).all()
    with pytest.raises(Exception):   # output inside the input
        ffi.fold_points_kary_device(cid, d_g, ck, k, y, d_g)
    d_g.free()
    d_o.free()
    # hadamard_gp: every element its own multiplier (com_r1cs_to_ipa.rs:110 uses neg_powers)
    n = 600
    s = co.powers(cid, xi, 0, n)
    s[3] = 0
    g = pts[:n].copy()
    g[7] = co.normalise(cid, np.zeros((1, 12), np.uint64))[0]
    assert (ffi.hadamard_points(cid, g, s) == co.hadamard_points(cid, g, s, threads=8)).all()
    es = H.scalars_abi(c, [0, 1, 2, c.r - 1, (c.r - 1) // 2, 1 << 254 if (1 << 254) < c.r else 5])
    assert (ffi.hadamard_points(cid, pts[:6], es) == co.hadamard_points(cid, pts[:6], es)).all()
    # the windowed GLV ladder (2P, 3P, 4P table, signed 3-bit digits): small multiples hit the table entries themselves
    # (accumulator == table point: the doubling branch), the cube roots of unity split into (0, +-1), and -small
    lam = next(pow(gg, (c.r - 1) // 3, c.r) for gg in range(2, 50) if pow(gg, (c.r - 1) // 3, c.r) != 1)
    sm = [3, 4, 5, 7, 8, 9, 12, 16, 31, 32, 33, c.r - 2, c.r - 3, c.r - 4, c.r - 5, c.r - 8, lam, lam * lam % c.r, c.r - lam,
          (lam + 1) % c.r, 2 * lam % c.r, 4 * lam % c.r, (3 * lam + 4) % c.r, (1 << 128) - 1, 1 << 128, (1 << 129) + 5]
    es = H.scalars_abi(c, sm)
    gp = np.concatenate([pts[:len(sm) - 1], pts[:1]])  # the last element repeats a base
    assert (ffi.hadamard_points(cid, gp, es) == co.hadamard_points(cid, gp, es)).all()


@pytest.mark.parametrize("k,ck", [(4, 40_000), (3, 20_000), (8, 9_000)], ids=["one-lane", "two-lanes", "four-lanes"])
def test_fold_points_kary_lane_splits_vs_c_oracle(ffi, k, ck):
    """The shared-chain fold gives short vectors 2 or 4 lanes an output (ck <= 2^15 / 2^14 on a 256-CU chip) and long
    ones a lane each: one size per schedule against Oracle B, with an identity generator and a repeated one."""
    from bulletproofs_amd import ark_bp
    cid = 0
    pts, sc = ark_bp.msm_bench_inputs(cid, k * ck)
    g = pts.copy()
    g[ck + 7] = co.normalise(cid, np.zeros((1, 12), np.uint64))[0]
    g[2 * ck + 11] = g[11]
    for y in (co.powers(cid, sc[0], 0, k), sc[3:3 + k]):
        want = co.fold_points_kary(cid, g, k, y, threads=16)
        assert (ffi.fold_points_kary(cid, g, k, y) == want).all()


def _cross_terms_by_the_reference_recipe(c, cid, A, B, q, a, b, k):
    """pos / neg exactly as ipa_to_bp_unroll.rs:189-215 writes them: k (k - 1) separate MSMs X[i, j] (Oracle B), summed
    (big-integer model's affine addition)."""
    ck = len(a) // k

    def ch(v, i):
        return v[i * ck:(i + 1) * ck]

    def x(i, j):
        pts = np.concatenate([ch(A, j), ch(B, i), q.reshape(1, 12)])
        sc = np.concatenate([ch(a, i), ch(b, j), co.ip(cid, ch(a, i), ch(b, j)).reshape(1, 4)])
        r = co.msm_pippenger(cid, pts, sc, threads=8)
        return H.jac_from_abi(c, r.reshape(1, 12))[0]

    def total(ps):
        acc = None
        for p in ps:
            acc = pm.aff_add(c, acc, p)
        return acc
    pos = [total([x(i + j + 1, j) for j in range(k - i - 1)]) for i in range(k - 1)]
    neg = [total([x(j, i + j + 1) for j in range(k - i - 1)]) for i in range(k - 1)]
    return pos, neg


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,k", [(1024, 8), (60, 3), (38, 19), (4, 2), (7, 7)])
def test_kary_cross_terms_batched(ffi, c, n, k):
    """bpgpu_kary_cross_terms (one batch of 2 (k - 1) merged MSMs; two batches at k = 19) against the reference's recipe of
    k (k - 1) separate MSMs: identical group elements, with zero padding (identity points, zero scalars) and a
    duplicated generator among the operands."""
    cid = c.curve_id
    A, B, q, a, b, _ = co.ipa_inputs(cid, n)
    ident = co.normalise(cid, np.zeros((1, 12), np.uint64))
    if n >= 38:
        A, a = A.copy(), a.copy()
        A[n - 1], a[n - 1] = ident[0], 0        # zero_pad_to_multiple leaves identities / zeros at the end
        A[3] = A[2]
        a[5] = 0
    pos, neg = ffi.kary_cross_terms(cid, A, B, q, a, b, k)
    want_pos, want_neg = _cross_terms_by_the_reference_recipe(c, cid, A, B, q, a, b, k)
    assert [H.xy_inf_to_affine(c, p) for p in pos] == want_pos
    assert [H.xy_inf_to_affine(c, p) for p in neg] == want_neg
    # k = 1: no cross terms, nothing written
    p1, n1 = ffi.kary_cross_terms(cid, A, B, q, a, b, 1)
    assert p1.shape == (0, 9) and n1.shape == (0, 9)


def test_kary_cross_terms_piped_upload(ffi):
    """Generators of 8 MiB and more go up in chunks while the inner products and the batch's sort run (the accumulation
    waits for the last chunk): n = 3 * 2^15, k = 3, one generator with Z = 3 (its chunk goes up as Jacobian rows),
    against the reference's recipe of k (k - 1) separate MSMs."""
    from bulletproofs_amd import ark_bp
    c, cid, n, k = pm.PALLAS, 0, 3 << 15, 3
    A, B, q, a, b, _ = ark_bp.ipa_inputs(cid, n)
    B = B.copy()
    xb, yb = H.jac_from_abi(c, B[70_000:70_001])[0]
    B[70_000] = H.jac_abi(c, [(xb * 9 % c.q, yb * 27 % c.q, 3)])[0]
    Bn = B.copy()
    Bn[70_000] = H.jac_abi(c, [(xb, yb)])[0]
    pos, neg = ffi.kary_cross_terms(cid, A, B, q, a, b, k)
    want_pos, want_neg = _cross_terms_by_the_reference_recipe(c, cid, A, Bn, q.reshape(12), a, b, k)
    assert [H.xy_inf_to_affine(c, p) for p in pos] == want_pos
    assert [H.xy_inf_to_affine(c, p) for p in neg] == want_neg
