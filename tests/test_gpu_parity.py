"""GPU parity tests proper: the HIP path, called through the C ABI, against the
big-integer oracle on the same seeded inputs.  Bit-exact (integer work)."""
import numpy as np
import pytest

from oracle import pasta_model as pm
from tests import helpers as H

pytestmark = pytest.mark.gpu

CURVES = [pm.PALLAS, pm.VESTA]


@pytest.fixture(scope="module")
def ffi():
    from bulletproofs_amd import ffi as f
    f.init(0)
    yield f
    f.set_window_bits(0)


def edge_scalars(c):
    r = c.r
    return [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, 1 << 15, (1 << 15) - 1, (1 << 16) + 1, (1 << 254) % r,
            (1 << 253) - 1, 0x8000800080008000800080008000800080008000800080008000800080008000 % r,
            0x7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff7fff % r]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_ip_and_fold_scalars(ffi, c):
    for n in [0, 1, 2, 3, 64, 257, 1000, 4096 + 17]:
        a = H.rand_scalars(c, n, 11 + n)
        b = H.rand_scalars(c, n, 12 + n)
        if n >= 3:
            a[0], b[0], a[1], b[2] = c.r - 1, c.r - 1, 0, 1
        got = H.scalars_from_abi(c, ffi.ip(c.curve_id, H.scalars_abi(c, a), H.scalars_abi(c, b)))[0]
        assert got == pm.ip(c, a, b), n
    for n in [1, 2, 5, 64, 1000]:
        v = H.rand_scalars(c, 2 * n, 21 + n)
        v[0] = 0
        v[-1] = c.r - 1
        x = H.rand_scalars(c, 1, 5)[0]
        xi = pow(x, -1, c.r)
        got = H.scalars_from_abi(c, ffi.fold_scalars(c.curve_id, H.scalars_abi(c, v), H.scalar_abi(c, x), H.scalar_abi(c, xi)))
        assert got == pm.fold_scalars(c, v, x, xi), n


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_small(ffi, c):
    cid = c.curve_id
    for n in [0, 1, 2, 3, 31, 32, 33, 257]:
        pts = H.rand_points(c, n, 100 + n)
        xs = H.rand_scalars(c, n, 200 + n)
        es = edge_scalars(c)
        for i in range(min(n, len(es))):
            xs[i] = es[i]
        if n >= 33:
            pts[5] = None                      # identity among the bases (zero padding, bp_iter.rs:59-62)
            pts[7] = pts[6]                    # duplicate base
            xs[7] = xs[6]                      # ... in the same buckets -> doubling inside a bucket
            pts[9] = pm.aff_neg(c, pts[8])     # P and -P with equal scalars -> bucket cancels to identity
            xs[9] = xs[8]
        want = pm.jac_to_affine(c, pm.msm_pippenger(c, [pm.to_jac(p) for p in pts], xs))
        got = H.xy_inf_to_affine(c, ffi.msm(cid, H.jac_abi(c, pts), H.scalars_abi(c, xs)))
        assert got == want, n
        got = H.xy_inf_to_affine(c, ffi.msm_affine(cid, H.aff_abi(c, pts), H.scalars_abi(c, xs)))
        assert got == want, n


def test_msm_window_sweep_and_jacobian_inputs(ffi):
    c = pm.PALLAS
    n = 300
    pts = H.rand_points(c, n, 42)
    xs = H.rand_scalars(c, n, 43)
    # non-trivial Z: scale (x, y, 1) -> (x z^2, y z^3, z)
    zs = H.rand_scalars(pm.VESTA, n, 44)  # Vesta's scalar field = Pallas' base field
    jac = [(p[0] * z * z % c.q, p[1] * z * z * z % c.q, z) for p, z in zip(pts, zs)]
    jac[3] = (5, 7, 0)  # identity with junk X, Y
    pts[3] = None
    want = pm.jac_to_affine(c, pm.msm_naive(c, [pm.to_jac(p) for p in pts], xs))
    for bits in [0, 2, 3, 5, 8, 11, 13, 16]:
        ffi.set_window_bits(bits)
        got = H.xy_inf_to_affine(c, ffi.msm(0, H.jac_abi(c, jac), H.scalars_abi(c, xs)))
        assert got == want, bits
    ffi.set_window_bits(0)
    # zip truncation (msm.rs:94-97) and all-zero scalars
    got = H.xy_inf_to_affine(c, ffi.msm(0, H.jac_abi(c, pts), H.scalars_abi(c, xs[:10])))
    assert got == pm.jac_to_affine(c, pm.msm_naive(c, [pm.to_jac(p) for p in pts[:10]], xs[:10]))
    assert H.xy_inf_to_affine(c, ffi.msm(0, H.jac_abi(c, pts), H.scalars_abi(c, [0] * n))) is None


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fold_points(ffi, c):
    n = 37
    g = H.rand_points(c, 2 * n, 77)
    g[0] = None                     # padding on the low side
    g[n + 1] = None                 # padding on the high side
    g[2] = None
    g[n + 2] = None                 # both identity
    g[n + 3] = g[3]                 # P == Q
    g[n + 4] = pm.aff_neg(c, g[4])  # P == -Q
    x = H.rand_scalars(c, 1, 9)[0]
    xi = pow(x, -1, c.r)
    for s_lo, s_hi in [(xi, x), (x, xi), (1, c.r - 1), (0, 5), (3, 0), (0, 0), (c.r - 1, c.r - 1)]:
        want = [pm.jac_to_affine(c, p) for p in pm.fold_points(c, [pm.to_jac(p) for p in g], s_lo, s_hi)]
        out = ffi.fold_points(c.curve_id, H.jac_abi(c, g), H.scalar_abi(c, s_lo), H.scalar_abi(c, s_hi))
        assert H.jac_from_abi(c, out) == want, (s_lo, s_hi)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 17, 5])
def test_bp_prove_session(ffi, c, n):
    """Bp::prove through the device-resident session vs the oracle's bp_prove: identical
    ls, rs, challenges, final_a, final_b (reference test sizes, src/lib.rs:188,257)."""
    rng = pm.test_rng()
    inst, wit = pm.sample_ipa(c, rng, n)
    fs_seed = rng.gen_bytes32()
    want = pm.bp_prove(c, inst, wit, pm.FiatShamirRng(fs_seed))

    fs = pm.FiatShamirRng(fs_seed)
    sess = ffi.IpaSession(c.curve_id, H.jac_abi(c, inst.a_gens), H.jac_abi(c, inst.b_gens), H.jac_abi(c, [inst.ip_gen]),
                          H.scalars_abi(c, wit.a), H.scalars_abi(c, wit.b))
    ls, rs, challs = [], [], []
    while len(sess) > 1:
        l, r = sess.round_lr()
        l, r = H.xy_inf_to_affine(c, l), H.xy_inf_to_affine(c, r)
        ls.append(l)
        rs.append(r)
        fs.absorb(pm.point_bytes_normalised(l))
        fs.absorb(pm.point_bytes_normalised(r))
        x = fs.rand_scalar(c)
        challs.append(x)
        sess.round_fold(H.scalar_abi(c, x), H.scalar_abi(c, pow(x, -1, c.r)))
    fa, fb = sess.finish()
    A, B, a, b = sess.export()
    sess.free()
    assert ls == want.ls and rs == want.rs and challs == want.challenges
    assert H.scalars_from_abi(c, fa)[0] == want.final_a and H.scalars_from_abi(c, fb)[0] == want.final_b
    # the proof verifies (reference: bp_iter.rs:130-178)
    assert pm.bp_verify(c, inst, want, pm.FiatShamirRng(fs_seed))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_skewed_scalars(ffi, c):
    """Skewed digit distributions (heavy buckets): repeated scalars, tiny scalars, all ones, scalars that
    differ only in their low bits, and window widths whose top window holds only a few values."""
    from oracle import c_oracle as co
    cid = c.curve_id
    n = 6000
    pts = co.gen_points(cid, 77, n, threads=4)
    pts[17] = pts[16]
    rs = np.random.default_rng(5)
    base = H.rand_scalars(c, 4, 31)
    cases = {
        "same": [base[0]] * n,
        "four values": [base[i % 4] for i in range(n)],
        "tiny": [int(v) for v in rs.integers(0, 1 << 16, n)],
        "ones": [1] * n,
        "minus ones": [c.r - 1] * n,
        "low bits differ": [(base[1] & ~0xFFFF) | int(v) for v in rs.integers(0, 1 << 16, n)],
        "zeros and one": [0] * (n - 1) + [base[2]],
    }
    for name, xs in cases.items():
        sc = H.scalars_abi(c, xs)
        want = co.msm_pippenger(cid, pts, sc, threads=4)
        for bits in (0, 12, 14):
            ffi.set_window_bits(bits)
            got = ffi.msm(cid, pts, sc)
            ok = (int(got[8]) == 0 and (got[:8] == want[:8]).all()) or (int(got[8]) == 1 and not want[8:].any())
            assert ok, (name, bits)
    ffi.set_window_bits(0)


def _xy(a):
    return None if int(a[8]) else a[:8].tobytes()


def test_msm_bench_recipe_2_16(ffi):
    """BASELINE config 1/2 plumbing: the reference's own input recipe (examples/msm_bench.rs:40-44, one
    test_rng stream: points then scalars) at 2^16, GPU vs the CPU restatement of pippenger_msm."""
    from oracle import c_oracle as co
    n = 1 << 16
    pts, sc = co.msm_bench_inputs(0, n)
    want = co.msm_pippenger(0, pts, sc, threads=16)
    got = ffi.msm(0, pts, sc)
    assert int(got[8]) == 0 and (got[:8] == want[:8]).all()
    # device-resident entry point, same answer
    d_p, d_s = ffi.points_to_device(0, pts), ffi.scalars_to_device(sc)
    assert (ffi.msm_device(0, d_p, d_s, n) == got).all()
    # round trip of the device point format
    assert (ffi.points_from_device(0, d_p, 1000) == pts[:1000]).all()


def test_msm_full_size_properties(ffi):
    """At BASELINE's full size (2^20) the oracle is too slow for a unit test, so use size-independent
    properties: linearity in the scalars, additivity over a split of the points (the multi-GPU shard
    identity), invariance under a permutation of the pairs, and a 2^18 slice against the oracle."""
    import bench
    from oracle import c_oracle as co
    n = 1 << 20
    pts = bench.synth_points(ffi, 0, n, seed=11)
    assert co.on_curve(0, pts[:: n // 64])
    s, t = bench.synth_scalars(n, 12), bench.synth_scalars(n, 13)
    d_p = ffi.points_to_device(0, pts)
    ms = ffi.msm_device(0, d_p, ffi.scalars_to_device(s), n)
    mt = ffi.msm_device(0, d_p, ffi.scalars_to_device(t), n)
    # s + t mod r on the host (Montgomery form is linear): python ints, vectorised by limbs is overkill here
    r = pm.PALLAS.r
    si = [int.from_bytes(row.tobytes(), "little") for row in s]
    ti = [int.from_bytes(row.tobytes(), "little") for row in t]
    st = np.frombuffer(b"".join(((a + b) % r).to_bytes(32, "little") for a, b in zip(si, ti)), np.uint64).reshape(-1, 4)
    mst = ffi.msm_device(0, d_p, ffi.scalars_to_device(st), n)
    assert _xy(ffi.points_sum(0, np.stack([ms, mt]))) == _xy(mst)
    # shard identity: the sum of the MSMs over two point ranges is the MSM over all points
    h = 3 * n // 8
    parts = [ffi.msm(0, pts[:h], s[:h]), ffi.msm(0, pts[h:], s[h:])]
    assert _xy(ffi.points_sum(0, np.stack(parts))) == _xy(ms)
    # permutation invariance
    perm = np.random.default_rng(1).permutation(n)
    assert _xy(ffi.msm(0, pts[perm], s[perm])) == _xy(ms)
    # a 2^18 slice against the CPU restatement
    k = 1 << 18
    want = co.msm_pippenger(0, pts[:k], s[:k], threads=16)
    got = ffi.msm(0, pts[:k], s[:k])
    assert int(got[8]) == 0 and (got[:8] == want[:8]).all()


def test_msm_2_24_sharded_like_eight_gpus(ffi):
    """BASELINE config 3 on one GPU: 2^24 pairs by the reference's msm_bench recipe (examples/msm_bench.rs:40-44,
    square roots on the device), cut into the eight contiguous point ranges the ranks of one node own
    (bulletproofs_amd/sharded.py); the sum of the eight 2^21-point MSMs (what all_gather + bpgpu_points_sum
    produce) equals the single 2^24-point MSM, which runs with another window width and bucket layout.  A shard
    prefix small enough for the CPU restatement is compared bit for bit."""
    from bulletproofs_amd import ark_bp, sharded
    from oracle import c_oracle as co
    n, world = 1 << 24, 8
    pts, sc = ark_bp.msm_bench_inputs(0, n)
    assert co.on_curve(0, pts[:: n // 64])
    whole = ffi.msm(0, pts, sc)
    parts = []
    for rank in range(world):
        lo, hi = sharded.shard_range(n, rank, world)
        assert hi - lo == 1 << 21
        parts.append(ffi.msm(0, pts[lo:hi], sc[lo:hi]))
    assert _xy(ffi.points_sum(0, np.stack(parts))) == _xy(whole) and int(whole[8]) == 0
    k = 1 << 17
    lo, _ = sharded.shard_range(n, 5, world)
    want = co.msm_pippenger(0, pts[lo:lo + k], sc[lo:lo + k], threads=16)
    got = ffi.msm(0, pts[lo:lo + k], sc[lo:lo + k])
    assert int(got[8]) == 0 and (got[:8] == want[:8]).all()


@pytest.mark.parametrize("table", [0, 2], ids=["notable", "table"])
@pytest.mark.parametrize("pair", [True, False], ids=["pairfold", "roundfold"])
@pytest.mark.parametrize("freeze", [0, 2, 8, 64, 1 << 15])
def test_bp_prove_deferred_fold_equivalence(ffi, freeze, pair, table):
    """Deferring the generator fold (frozen generators + challenge-product weights), folding two rounds
    at a time (k_fold_joint4) and running the short rounds over a fixed-base table of the frozen generators
    (one bucket set for all windows, several lanes per bucket) change nothing: same L, R, final scalars and
    exported generators for every freeze length, with and without pair folding, against the oracle."""
    from oracle import c_oracle as co
    c = pm.PALLAS
    n = 64
    A, B, q, a, b, seed = co.ipa_inputs(0, n)
    ls, rs, ch, fa, fb = co.bp_prove(0, A, B, q, a, b, seed, threads=4)
    ffi.set_freeze_len(freeze)
    ffi.set_pair_fold(pair)
    ffi.set_ipa_table(table)
    try:
        sess = ffi.IpaSession(0, A, B, q, a, b)
        k = 0
        while len(sess) > 1:
            l, r = sess.round_lr()
            assert (l[:8] == ls[k][:8]).all() and (r[:8] == rs[k][:8]).all(), (freeze, k)
            sess.round_fold(ch[k], co.scalar_inverse(0, ch[k]))
            k += 1
            if len(sess) == 4:
                # export in the middle of a (possibly frozen) session: folded generators are materialised
                gA, gB, ga, gb = sess.export()
                wantA = A
                wantB = B
                for j in range(k):
                    xi = co.scalar_inverse(0, ch[j])
                    wantA = co.fold_points(0, wantA, xi, ch[j])
                    wantB = co.fold_points(0, wantB, ch[j], xi)
                assert (gA == wantA).all() and (gB == wantB).all(), freeze
        ga, gb = sess.finish()
        sess.free()
        assert (ga == fa).all() and (gb == fb).all()
    finally:
        ffi.reset_tuning()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,freeze", [(2048, 16), (1000, 0), (4096, 256), (512, 2), (4096, 1 << 15), (1000, 1 << 15)])
def test_pair_folding_whole_proofs(ffi, c, n, freeze):
    """Sizes the oracle still proves in seconds: the proof with pair folding (two challenges applied by one
    k_fold_joint4 pass, odd and even numbers of long rounds, zero padding at n = 1000) equals the proof
    with one fold per round and the oracle's."""
    from oracle import c_oracle as co
    cid = c.curve_id
    A, B, q, a, b, seed = co.ipa_inputs(cid, n)
    np_ = 1 << (n - 1).bit_length()
    ident = co.normalise(cid, np.zeros((1, 12), np.uint64))
    Ap, Bp = np.concatenate([A, np.tile(ident, (np_ - n, 1))]), np.concatenate([B, np.tile(ident, (np_ - n, 1))])
    zs = np.zeros((np_ - n, 4), np.uint64)
    ls, rs, ch, fa, fb = co.bp_prove(cid, Ap, Bp, q, np.concatenate([a, zs]), np.concatenate([b, zs]), seed, threads=8)
    ffi.set_freeze_len(freeze)
    try:
        for pair, table in ((True, 2), (False, 2), (True, 0)):
            ffi.set_pair_fold(pair)
            ffi.set_ipa_table(table)  # a table whenever the generators freeze for good (also from the start)
            sess = ffi.IpaSession(cid, A, B, q, a, b)
            k = 0
            while len(sess) > 1:
                l, r = sess.round_lr()
                assert (l[:8] == ls[k][:8]).all() and (r[:8] == rs[k][:8]).all(), (pair, k)
                sess.round_fold(ch[k], co.scalar_inverse(cid, ch[k]))
                k += 1
            ga, gb = sess.finish()
            sess.free()
            assert (ga == fa).all() and (gb == fb).all(), pair
    finally:
        ffi.reset_tuning()


@pytest.mark.parametrize("pair", [False, True], ids=["roundfold", "pairfold"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_session_monic_fold_edge_cases(ffi, c, pair):
    """The session folds generators as A_lo + x^2 A_hi with the factor x^-1 carried as a scalar
    (k_fold_monic): L, R and the exported generators must equal the reference's
    x^-1 A_lo + x A_hi (bp_iter.rs:97-104) for awkward challenges -- 1, -1, both cube roots of
    unity (x^2 is then a unit of the GLV/Eisenstein expansion), a pair that is NOT an inverse pair
    (two-scalar fallback) -- with identity padding and P == +-Q pairs among the generators."""
    n = 32
    lam = next(pow(g, (c.r - 1) // 3, c.r) for g in range(2, 50) if pow(g, (c.r - 1) // 3, c.r) != 1)
    A = H.rand_points(c, n, 301)
    B = H.rand_points(c, n, 302)
    A[1] = None
    A[n // 2 + 2] = None
    A[3] = None
    A[n // 2 + 3] = None
    A[n // 2 + 4] = A[4]
    A[n // 2 + 5] = pm.aff_neg(c, A[5])
    B[n // 2 + 6] = B[6]
    q = H.rand_points(c, 1, 303)
    a = H.rand_scalars(c, n, 304)
    b = H.rand_scalars(c, n, 305)
    rnd = H.rand_scalars(c, 1, 306)[0]
    pairs = [(1, 1), (c.r - 1, c.r - 1), (lam, lam * lam % c.r), (rnd, pow(rnd, -1, c.r)), (3, 5)]
    ffi.set_freeze_len(0)
    ffi.set_pair_fold(pair)
    try:
        sess = ffi.IpaSession(c.curve_id, H.jac_abi(c, A), H.jac_abi(c, B), H.jac_abi(c, q), H.scalars_abi(c, a),
                              H.scalars_abi(c, b))
        jA, jB = [pm.to_jac(p) for p in A], [pm.to_jac(p) for p in B]
        for k, (x, xi) in enumerate(pairs):
            m = len(a) // 2
            c_l, c_r = pm.ip(c, a[:m], b[m:]), pm.ip(c, a[m:], b[:m])
            want_l = pm.msm_naive(c, jA[m:] + jB[:m] + [pm.to_jac(q[0])], a[:m] + b[m:] + [c_l])
            want_r = pm.msm_naive(c, jA[:m] + jB[m:] + [pm.to_jac(q[0])], a[m:] + b[:m] + [c_r])
            l, r = sess.round_lr()
            assert H.xy_inf_to_affine(c, l) == pm.jac_to_affine(c, want_l), k
            assert H.xy_inf_to_affine(c, r) == pm.jac_to_affine(c, want_r), k
            sess.round_fold(H.scalar_abi(c, x), H.scalar_abi(c, xi))
            a, b = pm.fold_scalars(c, a, x, xi), pm.fold_scalars(c, b, xi, x)
            jA, jB = pm.fold_points(c, jA, xi, x), pm.fold_points(c, jB, x, xi)
            if k in (1, 2, 4):
                gA, gB, ga, gb = sess.export()
                assert H.jac_from_abi(c, gA) == [pm.jac_to_affine(c, p) for p in jA], k
                assert H.jac_from_abi(c, gB) == [pm.jac_to_affine(c, p) for p in jB], k
                assert H.scalars_from_abi(c, ga) == a and H.scalars_from_abi(c, gb) == b
        sess.free()
    finally:
        ffi.reset_tuning()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 3, 8, 21, 64])
def test_verifier_msm(ffi, c, n):
    """bpgpu_ipa_verify_msm against the model's right-hand side of Bp::verify (bp_iter.rs:130-178):
    scalars_from_challenges expanded on the device, generators zero-padded to a power of two.  The
    value must match for an honest proof (= instance.result) and for a tampered one (some other point)."""
    cid = c.curve_id
    inst, wit = pm.sample_ipa(c, pm.test_rng(), n)
    proof = pm.bp_prove(c, inst, wit, pm.FiatShamirRng(b"\x07" * 32))
    A, B, q = H.jac_abi(c, inst.a_gens), H.jac_abi(c, inst.b_gens), H.jac_abi(c, [inst.ip_gen])

    def model_rhs(pf, challs):
        Ap = pm.zero_pad_to_two_power([pm.to_jac(g) for g in inst.a_gens], pm.JAC_INF)
        Bp = pm.zero_pad_to_two_power([pm.to_jac(g) for g in inst.b_gens], pm.JAC_INF)
        ci = [pow(x, -1, c.r) for x in challs]
        s, si = pm.scalars_from_challenges(c, challs), pm.scalars_from_challenges(c, ci)
        pts = Ap + Bp + [pm.to_jac(inst.ip_gen)]
        scs = [v * pf.final_a % c.r for v in s] + [v * pf.final_b % c.r for v in si] + [pf.final_a * pf.final_b % c.r]
        for j in range(len(pf.ls)):
            pts += [pm.to_jac(pf.ls[j]), pm.to_jac(pf.rs[j])]
            scs += [(-challs[j] * challs[j]) % c.r, (-ci[j] * ci[j]) % c.r]
        return pm.jac_to_affine(c, pm.msm_naive(c, pts, scs))

    def gpu_rhs(pf, challs, **kw):
        return H.xy_inf_to_affine(c, ffi.ipa_verify_msm(cid, A, B, q, H.jac_abi(c, pf.ls), H.jac_abi(c, pf.rs),
                                                        H.scalars_abi(c, challs), H.scalar_abi(c, pf.final_a),
                                                        H.scalar_abi(c, pf.final_b), **kw))

    assert gpu_rhs(proof, proof.challenges) == inst.result == model_rhs(proof, proof.challenges)
    bad = pm.BpProof(proof.ls, proof.rs, proof.final_b, (proof.final_a + 1) % c.r, proof.challenges)
    other = [(x * 3 + 1) % c.r for x in proof.challenges]
    assert gpu_rhs(bad, other) == model_rhs(bad, other)
    assert model_rhs(bad, other) != inst.result
    # generators already in HBM
    d_a, d_b, d_q = ffi.points_to_device(cid, A), ffi.points_to_device(cid, B), ffi.points_to_device(cid, q)
    assert gpu_rhs(proof, proof.challenges, d_gens=(d_a, d_b, d_q, n)) == inst.result
    for d in (d_a, d_b, d_q):
        d.free()
    if len(proof.ls):   # wrong number of rounds, zero challenge: errors (the reference panics)
        with pytest.raises(Exception):
            ffi.ipa_verify_msm(cid, A, B, q, H.jac_abi(c, proof.ls[:-1]), H.jac_abi(c, proof.rs[:-1]),
                               H.scalars_abi(c, proof.challenges[:-1]), H.scalar_abi(c, 1), H.scalar_abi(c, 1))
        with pytest.raises(Exception):
            gpu_rhs(proof, [0] * len(proof.challenges))
