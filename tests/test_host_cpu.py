"""No-GPU tests of the C++ host side (bulletproofs_amd/host/ark_bp.hpp): its Fiat-Shamir transcript
(Blake2s + ChaCha20 + rejection sampling) and scalar inversion against the oracle."""
import numpy as np
import pytest

from bulletproofs_amd import ark_bp
from oracle import pasta_model as pm
from tests import helpers as H


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_fiat_shamir_matches_oracle(c):
    pts = H.rand_points(c, 5, 91) + [None]
    for k in range(0, 7):
        seed = bytes((7 * i + k) & 0xFF for i in range(32))
        fs = pm.FiatShamirRng(seed)
        for p in pts[:k]:
            fs.absorb(pm.point_bytes_normalised(p))
        want = fs.rand_scalar(c)
        got = ark_bp.fs_challenge(c.curve_id, seed, H.jac_abi(c, pts[:k]))
        assert H.scalars_from_abi(c, got)[0] == want


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_scalar_inverse_and_zero_panics(c):
    for x in H.rand_scalars(c, 5, 3) + [1, c.r - 1]:
        got = H.scalars_from_abi(c, ark_bp.scalar_inverse(c.curve_id, H.scalar_abi(c, x)))[0]
        assert got == pow(x, -1, c.r)
    with pytest.raises(ark_bp.Panic):
        ark_bp.scalar_inverse(c.curve_id, np.zeros(4, np.uint64))   # x.inverse().unwrap() on zero


# the reference's own exact-value tests (src/util/mod.rs:154-201, Pallas Fr) on the host mirror
@pytest.mark.parametrize("xs,k,want", [([1, 2, 3, 4], 3, [1, 2, 3, 0, 4, 0]), ([1, 2, 3, 4, 5, 6], 3, [1, 2, 3, 4, 5, 6]),
                                       ([1, 2, 3, 4, 5], 3, [1, 2, 3, 4, 5, 0])])
def test_zero_intersperse_reference_known_answers(xs, k, want):
    c = pm.PALLAS
    got = ark_bp.pad(1, H.scalars_abi(c, xs), k)
    assert (got == H.scalars_abi(c, want)).all()


def test_zero_pad_to_multiple_host():
    c = pm.PALLAS
    for n, k in [(1, 3), (3, 3), (4, 3), (7, 4), (5, 8)]:
        xs = list(range(1, n + 1))
        assert H.scalars_from_abi(c, ark_bp.pad(0, H.scalars_abi(c, xs), k)) == pm.zero_pad_to_multiple(xs, k, 0)
    with pytest.raises(ark_bp.Panic):   # (len - 1) underflows in the reference
        ark_bp.pad(0, np.zeros((0, 4), np.uint64), 3)


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_bos_coster_msm_host(c):
    """util::msm::bos_coster_msm (src/util/msm.rs:34-81) of the host mirror against the model's Bos-Coster and naive MSM:
    zero scalars skipped, equal scalars, identity bases, a single entry, the empty input."""
    for n in [0, 1, 2, 3, 17, 64]:
        pts = H.rand_points(c, n, 700 + n)
        xs = H.rand_scalars(c, n, 800 + n)
        if n >= 3:
            xs[0], xs[1] = 0, c.r - 1
        if n >= 17:
            xs[5] = xs[4]
            pts[7] = None
            xs[9] = 1
        want = pm.jac_to_affine(c, pm.msm_bos_coster(c, [pm.to_jac(p) for p in pts], xs))
        assert want == pm.jac_to_affine(c, pm.msm_naive(c, [pm.to_jac(p) for p in pts], xs))
        got = ark_bp.bos_coster_msm(c.curve_id, H.jac_abi(c, pts), H.scalars_abi(c, xs))
        assert H.jac_from_abi(c, got)[0] == want, n
