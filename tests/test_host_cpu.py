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
