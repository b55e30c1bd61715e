"""Shared helpers for the parity tests: conversions between the oracle's Python integers
and the C ABI's ark-ff byte layout, and seeded input generation."""
import numpy as np

from oracle import pasta_model as pm


def scalars_abi(c, xs):
    return np.frombuffer(b"".join(pm.scalar_to_abi(c, x) for x in xs), dtype=np.uint64).reshape(-1, 4).copy()


def scalar_abi(c, x):
    return scalars_abi(c, [x])[0]


def scalars_from_abi(c, arr):
    raw = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [pm.scalar_from_abi(c, raw[i:i + 32]) for i in range(0, len(raw), 32)]


def jac_abi(c, pts):
    """pts: affine tuples / None, or Jacobian triples."""
    out = []
    for p in pts:
        if p is None or len(p) == 2:
            p = pm.to_jac(p)
        out.append(pm.point_to_abi_jac(c, p))
    return np.frombuffer(b"".join(out), dtype=np.uint64).reshape(-1, 12).copy() if out else np.zeros((0, 12), np.uint64)


def aff_abi(c, pts):
    return (np.frombuffer(b"".join(pm.point_to_abi_affine(c, p) for p in pts), dtype=np.uint64).reshape(-1, 8).copy()
            if pts else np.zeros((0, 8), np.uint64))


def xy_inf_to_affine(c, out):
    out = np.ascontiguousarray(out, dtype=np.uint64)
    if int(out[8]) != 0:
        assert not out[:8].any()
        return None
    return pm.point_from_abi_affine(c, out[:8].tobytes())


def jac_from_abi(c, arr):
    """(n, 12) u64 -> list of affine points; checks Z is 1 (or the (0,1,0) identity)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 12)
    res = []
    for row in arr:
        b = row.tobytes()
        X, Y, Z = (c.from_mont_q(int.from_bytes(b[i:i + 32], "little")) for i in (0, 32, 64))
        if Z == 0:
            assert (X, Y) == (0, 1)
            res.append(None)
        else:
            assert Z == 1
            res.append((X, Y))
    return res


def seeded_rng(seed: int):
    return pm.ChaChaRng(int(seed).to_bytes(32, "little"), 12)


def rand_points(c, n, seed, distinct=None):
    """n random points; generated as multiples / sums of a few ChaCha-sampled base points
    so that large n stays cheap in pure Python."""
    rng = seeded_rng(seed)
    k = min(n, distinct or 8)
    base = [pm.rand_point(c, rng) for _ in range(k)]
    pts = list(base)
    while len(pts) < n:
        i = rng.next_u32() % len(pts)
        j = rng.next_u32() % len(pts)
        pts.append(pm.aff_add(c, pts[i], pm.aff_add(c, pts[j], base[rng.next_u32() % k])))
    return pts[:n]


def rand_scalars(c, n, seed):
    rng = seeded_rng(seed)
    return [pm.rand_scalar(c, rng) for _ in range(n)]
