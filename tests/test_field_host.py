"""CPU build of the device arithmetic headers (field.cuh / curve.cuh, overflow-checked
accumulators) against the big-integer oracle.  No GPU needed."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle import pasta_model as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bulletproofs_amd", "lib", "libbpchk.so")

P = [pm.PALLAS_BASE, pm.PALLAS_SCALAR]
R256 = 1 << 256


@pytest.fixture(scope="module")
def chk():
    if not os.path.exists(LIB):
        pytest.skip("libbpchk.so not built (run python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(LIB)
    lib.bpchk_fe_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.bpchk_pt_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_uint64, ctypes.c_void_p]
    lib.bpchk_failures.restype = ctypes.c_long
    lib.bpchk_reset()
    return lib


def to_ark(v, p):
    return np.frombuffer((v * R256 % p).to_bytes(32, "little"), dtype=np.uint32).copy()


def from_ark(w, p):
    return int.from_bytes(w.tobytes(), "little") * pow(R256, -1, p) % p


def fe_op(lib, fid, op, a, b):
    p = P[fid]
    wa, wb, wo = to_ark(a, p), to_ark(b, p), np.zeros(8, dtype=np.uint32)
    assert lib.bpchk_fe_op(fid, op, wa.ctypes.data, wb.ctypes.data, wo.ctypes.data) == 0
    raw = int.from_bytes(wo.tobytes(), "little")
    assert raw < p, "non-canonical output"
    return from_ark(wo, p)


EDGE = lambda p: [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 254) - 1, (1 << 254) % p, R256 % p, (R256 * R256) % p,
                  (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232]


@pytest.mark.parametrize("fid", [0, 1])
def test_field_ops(chk, fid):
    p = P[fid]
    rnd = random.Random(1234 + fid)
    vals = EDGE(p) + [rnd.randrange(p) for _ in range(40)]
    pairs = [(a, b) for a in EDGE(p) for b in EDGE(p)] + [(rnd.choice(vals), rnd.choice(vals)) for _ in range(300)]
    for a, b in pairs:
        assert fe_op(chk, fid, 0, a, b) == a * b % p
        assert fe_op(chk, fid, 1, a, b) == a * a % p
        assert fe_op(chk, fid, 2, a, b) == (a + b) % p
        assert fe_op(chk, fid, 3, a, b) == (a - b) % p
        assert fe_op(chk, fid, 5, a, b) == (a + b) * (a - b) % p
        assert fe_op(chk, fid, 6, a, b) == (a - b) ** 2 % p
        assert fe_op(chk, fid, 7, a, b) == (b - a) * (a - b) % p
        assert fe_op(chk, fid, 8, a, b) == (-a) % p
        assert fe_op(chk, fid, 10, a, b) == (1 if a == b else 0)
        assert fe_op(chk, fid, 13, a, b) == (a * (a * b) + b * (b * b)) % p     # mul_add2: two products, one reduction
        assert fe_op(chk, fid, 14, a, b) == (a * (a * b) + b * (b * b)) % p     # ... and canon_small on the result
    for a in vals:
        if a:
            assert fe_op(chk, fid, 4, a, 0) == pow(a, -1, p)
            assert fe_op(chk, fid, 11, a, 0) == pow(a, -1, p)      # the Fermat ladder, cross-check of the gcd inverse
            assert fe_op(chk, fid, 12, a, 0) == pow(a, -1, p)      # lazy operand
    assert fe_op(chk, fid, 4, 0, 0) == 0                            # inv(0) = 0, as the Fermat ladder gives
    # division steps: values whose gcd run is long or lopsided (powers of two, all-ones, neighbours of p / 2 and of p)
    hard = [1 << k for k in range(0, 255, 7)] + [(1 << k) - 1 for k in range(2, 255, 11)] + [p - (1 << k) for k in range(0, 250, 13)]
    hard += [p // 2, p // 2 + 1, p // 3, 2 * p // 3, (p - 1) // 2 - 1, 3, 5, 7, p - 3]
    hard += [rnd.randrange(p) for _ in range(2000)]
    for a in hard:
        a %= p
        if a:
            assert fe_op(chk, fid, 4, a, 0) == pow(a, -1, p), hex(a)
    for a, b in pairs[:50]:
        assert fe_op(chk, fid, 9, a, b) == (a + 20 * (a + b)) * b % p
    # ark words (any 256-bit pattern, canonical or not) -> canonical integer, the digit kernel's conversion
    for raw in [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p, 3 * p + 5, (1 << 256) - 1, (1 << 255), R256 % p] + [rnd.getrandbits(256) for _ in range(300)]:
        wa = np.frombuffer(raw.to_bytes(32, "little"), dtype=np.uint32).copy()
        wo = np.zeros(8, dtype=np.uint32)
        assert chk.bpchk_fe_op(fid, 15, wa.ctypes.data, wa.ctypes.data, wo.ctypes.data) == 0
        assert int.from_bytes(wo.tobytes(), "little") == raw * pow(R256, -1, p) % p, hex(raw)
    assert chk.bpchk_failures() == 0, "an int64 accumulator / int32 limb bound was violated"


def rand_points(c, n, seed):
    rng = pm.ChaChaRng(bytes([seed]) * 32, 12)
    return [pm.rand_point(c, rng) for _ in range(n)]


def pts_buf(c, pts):
    return np.frombuffer(b"".join(pm.point_to_abi_affine(c, p) for p in pts), dtype=np.uint32).copy()


@pytest.mark.parametrize("curve", [pm.PALLAS, pm.VESTA])
def test_point_ops(chk, curve):
    fid = curve.curve_id  # base field id == curve id
    pts = rand_points(curve, 12, 7 + fid)
    out = np.zeros(16, dtype=np.uint32)

    def run(mode, plist, signs=None, k=0):
        buf = pts_buf(curve, plist) if plist else np.zeros(16, dtype=np.uint32)
        sg = np.array(signs if signs else [0] * max(1, len(plist)), dtype=np.int32)
        assert chk.bpchk_pt_op(fid, mode, buf.ctypes.data, sg.ctypes.data, len(plist), k, out.ctypes.data) == 0
        return pm.point_from_abi_affine(curve, out.tobytes())

    def expect(plist, signs=None):
        acc = None
        for i, p in enumerate(plist):
            acc = pm.aff_add(curve, acc, pm.aff_neg(curve, p) if signs and signs[i] else p)
        return acc

    cases = [
        (pts, None),
        (pts, [i & 1 for i in range(len(pts))]),
        ([pts[0], pts[0]], None),                      # doubling through madd
        ([pts[0], pts[0]], [0, 1]),                    # P - P = identity
        ([pts[0], pts[0], pts[1]], [0, 1, 0]),         # back from identity
        ([None, pts[2], None, pts[3]], None),          # identity operands
        ([pts[0], pts[1], pts[0], pts[1]], [0, 0, 1, 1]),
        ([pts[1], pts[1], pts[1], pts[1]], None),      # P+P then 2P+P, 3P+P
        ([], None),
    ]
    for plist, signs in cases:
        for mode in (0, 1, 3, 4):
            assert run(mode, plist, signs) == expect(plist, signs), (mode, signs)
    for k in [0, 1, 2, 3, 0xFFFFFFFFFFFFFFFF, 0x123456789ABCDEF]:
        assert run(2, [pts[4]], None, k) == pm.aff_mul(curve, pts[4], k)
        assert run(5, [pts[4]], None, k) == pm.aff_mul(curve, pts[4], k << 200)   # Jacobian ladder + 200 doublings
    assert chk.bpchk_failures() == 0
