#!/usr/bin/env python3
"""Generate bulletproofs_amd/csrc/pasta_consts.h from the two Pasta moduli.

The moduli are the only inputs (reference: analysis/pasta.sage:1-2); every other
constant (radix-2^29 limbs, Montgomery conversion factors for R = 2^256 (ark-ff's
in-memory form) and R' = 2^261 (the device's internal form), -p^-1 mod 2^64 ...)
is derived here.  Run from the repo root:  python tools/gen_consts.py
"""
import os

P0 = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # Pallas base / Vesta scalar
P1 = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # Pallas scalar / Vesta base

M29 = (1 << 29) - 1


def limbs29(v):
    return [(v >> (29 * i)) & M29 for i in range(9)]


def limbs32(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def limbs64(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def arr(vals, fmt):
    return "{" + ", ".join(fmt % v for v in vals) + "}"


def cube_root_of_unity(p):
    # smallest g with g^((p-1)/3) != 1
    g = 2
    while True:
        z = pow(g, (p - 1) // 3, p)
        if z != 1:
            return z
        g += 1


def emit_field(fid, p, name):
    assert p % (1 << 32) == 1
    l29 = limbs29(p)
    assert l29[0] == 1 and l29[5] == l29[6] == l29[7] == 0 and l29[8] == 1 << 22, l29
    out = []
    out.append(f"// {name}: p = 0x{p:064x}")
    out.append(f"template <> struct FieldConsts<{fid}> {{")
    out.append(f"  static constexpr int FID = {fid};")
    out.append(f"  static constexpr uint32_t P29[9] = {arr(l29, '0x%08xu')};")
    out.append(f"  static constexpr uint32_t P32[8] = {arr(limbs32(p), '0x%08xu')};")
    out.append(f"  static constexpr uint64_t P64[4] = {arr(limbs64(p), '0x%016xull')};")
    out.append(f"  // -p^-1 mod 2^64 (host 4x64 Montgomery, R = 2^256)")
    out.append(f"  static constexpr uint64_t INV64 = 0x{(-pow(p, -1, 1 << 64)) % (1 << 64):016x}ull;")
    consts = {
        "ONE261": (1 << 261) % p,             # 1 in device form
        "C256TO261": pow(2, 266, p),          # montmul261(x*2^256, .) = x*2^261
        "C261TO256": pow(2, 256, p),          # montmul261(x*2^261, .) = x*2^256
        "C256TOCANON": 32,                    # montmul261(x*2^256, 2^5) = x
        "C261TOCANON": 1,                     # montmul261(x*2^261, 1) = x
        "R3_261": pow(2, 783, p),             # montmul261(y, 2^783) = y*2^522: integer inverse of x*2^261 -> x^-1*2^261
        "CANONTO261": pow(2, 522, p),         # montmul261(x, .) = x*2^261
        "B261": 5 * (1 << 261) % p,           # curve coefficient b = 5 in device form
    }
    for k, v in consts.items():
        out.append(f"  static constexpr uint32_t {k}[9] = {arr(limbs29(v), '0x%08xu')};")
    hconsts = {
        "H_ONE": (1 << 256) % p,
        "H_R2": pow(2, 512, p),
        "H_B": 5 * (1 << 256) % p,
    }
    for k, v in hconsts.items():
        out.append(f"  static constexpr uint64_t {k}[4] = {arr(limbs64(v), '0x%016xull')};")
    out.append("};")
    return "\n".join(out)


def limbs64n(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def isqrt(n):
    import math
    return math.isqrt(n)


def glv_basis(r, lam):
    """Short basis of the lattice {(a, b): a + b*lam = 0 mod r} by the extended Euclidean algorithm."""
    seq = [(r, 0), (lam, 1)]  # (remainder, t) with remainder = s*r + t*lam
    while seq[-1][0] != 0:
        q = seq[-2][0] // seq[-1][0]
        seq.append((seq[-2][0] - q * seq[-1][0], seq[-2][1] - q * seq[-1][1]))
    rt = isqrt(r)
    l = max(i for i, (rem, _) in enumerate(seq) if rem >= rt)
    a1, b1 = seq[l + 1][0], -seq[l + 1][1]
    cands = [(seq[l][0], -seq[l][1]), (seq[l + 2][0], -seq[l + 2][1])]
    a2, b2 = min(cands, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for a, b in ((a1, b1), (a2, b2)):
        assert (a + b * lam) % r == 0
    return a1, b1, a2, b2


def glv_split(k, r, a1, b1, a2, b2, g1, g2, n1neg, n2neg):
    """What the host code does: c_i = +-((k * g_i) >> 384); (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2)."""
    c1 = (k * g1) >> 384
    c2 = (k * g2) >> 384
    if n1neg:
        c1 = -c1
    if n2neg:
        c2 = -c2
    return k - c1 * a1 - c2 * a2, -c1 * b1 - c2 * b2


def emit_curve(cid, q, r, name):
    """GLV constants of curve cid: base field q, scalar field r, endomorphism (x, y) -> (zeta x, y) = [lambda](x, y)."""
    import random
    # the matching pair (zeta, lambda): checked on the generator (-1, 2) with an affine ladder
    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % q == 0:
                return None
            m = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
        else:
            m = (y2 - y1) * pow(x2 - x1, -1, q) % q
        x3 = (m * m - x1 - x2) % q
        return (x3, (m * (x1 - x3) - y1) % q)

    def mul(P, k):
        acc = None
        for i in range(k.bit_length() - 1, -1, -1):
            acc = add(acc, acc)
            if (k >> i) & 1:
                acc = add(acc, P)
        return acc

    zeta = cube_root_of_unity(q)
    G = (q - 1, 2)
    lam = cube_root_of_unity(r)
    if mul(G, lam) != (zeta * G[0] % q, G[1]):
        lam = lam * lam % r
    assert mul(G, lam) == (zeta * G[0] % q, G[1])
    a1, b1, a2, b2 = glv_basis(r, lam)
    n1, n2 = b2, -b1
    g1 = ((abs(n1) << 384) + r // 2) // r
    g2 = ((abs(n2) << 384) + r // 2) // r
    rnd = random.Random(cid)
    worst = 0
    for _ in range(2000):
        k = rnd.randrange(r)
        k1, k2 = glv_split(k, r, a1, b1, a2, b2, g1, g2, n1 < 0, n2 < 0)
        assert (k1 + k2 * lam - k) % r == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    assert worst <= 129, worst
    M = (1 << 256) - 1
    out = []
    out.append(f"// {name}: endomorphism (x, y) -> (zeta x, y) = [lambda](x, y); GLV halves are at most {worst} bits")
    out.append(f"template <> struct CurveConsts<{cid}> {{")
    out.append(f"  static constexpr uint32_t ZETA261[9] = {arr(limbs29(zeta * (1 << 261) % q), '0x%08xu')};")
    out.append(f"  static constexpr uint64_t LAMBDA[4] = {arr(limbs64(lam), '0x%016xull')};  // canonical")
    for nm, v in (("A1", a1), ("B1", b1), ("A2", a2), ("B2", b2)):
        out.append(f"  static constexpr uint64_t GLV_{nm}[4] = {arr(limbs64(v & M), '0x%016xull')};  // two's complement, = {v}")
    out.append(f"  static constexpr uint64_t GLV_G1[5] = {arr(limbs64n(g1, 5), '0x%016xull')};  // round(2^384 |b2| / r)")
    out.append(f"  static constexpr uint64_t GLV_G2[5] = {arr(limbs64n(g2, 5), '0x%016xull')};  // round(2^384 |b1| / r)")
    out.append(f"  static constexpr bool GLV_N1_NEG = {'true' if n1 < 0 else 'false'};")
    out.append(f"  static constexpr bool GLV_N2_NEG = {'true' if n2 < 0 else 'false'};")
    out.append("};")
    return "\n".join(out)


def main():
    hdr = []
    hdr.append("// GENERATED by tools/gen_consts.py -- do not edit.")
    hdr.append("// Constants for the two Pasta prime fields, derived from the moduli alone.")
    hdr.append("#pragma once")
    hdr.append("#include <cstdint>")
    hdr.append("namespace bpgpu {")
    hdr.append("template <int FID> struct FieldConsts;")
    hdr.append(emit_field(0, P0, "field 0 = Pallas base field Fq = Vesta scalar field"))
    hdr.append(emit_field(1, P1, "field 1 = Pallas scalar field Fr = Vesta base field"))
    hdr.append("// Curve c (0 = Pallas, 1 = Vesta): base field id = c, scalar field id = 1 - c.")
    hdr.append("template <int CURVE> struct CurveConsts;")
    hdr.append(emit_curve(0, P0, P1, "Pallas"))
    hdr.append(emit_curve(1, P1, P0, "Vesta"))
    hdr.append("}  // namespace bpgpu")
    path = os.path.join(os.path.dirname(__file__), "..", "bulletproofs_amd", "csrc", "pasta_consts.h")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(hdr) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
