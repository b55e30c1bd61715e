#!/usr/bin/env python3
"""Generate tests/golden/*.json from Oracle A (oracle/pasta_model.py, Python big integers).

The reference holds no known-answer vectors for this path and cannot be built or
run here, so these fixtures are outputs of the build's own independent big-integer model -- they pin
the C oracle and the GPU path against silent drift, not against the reference ("parity unpinned").
All values are hex strings of the C-ABI byte layout (ark-ff Montgomery limbs, little-endian).

    python tools/gen_golden.py        # rewrites tests/golden/
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pasta_model as pm  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sc_hex(c, xs):
    return [pm.scalar_to_abi(c, x).hex() for x in xs]


def pt_hex(c, ps):
    return [pm.point_to_abi_jac(c, pm.to_jac(p)).hex() for p in ps]


def rng_for(tag):
    return pm.ChaChaRng((tag.encode() + bytes(32))[:32], 12)


def msm_cases(c):
    cases = []
    for n in (0, 1, 2, 3, 31, 32, 33, 100):
        rng = rng_for(f"msm-{c.name}-{n}")
        pts = [pm.rand_point(c, rng) for _ in range(n)]
        xs = [pm.rand_scalar(c, rng) for _ in range(n)]
        edge = [0, 1, c.r - 1, 2, (c.r - 1) // 2, 1 << 128, (1 << 254) % c.r]
        for i in range(min(n, len(edge))):
            xs[i] = edge[i]
        if n >= 33:
            pts[10] = None                     # identity base (zero padding)
            pts[12] = pts[11]                  # duplicate base, same scalar: doubling inside a bucket
            xs[12] = xs[11]
            pts[14] = pm.aff_neg(c, pts[13])   # P and -P with equal scalars cancel
            xs[14] = xs[13]
        res = pm.jac_to_affine(c, pm.msm_pippenger(c, [pm.to_jac(p) for p in pts], xs))
        assert res == pm.jac_to_affine(c, pm.msm_naive(c, [pm.to_jac(p) for p in pts], xs))
        cases.append({"n": n, "points": pt_hex(c, pts), "scalars": sc_hex(c, xs), "result": pt_hex(c, [res])[0]})
    return cases


def vector_cases(c):
    rng = rng_for(f"vec-{c.name}")
    n = 12
    a = [pm.rand_scalar(c, rng) for _ in range(2 * n)]
    b = [pm.rand_scalar(c, rng) for _ in range(2 * n)]
    a[0], b[0], a[1] = c.r - 1, c.r - 1, 0
    x = pm.rand_scalar(c, rng)
    xi = pow(x, -1, c.r)
    g = [pm.rand_point(c, rng) for _ in range(2 * n)]
    g[2] = None
    g[n + 3] = g[3]
    g[n + 4] = pm.aff_neg(c, g[4])
    return {
        "a": sc_hex(c, a), "b": sc_hex(c, b), "x": sc_hex(c, [x])[0], "x_inv": sc_hex(c, [xi])[0],
        "ip": sc_hex(c, [pm.ip(c, a, b)])[0],
        "fold_a": sc_hex(c, pm.fold_scalars(c, a, x, xi)),
        "fold_b": sc_hex(c, pm.fold_scalars(c, b, xi, x)),
        "gens": pt_hex(c, g),
        "fold_gens_A": pt_hex(c, [pm.jac_to_affine(c, p) for p in pm.fold_points(c, [pm.to_jac(p) for p in g], xi, x)]),
        "fold_gens_B": pt_hex(c, [pm.jac_to_affine(c, p) for p in pm.fold_points(c, [pm.to_jac(p) for p in g], x, xi)]),
    }


def prove_cases(c):
    """Bp::prove transcripts at the reference's own test sizes (src/lib.rs:188,257) with its sampling
    recipe (test_rng stream: a_gens, b_gens, ip_gen, a, b, fs seed), normalise-at-boundary rule."""
    cases = []
    for n in (1, 2, 4, 8, 16, 17):
        rng = pm.test_rng()
        inst, wit = pm.sample_ipa(c, rng, n)
        seed = rng.gen_bytes32()
        pf = pm.bp_prove(c, inst, wit, pm.FiatShamirRng(seed))
        assert pm.bp_verify(c, inst, pf, pm.FiatShamirRng(seed))
        cases.append({
            "n": n, "fs_seed": seed.hex(),
            "a_gens": pt_hex(c, inst.a_gens), "b_gens": pt_hex(c, inst.b_gens), "ip_gen": pt_hex(c, [inst.ip_gen])[0],
            "result": pt_hex(c, [inst.result])[0], "a": sc_hex(c, wit.a), "b": sc_hex(c, wit.b),
            "ls": pt_hex(c, pf.ls), "rs": pt_hex(c, pf.rs), "challenges": sc_hex(c, pf.challenges),
            "final_a": sc_hex(c, [pf.final_a])[0], "final_b": sc_hex(c, [pf.final_b])[0],
        })
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    for c in (pm.PALLAS, pm.VESTA):
        doc = {"curve": c.name, "curve_id": c.curve_id,
               "format": "hex of C-ABI bytes: scalar 32 B, jacobian point 96 B (Z = 1, identity = (0,1,0)); Montgomery limbs",
               "generated_by": "tools/gen_golden.py from oracle/pasta_model.py (parity unpinned: no reference vectors exist)",
               "msm": msm_cases(c), "vectors": vector_cases(c), "bp_prove": prove_cases(c)}
        path = os.path.join(OUT, f"{c.name}.json")
        with open(path, "w") as f:
            json.dump(doc, f, indent=0)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
