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


def next_row_cases(c):
    """The "next" rows: k-ary folds, powers, hadamard, hadamard_gp, the sparse products of the
    Com-R1CS compiler and the per-candidate half of G::rand."""
    rng = rng_for(f"next-{c.name}")
    k, ck = 3, 5
    v = [pm.rand_scalar(c, rng) for _ in range(k * ck)]
    v[0], v[1] = 0, c.r - 1
    y = [pm.rand_scalar(c, rng) for _ in range(k)]
    g = [pm.rand_point(c, rng) for _ in range(k * ck)]
    g[1] = None
    g[ck + 2] = g[2]
    x = pm.rand_scalar(c, rng)
    xp = pm.powers(c, x, 0, k)
    hs = [pm.rand_scalar(c, rng) for _ in range(k * ck)]
    hs[0], hs[3] = 0, 1
    mat = [[(pm.rand_scalar(c, rng), 0), (pm.rand_scalar(c, rng), 4)], [], [(c.r - 1, 2), (pm.rand_scalar(c, rng), 2), (1, 3)],
           [(pm.rand_scalar(c, rng), j) for j in range(ck)]]
    z = [pm.rand_scalar(c, rng) for _ in range(ck)]
    w = [pm.rand_scalar(c, rng) for _ in range(len(mat))]
    cand = [c.from_mont_q(pm.rand_field_mont(rng, c.q)) for _ in range(12)]
    great = [rng.gen_bool() for _ in cand]
    lifted = [pm.lift_x(c, xx, gg) for xx, gg in zip(cand, great)]
    fe_hex = lambda vals: [pm.mont_limbs_bytes(c.to_mont_q(t)).hex() for t in vals]
    jac = lambda ps: [pm.to_jac(p) for p in ps]
    aff = lambda js: [pm.jac_to_affine(c, j) for j in js]
    return {
        "k": k, "ck": ck, "v": sc_hex(c, v), "y": sc_hex(c, y), "x": sc_hex(c, [x])[0],
        "fold_scalars_kary": sc_hex(c, pm.fold_scalars_kary(c, v, k, y)),
        "gens": pt_hex(c, g),
        "fold_points_kary": pt_hex(c, aff(pm.fold_points_kary(c, jac(g), k, y))),
        "fold_points_kary_powers": pt_hex(c, aff(pm.fold_points_kary(c, jac(g), k, xp))),
        "powers_5_to_12": sc_hex(c, pm.powers(c, x, 5, 12)),
        "hadamard_scalars": sc_hex(c, hs),
        "hadamard": sc_hex(c, pm.hadamard(c, v, hs)),
        "hadamard_gp": pt_hex(c, aff(pm.hadamard_gp(c, jac(g), hs))),
        "mat_row_ptr": [0, 2, 2, 5, 5 + ck], "mat_col_idx": [col for row in mat for _, col in row],
        "mat_vals": sc_hex(c, [val for row in mat for val, _ in row]),
        "z": sc_hex(c, z), "mat_vec": sc_hex(c, pm.mat_vec_mult(c, mat, z)),
        "w": sc_hex(c, w), "vec_mat_out7": sc_hex(c, pm.vec_mat_mult(c, w, mat, 7)),
        "lift_x": fe_hex(cand), "lift_greatest": [int(b) for b in great],
        "lift_ok": [int(t is not None) for t in lifted], "lift_y": fe_hex([t or 0 for t in lifted]),
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


def faithful_z_cases(c):
    """What the model predicts an UNMODIFIED reference produces for `test_bp_ipa_pallas` (src/lib.rs:253-258:
    test_ipa(vec![1, 2, 4, 8, 16], 1, Bp) -- the reference's "vesta" test, :260-265, instantiates Pallas too; size 17, the
    non-power-of-two of :188, is appended here and does not change the five before it): raw Jacobian L, R
    (canonical X | Y | Z as ToBytes writes them), the challenges and final_a / final_b.  NOT produced by the reference:
    to be confirmed by a maintainer with `cargo test` (INTEGRATION.md 7 prints the same bytes from Rust)."""
    cases = []
    rng = pm.test_rng()  # test_ipa draws every size from ONE rng stream (src/lib.rs:111-118)
    for n in (1, 2, 4, 8, 16, 17):
        inst, wit = pm.sample_ipa(c, rng, n)
        seed = rng.gen_bytes32()
        pf = pm.bp_prove_faithful_z(c, inst, wit, pm.FiatShamirRng(seed))
        assert pm.bp_verify_faithful_z(c, inst, pf, pm.FiatShamirRng(seed))
        raw = lambda pts: [pm.point_bytes_raw(p).hex() for p in pts]
        cases.append({"n": n, "fs_seed": seed.hex(), "result_xyz_canonical": raw([pm.to_jac(inst.result)])[0],
                      "ls_xyz_canonical": raw(pf.ls), "rs_xyz_canonical": raw(pf.rs),
                      "challenges_canonical": [pm.fe_bytes(x).hex() for x in pf.challenges],
                      "final_a_canonical": pm.fe_bytes(pf.final_a).hex(), "final_b_canonical": pm.fe_bytes(pf.final_b).hex()})
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(os.path.join(OUT, "faithful_z"), exist_ok=True)
    for c in (pm.PALLAS, pm.VESTA):
        doc = {"curve": c.name,
               "WARNING": "self-generated by this repository's own model (oracle/pasta_model.py, bp_prove_faithful_z); NOT reference "
                          "output.  It restates what an unmodified ark-bp would print for test_ipa(&[1,2,4,8,16,17], 1, Bp) under the "
                          "UPSTREAM-RECALLED behaviour of ark-ff/ark-ec/ark-std/ark-marlin 0.3 (SURVEY.md 8c); confirm with cargo test.",
               "format": "hex of canonical little-endian bytes (into_repr): field element 32 B, point X | Y | Z 96 B raw Jacobian",
               "recipe": "rng = ark_std::test_rng(); for each size: IpaInstance::sample_from_length, fs_seed = rng.gen::<[u8;32]>() "
                         "(src/lib.rs:110-118), Bp::prove (src/proofs/bp_iter.rs:49-128)",
               "test_ipa": faithful_z_cases(c)}
        path = os.path.join(OUT, "faithful_z", f"{c.name}.json")
        with open(path, "w") as f:
            json.dump(doc, f, indent=0)
        print("wrote", path, os.path.getsize(path), "bytes")
    for c in (pm.PALLAS, pm.VESTA):
        doc = {"curve": c.name, "curve_id": c.curve_id,
               "format": "hex of C-ABI bytes: scalar 32 B, jacobian point 96 B (Z = 1, identity = (0,1,0)); Montgomery limbs",
               "generated_by": "tools/gen_golden.py from oracle/pasta_model.py (parity unpinned: no reference vectors exist)",
               "msm": msm_cases(c), "vectors": vector_cases(c), "bp_prove": prove_cases(c), "next_rows": next_row_cases(c)}
        path = os.path.join(OUT, f"{c.name}.json")
        with open(path, "w") as f:
            json.dump(doc, f, indent=0)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
