"""The HIP path (through the C ABI and the C++ host side) against the committed golden fixtures."""
import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    from bulletproofs_amd import ark_bp, ffi
    ffi.init(0)
    return ffi, ark_bp


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_gpu_matches_golden(libs, name):
    ffi, bp = libs
    doc = G.load(name)
    cid = doc["curve_id"]
    for case in doc["msm"]:
        pts, sc = G.points(case["points"]), G.scalars(case["scalars"])
        assert G.xy_inf_matches(ffi.msm(cid, pts, sc), case["result"]), case["n"]
        assert (bp.msm(cid, pts, sc) == G.points([case["result"]])[0]).all(), case["n"]
    v = doc["vectors"]
    a, b = G.scalars(v["a"]), G.scalars(v["b"])
    x, xi = G.scalars([v["x"]])[0], G.scalars([v["x_inv"]])[0]
    assert (ffi.ip(cid, a, b) == G.scalars([v["ip"]])[0]).all()
    assert (bp.scalar_inverse(cid, x) == xi).all()
    assert (ffi.fold_scalars(cid, a, x, xi) == G.scalars(v["fold_a"])).all()
    assert (ffi.fold_scalars(cid, b, xi, x) == G.scalars(v["fold_b"])).all()
    g = G.points(v["gens"])
    assert (ffi.fold_points(cid, g, xi, x) == G.points(v["fold_gens_A"])).all()
    assert (ffi.fold_points(cid, g, x, xi) == G.points(v["fold_gens_B"])).all()
    for case in doc["bp_prove"]:
        A, B, q = G.points(case["a_gens"]), G.points(case["b_gens"]), G.points([case["ip_gen"]])
        a, b = G.scalars(case["a"]), G.scalars(case["b"])
        seed = bytes.fromhex(case["fs_seed"])
        assert (bp.commit(cid, A, B, q, a, b) == G.points([case["result"]])[0]).all()
        pf = bp.bp_prove(cid, A, B, q, a, b, seed)
        assert (pf["ls"] == G.points(case["ls"])).all() and (pf["rs"] == G.points(case["rs"])).all()
        assert (pf["challenges"] == G.scalars(case["challenges"])).all()
        assert (pf["final_a"] == G.scalars([case["final_a"]])[0]).all()
        assert (pf["final_b"] == G.scalars([case["final_b"]])[0]).all()
        bp.bp_verify(cid, A, B, q, G.points([case["result"]]), pf, seed)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_gpu_matches_golden_next_rows(libs, name):
    ffi, bp = libs
    doc = G.load(name)
    cid, r = doc["curve_id"], doc["next_rows"]
    k = r["k"]
    v, y, g = G.scalars(r["v"]), G.scalars(r["y"]), G.points(r["gens"])
    x = G.scalars([r["x"]])[0]
    assert (ffi.fold_scalars_kary(cid, v, k, y) == G.scalars(r["fold_scalars_kary"])).all()
    assert (ffi.fold_points_kary(cid, g, k, y) == G.points(r["fold_points_kary"])).all()
    assert (ffi.fold_points_kary(cid, g, k, ffi.powers(cid, x, 0, k)) == G.points(r["fold_points_kary_powers"])).all()
    assert (ffi.powers(cid, x, 5, 12) == G.scalars(r["powers_5_to_12"])).all()
    hs = G.scalars(r["hadamard_scalars"])
    assert (ffi.hadamard(cid, v, hs) == G.scalars(r["hadamard"])).all()
    assert (ffi.hadamard_points(cid, g, hs) == G.points(r["hadamard_gp"])).all()
    row_ptr, col_idx = np.array(r["mat_row_ptr"], np.uint64), np.array(r["mat_col_idx"], np.uint64)
    vals = G.scalars(r["mat_vals"])
    assert (ffi.mat_vec_mult(cid, row_ptr, col_idx, vals, G.scalars(r["z"])) == G.scalars(r["mat_vec"])).all()
    assert (ffi.vec_mat_mult(cid, G.scalars(r["w"]), row_ptr, col_idx, vals, 7) == G.scalars(r["vec_mat_out7"])).all()
    ys, ok = ffi.curve_lift_x(cid, G.scalars(r["lift_x"]), np.array(r["lift_greatest"], np.uint8))
    assert ok.tolist() == r["lift_ok"] and (ys == G.scalars(r["lift_y"])).all()
