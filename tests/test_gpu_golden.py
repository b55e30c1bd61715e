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
