"""The one-process sharded MSM (bpgpu_sharded_init / bpgpu_msm_sharded / bpgpu_sharded_gens_*): config 3's
partitioning -- contiguous point ranges, one context per GPU, partials summed on the host -- rehearsed on
the one GPU of the test box with several contexts on device 0, against the oracle and against the
single-context MSM.  Bit-exact."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pasta_model as pm
from tests import helpers as H

pytestmark = pytest.mark.gpu

SHARDS = 4


@pytest.fixture(scope="module")
def ffi():
    from bulletproofs_amd import ffi as f
    f.init(0)
    f.sharded_init([0] * SHARDS)
    assert f.sharded_count() == SHARDS
    f.sharded_init([0] * SHARDS)  # idempotent for the same device list
    with pytest.raises(f.BpGpuError):
        f.sharded_init([0] * (SHARDS + 1))
    return f


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_sharded_small_against_model(ffi, c):
    # fewer points than contexts (empty ranges), ragged ranges, edge scalars, identity / duplicate / opposite bases
    for n in [0, 1, 2, 3, 5, 33, 257]:
        pts = H.rand_points(c, n, 300 + n)
        xs = H.rand_scalars(c, n, 400 + n)
        if n >= 33:
            xs[0], xs[1], xs[2] = 0, 1, c.r - 1
            pts[5] = None
            pts[7], xs[7] = pts[6], xs[6]
            pts[9], xs[9] = pm.aff_neg(c, pts[8]), xs[8]
        want = pm.jac_to_affine(c, pm.msm_pippenger(c, [pm.to_jac(p) for p in pts], xs))
        got = H.xy_inf_to_affine(c, ffi.msm_sharded(c.curve_id, H.jac_abi(c, pts), H.scalars_abi(c, xs)))
        assert got == want, n
        got = H.xy_inf_to_affine(c, ffi.msm_sharded_affine(c.curve_id, H.aff_abi(c, pts), H.scalars_abi(c, xs)))
        assert got == want, n


def test_sharded_recipe_against_oracle_and_single_context(ffi):
    n = (1 << 16) + 3  # not a multiple of the context count
    pts, sc = co.msm_bench_inputs(0, n)
    want = co.msm_pippenger(0, pts, sc, threads=8)
    got = ffi.msm_sharded(0, pts, sc)
    assert int(got[8]) == 0 and (got[:8] == want[:8]).all()
    assert (got == ffi.msm(0, pts, sc)).all()
    # generators resident on the contexts, two scalar vectors in a row
    gens = ffi.ShardedGens(0, pts)
    assert (gens.msm(sc) == got).all()
    sc2 = np.roll(sc, 7, axis=0)
    assert (gens.msm(sc2) == ffi.msm(0, pts, sc2)).all()
    gens.free()


def test_sharded_2_20_equals_single_context(ffi):
    n = 1 << 20
    from bulletproofs_amd import ark_bp
    pts, sc = ark_bp.msm_bench_inputs(0, n)
    single = ffi.msm(0, pts, sc)
    assert (ffi.msm_sharded(0, pts, sc) == single).all()
    # the shards are the ranges the multi-process path uses (bulletproofs_amd/sharded.py): their sum is the whole
    from bulletproofs_amd import sharded
    parts = []
    for g in range(SHARDS):
        lo, hi = sharded.shard_range(n, g, SHARDS)
        parts.append(ffi.msm(0, pts[lo:hi], sc[lo:hi]))
    assert (ffi.points_sum(0, np.stack(parts)) == single).all()
