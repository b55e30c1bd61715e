"""Library life cycle and the widest sharded configuration on the one GPU of the test box:
  * bpgpu_init -> MSM (upload large enough for the staging threads) -> bpgpu_shutdown -> bpgpu_init -> MSM, with
    bpgpu_sharded_init before and after: the persistent host threads (Horner workers, staging threads, shard legs)
    are stopped and started again and must never run a job of the previous life;
  * EIGHT contexts on device 0 (the node's shape, config 3): ragged ranges against the oracle, resident generators
    at 2^20 against the single-context MSM.
Bit-exact.  Ends with the library initialised on device 0 and no shard contexts, as the other modules expect."""
import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def test_restart_and_eight_contexts():
    from bulletproofs_amd import ark_bp, ffi
    ffi.shutdown()  # drop whatever earlier modules left (no session is open between modules)
    ffi.init(0)
    assert ffi.sharded_count() == 0
    ffi.sharded_init([0] * 8)
    assert ffi.sharded_count() == 8
    n = (1 << 17) + 5  # 12.6 MB of Jacobian points: above the staging threshold, not a multiple of 8
    pts, sc = co.msm_bench_inputs(0, n)
    want = co.msm_pippenger(0, pts, sc, threads=8)
    first = ffi.msm(0, pts, sc)
    assert int(first[8]) == 0 and (first[:8] == want[:8]).all()
    assert (ffi.msm_sharded(0, pts, sc) == first).all()
    assert (ffi.msm_sharded_affine(0, np.ascontiguousarray(pts[:, :8]), sc) == first).all()
    # fewer points than contexts: empty ranges
    assert (ffi.msm_sharded(0, pts[:3], sc[:3]) == ffi.msm(0, pts[:3], sc[:3])).all()
    # resident generators at 2^20 on the eight contexts, two scalar vectors
    big_p, big_s = ark_bp.msm_bench_inputs(0, 1 << 20)
    single = ffi.msm(0, big_p, big_s)
    gens = ffi.ShardedGens(0, big_p)
    assert (gens.msm(big_s) == single).all()
    s2 = np.roll(big_s, 11, axis=0)
    assert (gens.msm(s2) == ffi.msm(0, big_p, s2)).all()
    gens.free()
    # an IPA session keeps the shutdown away; after it is freed the shutdown goes through
    A, B, q, a, b, seed = co.ipa_inputs(0, 8)
    sess = ffi.IpaSession(0, A, B, q, a, b)
    ffi.shutdown()
    assert b"session" in ffi.load().bpgpu_last_error()
    l0, r0 = sess.round_lr()
    sess.free()
    for life in range(2):
        ffi.shutdown()
        assert ffi.sharded_count() == 0
        with pytest.raises(ffi.BpGpuError):
            ffi.msm(0, pts, sc)  # not initialised: fails loudly
        ffi.init(0)
        again = ffi.msm(0, pts, sc)  # fresh staging threads and Horner workers
        assert (again == first).all(), life
        ffi.sharded_init([0] * (2 + life))  # another shape than before is fine after a shutdown
        assert (ffi.msm_sharded(0, pts, sc) == first).all(), life
        sess = ffi.IpaSession(0, A, B, q, a, b)
        l1, r1 = sess.round_lr()
        sess.free()
        assert (l1 == l0).all() and (r1 == r0).all()
    ffi.shutdown()
    ffi.init(0)
    assert ffi.get_tuning() == {"window_bits": 0, "freeze_len": 1 << 14, "pair_fold": True, "ipa_table_min": 1 << 12}
