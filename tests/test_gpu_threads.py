"""The boundary's threading contract ("callable from any thread; internally serialised per
device" -- rayon worker threads may be the callers of util::msm, src/util/msm.rs:88): several host threads
call the C ABI at once (ctypes drops the GIL for the call), each on inputs of its own, and every result
must equal the CPU restatement's.  Bit-exact."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from bulletproofs_amd import ffi as f
    f.init(0)
    return f


def test_msm_and_ip_from_concurrent_threads(ffi):
    jobs = []
    for t, n in enumerate([1, 33, 257, 1000, 4096, 70000, 2, 513]):
        curve = t & 1
        pts, sc = co.msm_bench_inputs(curve, n)
        jobs.append((curve, pts, sc, co.msm_pippenger(curve, pts, sc, threads=4), np.array(co.ip(curve, sc, sc))))

    def work(job):
        curve, pts, sc, want, want_ip = job
        out = []
        for _ in range(3):
            got = ffi.msm(curve, pts, sc)
            # got = x, y, infinity flag; want = the restatement's normalised (X, Y, Z = 1)
            out.append(int(got[8]) == 0 and bool((got[:8] == want[:8]).all()))
            # a scalar-only call in between: the inner product of the scalars with themselves (util/mod.rs:17-23)
            out.append(bool((np.array(ffi.ip(curve, sc, sc)).ravel() == want_ip.ravel()).all()))
        return all(out)

    with ThreadPoolExecutor(len(jobs)) as ex:
        assert all(ex.map(work, jobs))


def test_ipa_sessions_from_concurrent_threads(ffi):
    """Two or more prover sessions open at once, each driven by its own thread (bp_iter.rs:49-128 per thread):
    the pooled session buffers serve one, the others allocate their own."""
    cases = []
    for t, n in enumerate([4, 16, 64, 8]):
        curve = t & 1
        A, B, q, a, b, seed = co.ipa_inputs(curve, n)
        cases.append((curve, A, B, q, a, b, co.bp_prove(curve, A, B, q, a, b, seed)))

    def prove(case):
        curve, A, B, q, a, b, (ls, rs, ch, fa, fb) = case
        sess = ffi.IpaSession(curve, A, B, q, a, b)
        ok, k = True, 0
        while len(sess) > 1:
            l, r = sess.round_lr()
            ok &= bool((l[:8] == ls[k][:8]).all() and (r[:8] == rs[k][:8]).all())
            sess.round_fold(ch[k], co.scalar_inverse(curve, ch[k]))
            k += 1
        ga, gb = sess.finish()
        sess.free()
        return ok and bool((ga == fa).all() and (gb == fb).all())

    with ThreadPoolExecutor(len(cases)) as ex:
        assert all(ex.map(prove, cases))
