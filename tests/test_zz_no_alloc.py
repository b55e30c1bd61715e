"""bpgpu_ip_device_noalloc (the entry point that allocates nothing) against bpgpu_ip_device: two calls in a
row, the second with a smaller input, through the C ABI.  Bit-exact."""
import numpy as np
import pytest

from oracle import pasta_model as pm
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
def test_ip_noalloc_matches_ip_device(c):
    from bulletproofs_amd import ffi
    ffi.init(0)
    d_out = ffi.DeviceBuffer(32)
    # a large call first (every block of the partial array in use), then smaller ones: stale partials of the
    # previous call must not leak into the next result
    for n in [(1 << 21) + 5, 4096 + 17, 3, 1, 0]:
        a = H.rand_scalars(c, min(n, 1000), 31 + n)
        b = H.rand_scalars(c, min(n, 1000), 32 + n)
        reps = -(-n // max(1, len(a))) if n else 0
        ha = np.tile(H.scalars_abi(c, a), (reps, 1))[:n] if n else np.zeros((0, 4), np.uint64)
        hb = np.tile(H.scalars_abi(c, b), (reps, 1))[:n] if n else np.zeros((0, 4), np.uint64)
        d_a, d_b = ffi.scalars_to_device(ha), ffi.scalars_to_device(hb)
        d_out.upload(np.full(4, 0xdeadbeefdeadbeef, np.uint64))
        want = ffi.ip_device(c.curve_id, d_a, d_b, n)
        ffi.ip_device_noalloc(c.curve_id, d_a, d_b, n, d_out)
        got = d_out.download(np.uint64, 4)
        assert (got == want).all(), n
        if n <= 4096 + 17:
            assert H.scalars_from_abi(c, got)[0] == pm.ip(c, (a * reps)[:n], (b * reps)[:n]), n
        d_a.free()
        d_b.free()
    with pytest.raises(ffi.BpGpuError):
        load = ffi.load()
        ffi._check(load.bpgpu_ip_device_noalloc(c.curve_id, d_out.ptr, d_out.ptr, 1 << 32, d_out.ptr))
    d_out.free()
