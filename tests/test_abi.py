"""No-GPU checks of the boundary: the C-ABI library loads, exports every symbol that
include/bpgpu.h declares, and fails loudly (no CPU fallback) when used without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from bulletproofs_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ffi.load()
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bpgpu.h but not exported"
    assert sorted(ffi.SIGNATURES) == syms, "ffi.SIGNATURES and include/bpgpu.h disagree"


def test_no_fallback_without_init():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ffi.BpGpuError):
        ffi.msm(0, np.zeros((1, 12), np.uint64), np.zeros((1, 4), np.uint64))
    with pytest.raises(ffi.BpGpuError):
        ffi.ip(0, np.zeros((1, 4), np.uint64), np.zeros((1, 4), np.uint64))


def test_size_limits_are_rejected_before_any_device_work():
    """32-bit guards of the bucket sort: over-long inputs are an argument error, with or without a
    GPU -- the check precedes every device call, so null data pointers are never touched."""
    lib = ffi.load()
    out = np.zeros(9, np.uint64)
    h = ctypes.c_void_p()
    big_msm, big_ipa = (1 << 28) + 1, (1 << 26) + 1
    assert lib.bpgpu_msm(0, None, None, big_msm, out.ctypes.data) == -1
    assert b"2^28" in lib.bpgpu_last_error()
    assert lib.bpgpu_msm_affine(0, None, None, big_msm, out.ctypes.data) == -1
    assert lib.bpgpu_msm_device(0, None, None, big_msm, out.ctypes.data) == -1
    assert lib.bpgpu_ipa_begin(0, None, None, None, None, None, big_ipa, ctypes.byref(h)) == -1
    assert b"2^26" in lib.bpgpu_last_error()
    assert lib.bpgpu_ipa_begin_device(0, None, None, None, None, None, big_ipa, ctypes.byref(h)) == -1
    assert lib.bpgpu_ip_device_noalloc(0, None, None, 1 << 32, None) == -1
    assert lib.bpgpu_sharded_init(0, None) == -1 and lib.bpgpu_sharded_init(65, None) == -1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bulletproofs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle involved", ""), f"{f} mentions oracle/"


def test_points_sum_host_helper():
    """bpgpu_points_sum is host-only (the combine step of the sharded MSM); check it against the oracle."""
    from oracle import pasta_model as pm
    from tests import helpers as H
    for c in (pm.PALLAS, pm.VESTA):
        pts = H.rand_points(c, 5, 3) + [None]
        pts.append(pm.aff_neg(c, pts[0]))
        rows = []
        for p in pts:
            row = np.zeros(9, np.uint64)
            if p is None:
                row[8] = 1
            else:
                row[:8] = np.frombuffer(pm.point_to_abi_affine(c, p), np.uint64)
            rows.append(row)
        want = None
        for p in pts:
            want = pm.aff_add(c, want, p)
        assert H.xy_inf_to_affine(c, ffi.points_sum(c.curve_id, np.stack(rows))) == want
        assert H.xy_inf_to_affine(c, ffi.points_sum(c.curve_id, np.stack([rows[0], rows[-1]]))) is None
        assert H.xy_inf_to_affine(c, ffi.points_sum(c.curve_id, np.zeros((0, 9), np.uint64))) is None


def test_static_archive_defines_every_declared_symbol():
    """libbpgpu.a is what the reference's build.rs links: same object as the shared library."""
    import subprocess
    arc = os.path.join(ROOT, "bulletproofs_amd", "lib", "libbpgpu.a")
    assert os.path.exists(arc), "run `make` (or __graft_entry__.build()) first"
    out = subprocess.run(["nm", "--defined-only", "-g", arc], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r"\bT (bpgpu_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in defined]
    assert not missing, f"not defined in libbpgpu.a: {missing}"


def test_tuning_knobs_have_getters_and_a_reset():
    """The four process-wide knobs can be read back and restored to the shipped defaults: tests that
    change a schedule restore the REAL defaults, whatever they are."""
    defaults = {"window_bits": 0, "freeze_len": 1 << 14, "pair_fold": True, "ipa_table_min": 1 << 12}
    ffi.reset_tuning()
    assert ffi.get_tuning() == defaults
    ffi.set_window_bits(13)
    ffi.set_freeze_len(1 << 9)
    ffi.set_pair_fold(False)
    ffi.set_ipa_table(0)
    assert ffi.get_tuning() == {"window_bits": 13, "freeze_len": 1 << 9, "pair_fold": False, "ipa_table_min": 0}
    ffi.set_ipa_table(1)  # 1 = the default threshold
    assert ffi.get_tuning()["ipa_table_min"] == 1 << 12
    with pytest.raises(ffi.BpGpuError):
        ffi.set_window_bits(21)
    ffi.reset_tuning()
    assert ffi.get_tuning() == defaults


def test_ip_index_is_64_bit():
    """bpgpu_ip_device[_noalloc] admit n up to 2^32 - 1; the kernel's loop index must not wrap below that
    (a 32-bit `i += gridDim.x * 256` would for n > 2^32 - 262144)."""
    src = open(os.path.join(ROOT, "bulletproofs_amd", "csrc", "kernels.cuh")).read()
    body = src[src.index("k_ip_partial("):src.index("k_ip_final(")]
    assert "for (uint64_t i" in body and "for (uint32_t i" not in body
    lib = ffi.load()
    assert lib.bpgpu_ip_device(0, None, None, 1 << 32, None) == -1
    assert lib.bpgpu_ip_device_noalloc(0, None, None, 1 << 32, None) == -1
