"""The Rust side of the boundary as files (rust/): no Rust toolchain exists here, so this checks on the CPU what a
compiler would about the FFI surface -- rust/bpgpu-sys/src/lib.rs declares exactly the functions of include/bpgpu.h
(names, order-independent; argument counts; pointer / size types), and gpu.rs / the patches call only declared
functions with the right number of arguments."""
import os
import re

from tests.rust_ffi_util import c_prototypes, rust_externs, rust_type

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(*p):
    return open(os.path.join(ROOT, *p)).read()


def test_extern_block_matches_header():
    c = {n: (r, a) for n, r, a in c_prototypes(read("include", "bpgpu.h"))}
    rs = {n: (r, a) for n, r, a in rust_externs(read("rust", "bpgpu-sys", "src", "lib.rs"))}
    assert len(c) >= 55
    assert sorted(c) == sorted(rs), (sorted(set(c) - set(rs)), sorted(set(rs) - set(c)))
    for name, (ret, args) in c.items():
        rret, rargs = rs[name]
        assert rret == rust_type(ret), (name, ret, rret)
        assert len(args) == len(rargs), name
        for (ct, cn), (rt, rn) in zip(args, rargs):
            assert rt == rust_type(ct), (name, cn, ct, rt)
            assert rn == cn, (name, cn, rn)


def test_header_constants_match():
    h, rs = read("include", "bpgpu.h"), read("rust", "bpgpu-sys", "src", "lib.rs")
    for name in ("BPGPU_E_ARG", "BPGPU_E_HIP", "BPGPU_E_NOINIT", "BPGPU_E_STATE"):
        cv = int(re.search(rf"#define {name} \((-?\d+)\)", h).group(1))
        rv = int(re.search(rf"pub const {name}: i32 = (-?\d+);", rs).group(1))
        assert cv == rv, name
    assert re.search(r"#define BPGPU_MSM_MAX_N \(\(size_t\)1 << 28\)", h) and "BPGPU_MSM_MAX_N: usize = 1 << 28" in rs
    assert re.search(r"#define BPGPU_IPA_MAX_N \(\(size_t\)1 << 26\)", h) and "BPGPU_IPA_MAX_N: usize = 1 << 26" in rs
    assert "#define BPGPU_PALLAS 0" in h and "pub const PALLAS: i32 = 0;" in rs
    assert "#define BPGPU_VESTA 1" in h and "pub const VESTA: i32 = 1;" in rs


def _calls(text):
    """(name, number of top-level arguments) of every `sys::bpgpu_*(...)` call"""
    out = []
    for m in re.finditer(r"sys::(bpgpu_[a-z0-9_]+)\s*\(", text):
        i, depth, args, cur = m.end(), 1, 0, False
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1
                cur = False
                i += 1
                continue
            if depth and not ch.isspace():
                cur = True
            i += 1
        out.append((m.group(1), args + (1 if cur else 0)))
    return out


def test_gpu_module_calls_declared_functions_with_right_arity():
    decl = {n: len(a) for n, _, a in c_prototypes(read("include", "bpgpu.h"))}
    calls = _calls(read("rust", "ark-bp-gpu", "gpu.rs"))
    assert len(calls) >= 12
    for name, nargs in calls:
        assert name in decl, name
        assert decl[name] == nargs, (name, decl[name], nargs)
    used = {n for n, _ in calls}
    for must in ("bpgpu_msm", "bpgpu_msm_sharded", "bpgpu_ipa_begin", "bpgpu_ipa_round_lr", "bpgpu_ipa_round_fold",
                 "bpgpu_ipa_finish", "bpgpu_ipa_export", "bpgpu_ipa_free", "bpgpu_ipa_verify_msm", "bpgpu_fold_points", "bpgpu_ip"):
        assert must in used, must


def test_patches_name_real_hooks():
    gpu = read("rust", "ark-bp-gpu", "gpu.rs")
    hooks = set(re.findall(r"pub fn ([a-z0-9_]+)", gpu))
    for patch in ("msm.rs.patch", "bp_iter.rs.patch", "bp_2ary_step.rs.patch"):
        text = read("rust", "patches", patch)
        called = set(re.findall(r"crate::gpu::([a-z0-9_]+)", text))
        assert called and called <= hooks, (patch, called - hooks)
        assert '#[cfg(feature = "gpu")]' in text
    crate = read("rust", "patches", "crate.patch")
    assert "bpgpu-sys" in crate and 'gpu = [ "bpgpu-sys" ]' in crate and "pub mod gpu;" in crate
    build = read("rust", "bpgpu-sys", "build.rs")
    assert "rustc-link-lib=static=bpgpu" in build and "amdhip64" in build and "bulletproofs_amd/lib/libbpgpu.a" in build
    # the Makefile target build.rs asks for exists
    assert re.search(r"^\$\(LIBDIR\)/libbpgpu\.a:", read("Makefile"), flags=re.M)
