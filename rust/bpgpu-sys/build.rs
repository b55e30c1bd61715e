// Builds (or finds) libbpgpu.a and tells cargo how to link it.
//
//   BPGPU_ROOT   checkout of this repository (default: two directories up from this crate)
//   ROCM_PATH    ROCm installation (default /opt/rocm)
//   BPGPU_NO_MAKE=1  do not run make: link the archive that is already there
//
// The archive is one hipcc compile of bulletproofs_amd/csrc/bpgpu.hip for gfx950
// (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -fPIC ... && ar rcs libbpgpu.a bpgpu.o): the repository's Makefile
// target `bulletproofs_amd/lib/libbpgpu.a`.  Its link line (static bpgpu + dylib amdhip64 + the C++ runtime) is the one
// bulletproofs_amd/bin/msm_bench_static is linked with in that Makefile.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let here = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let root = env::var("BPGPU_ROOT").map(PathBuf::from).unwrap_or_else(|_| here.join("../.."));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    let lib_dir = root.join("bulletproofs_amd/lib");
    if env::var("BPGPU_NO_MAKE").is_err() {
        let ok = Command::new("make")
            .arg("-C")
            .arg(&root)
            .arg("bulletproofs_amd/lib/libbpgpu.a")
            .status()
            .expect("could not run make")
            .success();
        assert!(ok, "make bulletproofs_amd/lib/libbpgpu.a failed (hipcc, gfx950)");
    }
    assert!(lib_dir.join("libbpgpu.a").exists(), "libbpgpu.a not found under {}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=static=bpgpu");
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64"); // HIP runtime
    println!("cargo:rustc-link-lib=dylib=stdc++"); // the library's host side is C++
    println!("cargo:rustc-link-lib=dylib=pthread");
    for f in ["bulletproofs_amd/csrc/bpgpu.hip", "bulletproofs_amd/csrc/kernels.cuh", "bulletproofs_amd/csrc/curve.cuh",
              "bulletproofs_amd/csrc/field.cuh", "include/bpgpu.h"] {
        println!("cargo:rerun-if-changed={}", root.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=BPGPU_ROOT");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
