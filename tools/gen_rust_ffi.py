#!/usr/bin/env python3
"""Prints the `extern "C"` block of rust/bpgpu-sys/src/lib.rs from include/bpgpu.h (one Rust declaration per C
prototype).  tests/test_rust_ffi.py checks the committed file against the header with its own parser."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.rust_ffi_util import c_prototypes, rust_type  # noqa: E402

for name, ret, args in c_prototypes(open(os.path.join(ROOT, "include", "bpgpu.h")).read()):
    a = ", ".join(f"{n}: {rust_type(t)}" for t, n in args)
    r = "" if ret == "void" else f" -> {rust_type(ret)}"
    print(f"    pub fn {name}({a}){r};")
