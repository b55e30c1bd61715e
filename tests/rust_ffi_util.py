"""Parsers shared by tools/gen_rust_ffi.py and tests/test_rust_ffi.py: C prototypes of include/bpgpu.h and the
`extern "C"` declarations of rust/bpgpu-sys/src/lib.rs, both reduced to (name, return type, [(type, name)])."""
import re

C_TO_RUST = {
    "int": "i32", "size_t": "usize", "uint64_t": "u64", "void": "()", "float": "f32", "double": "f64",
    "const char*": "*const c_char",
    "const uint64_t*": "*const u64", "uint64_t*": "*mut u64",
    "const uint8_t*": "*const u8", "uint8_t*": "*mut u8",
    "const int*": "*const i32", "float*": "*mut f32", "double*": "*mut f64", "size_t*": "*mut usize",
    "const void*": "*const c_void", "void*": "*mut c_void", "void**": "*mut *mut c_void",
    "bpgpu_ipa*": "*mut bpgpu_ipa", "const bpgpu_ipa*": "*const bpgpu_ipa", "bpgpu_ipa**": "*mut *mut bpgpu_ipa",
    "bpgpu_sharded_gens*": "*mut bpgpu_sharded_gens", "bpgpu_sharded_gens**": "*mut *mut bpgpu_sharded_gens",
}


def rust_type(c_type):
    return C_TO_RUST[c_type]


def _norm_c_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", "*", t)
    return t


def c_prototypes(text):
    """[(name, return type, [(arg type, arg name)])] of every bpgpu_* prototype; `T x[4]` counts as `T*`."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(bpgpu_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = _norm_c_type(m.group(1)), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[\d*\]$", a)
                if arr:
                    a = a[:arr.start()].strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                t = _norm_c_type(mm.group(1))
                if arr:
                    t += "*"
                parsed.append((t, mm.group(2)))
        out.append((name, ret, parsed))
    return out


def rust_externs(text):
    """[(name, return type or '()', [(arg type, arg name)])] of every `fn bpgpu_*` inside extern "C" blocks."""
    text = re.sub(r"//.*$", "", text, flags=re.M)
    out = []
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn\s+(bpgpu_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            args = []
            for a in [x.strip() for x in m.group(2).split(",") if x.strip()]:
                n, t = a.split(":", 1)
                args.append((re.sub(r"\s+", " ", t.strip()), n.strip()))
            out.append((m.group(1), (m.group(3) or "()").strip(), args))
    return out
