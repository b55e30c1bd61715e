"""A/B of Bp::verify from host vectors (dev tool): alternate an environment switch between calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi
ffi.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
switch = sys.argv[2] if len(sys.argv) > 2 else "BPGPU_VERIFY_NO_PIPE"
n = 1 << k
A, B, q, a, b, seed = ark_bp.ipa_inputs(0, n)
res = ark_bp.commit(0, A, B, q, a, b)
pf = ark_bp.bp_prove(0, A, B, q, a, b, seed)
out = {0: [], 1: []}
for rep in range(16):
    on = rep & 1
    if on: os.environ[switch] = "1"
    else: os.environ.pop(switch, None)
    ms = ark_bp.bp_verify(0, A, B, q, res, pf, seed)
    if rep >= 2: out[on].append(ms)
for on in (0, 1):
    v = sorted(out[on]); print(f"{switch}={on}: verify min {v[0]:.3f} median {v[len(v)//2]:.3f} max {v[-1]:.3f} ms")
