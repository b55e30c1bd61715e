#!/usr/bin/env python3
"""Bp::prove / Bp::verify at vector lengths up to the header's limit (BPGPU_IPA_MAX_N = 2^26), dev tool.

Generators are grown on the device (tools/maxsize_probe.py) and brought back as host Jacobian arrays, so the proof runs
through the host-vector entry points (two-phase generator upload, first round inside bpgpu_ipa_begin); the check is the
reference's own: Bp::verify accepts the proof (one MSM of 2N + 1 + 2 log2 N terms over the ORIGINAL generators) and
rejects a tampered one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bulletproofs_amd import ark_bp, ffi
from tools.maxsize_probe import View, grow_inputs


def main():
    ffi.init(0)
    curve = 0
    for lg in [int(a) for a in sys.argv[1:]] or [22]:
        n = 1 << lg
        t0 = time.perf_counter()
        d_pts, d_sc = grow_inputs(curve, 2 * n + 1)
        pts = ffi.points_from_device(curve, d_pts, 2 * n + 1)
        sc = d_sc.download(np.uint64, count=2 * n * 4).reshape(2 * n, 4)
        d_pts.free()
        d_sc.free()
        A, B, q = pts[:n], pts[n:2 * n], pts[2 * n:]
        a, b = sc[:n], sc[n:]
        t1 = time.perf_counter()
        result = ark_bp.commit(curve, A, B, q, a, b)
        seed = b"maxsize-ipa-probe".ljust(32, b"\0")
        pf = ark_bp.bp_prove(curve, A, B, q, a, b, seed)
        t2 = time.perf_counter()
        vms = ark_bp.bp_verify(curve, A, B, q, result, pf, seed)
        bad = dict(pf)
        bad["final_a"] = pf["final_a"].copy()
        bad["final_a"][0] ^= np.uint64(1)
        try:
            ark_bp.bp_verify(curve, A, B, q, result, bad, seed)
            rejected = False
        except ark_bp.Panic:
            rejected = True
        print(f"2^{lg}: inputs {t1 - t0:.1f} s, prove {pf['prove_ms']:.1f} ms ({len(pf['ls'])} rounds), verify {vms:.1f} ms, "
              f"tampered proof rejected: {rejected}", flush=True)
        if not rejected:
            sys.exit(1)


if __name__ == "__main__":
    main()
