"""Host-side cost of the one-process sharded MSM (bpgpu_sharded_*): the same 2^24-point MSM with resident generators as
ONE leg and as EIGHT legs, all on GPU 0 (a one-GPU box: the eight legs share the device, so the difference between the
two is what eight host threads, eight Horner tails, eight scalar uploads and the combine add on the host side -- plus what
eight smaller MSMs lose against one large one on the device).  python tools/bench_8ctx.py [log2n] > profiles/r03_bench_8ctx_one_device.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import ark_bp, ffi  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
ffi.init(0)
pts, sc = ark_bp.msm_bench_inputs(0, n)
out = {"workload": f"2^{lg}-point Pallas MSM, generators resident (bpgpu_sharded_gens_msm), scalars from host memory", "legs": {}}
ref = None
for P in (1, 2, 4, 8):
    ffi.shutdown()
    ffi.init(0)
    ffi.sharded_init([0] * P)
    gens = ffi.ShardedGens(0, pts)
    for _ in range(2):
        r = gens.msm(sc)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        r = gens.msm(sc)
        ts.append(time.perf_counter() - t)
    if ref is None:
        ref = r
    assert (r == ref).all(), "legs disagree"
    gens.free()
    out["legs"][str(P)] = {"contexts_on_device_0": P, "ms_median": sorted(ts)[2] * 1e3, "ms_min": min(ts) * 1e3,
                           "points_per_s": n / sorted(ts)[2]}
# the single-context device-resident MSM (scalars already in HBM) for scale
ffi.shutdown()
ffi.init(0)
d_p, d_s = ffi.points_to_device(0, pts), ffi.scalars_to_device(sc)
for _ in range(2):
    ffi.msm_device(0, d_p, d_s, n)
ts = []
for _ in range(5):
    t = time.perf_counter()
    r = ffi.msm_device(0, d_p, d_s, n)
    ts.append(time.perf_counter() - t)
assert (r == ref).all()
out["single_context_all_resident_ms"] = sorted(ts)[2] * 1e3
out["scalar_upload_bytes"] = n * 32
out["note"] = ("all legs of a sharded call run on ONE physical GPU here: not a scaling number.  legs[1] - "
               "single_context_all_resident_ms = the scalar upload (PCIe); legs[8] - legs[1] = host threads, eight tails, "
               "the combine, and the device-side loss of eight 2^21-point MSMs sharing the GPU against one 2^24-point MSM")
out["host_threads"] = {"cores_seen": os.cpu_count(), "BPGPU_HOST_CORES": os.environ.get("BPGPU_HOST_CORES")}
print(json.dumps(out))
