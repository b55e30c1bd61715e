#!/usr/bin/env python3
"""Headline benchmark: variable-base MSM over Pallas on MI355X (BASELINE.json metric
"MSM points/sec"), one process per GPU.

  python bench.py --gpus N --steps K --warmup W [--log2n 20] [--curve pallas]

A "step" is one MSM of 2^log2n (point, scalar) pairs per rank over operands already
resident in HBM (the IPA-relevant case: generators are fixed public parameters), result
returned to the host as a normalised point.  With N > 1 every rank owns a contiguous
point range (weak scaling: 2^log2n pairs per rank); each step ends with the one real
exchange the sharded MSM has, an all_gather of the 72-byte partial results over
RCCL/xGMI followed by their sum (bpgpu_points_sum).  Rank 0 prints ONE JSON line.

Inputs are synthetic.  N = 1: the reference's own recipe (examples/msm_bench.rs:40-44, one
ark_std::test_rng() stream: n points by G::rand, then n scalars), sampled through the host mirror
with the square roots taken on the device.  N > 1: every rank grows its own points from the
generator (-1, 2) by repeated generator-fold calls with seeded random multipliers (no oracle
involved), scalars uniform in [0, 2^254).  The cpu_baseline leg (rank 0, N = 1 only) times oracle/pasta_oracle.c --
a restatement of the reference's pippenger_msm -- on a bounded sample of the same bytes
and checks the GPU result of that sample against it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MADS_PER_FIELD_MUL = 136   # 8-limb CIOS = 2*8^2 + 8 multiply-adds
MULS_PER_MIXED_ADD = 11    # mixed Jacobian add


def synth_points(ffi, curve, n, seed):
    """n distinct curve points as (n, 12) u64 ark-form Jacobian (Z = 1), grown from the
    generator by fold_points with random multipliers: V <- V ++ (a V + b roll(V))."""
    rng = np.random.default_rng(seed)
    p_mod = [0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
             0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001][curve]
    R = 1 << 256

    def fe(v):
        return np.frombuffer((v * R % p_mod).to_bytes(32, "little"), dtype=np.uint64)

    V = np.concatenate([fe(p_mod - 1), fe(2), fe(1)]).reshape(1, 12).copy()

    def rand_scalar():
        s = rng.integers(0, 1 << 63, size=4, dtype=np.uint64)
        s[3] &= np.uint64((1 << 61) - 1)
        return s

    while len(V) < n:
        m = min(len(V), n - len(V))
        g = np.concatenate([V[:m], np.roll(V, 1, axis=0)[:m]])
        U = ffi.fold_points(curve, g, rand_scalar(), rand_scalar())
        V = np.concatenate([V, U])
    return V[:n]


def synth_scalars(n, seed):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 62) - 1)  # < 2^254 < r: valid Montgomery limbs of a uniform element
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20, help="log2 of the points per rank")
    ap.add_argument("--curve", default="pallas", choices=["pallas", "vesta"])
    ap.add_argument("--window", type=int, default=0, help="override the MSM window width (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="log2 of the CPU-baseline sample size")
    ap.add_argument("--ipa-log2n", type=int, default=20, help="log2 N of the secondary IPA-prove measurement (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --one-device rehearses N > 1 ranks on a single GPU")
    ap.add_argument("--grown-points", action="store_true",
                    help="N = 1 only: use the per-rank synthetic inputs of the N > 1 runs instead of the msm_bench recipe")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    from bulletproofs_amd import ffi
    ffi.init(dev_index)
    if args.window:
        ffi.set_window_bits(args.window)
    curve = 0 if args.curve == "pallas" else 1
    n = 1 << args.log2n

    if world == 1 and not args.grown_points:
        # the reference's own recipe (examples/msm_bench.rs:40-44): one test_rng() stream, n points by G::rand
        # (square roots of the candidates on the device), then n scalars
        from bulletproofs_amd import ark_bp
        pts, sc = ark_bp.msm_bench_inputs(curve, n)
        recipe = "examples/msm_bench.rs:40-44 (ark_std::test_rng stream: n x G::rand, n x Fr::rand)"
    else:
        pts = synth_points(ffi, curve, n, seed=1000 + rank)
        sc = synth_scalars(n, seed=2000 + rank)
        recipe = "per-rank seeded: points grown from the generator by random generator folds, scalars uniform below 2^254"
    d_pts = ffi.points_to_device(curve, pts, fmt=0)
    d_sc = ffi.scalars_to_device(sc)

    from bulletproofs_amd import sharded
    dev = torch.device("cuda", dev_index) if args.backend == "nccl" else None

    def step():
        part = ffi.msm_device(curve, d_pts, d_sc, n)
        return sharded.combine_partials(curve, part, dev), part

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phase = np.zeros(8)
    entries = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result, part = step()
        phase += np.array(ffi.last_msm_timings())
        entries += ffi.last_msm_entries()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phase /= max(1, args.steps)

    # PCIe-inclusive rate (host buffers in, like examples/msm_bench.rs:45-48), a few repetitions
    h2h = None
    if rank == 0:
        ffi.msm(curve, pts, sc)
        samples = []
        for _ in range(5):
            t1 = time.perf_counter()
            ffi.msm(curve, pts, sc)
            samples.append(time.perf_counter() - t1)
        h2h = sorted(samples)[len(samples) // 2]  # median: the staging threads share the host with everything else

    out = None
    if rank == 0:
        mad_peak = ffi.measure_mad_peak()
        c_bits = args.window or ffi.auto_window_bits(n)
        W = (256 + c_bits - 1) // c_bits
        accum_s = phase[3] * 1e-3
        # additions the kernel actually performed: one per non-zero digit (measured by the library).  With
        # c = 17 the 16th window is empty for Pasta scalars (15 x 17 = 255 bits, bit 254 is clear below r), so
        # this is 15 n, not windows x n
        adds = entries / max(1, args.steps)
        algo_mads = adds * MULS_PER_MIXED_ADD * MADS_PER_FIELD_MUL
        achieved = algo_mads / accum_s if accum_s > 0 else 0.0
        out = {
            "metric": "MSM points/sec (Pallas, variable-base)" if curve == 0 else "MSM points/sec (Vesta, variable-base)",
            "value": world * n * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (255-bit Montgomery, 9x29-bit, int64 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"2^{args.log2n}-point {args.curve} MSM per GPU, operands resident in HBM",
                       "inputs": recipe, "points_per_gpu": n, "total_points": world * n, "window_bits": c_bits, "windows": W,
                       "parallelism": f"point-range shards x{world}, all_gather of 72 B partials" if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "kernel": "k_accum (bucket accumulation, mixed XYZZ adds)",
                         "achieved": achieved / 1e12, "peak": mad_peak / 1e12, "unit": "Tmad/s",
                         "frac": achieved / mad_peak if mad_peak else None,
                         "traffic": pmc_traffic("k_accum", False) if (args.log2n == 20 and curve == 0 and not args.window) else None,
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC of this command (profiles/r01_pmc_summary.csv): "
                                         "FETCH_SIZE + WRITE_SIZE, raw (64-byte gathers); algorithmic = mixed additions x 68 B",
                         "algorithmic_bytes": adds * 68, "mixed_additions_per_launch": adds,
                         "avg_kernel_ms": phase[3],
                         # the same kernel against the HBM roof (8 TB/s): far below it, which is why the bound is "valu"
                         "hbm_view": {"achieved": adds * 68 / accum_s / 1e9 if accum_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                                      "frac": adds * 68 / accum_s / 8e12 if accum_s > 0 else 0.0},
                         "note": "algorithmic mads = non-zero digits (mixed additions) * 11 field-mul * 136 mad (SURVEY 8d); peak = measured v_mad_u64_u32 issue rate"},
            "phases_ms": {"digits+hist": phase[0], "scan+scatter_to_bins": phase[1], "sort_in_bins": phase[2], "accumulate": phase[3],
                          "reduce": phase[5], "host_tail": phase[6], "gpu_total": phase[7]},
            "host_to_host": {"ms": h2h * 1e3, "points_per_s": n / h2h,
                             "note": "host buffers in, PCIe + Jacobian->affine conversion included (msm_bench.rs:45-48 window); median of 5"},
        }
        if world == 1:
            out["roofline_hbm"] = scalar_kernels_leg(ffi, curve)
        if world == 1 and args.ipa_log2n:
            out["ipa_prove"] = ipa_prove_leg(ffi, curve, args.ipa_log2n)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ffi, curve, pts, sc, min(args.cpu_log2n, args.log2n))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary.csv")


def pmc_traffic(kernel, streaming):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/profile_round.sh -> tools/pmc_summary.py; FETCH_SIZE and WRITE_SIZE are in KiB).  For wide
    coalesced streaming reads gfx950's FETCH_SIZE reports half the bytes (MI355X_MICROARCH.md, HBM):
    doubled when `streaming`; left raw for 64-byte gathers, which that note calls uncalibrated.
    None when the summary is missing."""
    try:
        vals = {}
        for line in open(PMC_SUMMARY):
            f = line.strip().split(",")
            if len(f) == 4 and f[0].startswith(kernel) and f[2] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[2]] = float(f[3]) * 1024.0
        if len(vals) != 2:
            return None
        return vals["FETCH_SIZE"] * (2.0 if streaming else 1.0) + vals["WRITE_SIZE"]
    except OSError:
        return None


def scalar_kernels_leg(ffi, curve, log2n=24):
    """HBM roofline of the two streaming kernels of the fold (north-star: "HBM GB/s for scalar folding"):
    k_fold_scalars (a' = x a_lo + x^-1 a_hi, bp_iter.rs:88-95: 64 B read + 32 B written per output) and
    k_ip_partial (<a, b>, util/mod.rs:17-23: 64 B read per term), on vectors far larger than the 256 MiB
    Infinity Cache, timed with HIP events on the library's stream."""
    n = 1 << log2n
    v = synth_scalars(2 * n, 91)
    d_v = ffi.scalars_to_device(v)
    d_o = ffi.DeviceBuffer(n * 32)
    x, xi = synth_scalars(1, 92)[0], synth_scalars(1, 93)[0]
    best_fold, best_ip = 1e9, 1e9
    for _ in range(5):
        ffi.fold_scalars_device(curve, d_v, n, x, xi, d_o)
        best_fold = min(best_fold, ffi.last_op_ms())
    d_b = ffi.DeviceBuffer(n * 32)
    d_b.upload(v[n:])
    for _ in range(5):
        ffi.ip_device(curve, d_o, d_b, n)
        best_ip = min(best_ip, ffi.last_op_ms())
    for b in (d_v, d_o, d_b):
        b.free()
    peak = 8000.0  # GB/s, HBM3E spec (MI355X_MICROARCH.md); about 6300 GB/s is achievable
    fold_gbs = n * 96 / (best_fold * 1e-3) / 1e9
    ip_gbs = n * 64 / (best_ip * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_fold_scalars", "achieved": fold_gbs, "peak": peak, "unit": "GB/s",
            "frac": fold_gbs / peak, "traffic": pmc_traffic("k_fold_scalars", True) if (log2n == 24 and curve == 0) else None,
            "traffic_note": "bytes per launch, rocprofv3 PMC (profiles/r01_pmc_summary.csv): 2 x FETCH_SIZE + WRITE_SIZE; algorithmic = outputs x 96 B",
            "avg_kernel_ms": best_fold,
            "outputs": n, "algorithmic_bytes_per_output": 96,
            "k_ip": {"achieved": ip_gbs, "frac": ip_gbs / peak, "kernel_ms": best_ip, "terms": n,
                     "algorithmic_bytes_per_term": 64}}


def ipa_prove_leg(ffi, curve, log2n):
    """Secondary metric of BASELINE.json: classic Bulletproofs IPA prove (Bp::prove, reference
    src/proofs/bp_iter.rs:49-128) at N = 2^log2n through the C++ host side; the timed window is the
    reference's own (around i.prove, src/lib.rs:119-121): host vectors in, proof out, Fiat-Shamir
    on the host between rounds."""
    from bulletproofs_amd import ark_bp
    n = 1 << log2n
    pts = synth_points(ffi, curve, 2 * n + 1, seed=77)
    a, b = synth_scalars(n, 78), synth_scalars(n, 79)
    seed = bytes(range(32))
    ms = []
    for _ in range(3):
        pf = ark_bp.bp_prove(curve, pts[:n], pts[n:2 * n], pts[2 * n:], a, b, seed)
        ms.append(pf["prove_ms"])
    best = min(ms)
    ref_s = 527.8 if (log2n == 20 and curve == 0) else None  # analysis/main/data.csv:32 (unstated CPU)
    ref_v = 22.88 if (log2n == 20 and curve == 0) else None  # same row, verify
    # Bp::verify (bp_iter.rs:130-178; window of src/lib.rs:127-131) on that proof, and both with the
    # generators -- fixed public parameters -- already in HBM
    result = ark_bp.commit(curve, pts[:n], pts[n:2 * n], pts[2 * n:], a, b)
    v_ms = min(ark_bp.bp_verify(curve, pts[:n], pts[n:2 * n], pts[2 * n:], result, pf, seed) for _ in range(3))
    gens = ark_bp.DeviceGens(curve, pts[:n], pts[n:2 * n], pts[2 * n:])
    pc_ms = min(ark_bp.bp_prove_cached(gens, a, b, seed)["prove_ms"] for _ in range(3))
    vc_ms = min(ark_bp.bp_verify_cached(gens, result, pf, seed) for _ in range(3))
    gens.free()
    return {"n": n, "prove_ms": best, "runs_ms": ms, "rounds": len(pf["ls"]), "proof_elements": 2 * (1 + len(pf["ls"])),
            "reference_published_s": ref_s, "speedup_vs_reference_published": (ref_s * 1e3 / best) if ref_s else None,
            "verify_ms": v_ms, "reference_published_verify_s": ref_v,
            "prove_ms_generators_resident": pc_ms, "verify_ms_generators_resident": vc_ms,
            "note": "host-resident inputs, upload + 2*log2(N) MSMs + folds + host Fiat-Shamir; verify = transcript replay + "
                    "one device MSM of 2N+1+2log2(N) terms; reference numbers are from unknown CPU hardware"}


def cpu_baseline(ffi, curve, pts, sc, log2s):
    """Oracle B (restatement of pippenger_msm, threads over windows like rayon at msm.rs:119)
    on the first 2^log2s pairs of the bench inputs; the only place bench.py touches oracle/."""
    import ctypes
    from oracle import c_oracle as co
    ns = 1 << log2s
    threads = min(16, os.cpu_count() or 1)
    co.load(threads)
    t0 = time.perf_counter()
    want = co.msm_pippenger(curve, pts[:ns], sc[:ns], threads=threads)
    dt = time.perf_counter() - t0
    got = ffi.msm(curve, pts[:ns], sc[:ns])
    ok = (int(got[8]) == 0 and bool((got[:8] == want[:8]).all())) or (int(got[8]) == 1 and not want[8:].any())
    t1 = time.perf_counter()
    co.msm_pippenger(curve, pts[:ns], sc[:ns], threads=1) if log2s <= 15 else None
    dt1 = time.perf_counter() - t1 if log2s <= 15 else None
    return {"value": ns / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"first 2^{log2s} pairs of the bench inputs, window c={co.load(threads).po_pippenger_window(ctypes.c_size_t(ns))}, {dt:.2f} s",
            "single_thread_points_per_s": (ns / dt1) if dt1 else None,
            "gpu_matches_cpu_on_sample": bool(ok)}


if __name__ == "__main__":
    main()
