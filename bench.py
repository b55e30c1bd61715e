#!/usr/bin/env python3
"""Headline benchmark: variable-base MSM over Pallas on MI355X (BASELINE.json metric
"MSM points/sec"), one process per GPU.

  python bench.py --gpus N --steps K --warmup W [--log2n 20] [--curve pallas]

A "step" is one MSM over operands already resident in HBM (the IPA-relevant case: generators are
fixed public parameters), result returned to the host as a normalised point.
N = 1: 2^20 (point, scalar) pairs on one GPU (BASELINE config 2).
N > 1: BASELINE config 3, STRONG scaling: ONE MSM of 2^24 pairs in total (--log2n sets the total), cut into
contiguous point ranges [g n / N, (g+1) n / N), one per rank; every step ends with the one real exchange the
sharded MSM has, an all_gather of the 72-byte partial results over RCCL/xGMI followed by their sum
(bpgpu_points_sum).  Rank 0 prints ONE JSON line.

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

# Two ways to count the multiply-adds of a group operation:
#  * the SURVEY 8(d) yardstick (config-independent): field mul = 8-limb CIOS = 2*8^2 + 8 = 136 mads; mixed
#    Jacobian add = 11 field muls, doubling = 7, full add = 16;
#  * what the compiled kernels execute (curve.cuh / field.cuh: 9 x 29-bit limbs, p = 2^254 + t): a product is 81
#    mads (a square 45), a Montgomery reduction 45; XYZZ mixed add madd-2008-s = 6 products + 2 squares + one
#    two-product accumulation (Y3) with 9 reductions = 1143 mads; doubling dbl-2008-s-1 = 4 products + 3 squares
#    (Y3 two products, one reduction) with 8 reductions = 981 mads.
# `frac` uses the yardstick (comparable across implementations); `frac_executed` uses the second count.
MADS_PER_FIELD_MUL = 136
MULS_PER_MIXED_ADD = 11
MULS_PER_DOUBLE = 7
EXEC_MADS_MIXED_ADD = 6 * 81 + 2 * 45 + 2 * 81 + 9 * 45   # 1143
EXEC_MADS_DOUBLE = 4 * 81 + 3 * 45 + 2 * 81 + 8 * 45      # 981
EXEC_MADS_FIELD_MUL = 81 + 45                              # 126
# the generator folds and the fixed-base table run their chains in Jacobian coordinates (curve.cuh, jdbl / jmadd):
# doubling 4 squares + 3 products with 7 reductions, mixed addition 3 squares + 6 products + the two-product
# accumulation with 10 reductions
EXEC_MADS_JAC_DOUBLE = 4 * 45 + 3 * 81 + 7 * 45            # 738
EXEC_MADS_JAC_MIXED_ADD = 3 * 45 + 6 * 81 + 2 * 81 + 10 * 45  # 1233


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
    ap.add_argument("--log2n", type=int, default=None,
                    help="log2 of the MSM size: points on the one GPU (default 20), or TOTAL points over all ranks when "
                         "--gpus > 1 (default 24: config 3)")
    ap.add_argument("--weak", action="store_true", help="N > 1: --log2n is per rank (weak scaling) instead of the total")
    ap.add_argument("--curve", default="pallas", choices=["pallas", "vesta"])
    ap.add_argument("--window", type=int, default=0, help="override the MSM window width (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="log2 of the CPU-baseline sample size")
    ap.add_argument("--ipa-log2n", type=int, default=20, help="log2 N of the secondary IPA-prove measurement (0 = skip)")
    ap.add_argument("--ipa-cpu-log2n", type=int, default=18,
                    help="log2 N of the CPU baseline of Bp::prove (Oracle B on the host cores; 2^18 takes about 8 s on 16 threads, 2^20 half a minute)")
    ap.add_argument("--second-log2n", type=int, default=24,
                    help="N = 1: log2 of the second MSM size of BASELINE's metric, reported as msm_2_<log2n> (0 = skip)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the roofline_next leg (SURVEY 8 f3/f4 kernels)")
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
    from bulletproofs_amd import sharded
    if args.log2n is None:
        args.log2n = 20 if world == 1 else 24
    strong = world > 1 and not args.weak
    if strong:
        total = 1 << args.log2n
        lo, hi = sharded.shard_range(total, rank, world)
        n = hi - lo
    else:
        n = 1 << args.log2n
        total = world * n

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

    dev = torch.device("cuda", dev_index) if args.backend == "nccl" else None

    gather_s = [0.0]

    def step():
        part = ffi.msm_device(curve, d_pts, d_sc, n)
        t_g = time.perf_counter()
        total_pt = sharded.combine_partials(curve, part, dev)  # N > 1: the all_gather of the 72-byte partials + their sum
        gather_s[0] += time.perf_counter() - t_g
        return total_pt, part

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phase = np.zeros(8)
    entries = 0
    sync()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result, part = step()
        phase += np.array(ffi.last_msm_timings())
        entries += ffi.last_msm_entries()
    sync()
    dt = time.perf_counter() - t0
    phase /= max(1, args.steps)
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank's accumulate-phase time and addition count, for the per-GPU roofline fractions
        mine = torch.tensor([phase[3], entries / max(1, args.steps), float(n), phase[7]], dtype=torch.float64,
                            device="cuda" if args.backend == "nccl" else "cpu")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in r.cpu()] for r in allr]
        # which physical device every rank ran on (a SCALE record must show N distinct GPUs) and its exchange time
        idents = [None] * world
        dist.all_gather_object(idents, {"rank": rank, "local_rank": local_rank, "device_index": dev_index,
                                        "pci": device_identity(torch, dev_index), "allgather_us": gather_s[0] / max(1, args.steps) * 1e6})

    # PCIe-inclusive rate (host buffers in, like examples/msm_bench.rs:45-48), a few repetitions
    h2h = None
    if rank == 0:
        for _ in range(3):  # the first calls on freshly allocated host arrays run at half speed (tools/h2h_probe.py)
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
        yard = MULS_PER_MIXED_ADD * MADS_PER_FIELD_MUL
        # additions the kernel actually performed: one per non-zero digit (measured by the library).  With
        # c = 17 the 16th window is empty for Pasta scalars (15 x 17 = 255 bits, bit 254 is clear below r), so
        # this is 15 n, not windows x n
        adds = entries / max(1, args.steps)
        algo_mads = adds * MULS_PER_MIXED_ADD * MADS_PER_FIELD_MUL
        achieved = algo_mads / accum_s if accum_s > 0 else 0.0
        out = {
            "metric": "MSM points/sec (Pallas, variable-base)" if curve == 0 else "MSM points/sec (Vesta, variable-base)",
            "value": total * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (255-bit Montgomery, 9x29-bit, int64 accumulate)",
            "data": "synthetic",
            "config": {"workload": (f"2^{args.log2n}-point {args.curve} MSM sharded across {world} GPUs (contiguous point ranges), "
                                    "operands resident in HBM" if strong else
                                    f"2^{args.log2n}-point {args.curve} MSM per GPU, operands resident in HBM"),
                       "inputs": recipe, "points_per_gpu": n, "total_points": total, "window_bits": c_bits, "windows": W,
                       "parallelism": f"point-range shards x{world}, all_gather of 72 B partials" if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "kernel": "k_accum (bucket accumulation, mixed XYZZ adds)",
                         "achieved": achieved / 1e12, "peak": mad_peak / 1e12, "unit": "Tmad/s",
                         "frac": achieved / mad_peak if mad_peak else None,
                         "frac_executed": (adds * EXEC_MADS_MIXED_ADD / accum_s / mad_peak) if (mad_peak and accum_s > 0) else None,
                         "mads_per_mixed_add": {"yardstick": yard, "executed": EXEC_MADS_MIXED_ADD},
                         # counters cannot be read from inside the process: the figure comes from separate rocprofv3 --pmc passes of
                         # this same command, committed under profiles/ -- named for what it is, with the file's git blob hash
                         "traffic": None,
                         "traffic_from_profile": traffic_from_profile("k_accum", False) if (world == 1 and args.log2n == 20 and curve == 0 and not args.window) else None,
                         "algorithmic_bytes": adds * 68, "mixed_additions_per_launch": adds,
                         "avg_kernel_ms": phase[3],
                         # the same kernel against the HBM roof (8 TB/s): far below it, which is why the bound is "valu"
                         "hbm_view": {"achieved": adds * 68 / accum_s / 1e9 if accum_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                                      "frac": adds * 68 / accum_s / 8e12 if accum_s > 0 else 0.0},
                         "note": "frac: algorithmic mads = non-zero digits (mixed additions) * 11 field-mul * 136 mad (SURVEY 8d yardstick); "
                                 "frac_executed: the 1143 multiply-adds the compiled XYZZ mixed add issues per addition; peak = measured v_mad_u64_u32 issue rate"},
            "phases_ms": {"digits+hist": phase[0], "scan+scatter_to_bins": phase[1], "sort_in_bins": phase[2], "accumulate": phase[3],
                          "reduce": phase[5], "host_tail": phase[6], "gpu_total": phase[7]},
            "host_to_host": {"ms": h2h * 1e3, "points_per_s": n / h2h,
                             "note": "host buffers in, PCIe + Jacobian->affine conversion included (msm_bench.rs:45-48 window); median of 5"},
        }
        if per_rank is not None:
            out["per_gpu"] = [{"rank": r, "points": int(v[2]), "k_accum_ms": v[0], "gpu_total_ms": v[3],
                               "k_accum_frac": (v[1] * yard / (v[0] * 1e-3) / mad_peak) if (v[0] > 0 and mad_peak) else None,
                               "device_index": idents[r]["device_index"], "pci": idents[r]["pci"],
                               "allgather_us": idents[r]["allgather_us"]}
                              for r, v in enumerate(per_rank)]
            out["exchange"] = {"collective": "all_gather of one 72-byte partial per rank, then bpgpu_points_sum on every rank",
                               "backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                               "distinct_devices": len({i["pci"] for i in idents}),
                               "allgather_us_max": max(i["allgather_us"] for i in idents),
                               "note": "allgather_us = host wall time per step of D2H-free path: 72 B host -> device tensor, all_gather, "
                                       "device -> host, host sum; includes waiting for the slowest rank's MSM"}
        if world == 1 and args.second_log2n and args.second_log2n != args.log2n:
            for b in (d_pts, d_sc):
                b.free()
            out[f"msm_2_{args.second_log2n}"] = second_size_leg(ffi, curve, args.second_log2n, mad_peak)
        if world == 1:
            out["roofline_hbm"] = scalar_kernels_leg(ffi, curve, mad_peak)
        if world == 1 and not args.no_next_rows:
            out["roofline_next"] = next_rows_leg(ffi, curve, mad_peak)
        if world == 1 and args.ipa_log2n:
            out["roofline_fold"] = fold_kernels_leg(ffi, curve, args.ipa_log2n, mad_peak)
            out["ipa_prove"] = ipa_prove_leg(ffi, curve, args.ipa_log2n, 0 if args.no_cpu_baseline else args.ipa_cpu_log2n)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ffi, curve, pts, sc, min(args.cpu_log2n, args.log2n))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_pmc_summary.csv")
if not os.path.exists(PMC_SUMMARY):
    PMC_SUMMARY = os.path.join(ROOT, "profiles", "r02_pmc_summary.csv")


def device_identity(torch, index):
    """PCI address (or, failing that, the UUID / name) of the HIP device a rank runs on."""
    try:
        p = torch.cuda.get_device_properties(index)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is not None:
            return f"{dom or 0:04x}:{bus:02x}:{dev or 0:02x}"
        return str(getattr(p, "uuid", p.name))
    except Exception as e:  # identity is evidence, not a dependency of the measurement
        return f"unknown ({type(e).__name__})"


def git_blob_sha1(path):
    """The hash `git hash-object` gives the file: ties a number quoted from profiles/ to the committed bytes."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def traffic_from_profile(kernel, streaming):
    b = pmc_traffic(kernel, streaming)
    if b is None:
        return None
    return {"bytes_per_launch": b, "file": os.path.relpath(PMC_SUMMARY, ROOT), "git_blob_sha1": git_blob_sha1(PMC_SUMMARY),
            "counters": ("2 x FETCH_SIZE + WRITE_SIZE (wide streaming reads: gfx950 reports half, MI355X_MICROARCH.md)" if streaming
                         else "FETCH_SIZE + WRITE_SIZE, raw (64-byte gathers)"),
            "note": "NOT measured in this run: separate rocprofv3 --pmc passes of this command (tools/profile_round.sh), committed"}


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


def scalar_kernels_leg(ffi, curve, mad_peak, log2n=24):
    """Both ceilings of the two streaming kernels of the fold (north-star: "HBM GB/s for scalar folding"; SURVEY 8d:
    "report both ceilings and which one binds"): k_fold_scalars (a' = x a_lo + x^-1 a_hi, bp_iter.rs:88-95: 64 B read +
    32 B written and two field multiplications per output) and k_ip_partial (<a, b>, util/mod.rs:17-23: 64 B read and one
    multiplication per term), on vectors far larger than the 256 MiB Infinity Cache, timed with HIP events on the
    library's stream."""
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

    def valu(mults, executed, ms):
        # integer ceiling of the same launch: yardstick (136 mads per multiplication) and the executed multiply-adds
        if not mad_peak:
            return None
        return {"achieved_Tmad_s": n * mults * MADS_PER_FIELD_MUL / (ms * 1e-3) / 1e12, "peak_Tmad_s": mad_peak / 1e12,
                "frac": n * mults * MADS_PER_FIELD_MUL / (ms * 1e-3) / mad_peak,
                "frac_executed": n * executed / (ms * 1e-3) / mad_peak}
    # the fold's two products share one reduction (Fp::mul_add2): 2 x 81 + 45 multiply-adds
    fold_valu, ip_valu = valu(2, 2 * 81 + 45, best_fold), valu(1, EXEC_MADS_FIELD_MUL, best_ip)
    return {"bound": "hbm", "kernel": "k_fold_scalars", "achieved": fold_gbs, "peak": peak, "unit": "GB/s",
            "frac": fold_gbs / peak, "traffic": None,
            "traffic_from_profile": traffic_from_profile("k_fold_scalars", True) if (log2n == 24 and curve == 0) else None,
            "avg_kernel_ms": best_fold,
            "outputs": n, "algorithmic_bytes_per_output": 96,
            "valu_ceiling": fold_valu,
            "binds": "hbm: the kernel issues ~360 VALU instructions per output (unpack, two products with one reduction, "
                     "conditional-add canonicalisation, pack: profiles/, PMC SQ_INSTS_VALU), 0.17 ms of issue time against a "
                     "0.31-0.32 ms kernel; it was co-bound at ~580 instructions per output before the products shared a reduction",
            "k_ip": {"achieved": ip_gbs, "frac": ip_gbs / peak, "kernel_ms": best_ip, "terms": n,
                     "algorithmic_bytes_per_term": 64, "valu_ceiling": ip_valu, "binds": "hbm"}}


def second_size_leg(ffi, curve, log2n, mad_peak, steps=5, warmup=2):
    """BASELINE's metric names two sizes (2^20 and 2^24 points); this is the second one in the same run: same recipe
    (examples/msm_bench.rs:40-44), operands resident in HBM, same phases and roofline as the headline object."""
    from bulletproofs_amd import ark_bp
    n = 1 << log2n
    t0 = time.perf_counter()
    pts, sc = ark_bp.msm_bench_inputs(curve, n)
    gen_s = time.perf_counter() - t0
    d_pts = ffi.points_to_device(curve, pts, fmt=0)
    d_sc = ffi.scalars_to_device(sc)
    del pts, sc
    for _ in range(warmup):
        ffi.msm_device(curve, d_pts, d_sc, n)
    phase, entries = np.zeros(8), 0
    t0 = time.perf_counter()
    for _ in range(steps):
        ffi.msm_device(curve, d_pts, d_sc, n)
        phase += np.array(ffi.last_msm_timings())
        entries += ffi.last_msm_entries()
    dt = time.perf_counter() - t0
    phase /= steps
    adds = entries / steps
    for b in (d_pts, d_sc):
        b.free()
    c_bits = ffi.auto_window_bits(n)
    accum_s = phase[3] * 1e-3
    yard = MULS_PER_MIXED_ADD * MADS_PER_FIELD_MUL
    return {"metric": "MSM points/sec (Pallas, variable-base)" if curve == 0 else "MSM points/sec (Vesta, variable-base)",
            "value": n * steps / dt, "unit": "points/s", "n": n, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "window_bits": c_bits, "windows": (256 + c_bits - 1) // c_bits,
            "inputs": "examples/msm_bench.rs:40-44 recipe", "input_generation_s": gen_s,
            "phases_ms": {"digits+hist": phase[0], "scan+scatter_to_bins": phase[1], "sort_in_bins": phase[2], "accumulate": phase[3],
                          "reduce": phase[5], "host_tail": phase[6], "gpu_total": phase[7]},
            "roofline": {"bound": "valu", "kernel": "k_accum", "achieved": adds * yard / accum_s / 1e12, "peak": mad_peak / 1e12,
                         "unit": "Tmad/s", "frac": adds * yard / accum_s / mad_peak,
                         "frac_executed": adds * EXEC_MADS_MIXED_ADD / accum_s / mad_peak,
                         "mixed_additions_per_launch": adds, "avg_kernel_ms": phase[3], "algorithmic_bytes": adds * 68,
                         "traffic": None}}


def next_rows_leg(ffi, curve, mad_peak):
    """Kernel times of the 'next' rows of SURVEY 8 (f3 k-ary fold, f4 Com-R1CS vector work), HIP events on the library's
    stream around each operation's kernels (bpgpu_last_op_ms); inputs by the msm_bench recipe / seeded numpy."""
    from bulletproofs_amd import ark_bp
    out = {}
    r_mod = [0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
             0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001][curve]
    # hadamard_gp (util/mod.rs:70-78, used at com_r1cs_to_ipa.rs:110): 2^18 independent scalar multiplications
    n = 1 << 18
    pts, sc = ark_bp.msm_bench_inputs(curve, n)
    best = 1e9
    for _ in range(3):
        ffi.hadamard_points(curve, pts, sc)
        best = min(best, ffi.last_op_ms())
    # the reference's ladder per output: 255 doublings + one addition per set bit (127.5 on average)
    yard = (255 * MULS_PER_DOUBLE + 127.5 * MULS_PER_MIXED_ADD) * MADS_PER_FIELD_MUL
    # executed: 44 windows x 3 Jacobian doublings, two digits a window (7 of 8 non-zero), the 2P / 3P / 4P table
    execd = (132 + 2) * EXEC_MADS_JAC_DOUBLE + (2 * 44 * 7 / 8 + 1) * EXEC_MADS_JAC_MIXED_ADD + 88 * 7 / 8 / 2 * EXEC_MADS_FIELD_MUL
    out["k_hadamard_points"] = {"bound": "valu", "outputs": n, "kernel_ms": best, "outputs_per_s": n / (best * 1e-3),
                                "achieved": n * yard / (best * 1e-3) / 1e12, "peak": mad_peak / 1e12, "unit": "Tmad/s",
                                "frac": n * yard / (best * 1e-3) / mad_peak, "frac_executed": n * execd / (best * 1e-3) / mad_peak,
                                "note": "frac prices the REFERENCE's double-and-add (255 doublings + 127.5 additions, 7 / 11 field muls x 136 "
                                        "mads) against the kernel's time; the kernel itself runs a GLV-split, signed 3-bit windowed chain "
                                        "(132 Jacobian doublings + about 78 mixed additions): frac_executed"}
    # k-ary generator fold (ipa_to_bp_unroll.rs:253-260): ck = 2^16, k = 8, y = powers(x, 0..8) -> one launch of k_fold_multi
    ck, k = 1 << 16, 8
    x = synth_scalars(1, 31)[0]
    y = ffi.powers(curve, x, 0, k)
    d_g = ffi.points_to_device(curve, np.concatenate([pts, pts])[:ck * k])
    d_o = ffi.DeviceBuffer(ck * 64)
    best = 1e9
    for _ in range(3):
        ffi.fold_points_kary_device(curve, d_g, ck, k, y, d_o)
        best = min(best, ffi.last_op_ms())
    outputs, dbls, adds, kind = ffi.last_fold_ops()  # group operations of the whole call
    assert kind == 4, "unexpected fold kernel"
    yard = (dbls * MULS_PER_DOUBLE + adds * MULS_PER_MIXED_ADD) * MADS_PER_FIELD_MUL
    execd = dbls * EXEC_MADS_JAC_DOUBLE + adds * EXEC_MADS_JAC_MIXED_ADD
    # the reference's schedule for the same outputs: k - 1 scalar multiplications by double-and-add (mul of a 255-bit
    # power of x: 255 doublings + 127.5 additions) and k - 1 additions per output (scale_vec + sum_vecs, util/mod.rs:25-55)
    yard_ref = ck * (k - 1) * (255 * MULS_PER_DOUBLE + 128.5 * MULS_PER_MIXED_ADD) * MADS_PER_FIELD_MUL
    out["fold_points_kary"] = {"bound": "valu", "ck": ck, "k": k, "launches": -(-(k - 1) // 7), "kernel_ms": best,
                               "outputs_per_s": ck / (best * 1e-3), "doublings_per_output": dbls / outputs,
                               "mixed_additions_per_output": adds / outputs,
                               "achieved": yard / (best * 1e-3) / 1e12, "peak": mad_peak / 1e12,
                               "unit": "Tmad/s", "frac": yard / (best * 1e-3) / mad_peak, "frac_executed": execd / (best * 1e-3) / mad_peak,
                               "frac_reference_schedule": yard_ref / (best * 1e-3) / mad_peak,
                               "note": "y = powers(x, 0..k): the k - 1 shared multipliers ride ONE doubling chain (k_fold_multi, operands "
                                       "in global memory, one wave per SIMD at this size: latency-bound); frac / frac_executed price the "
                                       "group operations the call executed (bpgpu_last_fold_ops) by the SURVEY yardstick / the compiled "
                                       "Jacobian counts; frac_reference_schedule prices the reference's k - 1 double-and-add "
                                       "multiplications per output against the same time"}
    d_g.free()
    d_o.free()
    # one whole k-ary round (bp_rec_kary.rs:27-143) from host vectors at n = 2^17, k = 8: the 2 (k - 1) cross-term MSMs as
    # one batch, the two scalar folds and the two generator folds -- as the five C-ABI calls, and as the host mirror's
    # prove_step (which, like the reference, also clones and pads its vectors and runs the transcript)
    nk = 1 << 17
    A, B, q = pts[:nk], pts[nk:2 * nk], synth_points(ffi, curve, 3, seed=41)[2:]
    a, b = sc[:nk], sc[nk:2 * nk]
    yi = ffi.powers(curve, ark_bp.scalar_inverse(curve, x), 0, k)

    def abi_calls():
        ffi.kary_cross_terms(curve, A, B, q, a, b, k)
        ffi.fold_scalars_kary(curve, a, k, y)
        ffi.fold_scalars_kary(curve, b, k, yi)
        ffi.fold_points_kary(curve, A, k, yi)
        ffi.fold_points_kary(curve, B, k, y)
    res = ark_bp.commit(curve, A, B, q, a, b)
    seed = b"bench k-ary round".ljust(32, b"\0")
    best_abi = best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        abi_calls()
        best_abi = min(best_abi, time.perf_counter() - t0)
        t0 = time.perf_counter()
        step = ark_bp.kary_prove_step(curve, k, A, B, q, res, a, b, seed)
        best = min(best, time.perf_counter() - t0)
    v = ark_bp.kary_verify_step(curve, k, A, B, q, res, step["pos_cross"], step["neg_cross"], seed)
    agree = bool((v["a_gens"] == step["a_gens"]).all() and (v["b_gens"] == step["b_gens"]).all() and (v["result"] == step["result"]).all())
    out["kary_round"] = {"n": nk, "k": k, "abi_calls_ms": best_abi * 1e3, "host_mirror_prove_step_ms": best * 1e3,
                         "cross_term_msms": 2 * (k - 1),
                         "note": "host vectors in, folded vectors out (PCIe both ways included).  abi_calls_ms: bpgpu_kary_cross_terms + "
                                 "bpgpu_fold_scalars_kary x 2 + bpgpu_fold_points_kary x 2 on the caller's arrays; "
                                 "host_mirror_prove_step_ms: KaryBp::prove_step of the C++ mirror, which clones, pads and concatenates "
                                 "its vectors the way the reference does (fresh allocations, first-touch page faults) and hashes the transcript",
                         "verifier_step_reaches_the_same_instance": agree}
    # k-ary scalar fold (ipa_to_bp_unroll.rs:248-252): ck = 2^20, k = 8: 32 (k + 1) B per output
    ck = 1 << 20
    d_v = ffi.scalars_to_device(synth_scalars(ck * k, 32))
    d_o = ffi.DeviceBuffer(ck * 32)
    best = 1e9
    for _ in range(5):
        ffi.fold_scalars_kary_device(curve, d_v, ck, k, y, d_o)
        best = min(best, ffi.last_op_ms())
    gbs = ck * 32 * (k + 1) / (best * 1e-3) / 1e9
    out["fold_scalars_kary"] = {"bound": "hbm", "ck": ck, "k": k, "kernel_ms": best, "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                "frac": gbs / 8000.0, "algorithmic_bytes_per_output": 32 * (k + 1),
                                "valu_frac": ck * k * MADS_PER_FIELD_MUL / (best * 1e-3) / mad_peak}
    d_v.free()
    d_o.free()
    # mat_vec_mult (com_r1cs.rs:43-57 at com_r1cs_to_ipa.rs:63-64) on an R1CS-shaped matrix: 2^20 rows of 1..5 entries
    rows = 1 << 20
    rng = np.random.default_rng(33)
    lens = rng.integers(1, 6, size=rows)
    row_ptr = np.zeros(rows + 1, np.uint64)
    row_ptr[1:] = np.cumsum(lens)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, rows, size=nnz).astype(np.uint64)
    vals, vec = synth_scalars(nnz, 34), synth_scalars(rows, 35)
    best = 1e9
    for _ in range(3):
        ffi.mat_vec_mult(curve, row_ptr, col, vals, vec)
        best = min(best, ffi.last_op_ms())
    bytes_ = nnz * (32 + 32 + 4) + rows * (4 + 32)
    out["k_spmv"] = {"bound": "hbm", "rows": rows, "entries": nnz, "kernel_ms": best, "achieved": bytes_ / (best * 1e-3) / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": bytes_ / (best * 1e-3) / 8e12,
                     "algorithmic_bytes": bytes_, "note": "68 B per entry (value, gathered vector element, index) + 36 B per row; the vector "
                                                          "(32 MiB) sits in the Infinity Cache, so the gathers are not HBM reads"}
    return out


def fold_kernels_leg(ffi, curve, log2n, mad_peak):
    """Integer roofline of the generator-fold kernels (a6, bp_iter.rs:97-104) as the prover launches them at N = 2^log2n:
    k_fold_joint4 (two rounds at once, 2 x N/4 outputs) and k_fold_monic (one round, 2 x N/2 outputs).  Group operations
    are the ones the launch executed (bpgpu_last_fold_ops: doublings = recoded length, mixed additions = non-zero digits),
    priced by the SURVEY yardstick (7 / 11 field muls x 136 mads) and by the executed count; kernel time from HIP events."""
    n = 1 << log2n
    if n < 8:
        return None
    r_mod = [0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
             0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001][curve]
    R = 1 << 256

    def ark(v):
        return np.frombuffer((v * R % r_mod).to_bytes(32, "little"), dtype=np.uint64)
    pts = synth_points(ffi, curve, 2 * n + 1, seed=77)
    a, b = synth_scalars(n, 78), synth_scalars(n, 79)
    d_a, d_b, d_q = (ffi.points_to_device(curve, pts[:n]), ffi.points_to_device(curve, pts[n:2 * n]),
                     ffi.points_to_device(curve, pts[2 * n:]))
    out = {}
    rng = np.random.default_rng(5)
    for name, pair in (("k_fold_joint4", True), ("k_fold_monic", False)):
        ffi.set_pair_fold(pair)
        best, ops = 1e9, None
        for _ in range(2):
            sess = ffi.IpaSession.from_device(curve, d_a, d_b, d_q, a, b)
            for _round in range(2 if pair else 1):
                sess.round_lr()
                x = int.from_bytes(rng.bytes(31), "little") | 1
                sess.round_fold(ark(x), ark(pow(x, -1, r_mod)))
            ms, ops = ffi.last_op_ms(), ffi.last_fold_ops()
            sess.free()
            best = min(best, ms)
        outputs, dbls, adds, kind = ops
        assert kind == (2 if pair else 1), "unexpected fold kernel"
        yard = (dbls * MULS_PER_DOUBLE + adds * MULS_PER_MIXED_ADD) * MADS_PER_FIELD_MUL
        execd = dbls * EXEC_MADS_JAC_DOUBLE + adds * EXEC_MADS_JAC_MIXED_ADD
        out[name] = {"bound": "valu", "outputs": outputs, "doublings_per_output": dbls / outputs,
                     "mixed_additions_per_output": adds / outputs, "kernel_ms": best,
                     "achieved": yard / (best * 1e-3) / 1e12, "peak": mad_peak / 1e12, "unit": "Tmad/s",
                     "frac": yard / (best * 1e-3) / mad_peak if mad_peak else None,
                     "frac_executed": execd / (best * 1e-3) / mad_peak if mad_peak else None}
    ffi.set_pair_fold(True)
    for buf in (d_a, d_b, d_q):
        buf.free()
    out["note"] = ("group operations executed by the launch (doublings = length of the recoded shared multipliers, mixed additions = "
                   "their non-zero digits + 1) x SURVEY 8(d) costs (7 / 11 field muls x 136 mads); frac_executed uses the compiled "
                   "counts of the Jacobian chain (738 / 1233 mads); the per-generator alpha point and the batched inversions (about 25 "
                   "multiplications per output) are not counted")
    return out


def ipa_prove_leg(ffi, curve, log2n, cpu_log2n=0):
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
    cpu = ipa_cpu_baseline(ffi, curve, pts, n, a, b, seed, min(cpu_log2n, log2n)) if cpu_log2n else None
    return {"n": n, "prove_ms": best, "runs_ms": ms, "rounds": len(pf["ls"]), "proof_elements": 2 * (1 + len(pf["ls"])),
            "cpu_baseline": cpu,
            "reference_published_s": ref_s, "speedup_vs_reference_published": (ref_s * 1e3 / best) if ref_s else None,
            "verify_ms": v_ms, "reference_published_verify_s": ref_v,
            "prove_ms_generators_resident": pc_ms, "verify_ms_generators_resident": vc_ms,
            "note": "host-resident inputs, upload + 2*log2(N) MSMs + folds + host Fiat-Shamir; verify = transcript replay + "
                    "one device MSM of 2N+1+2log2(N) terms; reference numbers are from unknown CPU hardware"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def ipa_cpu_baseline(ffi, curve, pts, n, a, b, seed, log2s):
    """Oracle B's Bp::prove (restatement of src/proofs/bp_iter.rs:49-128; window of src/lib.rs:119-121) on the host
    cores, on the first 2^log2s generators / witness elements of the bench inputs, next to the GPU prove of the SAME
    sample, whose proof is checked against it element by element."""
    from bulletproofs_amd import ark_bp
    from oracle import c_oracle as co
    ns = 1 << log2s
    threads = min(16, os.cpu_count() or 1)
    co.load(threads)
    A, B, q = pts[:ns], pts[n:n + ns], pts[2 * n:]
    t0 = time.perf_counter()
    ls, rs, ch, fa, fb = co.bp_prove(curve, A, B, q, a[:ns], b[:ns], seed, threads=threads)
    cpu_s = time.perf_counter() - t0
    g = min((ark_bp.bp_prove(curve, A, B, q, a[:ns], b[:ns], seed) for _ in range(2)), key=lambda p: p["prove_ms"])
    same = (len(g["ls"]) == len(ls) and all((g["ls"][k][:8] == ls[k][:8]).all() and (g["rs"][k][:8] == rs[k][:8]).all()
                                            for k in range(len(ls)))
            and (g["final_a"] == fa).all() and (g["final_b"] == fb).all())
    # Bp::prove doubles per doubling of N (the reference's own data: 0.58 s at 2^10 -> 528 s at 2^20, analysis/main/data.csv:12,32)
    full = int(np.log2(n))
    return {"value": cpu_s * 1e3, "unit": "ms", "n": ns, "cores": threads, "cpu": cpu_model(), "kind": "port",
            "sample": f"Bp::prove over the first 2^{log2s} generators / witness elements of the bench inputs",
            "gpu_prove_ms_same_sample": g["prove_ms"], "speedup_same_sample": cpu_s * 1e3 / g["prove_ms"],
            "gpu_proof_matches_cpu": bool(same),
            "extrapolated_ms_at_full_size": cpu_s * 1e3 * (1 << (full - log2s)),
            "extrapolation": f"EXTRAPOLATED, not measured: x{1 << (full - log2s)} for the {full - log2s} doublings to N = 2^{full} "
                             "(prove time doubles per doubling: analysis/main/data.csv:12-32); the reference's published 527.8 s "
                             "is from unknown hardware"}


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
    # one host thread, on a 2^16 sample (the reference's config 1 size) so that the bench stays short
    n1 = min(ns, 1 << 16)
    t1 = time.perf_counter()
    co.msm_pippenger(curve, pts[:n1], sc[:n1], threads=1)
    dt1 = time.perf_counter() - t1
    return {"value": ns / dt, "unit": "points/s", "cores": threads, "cpu": cpu_model(), "kind": "port",
            "sample": f"first 2^{log2s} pairs of the bench inputs, window c={co.load(threads).po_pippenger_window(ctypes.c_size_t(ns))}, {dt:.2f} s",
            "single_thread_points_per_s": n1 / dt1, "single_thread_sample": f"first {n1} pairs, {dt1:.2f} s",
            "gpu_matches_cpu_on_sample": bool(ok)}


if __name__ == "__main__":
    main()
