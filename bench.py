#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[1]: BLS12-381 2^20-point G1 MSM (Pippenger, c = 16)
per MI355X, with the batched-pairing rate reported beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one MSM over the rank's resident (point, scalar) shard: kernels, D2H of the window sums,
host Horner tail, and for N > 1 the all-gather of the per-rank partial sums over RCCL plus the local
EC addition.  The timed steps run one at a time (so the per-kernel HIP-event times behind `roofline` are
unshared); the throughput of the same steps issued two-deep through mlhip_msm_launch / mlhip_msm_finish
on two streams is reported as an extra, not as `value`.  Points and scalars are in HBM before the timed region starts.  Weak scaling: every rank
holds 2^20 pairs, `value` = N * 2^20 * K / (max over ranks of the K-step time).

Inputs are synthetic and produced by the product itself: P_i = [k_i]G from the batched scalar-mul
kernel, k_i and the MSM scalars from torch's generator (seeded per rank).  The oracle (oracle/cref) is
used ONLY for the `cpu_baseline` leg: the same workload timed on the host cores of rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mathlib_amd import _lib, dist as mdist  # noqa: E402
from mathlib_amd.driver import Curve  # noqa: E402

CURVE = _lib.CURVE_BLS12_381
N_PER_GPU = 1 << 20
WINDOW_C = 16
N_PAIRINGS = 1 << 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MSM_BYTES_PER_UNIT = 128  # 96 B affine point + 32 B scalar
PAIRING_BYTES_PER_UNIT = 864  # 96 + 192 in, 576 out
INT_MAC_PEAK = 3.19e13  # measured v_mad_u64_u32 lane-ops/s, profiles/r01_ubench_int.txt
MACS_PER_FP_MUL = 288  # 2 * 12^2 32x32->64 multiply-accumulates per 384-bit Montgomery product


def rand_scalars(n: int, gen: torch.Generator, device) -> torch.Tensor:
    """n x 32 bytes: uniform 256-bit integers (the device reduces them mod r, as fr.SetBigInt does)"""
    lo = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=device)
    return lo.view(torch.uint8).reshape(n, 32).contiguous()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pairing", action="store_true")
    ap.add_argument("--host-paths", action="store_true", help="also time the PCIe-inclusive entry points (SURVEY 8d protocol b and c: scalars from the host with resident bases, everything from the host) and report them in extra")
    ap.add_argument("--skewed", action="store_true", help="also time the skewed-scalar MSM (all scalars < 2^32, 1 %% duplicates) and report it in extra; off by default so that the rocprofv3 kernel averages of the plain command cover the timed steps only")
    ap.add_argument("--pipelined-extra", action="store_true", help="also time the steps two-deep on two streams (extra only)")
    ap.add_argument("--sequential", action="store_true", help="one MSM in flight at a time (default: the K steps are issued two-deep through launch/finish on two plans and streams)")
    ap.add_argument("--pipelined", action="store_true", help="accepted for compatibility: two in flight is the default")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # MLHIP_BENCH_REHEARSAL=1: several ranks share GPU 0 and exchange over gloo -- only for rehearsing the
    # N > 1 code path on a one-GPU box (RCCL needs one device per rank); never a measurement.
    rehearsal = os.environ.get("MLHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.load()
    _lib.check(lib.mlhip_set_device(local_rank))
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic inputs, resident in HBM
    fpb, g1b, g2b, gtb = _lib.sizes(CURVE)
    curve = Curve(CURVE)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6D6C686970 + rank)
    n = N_PER_GPU
    k = rand_scalars(n, gen, dev)
    base = torch.frombuffer(bytearray(curve.GenG1().raw), dtype=torch.uint8).to(dev)
    points = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(CURVE, _lib.GROUP_G1, base.data_ptr(), 0, k.data_ptr(), 0, n, points.data_ptr(), stream))
    scalars = rand_scalars(n, gen, dev)
    torch.cuda.synchronize()

    # two plans on two streams (the second one only for the pipelined extra)
    plans = [_lib.MsmPlan(CURVE, _lib.GROUP_G1, n, WINDOW_C) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for pl in plans:
        pl.set_profiling(True)
    phase = {}

    def finalize(j: int, record: bool) -> bytes:
        part = plans[j].finish()
        if record:
            for kname, v in plans[j].timings().items():
                phase[kname] = phase.get(kname, 0.0) + v
        return mdist.combine_partials(CURVE, _lib.GROUP_G1, part, dev)

    def run_steps(k: int, record: bool) -> bytes:
        pending, res = None, None
        for i in range(k):
            j = i & 1
            plans[j].launch(points.data_ptr(), scalars.data_ptr(), n, False, streams[j].cuda_stream)
            if pending is not None:
                res = finalize(pending, record)
            pending = j
        if pending is not None:
            res = finalize(pending, record)
        return res

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    def run_sequential(k: int, record: bool) -> bytes:
        res = None
        for _ in range(k):
            plans[0].launch(points.data_ptr(), scalars.data_ptr(), n, False, streams[0].cuda_stream)
            res = finalize(0, record)
        return res

    # ---- timed region: the K steps, two in flight (step i+1 is launched before step i is finished: its sort
    # kernels run under the host tail and the latency-bound end of the previous reduction); --sequential keeps one in
    # flight.  Per-kernel HIP events stay on the stream each kernel is launched on.
    timed = run_sequential if args.sequential else run_steps
    timed(args.warmup, False)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = timed(args.steps, True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- extra (not the headline): the same K steps issued two-deep through launch/finish on two streams
    pipelined = None
    if args.pipelined_extra and args.sequential:
        saved = dict(phase)
        run_steps(2, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.steps, False)
        torch.cuda.synchronize()
        pipelined = n * args.steps / (time.perf_counter() - t1)
        phase.clear()
        phase.update(saved)
    steps = max(args.steps, 1)
    phase = {kname: v / steps for kname, v in phase.items()}
    value = world * n * args.steps / elapsed

    # ---- roofline of the dominant kernel (the bucket accumulation): algorithmic bytes / its HIP-event duration
    acc_ms = phase.get("accumulate", 0.0)
    acc32 = os.environ.get("MLHIP_ACC32", "") == "1"
    acc_kernel = "k_accumulate<FpField<Bls381>>" if acc32 else "k_accumulate28<Bls381>"
    # multiplier instructions per mixed addition: 10 x 288 v_mad_u64_u32 (+ as many v_addc) in the boundary form;
    # 8 x 196 + 2 x 105 product and 9 x 210 reduction v_mad_i64_i32 + 9 x 14 v_mul_lo_u32 in the carry-free form
    v_mad_per_madd = 10 * MACS_PER_FP_MUL if acc32 else (8 * 196 + 2 * 105 + 9 * 210 + 9 * 14)
    achieved = (MSM_BYTES_PER_UNIT * n) / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    # HBM traffic of the dominant kernel from the committed PMC passes (profiles/r01_pmc_traffic.json): the
    # counters cannot be read from inside this process, so this is the per-launch figure of the same workload
    traffic, traffic_note = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pm = json.load(f)
        kk = next(v for k_, v in pm["kernels"].items() if k_.startswith(acc_kernel.split("<")[0] + "<") and "Fp2" not in k_)
        traffic = (kk["FETCH_SIZE"] + kk["WRITE_SIZE"]) * 1024.0
        traffic_note = "bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB from profiles/r01_pmc_traffic.json, uncorrected (gather pattern, Infinity-Cache hits included): every point row is re-read once per window (16 x 112 B x n), plus the 96 MiB bucket array written once"
    except Exception:
        pass
    roofline = {
        "bound": "hbm",
        "kernel": acc_kernel,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_note": traffic_note,
        "avg_kernel_ms": acc_ms,
        "phase_ms": phase,
        # the path is integer-ALU bound, not HBM bound: W = 16 mixed additions per scalar, each V_MAD_PER_MADD
        # multiplier instructions => fraction of the measured v_mad issue peak (profiles/r01_ubench_int.txt).  The
        # kernel is power limited: its effective clock is ~1.9-2.2 GHz, not 2.4 (profiles/r01_pmc_clocks.txt).
        "int_alu": {
            "form": "carry-free 28-bit limbs (fp28.h)" if not acc32 else "saturated 32-bit limbs (fp.h)",
            "fp_mul_per_s": (n * 16 * 10) / (acc_ms * 1e-3) if acc_ms > 0 else 0.0,
            "v_mad_per_mixed_add": v_mad_per_madd,
            "v_mad_frac_of_measured_peak": ((n * 16 * v_mad_per_madd) / (acc_ms * 1e-3)) / INT_MAC_PEAK if acc_ms > 0 else 0.0,
        },
    }

    extra = {}
    if pipelined is not None:
        extra["msm_pipelined_depth2_scalar_muls_per_s_per_gpu"] = pipelined
    # ---- skewed scalars (BASELINE configs[1], second distribution: all scalars < 2^32 and 1 % duplicated pairs):
    # the carry bucket of the third window then holds half of the entries; reported beside the headline, never as it
    if args.skewed:
        sk = scalars.clone().view(torch.int64).reshape(n, 4)
        sk[:, 1:] = 0
        sk[:, 0] &= 0xFFFFFFFF
        sk[::100] = sk[0]
        sk = sk.view(torch.uint8).reshape(n, 32).contiguous()
        pts_sk = points.clone().reshape(n, g1b)
        pts_sk[::100] = pts_sk[0]
        pts_sk = pts_sk.reshape(-1).contiguous()
        best_sk = None
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            plans[0].launch(pts_sk.data_ptr(), sk.data_ptr(), n, False, streams[0].cuda_stream)
            plans[0].finish()
            dt = (time.perf_counter() - t1) * 1e3
            best_sk = dt if best_sk is None or dt < best_sk else best_sk
        extra["skewed_scalars_below_2^32_1pct_duplicates_ms_per_msm"] = best_sk
    # ---- PCIe-inclusive protocols (never the headline): (b) resident bases, scalars from host memory per call;
    # (c) points and scalars from host memory per call -- the reference-shaped MultiScalarMul on host slices
    if args.host_paths and rank == 0:
        import ctypes

        hp = points.cpu().numpy().tobytes()
        hs = scalars.cpu().numpy().tobytes()
        out_h = ctypes.create_string_buffer(g1b)
        handle = ctypes.c_void_p()
        _lib.check(lib.mlhip_bases_create(CURVE, _lib.GROUP_G1, hp, n, 16, ctypes.byref(handle)))
        tb, tc = [], []
        for _ in range(5):
            t1 = time.perf_counter()
            _lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out_h))
            tb.append((time.perf_counter() - t1) * 1e3)
        same_b = out_h.raw == res
        _lib.check(lib.mlhip_bases_destroy(handle))
        for _ in range(5):
            t1 = time.perf_counter()
            _lib.check(lib.mlhip_msm_g1(CURVE, hp, hs, 0, n, 16, out_h))
            tc.append((time.perf_counter() - t1) * 1e3)
        extra["pcie_inclusive"] = {
            "scalars_from_host_resident_bases_ms": min(tb[1:]),
            "points_and_scalars_from_host_ms": min(tc[1:]),
            "match_resident_result": bool(same_b and out_h.raw == res),
        }
    # ---- batched pairing (BASELINE configs[2]): 65 536 x (Miller loop + final exponentiation)
    if not args.no_pairing:
        npair = N_PAIRINGS
        g2base = torch.frombuffer(bytearray(curve.GenG2().raw), dtype=torch.uint8).to(dev)
        q = torch.empty(npair * g2b, dtype=torch.uint8, device=dev)
        _lib.check(lib.mlhip_scalar_mul_device(CURVE, _lib.GROUP_G2, g2base.data_ptr(), 0, scalars.data_ptr(), 0, npair, q.data_ptr(), stream))
        gt = torch.empty(npair * gtb, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(3):
            ev0.record()
            _lib.check(lib.mlhip_pairing_batch_device(CURVE, points.data_ptr(), q.data_ptr(), npair, gt.data_ptr(), stream))
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1)
            best = ms if best is None or ms < best else best
        extra["pairings_per_s_per_gpu"] = npair / (best * 1e-3)
        extra["pairing_batch"] = npair
        extra["pairing_kernel_ms"] = best
        extra["pairing_roofline"] = {
            "bound": "hbm",
            "achieved": PAIRING_BYTES_PER_UNIT * npair / (best * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": PAIRING_BYTES_PER_UNIT * npair / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }

    # ---- CPU baseline: the oracle's C restatement on the same workload, host cores of this box
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cref  # checker / baseline only

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(cores, 64))
        hp = points.cpu().numpy()
        hs = scalars.cpu().numpy()
        t1 = time.perf_counter()
        ref = cref.msm(CURVE, 1, hp, hs, n, False, WINDOW_C, threads)
        dt = time.perf_counter() - t1
        cpu_baseline = {
            "value": n / dt,
            "unit": "scalar-muls/s",
            "cores": threads,
            "kind": "port",
            "sample": "the full 2^20-point workload (same points and scalars), oracle/cref Pippenger c=16, %d pthreads, %.2f s" % (threads, dt),
            "matches_gpu_result": bool(ref == res),
        }
        if not args.no_pairing:
            ns = 2048
            t1 = time.perf_counter()
            cpu_gt = cref.pairing_batch(CURVE, hp[: ns * g1b], q[: ns * g2b].cpu().numpy(), ns, threads)
            dtp = time.perf_counter() - t1
            cpu_baseline["pairings_per_s"] = ns / dtp
            cpu_baseline["pairings_match_gpu_result"] = bool(cpu_gt == bytes(gt[: ns * gtb].cpu().numpy().tobytes()))
            cpu_baseline["pairing_sample"] = "%d of the 65 536 pairs, %d pthreads, %.2f s" % (ns, threads, dtp)

    if rank == 0:
        line = {
            "metric": "G1 scalar-muls/sec (BLS12-381 2^20-point MSM per GPU), pairings/sec reported in extra",
            "value": value,
            "unit": "scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {
                "workload": "BLS12-381 2^20-point G1 MSM per GPU, Pippenger c=16 (BASELINE configs[1]); inputs resident in HBM",
                "curve": "BLS12-381",
                "points_per_gpu": n,
                "window_c": WINDOW_C,
                "parallelism": "pairs sharded contiguously across ranks; one all-gather of 96-byte partial sums over RCCL + local EC add",
                "msms_in_flight": 1 if args.sequential else 2,
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
