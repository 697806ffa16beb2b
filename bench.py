#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric (G1 scalar-muls/sec + pairings/sec) on BASELINE.json's configs.

    python bench.py [--config 2|3|4|5] --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

--config (SURVEY.md 8d numbering; default 2, the configuration BASELINE.json's metric is quoted on):
  2  BLS12-381 2^20-point G1 MSM per GPU, Pippenger c = 16          (BASELINE configs[1], weak scaling)
  3  BLS12-381 batch of 65 536 optimal-ate pairings + FExp per GPU  (BASELINE configs[2], weak scaling)
  4  BLS12-381 2^24-point G1 + G2 MSM (shared scalars), the pairs sharded contiguously over the N ranks
                                                                    (BASELINE configs[3], strong scaling)
  5  BLS12-377 2^22-point G1 MSM sharded over the N ranks           (BASELINE configs[4], strong scaling)
At N = 1 configs 4 and 5 run whole on one GPU.

One "step" = one pass of the hot path over the rank's resident shard: the MSM kernels, D2H of the window sums, the
host Horner tail and, for N > 1, ONE all-gather of the per-rank partial sums over RCCL plus the local EC addition
(config 4: the G1 and the G2 MSM of the step are both in flight, their partials share the all-gather).  Timed
protocol: W warm-up steps, then exactly K steps, ONE step in flight at a time, bracketed by barrier +
synchronize; `value` = units of all ranks / max over ranks of the K-step time -- protocol (a), inputs resident in HBM
when the timed region starts.  `extra` carries median / min of the K per-step times and, measured after the timed
region without any flag: protocol (b) (resident bases, scalars from host memory per call: mlhip_bases_msm), protocol
(c) (points and scalars from host memory: mlhip_msm_g1, what the reference's MultiScalarMul maps to), the
skewed-scalar MSM (all scalars < 2^32, 1 % duplicated pairs), the same steps issued two-deep on two plans and streams,
and (config 2) the 65 536-pairing batch.  The PCIe-inclusive rates are never `value`.  --kernels-only skips every
extra and the CPU baseline, so that a rocprofv3 run of it averages exactly the timed steps' kernels.

`roofline` is for the dominant kernel (the bucket accumulation; the fused pairing kernel for config 3): algorithmic
bytes of one launch / its average duration, from HIP events recorded on the stream the kernel is launched on, in the
timed steps themselves (config 4: in a separate pass with one MSM in flight, so the times are unshared).

Inputs are synthetic and produced by the product itself: P_i = [k_i]G from the batched scalar-mul kernel, k_i and the
MSM scalars from torch's generator (seeded per config and global pair index, so a shard is the same data whatever N is).
The oracle (oracle/cref) is used ONLY for the `cpu_baseline` leg, after the timed region, on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mathlib_amd import _lib, dist as mdist  # noqa: E402
from mathlib_amd.driver import Curve  # noqa: E402

G1, G2 = _lib.GROUP_G1, _lib.GROUP_G2
WINDOW_C = 16
N_PAIRINGS = 1 << 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0  # the same guide's measured copy rate (BASELINE.md section 4 asks for both denominators)
INT_MAC_PEAK = 3.19e13  # measured v_mad_u64_u32 lane-ops/s, profiles/r01_ubench_int.txt
PAIRING_BYTES_PER_UNIT = 864  # 96 + 192 in, 576 out

CONFIGS = {
    2: dict(curve=_lib.CURVE_BLS12_381, curve_name="BLS12-381", log_n=20, groups=(G1,), scaling="weak",
            workload="BLS12-381 2^20-point G1 MSM per GPU, Pippenger c=16 (BASELINE configs[1]); inputs resident in HBM"),
    3: dict(curve=_lib.CURVE_BLS12_381, curve_name="BLS12-381", log_n=16, groups=(), scaling="weak",
            workload="BLS12-381 batch of 65 536 optimal-ate pairings + FExp per GPU (BASELINE configs[2]); inputs resident in HBM"),
    4: dict(curve=_lib.CURVE_BLS12_381, curve_name="BLS12-381", log_n=24, groups=(G1, G2), scaling="strong",
            workload="BLS12-381 2^24-point G1 + G2 MSM (shared scalars) sharded over the ranks, Pippenger c=16 (BASELINE configs[3]); inputs resident in HBM"),
    5: dict(curve=_lib.CURVE_BLS12_377, curve_name="BLS12-377", log_n=22, groups=(G1,), scaling="strong", window_c=0,
            workload="BLS12-377 2^22-point G1 MSM sharded over the ranks (BASELINE configs[4], which names no window: the library "
                     "picks it per shard -- c=17, 15 windows, from 2^22 pairs on, c=16 below); inputs resident in HBM"),
}
# algorithmic bytes per scalar-mul: affine point + 32-byte scalar
MSM_BYTES = {(0, G1): 96, (1, G1): 128, (2, G1): 128, (0, G2): 160, (1, G2): 224, (2, G2): 224}
# gnark-crypto's BLS12-377 G2 generator is not in the reference tree; any r-torsion point serves as the base of the
# synthetic points: the golden file's generator (tests/golden/bls12_377.json, produced by oracle/pyref.py) is read at
# run time only for config runs on that curve's G2, which no BASELINE config asks for.


def rand_scalars(n: int, gen: torch.Generator, device) -> torch.Tensor:
    """n x 32 bytes: uniform 256-bit integers (the device reduces them mod r, as fr.SetBigInt does)"""
    lo = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=device)
    return lo.view(torch.uint8).reshape(n, 32).contiguous()


def seeded_scalars(n_total: int, lo: int, hi: int, seed: int, dev) -> torch.Tensor:
    """rows [lo, hi) of a stream of n_total rows, generated in blocks of 2^18 rows each seeded by its block index
    (so a shard holds the same rows whatever the number of ranks)"""
    blk = 1 << 18
    out = []
    gen = torch.Generator(device=dev)
    for b in range(lo // blk, (hi + blk - 1) // blk):
        gen.manual_seed(seed * 1000003 + b)
        rows = rand_scalars(min(blk, n_total - b * blk), gen, dev)
        a, z = max(lo, b * blk) - b * blk, min(hi, (b + 1) * blk) - b * blk
        out.append(rows[a:z])
    return torch.cat(out).contiguous() if out else torch.empty((0, 32), dtype=torch.uint8, device=dev)


def self_launch(n_ranks: int) -> int:
    """python -m torch.distributed.run --standalone --nproc-per-node N bench.py <the same arguments>, as a child"""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port (several benches may share a node)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--log-n", type=int, default=0, help="override the config's size (rehearsals only: the line is marked reduced and is not a measurement of the config)")
    ap.add_argument("--kernels-only", action="store_true", help="timed steps only: no extras, no pairing batch, no CPU baseline (rocprofv3 passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pairing", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip protocol (b)/(c), the skewed run and the two-deep run")
    ap.add_argument("--separate-sorts", action="store_true", help="config 4: launch the G1 and the G2 MSM of a step as two independent MSMs (each sorts the shared scalars itself) instead of mlhip_msm_launch_shared")
    ap.add_argument("--pipelined", action="store_true", help="issue the timed steps two-deep (launch of step i+1 before finish of step i); default: one step in flight")
    # accepted for compatibility with round-1 command lines
    ap.add_argument("--sequential", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--host-paths", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--skewed", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pipelined-extra", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.kernels_only:
        args.no_cpu_baseline = args.no_pairing = args.no_extras = True

    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (never an exec, and before
        # anything in this process has touched the GPU), relay rank 0's JSON line and leave with the child's exit code
        sys.exit(self_launch(args.gpus))
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

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        import torch.distributed as dist

        t = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    CURVE = cfg["curve"]
    fpb, g1b, g2b, gtb = _lib.sizes(CURVE)
    curve = Curve(CURVE)
    log_n = args.log_n or cfg["log_n"]
    reduced = bool(args.log_n) and args.log_n != cfg["log_n"]
    n_cfg = 1 << log_n
    if cfg["scaling"] == "weak":
        n_total, lo, hi = n_cfg * world, n_cfg * rank, n_cfg * (rank + 1)
    else:
        n_total = n_cfg
        lo, hi = mdist.shard_bounds(n_total, rank, world)
    n = hi - lo
    seed = 0x6D6C68 + args.config

    # ---- synthetic inputs of this rank's shard, resident in HBM: P_i = [k_i]G (G1 and, config 3/4, Q_i = [k'_i]G2)
    def gen_points(group: int, k: torch.Tensor) -> torch.Tensor:
        if group == G1:
            raw = curve.GenG1().raw
        elif CURVE == _lib.CURVE_BLS12_377:
            with open(os.path.join(ROOT, "tests", "golden", "bls12_377.json")) as f:
                raw = bytes.fromhex(json.load(f)["g2_gen"])
        else:
            raw = curve.GenG2().raw
        base = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        out = torch.empty(k.shape[0] * (g1b if group == G1 else g2b), dtype=torch.uint8, device=dev)
        _lib.check(lib.mlhip_scalar_mul_device(CURVE, group, base.data_ptr(), 0, k.data_ptr(), 0, k.shape[0], out.data_ptr(), stream))
        torch.cuda.synchronize()
        return out

    groups = cfg["groups"]
    points = {}
    if G1 in groups or args.config in (2, 3):
        points[G1] = gen_points(G1, seeded_scalars(n_total, lo, hi, seed * 3 + 1, dev))
    if G2 in groups:
        points[G2] = gen_points(G2, seeded_scalars(n_total, lo, hi, seed * 3 + 2, dev))
    scalars = seeded_scalars(n_total, lo, hi, seed * 3, dev)
    torch.cuda.synchronize()

    extra = {}
    phase = {}
    step_ms = []
    exchange_ms = []  # per timed step: the all-gather of the partial sums + the local EC additions (N > 1 only)
    res = {}

    # =============================================================== config 3: the pairing batch is the step
    def pairing_setup():
        npair = min(N_PAIRINGS if not reduced else n_cfg, n) if args.config == 3 else min(N_PAIRINGS, n)
        q = gen_points(G2, scalars[:npair].contiguous())
        gt = torch.empty(npair * gtb, dtype=torch.uint8, device=dev)
        return npair, q, gt

    if args.config == 3:
        npair, q, gt = pairing_setup()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

        def pairing_step(i, record):
            if record:
                ev[i][0].record()
            _lib.check(lib.mlhip_pairing_batch_device(CURVE, points[G1].data_ptr(), q.data_ptr(), npair, gt.data_ptr(), stream))
            if record:
                ev[i][1].record()
            torch.cuda.synchronize()

        for i in range(args.warmup):
            pairing_step(i, False)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            t1 = time.perf_counter()
            pairing_step(i, True)
            step_ms.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        kernel_ms = statistics.mean(a.elapsed_time(b) for a, b in ev) if ev else 0.0
        units_per_step = npair * world
        achieved = PAIRING_BYTES_PER_UNIT * npair / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
        roofline = {
            "bound": "hbm", "kernel": "k_pairing_lp28<Bls381,2,1> (fused Miller loop + final exponentiation, one pairing per lane pair, carry-free 28-bit limbs)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_GBS,
            "traffic": None, "avg_kernel_ms": kernel_ms,
            "note": "864 algorithmic bytes per pairing; the kernel is integer-issue bound (DESIGN.md section 4)",
        }
        roofline["traffic"], roofline["traffic_note"] = _pmc(3, "k_pairing_lp28<Bls381, 2")
        # the honest bound is integer issue: 64-bit integer VALU instructions per pairing (v_mad_i64_i32 and the 64-bit
        # column shifts; the kernel is straight-line, the count does not depend on the data) from the PMC pass
        # profiles/r03_pmc_issue.txt: SQ_INSTS_VALU_INT64 5.9215e9 wave-instructions per 65 536 pairings, 69 % of its VALU
        # instructions -- against the same measured issue peak as the MSM kernels
        int64_per_pairing = 5.9215e9 * 64 / 65536
        roofline["int_alu"] = {
            "form": "carry-free 28-bit limbs on lane pairs (fp2_lanes28.h)",
            "int64_valu_per_pairing": int64_per_pairing, "share_of_valu_instructions": 5.9215 / 8.5283,
            "frac_of_measured_peak": (int64_per_pairing * npair / (kernel_ms * 1e-3)) / INT_MAC_PEAK if kernel_ms else 0.0,
            "source": "profiles/r03_pmc_issue.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 ...)",
        }
        unit, metric = "pairings/s", "pairings/sec (BLS12-381, Miller loop + final exponentiation, batch of 65 536 per GPU, inputs resident in HBM)"
        res_check = None
    else:
        # =========================================================== MSM configs
        # one plan + stream per (slot, group); slot 1 only for the two-deep runs
        nslots = 2
        window_c = cfg.get("window_c", WINDOW_C)  # 0: the library's pick for this shard's size
        plans = {(s, g): _lib.MsmPlan(CURVE, g, n, window_c) for s in range(nslots) for g in groups}
        window_c, n_windows = plans[(0, groups[0])].window()
        streams = {(s, g): torch.cuda.Stream(device=dev) for s in range(nslots) for g in groups}
        for pl in plans.values():
            pl.set_profiling(True)

        # config 4: the G1 and the G2 MSM of a step share their scalars -- one sort per tile for both (mlhip_msm_launch_shared)
        shared = len(groups) > 1 and not args.separate_sorts

        def launch(slot):
            if shared:
                plans[(slot, G1)].launch_shared(plans[(slot, G2)], points[G1].data_ptr(), points[G2].data_ptr(), scalars.data_ptr(),
                                                n, False, streams[(slot, G1)].cuda_stream)
                return
            for g in groups:
                plans[(slot, g)].launch(points[g].data_ptr(), scalars.data_ptr(), n, False, streams[(slot, g)].cuda_stream)

        def finish(slot, record):
            parts = []
            for g in groups:
                parts.append((g, plans[(slot, g)].finish()))
                if record:
                    for kname, v in plans[(slot, g)].timings().items():
                        if kname in ("window_c", "digits_per_scalar", "edwards", "tables"):
                            continue  # (not times: the plan's geometry and path flags)
                        phase[(g, kname)] = phase.get((g, kname), 0.0) + v
            tx = time.perf_counter()
            totals = mdist.combine_many(CURVE, parts, dev)  # one all-gather (RCCL) + local EC adds; identity at N = 1
            if record:
                exchange_ms.append((time.perf_counter() - tx) * 1e3)
            return dict(zip(groups, totals))

        def run_sequential(k, record):
            out = None
            for _ in range(k):
                t1 = time.perf_counter()
                launch(0)
                out = finish(0, record)
                if record:
                    step_ms.append((time.perf_counter() - t1) * 1e3)
            return out

        def run_two_deep(k, record):
            pending, out = None, None
            t1 = time.perf_counter()
            for i in range(k):
                launch(i & 1)
                if pending is not None:
                    out = finish(pending, record)
                    if record:
                        t2 = time.perf_counter()
                        step_ms.append((t2 - t1) * 1e3)
                        t1 = t2
                pending = i & 1
            if pending is not None:
                out = finish(pending, record)
                if record:
                    step_ms.append((time.perf_counter() - t1) * 1e3)
            return out

        timed = run_two_deep if args.pipelined else run_sequential
        timed(args.warmup, False)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = timed(args.steps, True)
        torch.cuda.synchronize()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        steps_n = max(args.steps, 1)
        # per-kernel times: unshared only when one MSM is in flight -- config 4 (G1 and G2 of a step overlap) and
        # --pipelined take them from a separate pass, one MSM at a time
        if (len(groups) > 1 and not shared) or args.pipelined:
            phase.clear()
            reps = 3
            for _ in range(reps):
                if shared:
                    launch(0)
                for g in groups:
                    if not shared:
                        plans[(0, g)].launch(points[g].data_ptr(), scalars.data_ptr(), n, False, streams[(0, g)].cuda_stream)
                    plans[(0, g)].finish()
                    for kname, v in plans[(0, g)].timings().items():
                        phase[(g, kname)] = phase.get((g, kname), 0.0) + v
            phase_avg = {k_: v / reps for k_, v in phase.items()}
        else:
            phase_avg = {k_: v / steps_n for k_, v in phase.items()}
        units_per_step = n_total
        dom = G2 if G2 in groups else G1  # the dominant kernel of the step
        acc_ms = phase_avg.get((dom, "accumulate"), 0.0)
        acc32 = os.environ.get("MLHIP_ACC32", "") == "1"
        cname = {0: "Bn254", 1: "Bls381", 2: "Bls377"}[CURVE]
        red32 = os.environ.get("MLHIP_REDUCE32", "") == "1"  # the boundary-form reduction: the accumulation then converts its buckets itself
        if dom == G1:
            # (the carry-free reduction reads the accumulators as the kernel leaves them: the segment form of the kernel
            # runs even for one pass)
            acc_kernel = ("k_accumulate<FpField<%s>>" if acc32 else ("k_accumulate28<%s>" if red32 else "k_accumulate28_seg<%s>")) % cname
        else:
            acc_kernel = ("k_accumulate28_lp<%s>" if (CURVE == 1 and not acc32) else "k_accumulate_lp<%s>") % cname
        # from 2^22 (G1) / 2^23 (G2) points on the library accumulates tile by tile (msm_plan.h: resident_tiles): the
        # dominant kernel is then launched `tiles` times per MSM (its _seg form), each launch over n / tiles units
        tiles = max(1, int(round(phase_avg.get((dom, "tiles"), 1.0))))
        if tiles > 1 and not acc32 and "_seg<" not in acc_kernel:
            acc_kernel = acc_kernel.replace("<", "_seg<", 1)
        bytes_unit = MSM_BYTES[(CURVE, dom)]
        achieved = bytes_unit * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": acc_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_GBS,
            "traffic": None, "avg_kernel_ms": acc_ms / tiles,
            "algorithmic_bytes_per_launch": bytes_unit * n // tiles, "launches_per_msm": tiles,
            "phase_ms": {("g1_" if g == G1 else "g2_") + k_: v for (g, k_), v in sorted(phase_avg.items()) if k_ not in ("tiles", "edwards")},
        }
        roofline["traffic"], roofline["traffic_note"] = _pmc(args.config, acc_kernel.split("<")[0] + "<" + cname)
        if reduced:
            roofline["traffic"], roofline["traffic_note"] = None, "reduced size: the PMC passes are of the full config"
        if dom == G1 and not acc32 and fpb == 48:
            # the path is integer-ALU bound, not HBM bound: W mixed additions per scalar (16 at c = 16), each 8 x 196 + 2 x 105
            # product and 9 x 210 reduction v_mad_i64_i32 + 9 x 14 v_mul_lo_u32 in the carry-free form => fraction of
            # the measured v_mad issue peak (profiles/r01_ubench_int.txt); the kernel is power limited (~1.9 GHz)
            v_mad = 8 * 196 + 2 * 105 + 9 * 210 + 9 * 14
            roofline["int_alu"] = {
                "form": "carry-free 28-bit limbs (fp28.h)",
                "fp_mul_per_s": (n * n_windows * 10) / (acc_ms * 1e-3) if acc_ms > 0 else 0.0,
                "v_mad_per_mixed_add": v_mad, "mixed_adds_per_scalar": n_windows,
                "v_mad_frac_of_measured_peak": ((n * n_windows * v_mad) / (acc_ms * 1e-3)) / INT_MAC_PEAK if acc_ms > 0 else 0.0,
            }
        if dom == G2 and CURVE == 1 and not acc32:
            # G2 on lane pairs, carry-free form (ec28_lp.h: xyzz28_lp_madd): per lane 8 dual products (2 x 196 product +
            # 210 reduction v_mad_i64_i32 + 14 v_mul_lo_u32 each) and 2 single products (196 + 210 + 14), two lanes per addition
            v_mad = 2 * (8 * (2 * 196 + 210 + 14) + 2 * (196 + 210 + 14))
            roofline["int_alu"] = {
                "form": "carry-free 28-bit limbs on lane pairs (ec28_lp.h)",
                "fp2_mul_per_s": (n * n_windows * 10) / (acc_ms * 1e-3) if acc_ms > 0 else 0.0,
                "v_mad_per_mixed_add": v_mad, "mixed_adds_per_scalar": n_windows,
                "v_mad_frac_of_measured_peak": ((n * n_windows * v_mad) / (acc_ms * 1e-3)) / INT_MAC_PEAK if acc_ms > 0 else 0.0,
            }
        if len(groups) > 1:
            g1_ms = phase_avg.get((G1, "accumulate"), 0.0)
            g1_tiles = max(1, int(round(phase_avg.get((G1, "tiles"), 1.0))))
            roofline["g1_kernel"] = {"kernel": ("k_accumulate28<%s>" if (red32 and g1_tiles == 1) else "k_accumulate28_seg<%s>") % cname,
                                     "avg_kernel_ms": g1_ms / g1_tiles, "launches_per_msm": g1_tiles,
                                     "achieved": MSM_BYTES[(CURVE, G1)] * n / (g1_ms * 1e-3) / 1e9 if g1_ms else 0.0}
        unit = "scalar-muls/s"
        if len(groups) > 1:
            metric = "(point, scalar) pairs/sec through the G1 MSM and the G2 MSM of a step (BLS12-381 2^24 pairs, shared scalars): each pair is one G1 and one G2 scalar-mul"
        else:
            metric = "G1 scalar-muls/sec (%s 2^%d-point MSM%s, inputs resident in HBM: SURVEY 8d protocol a)" % (
                cfg["curve_name"], log_n, " per GPU" if cfg["scaling"] == "weak" else ", sharded over the ranks")
        if args.config == 2:
            metric += "; protocol b (scalars from host memory) in extra.headline_protocol_b, pairings/sec in extra"
        res_check = res

        # ---- extras, after the timed region (never the headline) -------------------------------------------------
        if not args.no_extras:
            # the same steps two-deep through launch / finish on two plans and streams (a prover issues MSMs back to back)
            other = run_sequential if args.pipelined else run_two_deep
            saved_steps, saved_phase = list(step_ms), dict(phase)
            other(2, False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            other(args.steps, False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            step_ms[:] = saved_steps
            phase.clear()
            phase.update(saved_phase)
            extra["one_step_in_flight_units_per_s_per_gpu" if args.pipelined else "two_steps_in_flight_units_per_s_per_gpu"] = n * args.steps / dt
            # skewed scalars (SURVEY 8d, config 2's second distribution): all scalars < 2^32 and 1 % duplicated pairs
            g0 = groups[0]
            ptsz = g1b if g0 == G1 else g2b
            sk = scalars.clone().view(torch.int64).reshape(n, 4)
            sk[:, 1:] = 0
            sk[:, 0] &= 0xFFFFFFFF
            sk[::100] = sk[0]
            sk = sk.view(torch.uint8).reshape(n, 32).contiguous()
            pts_sk = points[g0].clone().reshape(n, ptsz)
            pts_sk[::100] = pts_sk[0]
            pts_sk = pts_sk.reshape(-1).contiguous()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                plans[(0, g0)].launch(pts_sk.data_ptr(), sk.data_ptr(), n, False, streams[(0, g0)].cuda_stream)
                plans[(0, g0)].finish()
                ts.append((time.perf_counter() - t1) * 1e3)
            extra["skewed_scalars_below_2^32_1pct_duplicates"] = {"ms_per_msm_median": statistics.median(ts[1:]), "ms_per_msm_min": min(ts[1:]), "group": "G1" if g0 == G1 else "G2"}
            del sk, pts_sk
            # BLS12-377 G1 (config 5): the same steps with the caller's SRS promise (mlhip_msm_plan_assume_srs: the points
            # are fixed at their address and lie in the prime-order subgroup -- true of this shard, P_i = [k_i]G): bucket
            # sums in twisted Edwards coordinates, the converted copy of the points kept between launches.  Never `value`:
            # the reference's MultiScalarMul takes any curve points in fresh slices, and `value` is measured that way.
            if CURVE == _lib.CURVE_BLS12_377 and tuple(groups) == (G1,):
                pl = plans[(0, G1)]
                pl.assume_srs(True)
                ts, ed_ph = [], {}
                for i in range(args.steps + 2):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    pl.launch(points[G1].data_ptr(), scalars.data_ptr(), n, False, streams[(0, G1)].cuda_stream)
                    out_srs = pl.finish()
                    ts.append((time.perf_counter() - t1) * 1e3)
                    ed_ph = pl.timings()
                pl.assume_srs(False)
                acc_ed = ed_ph.get("accumulate", 0.0)
                extra["fixed_srs_promise"] = {
                    "ms_per_msm_median": statistics.median(ts[2:]), "ms_per_msm_min": min(ts[2:]),
                    "scalar_muls_per_s": n / (statistics.median(ts[2:]) * 1e-3), "edwards_bucket_sums": ed_ph.get("edwards", 0.0) == 1.0,
                    "kernel": "k_accumulate_ed28_seg<Bls377>", "accumulate_ms": acc_ed, "tiles": int(ed_ph.get("tiles", 1)),
                    # 7 products per unified mixed addition: 7 x (196 + 196) v_mad_i64_i32 + 7 x 14 v_mul_lo_u32
                    "v_mad_frac_of_measured_peak": ((n * n_windows * (7 * 392 + 7 * 14)) / (acc_ed * 1e-3)) / INT_MAC_PEAK if acc_ed > 0 else 0.0,
                    "same_result_as_the_timed_steps": out_srs == res[G1] if world == 1 else None,
                }
            # PCIe-inclusive protocols of SURVEY 8d on this rank's G1 shard: (b) resident bases + scalars from host
            # memory per call, (c) points and scalars from host memory per call (the reference-shaped MultiScalarMul)
            if rank == 0 and G1 in groups:
                hp = points[G1].cpu().numpy().tobytes()
                hs = scalars.cpu().numpy().tobytes()
                out_h = ctypes.create_string_buffer(g1b)
                handle = ctypes.c_void_p()
                _lib.check(lib.mlhip_bases_create(CURVE, G1, hp, n, cfg.get("window_c", WINDOW_C), ctypes.byref(handle)))
                # back-to-back calls, as a prover issues them: the copies to the host just above left the GPU idle for ~0.1 s,
                # and the first calls after that run at idle clocks (a 3 ms MSM does not ramp them up) -- W untimed calls first
                tb, tc = [], []
                nwarm, ntimed = max(args.warmup, 4), max(args.steps // 2, 8)
                for i in range(nwarm + ntimed):
                    t1 = time.perf_counter()
                    _lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out_h))
                    if i >= nwarm:
                        tb.append((time.perf_counter() - t1) * 1e3)
                same_b = world > 1 or out_h.raw == res[G1]
                _lib.check(lib.mlhip_bases_destroy(handle))
                # ... and the same handle with the geometry left to the library (window_c = 0): from 2^17 bases on it keeps
                # shifted-base tables (include/mlhip.h: mlhip_bases_create; msm_fold.h) -- 13 digits of 20 bits into ONE bucket
                # set instead of 16 windows.  Not BASELINE's "c = 16", so never `value`; timed as protocol (b) and with the scalars
                # resident (mlhip_bases_msm_device), phases from the handle's plan.
                t1 = time.perf_counter()
                _lib.check(lib.mlhip_bases_create(CURVE, G1, hp, n, 0, ctypes.byref(handle)))
                create_ms = (time.perf_counter() - t1) * 1e3
                tplan = lib.mlhip_bases_plan(handle)
                _lib.check(lib.mlhip_msm_plan_set_profiling(tplan, 1))
                tbt, tdt = [], []
                for i in range(nwarm + ntimed):
                    t1 = time.perf_counter()
                    _lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out_h))
                    if i >= nwarm:
                        tbt.append((time.perf_counter() - t1) * 1e3)
                same_t = world > 1 or out_h.raw == res[G1]
                for i in range(nwarm + ntimed):
                    t1 = time.perf_counter()
                    _lib.check(lib.mlhip_bases_msm_device(handle, scalars.data_ptr(), 0, n, stream, out_h))
                    if i >= nwarm:
                        tdt.append((time.perf_counter() - t1) * 1e3)
                same_t = same_t and (world > 1 or out_h.raw == res[G1])
                tph = _lib.plan_timings(lib, tplan)
                _lib.check(lib.mlhip_bases_destroy(handle))
                extra["resident_bases_library_geometry"] = {
                    "shifted_base_tables": tph.get("tables", 0.0) == 1.0, "digit_bits": tph.get("window_c"), "digits_per_scalar": tph.get("digits_per_scalar"),
                    "create_ms": create_ms,
                    "protocol_b_scalars_from_host_ms": {"median": statistics.median(tbt), "min": min(tbt), "max": max(tbt)},
                    "scalars_resident_ms": {"median": statistics.median(tdt), "min": min(tdt), "max": max(tdt)},
                    "protocol_b_scalar_muls_per_s": n / (statistics.median(tbt) * 1e-3),
                    "scalars_resident_scalar_muls_per_s": n / (statistics.median(tdt) * 1e-3),
                    "phase_ms_scalars_resident": {k: tph[k] for k in ("digits", "sort", "accumulate", "reduce", "device_total", "host_tail") if k in tph},
                    "edwards_bucket_sums": tph.get("edwards", 0.0) == 1.0, "group": "G1", "pairs": n, "match_resident_result": bool(same_t),
                    "note": "not BASELINE's c = 16 geometry: reported beside `value`, never as it",
                }
                for i in range(nwarm + ntimed):
                    t1 = time.perf_counter()
                    _lib.check(lib.mlhip_msm_g1(CURVE, hp, hs, 0, n, cfg.get("window_c", WINDOW_C), out_h))
                    if i >= nwarm:
                        tc.append((time.perf_counter() - t1) * 1e3)
                same_c = world > 1 or out_h.raw == res[G1]
                extra["pcie_inclusive"] = {
                    "protocol_b_scalars_from_host_resident_bases_ms": {"median": statistics.median(tb), "min": min(tb), "max": max(tb)},
                    "protocol_c_points_and_scalars_from_host_ms": {"median": statistics.median(tc), "min": min(tc), "max": max(tc)},
                    "protocol_b_scalar_muls_per_s": n / (statistics.median(tb) * 1e-3),
                    "protocol_c_scalar_muls_per_s": n / (statistics.median(tc) * 1e-3),
                    "group": "G1", "pairs": n, "calls_timed": ntimed, "calls_untimed_first": nwarm,
                    "match_resident_result": bool(same_b and same_c),
                }
                if G2 in groups:
                    # config 4's G2 half over a handle made from the points already on the device (mlhip_bases_create_device),
                    # scalars resident: with the G1 figure above, what the step costs over shifted-base tables (two handles, two
                    # sorts -- the shared-scalar sort of `value` is not available to them)
                    t1 = time.perf_counter()
                    _lib.check(lib.mlhip_bases_create_device(CURVE, G2, points[G2].data_ptr(), n, 0, ctypes.byref(handle)))
                    create2_ms = (time.perf_counter() - t1) * 1e3
                    g2plan = lib.mlhip_bases_plan(handle)
                    _lib.check(lib.mlhip_msm_plan_set_profiling(g2plan, 1))
                    out_g2 = ctypes.create_string_buffer(g2b)
                    tg2 = []
                    for i in range(2 + 3):
                        t1 = time.perf_counter()
                        _lib.check(lib.mlhip_bases_msm_device(handle, scalars.data_ptr(), 0, n, stream, out_g2))
                        if i >= 2:
                            tg2.append((time.perf_counter() - t1) * 1e3)
                    g2ph = _lib.plan_timings(lib, g2plan)
                    _lib.check(lib.mlhip_bases_destroy(handle))
                    g1t = extra["resident_bases_library_geometry"]["scalars_resident_ms"]["median"]
                    extra["resident_bases_library_geometry"]["g2"] = {
                        "shifted_base_tables": g2ph.get("tables", 0.0) == 1.0, "digits_per_scalar": g2ph.get("digits_per_scalar"), "create_ms": create2_ms,
                        "scalars_resident_ms": {"median": statistics.median(tg2), "min": min(tg2), "max": max(tg2)},
                        "phase_ms": {k: g2ph[k] for k in ("digits", "sort", "accumulate", "reduce", "device_total", "host_tail") if k in g2ph},
                        "match_resident_result": bool(world > 1 or out_g2.raw == res[G2]),
                        "g1_plus_g2_ms": g1t + statistics.median(tg2), "pairs_per_s": n / ((g1t + statistics.median(tg2)) * 1e-3),
                    }
                extra["headline_protocol_b"] = extra["pcie_inclusive"]["protocol_b_scalar_muls_per_s"]
                extra["headline_protocol_b_library_geometry"] = extra["resident_bases_library_geometry"]["protocol_b_scalar_muls_per_s"]
                del hp, hs
                lib.mlhip_release_cache()
        # ---- batched pairing beside the MSM headline (BASELINE configs[2]): 65 536 x (Miller loop + FExp)
        if args.config == 2 and not args.no_pairing:
            npair, q, gt = pairing_setup()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tp = []
            for _ in range(4):
                ev0.record()
                _lib.check(lib.mlhip_pairing_batch_device(CURVE, points[G1].data_ptr(), q.data_ptr(), npair, gt.data_ptr(), stream))
                ev1.record()
                torch.cuda.synchronize()
                tp.append(ev0.elapsed_time(ev1))
            best = min(tp[1:])
            extra["pairings_per_s_per_gpu"] = npair / (statistics.median(tp[1:]) * 1e-3)
            extra["pairing_batch"] = npair
            extra["pairing_kernel_ms"] = {"median": statistics.median(tp[1:]), "min": best}
            extra["pairing_roofline"] = {
                "bound": "hbm", "achieved": PAIRING_BYTES_PER_UNIT * npair / (best * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": PAIRING_BYTES_PER_UNIT * npair / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": PAIRING_BYTES_PER_UNIT * npair / (best * 1e-3) / 1e9 / HBM_MEASURED_GBS,
                "kernel": "k_pairing_lp28<Bls381,2,1>",
            }
            extra["pairing_roofline"]["traffic"], extra["pairing_roofline"]["traffic_note"] = _pmc(3, "k_pairing_lp28<Bls381, 2")
            # the north star names BN254 beside BLS12-381: the same batch on that curve (P_i = [k_i]G1, Q_i = [k_i]G2 made on the
            # device), rank 0 only, never part of `value`
            if rank == 0:
                bn = _lib.CURVE_BN254
                _, bg1, bg2, bgt = _lib.sizes(bn)
                with open(os.path.join(ROOT, "tests", "golden", "bn254.json")) as fbn:
                    gbn = json.load(fbn)
                bp, bq = torch.empty(npair * bg1, dtype=torch.uint8, device=dev), torch.empty(npair * bg2, dtype=torch.uint8, device=dev)
                for grp, key, dst in ((G1, "g1_gen", bp), (G2, "g2_gen", bq)):
                    base = torch.frombuffer(bytearray(bytes.fromhex(gbn[key])), dtype=torch.uint8).to(dev)
                    _lib.check(lib.mlhip_scalar_mul_device(bn, grp, base.data_ptr(), 0, scalars[:npair].contiguous().data_ptr(), 0, npair, dst.data_ptr(), stream))
                bgt_out = torch.empty(npair * bgt, dtype=torch.uint8, device=dev)
                tb = []
                for _ in range(4):
                    ev0.record()
                    _lib.check(lib.mlhip_pairing_batch_device(bn, bp.data_ptr(), bq.data_ptr(), npair, bgt_out.data_ptr(), stream))
                    ev1.record()
                    torch.cuda.synchronize()
                    tb.append(ev0.elapsed_time(ev1))
                extra["pairings_per_s_bn254"] = npair / (statistics.median(tb[1:]) * 1e-3)
                del bp, bq, bgt_out
                # ... and the G1 MSM of this config's size on BN254 (driver/gurvy/bn254.go:232-245): 2^20 pairs, c = 16, inputs
                # resident, one MSM in flight -- the protocol of `value`, never part of it
                base = torch.frombuffer(bytearray(bytes.fromhex(gbn["g1_gen"])), dtype=torch.uint8).to(dev)
                bpts = torch.empty(n * bg1, dtype=torch.uint8, device=dev)
                _lib.check(lib.mlhip_scalar_mul_device(bn, G1, base.data_ptr(), 0, seeded_scalars(n_total, lo, hi, seed * 3 + 7, dev).data_ptr(), 0, n, bpts.data_ptr(), stream))
                torch.cuda.synchronize()
                bplan = _lib.MsmPlan(bn, G1, n, WINDOW_C)
                tm = []
                for _ in range(6):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    bplan.launch(bpts.data_ptr(), scalars.data_ptr(), n, False, stream)
                    bplan.finish()
                    tm.append((time.perf_counter() - t1) * 1e3)
                bplan.close()
                extra["bn254_g1_msm"] = {"pairs": n, "window_c": WINDOW_C, "ms_per_msm_median": statistics.median(tm[1:]), "ms_per_msm_min": min(tm[1:]),
                                         "scalar_muls_per_s": n / (statistics.median(tm[1:]) * 1e-3), "algorithmic_bytes_per_scalar_mul": MSM_BYTES[(bn, G1)]}
                del bpts

        # ---- batched G1.Mul beside the MSM headline (north_star's double-and-add kernel): one base per scalar
        # (signed 4-bit windows, 2^17 products) and one base for all scalars (fixed-base table, 2^20 products, the table of the
        # generator still on the device from the input generation or built by the first of these calls)
        if args.config == 2 and rank == 0 and not args.no_pairing and not args.kernels_only:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nm = min(n, 1 << 17)
            gen1 = torch.frombuffer(bytearray(curve.GenG1().raw), dtype=torch.uint8).to(dev)
            dst = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
            times = {"per_point": [], "one_base": []}
            for _ in range(3):
                ev0.record()
                _lib.check(lib.mlhip_scalar_mul_device(CURVE, G1, points[G1].data_ptr(), 1, scalars.data_ptr(), 0, nm, dst.data_ptr(), stream))
                ev1.record()
                torch.cuda.synchronize()
                times["per_point"].append(ev0.elapsed_time(ev1))
                ev0.record()
                _lib.check(lib.mlhip_scalar_mul_device(CURVE, G1, gen1.data_ptr(), 0, scalars.data_ptr(), 0, n, dst.data_ptr(), stream))
                ev1.record()
                torch.cuda.synchronize()
                times["one_base"].append(ev0.elapsed_time(ev1))
            extra["batched_g1_mul"] = {
                "per_point_bases_muls_per_s": nm / (min(times["per_point"][1:]) * 1e-3), "per_point_bases_n": nm,
                "one_base_muls_per_s": n / (min(times["one_base"][1:]) * 1e-3), "one_base_n": n,
                "one_base_ms": {"first_call": times["one_base"][0], "later_calls_min": min(times["one_base"][1:])},
            }
            del dst

    steps_n = max(args.steps, 1)
    value = units_per_step * args.steps / elapsed
    if step_ms:
        extra["ms_per_step_median"] = statistics.median(step_ms)
        extra["ms_per_step_min"] = min(step_ms)
        extra["ms_per_step_max"] = max(step_ms)
    if world > 1:
        # where an N > 1 line's efficiency went: every rank's own median step and, within it, the exchange (all-gather of
        # the 96 / 192-byte partial sums over RCCL + the local EC additions; it also absorbs the wait for the slowest rank)
        # and the kernels' device time -- one all_gather_object after the timed region
        import torch.distributed as dist

        mine = {"rank": rank, "ms_per_step_median": statistics.median(step_ms) if step_ms else None,
                "exchange_ms_median": statistics.median(exchange_ms) if exchange_ms else None,
                "exchange_ms_max": max(exchange_ms) if exchange_ms else None,
                "device_ms_per_step": sum(v for (g, k_), v in phase.items() if k_ == "device_total") / steps_n if phase else None}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            med = [r["ms_per_step_median"] for r in allr if r["ms_per_step_median"] is not None]
            exm = [r["exchange_ms_median"] for r in allr if r["exchange_ms_median"] is not None]
            exx = [r["exchange_ms_max"] for r in allr if r["exchange_ms_max"] is not None]
            dvm = [r["device_ms_per_step"] for r in allr if r["device_ms_per_step"] is not None]
            extra["per_rank_ms_per_step"] = {"min": min(med), "max": max(med), "all": med} if med else None
            extra["exchange_ms"] = {"median": statistics.median(exm), "max": max(exx), "per_rank_median": exm,
                                    "what": "perf_counter around mathlib_amd.dist.combine_many in every timed step: H2D of the partials, "
                                            "ONE all-gather, D2H, the local EC additions on the host; includes waiting for the slowest rank"} if exm else None
            extra["per_rank_device_ms_per_step"] = {"min": min(dvm), "max": max(dvm)} if dvm else None

    # ---- CPU baseline: the oracle's C restatement on a bounded sample of the same workload, host cores of this box
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cref  # checker / baseline only

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(cores, 64))
        if args.config == 3:
            ns = 2048
            hp = points[G1][: ns * g1b].cpu().numpy()
            hq = q[: ns * g2b].cpu().numpy()
            t1 = time.perf_counter()
            cpu_gt = cref.pairing_batch(CURVE, hp, hq, ns, threads)
            dt = time.perf_counter() - t1
            cpu_baseline = {"value": ns / dt, "unit": "pairings/s", "cores": threads, "kind": "port",
                            "sample": "%d of the 65 536 pairs, oracle/cref, %d pthreads, %.2f s" % (ns, threads, dt),
                            "matches_gpu_result": bool(cpu_gt == bytes(gt[: ns * gtb].cpu().numpy().tobytes()))}
        else:
            rates, notes, ok = [], [], True
            for g in groups:
                ns = min(n, 1 << 20) if g == G1 else min(n, 1 << 18)
                ptsz = g1b if g == G1 else g2b
                hp = points[g][: ns * ptsz].cpu().numpy()
                hs = scalars[:ns].cpu().numpy()
                t1 = time.perf_counter()
                ref = cref.msm(CURVE, g, hp, hs, ns, False, WINDOW_C, threads)
                dt = time.perf_counter() - t1
                rates.append(ns / dt)
                notes.append("%s: first %d pairs, %.2f s" % ("G1" if g == G1 else "G2", ns, dt))
                if ns == n:
                    ok = ok and ref == res_check[g]
                else:  # the GPU on the same sample
                    pl = _lib.MsmPlan(CURVE, g, ns, cfg.get("window_c", WINDOW_C))
                    ok = ok and ref == pl.run(points[g].data_ptr(), scalars.data_ptr(), ns, False, stream)
                    pl.close()
            cpu_baseline = {
                "value": 1.0 / sum(1.0 / r for r in rates), "unit": unit, "cores": threads, "kind": "port",
                "sample": "oracle/cref Pippenger c=16, %d pthreads; %s" % (threads, "; ".join(notes)),
                "matches_gpu_result": bool(ok),
            }
            if args.config == 2 and not args.no_pairing:
                ns = 2048
                t1 = time.perf_counter()
                cpu_gt = cref.pairing_batch(CURVE, points[G1][: ns * g1b].cpu().numpy(), q[: ns * g2b].cpu().numpy(), ns, threads)
                dtp = time.perf_counter() - t1
                cpu_baseline["pairings_per_s"] = ns / dtp
                cpu_baseline["pairings_match_gpu_result"] = bool(cpu_gt == bytes(gt[: ns * gtb].cpu().numpy().tobytes()))
                cpu_baseline["pairing_sample"] = "%d of the 65 536 pairs, %d pthreads, %.2f s" % (ns, threads, dtp)

    if rank == 0:
        par = {
            "weak": "every rank holds its own 2^%d pairs; one all-gather of the partial sums over RCCL + local EC add" % log_n,
            "strong": "the 2^%d pairs sharded contiguously across ranks; one all-gather of the %s-byte partial sums over RCCL + local EC add"
            % (log_n, "+".join(str(g1b if g == G1 else g2b) for g in groups)),
        }[cfg["scaling"]]
        if args.config == 3:
            par = "independent pairings split across ranks, no collective"
        line = {
            "metric": metric,
            "value": value,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps_n * 1e3,
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo, not a measurement)" if rehearsal else "")
            + (" (REDUCED SIZE --log-n %d: not a measurement of the config)" % log_n if reduced else ""),
            "config": {
                "workload": cfg["workload"],
                "baseline_config": args.config,
                "curve": cfg["curve_name"],
                "pairs_total": n_total,
                "pairs_per_gpu": n,
                "window_c": window_c if args.config != 3 else None,
                "parallelism": par,
                "protocol": "SURVEY 8d (a): inputs resident in HBM; kernels + D2H of the window sums + host tail" + ("" if args.config == 3 else "; %d step(s) in flight" % (2 if args.pipelined else 1))
                + ("; G1 and G2 share their scalars: one sort per tile for both (mlhip_msm_launch_shared)" if args.config == 4 and not args.separate_sorts else ""),
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def _pmc(config: int, kernel_prefix: str):
    """HBM-side traffic of a kernel from the committed PMC passes (the counters cannot be read from inside this process):
    (bytes per launch or None, note).  The file names the sources it was taken on (mathlib_amd.build.source_hash):
    when the kernels have changed since, the figure is withheld rather than quoted for code it was not measured on."""
    from mathlib_amd.build import source_hash

    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pm = json.load(f)
            kk = next(v for k_, v in pm["configs"][str(config)].items() if k_.startswith(kernel_prefix) and "Fp2" not in k_)
        except Exception:
            continue
        if pm.get("source_hash") != source_hash():
            return (None, "profiles/%s was taken on other kernel sources (hash %s, commit %s; this tree: %s): traffic withheld"
                    % (name, pm.get("source_hash"), pm.get("commit"), source_hash()))
        return ((kk["FETCH_SIZE"] + kk["WRITE_SIZE"]) * 1024.0,
                "bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB from profiles/%s (separate --pmc passes of `bench.py --config %d "
                "--kernels-only`, commit %s, source hash %s = this tree), uncorrected: the x2 gfx950 correction applies to wide "
                "coalesced streaming reads, this kernel gathers 16-byte pieces of random rows / spills through scratch "
                "(uncalibrated pattern); Infinity-Cache hits are counted" % (name, config, pm.get("commit"), pm.get("source_hash")))
    return (None, "no PMC pass committed for this config")


if __name__ == "__main__":
    main()
