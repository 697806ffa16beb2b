"""The N > 1 MSM path end to end with the GPU producing the partials (BASELINE configs[3] / [4] at world_size 2):
two ranks share device 0 and exchange over gloo (RCCL needs one device per rank; this box has one).  Every rank
runs the single-GPU pipeline on its contiguous shard (mathlib_amd.dist.shard_bounds), the partial sums go through
mathlib_amd.dist.combine_partials (all-gather + mlhip_g1_sum / mlhip_g2_sum), and the total must equal both the
single-GPU MSM over the whole input and the C oracle.  tests/test_dist_gloo.py is the CPU-only form of the same
exchange (its partials come from the oracle)."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, cases, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mathlib_amd import _lib, dist as mdist
    from oracle import cref

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    ok = True
    msgs = []
    for curve, group, n, c in cases:
        # every rank derives the same whole input, then keeps only its shard on the device
        pts = cref.gen_points(curve, group, 1234 + curve, 77 + group, n)
        sc = np.random.default_rng(1000 + n).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        ps = len(pts) // n
        lo, hi = mdist.shard_bounds(n, rank, world)
        d_pts = torch.frombuffer(bytearray(pts[lo * ps : hi * ps]), dtype=torch.uint8).to(dev)
        d_sc = torch.from_numpy(sc[lo:hi].copy().view(np.uint8).reshape(-1)).to(dev)
        plan = _lib.MsmPlan(curve, group, hi - lo, c)
        part = plan.run(d_pts.data_ptr(), d_sc.data_ptr(), hi - lo, False, st)
        plan.close()
        total = mdist.combine_partials(curve, group, part)
        want = cref.msm(curve, group, pts, sc, n, False, 0, 8)
        if total != want:
            ok = False
            msgs.append("combine != oracle for %r" % ((curve, group, n, c),))
        if rank == 0:  # the same input as ONE MSM on the device
            whole = _lib.MsmPlan(curve, group, n, c)
            d_all = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
            d_sall = torch.from_numpy(sc.view(np.uint8).reshape(-1).copy()).to(dev)
            single = whole.run(d_all.data_ptr(), d_sall.data_ptr(), n, False, st)
            whole.close()
            if single != total:
                ok = False
                msgs.append("combine != single-GPU MSM for %r" % ((curve, group, n, c),))
    ret[rank] = (ok, msgs)
    dist.barrier()
    dist.destroy_process_group()


def test_gpu_shards_combine_to_the_single_msm():
    import torch.multiprocessing as mp

    world = 2
    # (curve, group, n, window): BLS12-381 G1 and G2 (config 4's two halves), BLS12-377 G1 (config 5), ragged n
    cases = [(1, 1, 40001, 16), (1, 2, 9001, 12), (2, 1, 30011, 16), (0, 1, 5003, 0)]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, cases, ret), nprocs=world, join=True)
    for r in range(world):
        ok, msgs = ret.get(r, (False, ["rank %d did not report" % r]))
        assert ok, msgs


def _rccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from mathlib_amd import dist as mdist

    ok = True
    for rep in range(3):  # the staging tensors are created once and reused
        for nb in (96, 288):
            local = bytes((7 * i + rep + nb) & 255 for i in range(nb))
            ok &= mdist._all_gather_bytes(local, None) == local * world
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_over_rccl_one_rank():
    """The exchange step's device branch (pinned staging -> all-gather on RCCL -> pinned staging) with the nccl backend:
    one rank, because this box has one GPU and RCCL wants one device per rank -- the collective, the stream ordering
    around it and the buffer reuse are the ones every rank of an N-GPU run executes."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret.get(0, False)


@pytest.mark.parametrize("config,log_n", [(2, 14), (4, 15), (3, 12)])
def test_bench_plain_gpus_n_command(config, log_n):
    """The driver's command shape for N > 1, `python bench.py --gpus 2 ...` with no torchrun in front: bench.py starts its
    two ranks itself (a child torch.distributed.run), rank 0's JSON line comes back on stdout and the exit code is the
    child's.  On a one-GPU box the ranks share device 0 and exchange over gloo (MLHIP_BENCH_REHEARSAL=1: the line says
    so and is not a measurement); the N > 1 code path -- shard bounds, one all-gather of the partial sums, the local
    EC additions, the max over ranks -- is the one eight GPUs run."""
    import json
    import subprocess

    env = dict(os.environ, MLHIP_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", str(config), "--log-n", str(log_n),
                        "--kernels-only", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert "REHEARSAL" in d["data"] and d["config"]["baseline_config"] == config
    assert d["config"]["pairs_total"] == 2 * d["config"]["pairs_per_gpu"] == 2 << log_n if d["scaling"] == "weak" else d["config"]["pairs_total"] == 2 * d["config"]["pairs_per_gpu"] == 1 << log_n


@pytest.mark.parametrize("config,ranks", [(4, 3), (5, 4)])
def test_bench_multi_rank_rehearsal_is_self_explaining(config, ranks):
    """the strong-scaling configs as the driver will run them at N = 8, rehearsed with the ranks a
    one-GPU box allows beside the test process itself (its process guard admits 6 processes on the card: 3 ranks --
    ragged shards of 2^18 pairs -- and 4), gloo on device 0; tests/test_dist_gloo.py has the 8-rank exchange on the CPU.
    The line must say where an N > 1 run's time went: every rank's median step, the exchange (all-gather + local EC
    additions) timed by itself, the kernels' device time per rank."""
    import json
    import subprocess

    env = dict(os.environ, MLHIP_BENCH_REHEARSAL="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--config", str(config), "--log-n", "18",
                        "--kernels-only", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["value"] > 0 and "REHEARSAL" in d["data"]
    assert d["config"]["pairs_total"] == 1 << 18 and d["config"]["pairs_per_gpu"] in ((1 << 18) // ranks, (1 << 18) // ranks + 1)
    ex = d["extra"]
    assert len(ex["per_rank_ms_per_step"]["all"]) == ranks and ex["per_rank_ms_per_step"]["min"] <= ex["per_rank_ms_per_step"]["max"]
    assert len(ex["exchange_ms"]["per_rank_median"]) == ranks and 0 < ex["exchange_ms"]["median"] <= ex["exchange_ms"]["max"]
    assert ex["exchange_ms"]["median"] < ex["per_rank_ms_per_step"]["max"]
    assert 0 < ex["per_rank_device_ms_per_step"]["min"] <= ex["per_rank_device_ms_per_step"]["max"]


def test_host_pool_is_sized_per_rank():
    """api.hip: host_pool_start -- the host-tail worker pool is sized from the process's affinity mask divided by
    LOCAL_WORLD_SIZE (8 ranks of `bench.py --gpus 8` share one host), halved when ranks share the host; MLHIP_HOST_THREADS
    overrides.  The workers are named, so a fresh process can count them after one MSM."""
    import subprocess

    code = r"""
import os, sys, glob, ctypes
sys.path.insert(0, %r)
import numpy as np
from mathlib_amd import _lib
from oracle import cref
lib = _lib.load()
n = 3000
pts = cref.gen_points(1, 1, 5, 6, n)
sc = np.random.default_rng(1).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
out = ctypes.create_string_buffer(96)
_lib.check(lib.mlhip_msm_g1(1, pts, sc.tobytes(), 0, n, 12, out))
assert out.raw == cref.msm(1, 1, pts, sc, n, False, 0, 4)
names = [open(f).read().strip() for f in glob.glob('/proc/self/task/*/comm')]
print('WORKERS', sum(1 for x in names if x == 'mlhip-host'), len(os.sched_getaffinity(0)))
""" % ROOT

    def workers(**env_over):
        env = dict(os.environ)
        for k in ("LOCAL_WORLD_SIZE", "MLHIP_HOST_THREADS"):
            env.pop(k, None)
        env.update(env_over)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        w, cores = [ln for ln in r.stdout.splitlines() if ln.startswith("WORKERS")][0].split()[1:]
        return int(w), int(cores)

    w, cores = workers()
    assert w == min(8, max(1, cores)) - 1, (w, cores)
    w, cores = workers(LOCAL_WORLD_SIZE="4")
    assert w == min(8, max(1, cores // 4 // 2)) - 1, (w, cores)
    w, _ = workers(LOCAL_WORLD_SIZE="4", MLHIP_HOST_THREADS="3")
    assert w == 2
