"""world_size-2 and world_size-8 rehearsals (gloo, CPU) of the multi-GPU MSM path: contiguous sharding of the pairs and
the all-gather + local EC-add combination (mathlib_amd/dist.py).  On the CPU the per-rank partial MSM
is produced by the oracle (there is no GPU here); the exchange and the combine are the product code."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np

    from mathlib_amd import _lib, dist as mdist
    from oracle import cref

    ok = True
    for curve, group in ((1, 1), (1, 2), (0, 1)):
        pts = cref.gen_points(curve, group, 4242, 99, n)
        sc = np.random.default_rng(5).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        ps = len(pts) // n
        lo, hi = mdist.shard_bounds(n, rank, world)
        part = cref.msm(curve, group, pts[lo * ps : hi * ps], sc[lo:hi].copy(), hi - lo)
        total = mdist.combine_partials(curve, group, part)
        ok = ok and total == cref.msm(curve, group, pts, sc, n)
    # configs[3]: the G1 and the G2 partial travel in ONE all-gather
    n2 = n // 2
    g1p = cref.gen_points(1, 1, 17, 3, n2)
    g2p = cref.gen_points(1, 2, 19, 5, n2)
    sc = np.random.default_rng(6).integers(0, 1 << 63, size=(n2, 4), dtype=np.uint64)
    lo, hi = mdist.shard_bounds(n2, rank, world)
    parts = [(1, cref.msm(1, 1, g1p[lo * 96 : hi * 96], sc[lo:hi].copy(), hi - lo)), (2, cref.msm(1, 2, g2p[lo * 192 : hi * 192], sc[lo:hi].copy(), hi - lo))]
    t1, t2 = mdist.combine_many(1, parts)
    ok = ok and t1 == cref.msm(1, 1, g1p, sc, n2) and t2 == cref.msm(1, 2, g2p, sc, n2)
    # degenerate: every rank holds the identity
    zero = bytes(96)
    ok = ok and mdist.combine_partials(1, 1, zero) == zero
    ret[rank] = ok
    dist.destroy_process_group()


def test_shard_bounds_cover_the_range():
    from mathlib_amd.dist import shard_bounds

    for n in (0, 1, 7, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_two_rank_combine_matches_single_msm():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, 301, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_eight_rank_combine_ragged():
    """The shape of the one hardware run at N = 8 (BASELINE configs[3] and [4]): eight ranks, n not a multiple of eight
    (shards of 37 and 38 pairs), G1 / G2 / BN254 partials, config 4's two partials in ONE all-gather, identity partials."""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, 301, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
