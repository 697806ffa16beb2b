"""Multi-GPU combination step of a sharded MSM (BASELINE configs 4 and 5).

The n (point, scalar) pairs are split contiguously across ranks, one process per GPU; every rank
runs the whole single-GPU pipeline on its shard down to ONE affine partial sum.  The only exchange is
an all-gather of those partial sums (96 B for G1, 192 B for G2 per rank -- latency-bound on xGMI),
followed by a local elliptic-curve addition of the world_size partials.  Elliptic-curve addition is
not an RCCL reduction operator, so the north-star's "bucket all-reduce" is all-gather + local add.
Works with the nccl (= RCCL) backend on GPU tensors and with gloo on CPU tensors (tests).
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib


def shard_bounds(n: int, rank: int, world: int):
    """contiguous shard [lo, hi) of rank"""
    return n * rank // world, n * (rank + 1) // world


_bufs: dict = {}  # (device, nbytes, world) -> (pinned source, device source, device gathered, pinned gathered)


def _all_gather_bytes(local: bytes, device: torch.device | None) -> bytes:
    """every rank's `local` (same length on all ranks), concatenated in rank order: ONE collective.  The staging
    tensors are kept between calls (a step's exchange is 96 + 192 bytes: allocation and synchronous copies would cost
    more than the collective): pinned host -> device, all-gather on the current stream, device -> pinned host, one wait."""
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    nb = len(local)
    if device.type != "cuda":  # gloo on CPU tensors (tests, rehearsals)
        mine = torch.frombuffer(bytearray(local), dtype=torch.uint8)
        try:
            flat = torch.empty(world * nb, dtype=torch.uint8)
            dist.all_gather_into_tensor(flat, mine)
            return bytes(flat.numpy().tobytes())
        except (RuntimeError, NotImplementedError):  # a backend without the flat form
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            return b"".join(bytes(t.numpy().tobytes()) for t in gathered)
    key = (device.index, nb, world)
    if key not in _bufs:
        _bufs[key] = (torch.empty(nb, dtype=torch.uint8).pin_memory(), torch.empty(nb, dtype=torch.uint8, device=device),
                      torch.empty(world * nb, dtype=torch.uint8, device=device), torch.empty(world * nb, dtype=torch.uint8).pin_memory())
    h_src, d_src, d_all, h_all = _bufs[key]
    h_src.copy_(torch.frombuffer(bytearray(local), dtype=torch.uint8))
    with torch.cuda.device(device):
        d_src.copy_(h_src, non_blocking=True)
        dist.all_gather_into_tensor(d_all, d_src)
        h_all.copy_(d_all, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return bytes(h_all.numpy().tobytes())


def combine_many(curve: int, parts, device: torch.device | None = None):
    """parts = [(group, local_affine_bytes), ...] -- e.g. the G1 and the G2 partial of BASELINE configs[3]: ONE
    all-gather of the concatenated partials (96 + 192 bytes per rank), then one local EC addition per part."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [p for _, p in parts]
    world = dist.get_world_size()
    blob = _all_gather_bytes(b"".join(p for _, p in parts), device)
    stride = len(blob) // world
    totals, off = [], 0
    for group, p in parts:
        mine = b"".join(blob[r * stride + off : r * stride + off + len(p)] for r in range(world))
        out = ctypes.create_string_buffer(len(p))
        fn = _lib.load().mlhip_g1_sum if group == _lib.GROUP_G1 else _lib.load().mlhip_g2_sum
        _lib.check(fn(curve, mine, world, out))
        totals.append(out.raw)
        off += len(p)
    return totals


def combine_partials(curve: int, group: int, local_affine: bytes, device: torch.device | None = None) -> bytes:
    """all-gather every rank's partial MSM result and add them; every rank gets the total."""
    return combine_many(curve, [(group, local_affine)], device)[0]
