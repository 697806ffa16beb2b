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


def combine_partials(curve: int, group: int, local_affine: bytes, device: torch.device | None = None) -> bytes:
    """all-gather every rank's partial MSM result and add them; every rank gets the total."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_affine
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.frombuffer(bytearray(local_affine), dtype=torch.uint8).to(device)
    try:  # one flat buffer: one collective, one copy back to the host
        flat = torch.empty(world * mine.numel(), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(flat, mine)
        blob = bytes(flat.cpu().numpy().tobytes())
    except (RuntimeError, NotImplementedError):  # a backend without the flat form
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in gathered)
    out = ctypes.create_string_buffer(len(local_affine))
    fn = _lib.load().mlhip_g1_sum if group == _lib.GROUP_G1 else _lib.load().mlhip_g2_sum
    _lib.check(fn(curve, blob, world, out))
    return out.raw
