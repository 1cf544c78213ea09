"""Frame sharding over ranks and reassembly of the extracted bit stream.

One process per GPU.  Rank r owns the contiguous frame range `batch.shard_frames(F, world, r)`;
frame k of the clip carries stream bits [k*cap, (k+1)*cap), so
  * embed needs no exchange: every rank reads the shared packed payload at bit offset
    first_frame * cap (the `bit_offset` argument of svs_embed_dev);
  * extract ends with ONE collective: the ranks' packed streams are gathered to rank `dst`
    (RCCL when the tensors live on GPUs - backend "nccl" is RCCL on ROCm - or gloo on CPU
    tensors in the tests) and concatenated in rank order.
The reference has no distributed code; this module is what the MI355X build adds.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import batch


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_frames: int) -> tuple[int, int]:
    """(first_frame, n_local_frames) of the calling rank."""
    rank, size = world()
    return batch.shard_frames(n_frames, size, rank)


def payload_bit_offset(first_frame: int, height: int, width: int, n_ac) -> int:
    return first_frame * batch.capacity_bits(1, height, width, n_ac)


def gather_packed(local_packed: torch.Tensor, n_bytes: int, dst: int = 0, group=None, recv=None):
    """The one data-path collective: equal-sized packed streams -> list of tensors on `dst`
    (rank order).  Stays on the device; `recv` may be a preallocated list to reuse."""
    rank, size = world()
    if size == 1:
        return [local_packed[:n_bytes]]
    if rank == dst and recv is None:
        recv = [torch.empty(n_bytes, dtype=torch.uint8, device=local_packed.device) for _ in range(size)]
    dist.gather(local_packed[:n_bytes], recv if rank == dst else None, dst=dst, group=group)
    return recv if rank == dst else None


def gather_stream(local_packed: torch.Tensor, local_bits: int, dst: int = 0, group=None):
    """Gather every rank's packed MSB-first stream to `dst` and join them in rank order.

    local_packed : uint8 tensor (CPU or GPU) holding at least ceil(local_bits / 8) bytes.
    Returns (packed uint8 numpy array, total_bits) on `dst`, (None, total_bits) elsewhere.
    Ranks may hold different numbers of bits; streams are joined at bit granularity."""
    rank, size = world()
    n_local = (local_bits + 7) // 8
    if size == 1:
        return local_packed[:n_local].cpu().numpy(), local_bits
    dev = local_packed.device
    counts = torch.zeros(size, dtype=torch.int64, device=dev)
    counts[rank] = local_bits
    dist.all_reduce(counts, group=group)                     # every rank learns every stream length
    counts = [int(c) for c in counts.tolist()]
    longest = (max(counts) + 7) // 8
    send = torch.zeros(longest, dtype=torch.uint8, device=dev)
    send[:n_local] = local_packed[:n_local]
    recv = gather_packed(send, longest, dst=dst, group=group)
    total = sum(counts)
    if rank != dst:
        return None, total
    parts = [t[: (c + 7) // 8].cpu().numpy() for t, c in zip(recv, counts)]
    if all(c % 8 == 0 for c in counts[:-1]):
        return np.concatenate(parts), total                  # byte aligned: plain concatenation
    bits = np.concatenate([np.unpackbits(p, count=c) for p, c in zip(parts, counts)])
    return np.packbits(bits), total
