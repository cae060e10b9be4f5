"""Sharding a batch of independent proofs across the GPUs of one node.

Each proof is verified in isolation, so the batch partitions by contiguous index ranges with NO data-path
collective; the only exchange is the final gather of the per-rank accept bytes on rank 0 (RCCL over xGMI on a GPU
node; `gloo` in the CPU tests).  One process per GPU, launched by torch.distributed.run.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of proofs owned by `rank`: [rank*n/world, (rank+1)*n/world)."""
    return n * rank // world, n * (rank + 1) // world


def slice_batch(proofs: bytes, proof_off: List[int], instances: bytes, committed: Optional[bytes], n_pi: int,
                lo: int, hi: int):
    """The sub-batch [lo, hi) in the C-ABI layout (offsets rebased to 0)."""
    base = proof_off[lo]
    off = [o - base for o in proof_off[lo:hi + 1]]
    return (proofs[base:proof_off[hi]], off, instances[32 * n_pi * lo:32 * n_pi * hi],
            committed[48 * lo:48 * hi] if committed else None)


def verify_sharded(verify_fn: Callable[[bytes, List[int], bytes, Optional[bytes]], bytes], proofs: bytes,
                   proof_off: List[int], instances: bytes, committed: Optional[bytes], n_pi: int,
                   device=None) -> Optional[bytes]:
    """Every rank holds the whole batch description, verifies its own range with `verify_fn` (the C-ABI call on
    its GPU) and the accept bytes are gathered on rank 0 (returns None on the other ranks).
    Works without torch.distributed initialised (world = 1)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    n = len(proof_off) - 1
    lo, hi = shard_range(n, rank, world)
    sub = slice_batch(proofs, proof_off, instances, committed, n_pi, lo, hi)
    mine = verify_fn(*sub) if hi > lo else b""
    if world == 1:
        return bytes(mine)
    # pad every shard to the same length for the gather
    width = max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))
    buf = torch.zeros(width, dtype=torch.uint8, device=device)
    if mine:
        buf[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(buf.device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    out = bytearray()
    for r in range(world):
        l, h = shard_range(n, r, world)
        out += bytes(gathered[r][:h - l].cpu().numpy().tobytes())
    return bytes(out)
