"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).

The hot path shards by image: inference needs NO data-path collective -- each
rank runs its slice of the batch -- and only the optional gather of the uint8 masks to rank 0 uses
RCCL.  (Training adds one exchange step, the gradient all-reduce over the flat parameter arena.)
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n images for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def predict_sharded(images: torch.Tensor, predict_fn: Callable[[torch.Tensor], torch.Tensor],
                    gather: bool = True) -> Optional[torch.Tensor]:
    """Batch-split inference: every rank holds the same global batch `images` [B, 3, S, S] (or at least its
    own slice), runs `predict_fn` on rows shard_range(B) and, if `gather`, rank 0 receives the uint8 masks
    of the whole batch in order.  `predict_fn` maps [b, 3, S, S] -> uint8 [b, S, S] on the same device."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = images.shape[0]
    lo, hi = shard_range(B, rank, world)
    mine = predict_fn(images[lo:hi]) if hi > lo else images.new_zeros((0,) + images.shape[2:], dtype=torch.uint8)
    if not gather or world == 1:
        return mine
    q = (B + world - 1) // world  # pad every shard to the largest so one all_gather moves everything
    padded = torch.zeros((q,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    padded[: hi - lo] = mine
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        l, h = shard_range(B, r, world)
        out.append(parts[r][: h - l])
    return torch.cat(out)


def allreduce_grads(flat_grad: torch.Tensor, bucket_mb: float = 64.0) -> None:
    """Sum `flat_grad` (the gradient of the flat parameter arena) over all ranks, in place.

    The arena is laid out in forward order, so its TAIL (head, last layers) is what the backward pass
    finishes first: buckets are issued from the tail towards the front, each as one large asynchronous
    all-reduce (xGMI rings are per-link bound, so few large messages beat many small ones), then waited
    for together.  The optimizer applies 1/world (`FusedAdam.step(grad_scale=1/world)`), which makes the
    update equal to the single-process update on the global batch (mean loss over all pixels)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    n = flat_grad.numel()
    per = max(1, int(bucket_mb * (1 << 20) / flat_grad.element_size()))
    works = []
    hi = n
    while hi > 0:
        lo = max(0, hi - per)
        works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        hi = lo
    for w in works:
        w.wait()
