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


class BucketReducer:
    """Gradient all-reduce overlapped with the backward.

    `ranges` are the arena ranges in the order `vitseg_backward` finishes them (`_lib.grad_buckets`: head, layers
    L-1 .. 0, embeddings); consecutive ones are merged until a message reaches `min_mb` -- they are adjacent in
    the arena, so a merged bucket is still one contiguous slice and one RCCL call (xGMI rings are per-link
    bound: few large messages).  `reduce()` issues one asynchronous all-reduce per bucket; on the GPU each is
    enqueued behind the event the backward records when that bucket's last gradient kernel has been launched
    (`events[i]`), on a side stream, so the ring for layer l runs while layers l-1 .. 0 are still computing."""

    def __init__(self, ranges, min_mb: float = 48.0, elem_bytes: int = 4):
        self.ranges = list(ranges)
        self.groups = []  # (lo, hi, index of the bucket whose event completes the group)
        lo = hi = None
        for i, (off, n) in enumerate(self.ranges):
            if lo is None:
                lo, hi = off, off + n
            else:
                if off + n != lo and off != hi:
                    raise ValueError("gradient buckets are not adjacent in the arena")
                lo, hi = min(lo, off), max(hi, off + n)
            if (hi - lo) * elem_bytes >= min_mb * (1 << 20) or i == len(self.ranges) - 1:
                self.groups.append((lo, hi, i))
                lo = hi = None

    def reduce(self, flat: torch.Tensor, events=None, comm_stream=None, group=None):
        """Starts the all-reduces (sum) of every bucket of `flat`; returns the work handles (wait() on each before
        the gradients are read).  With `events` (one torch.cuda.Event per range) bucket i is reduced on
        `comm_stream` as soon as events[last range of i] has fired on the compute stream."""
        works = []
        for lo, hi, last in self.groups:
            piece = flat[lo:hi]
            if events is not None:
                comm_stream.wait_event(events[last])
                with torch.cuda.stream(comm_stream):
                    works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group, async_op=True))
            else:
                works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group, async_op=True))
        return works


def sync_grads(model) -> None:
    """Makes `model.arena.grad` the SUM over ranks: a no-op when the backward already reduced it bucket by bucket
    (`model.grad_sync == "overlap"`), else the flat all-reduce above.  Callers then step with grad_scale=1/world."""
    if getattr(model, "_grads_reduced", False):
        model._grads_reduced = False
        return
    allreduce_grads(model.arena.grad)
