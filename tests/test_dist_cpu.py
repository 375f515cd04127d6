"""CPU, world_size 2 over gloo: the N>1 inference path (batch split + ordered mask gather).
The per-rank compute is stubbed (there is no GPU here); what is tested is the sharding logic the
8-GPU run uses unchanged with backend nccl (= RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visiontransformer_amd.dist import predict_sharded, shard_range


def test_shard_range_partitions():
    for n in (1, 7, 32, 33, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    images = torch.rand(B, 3, 8, 8, generator=g)  # same global batch on every rank

    def fake_predict(x):  # stands in for ViTSegmentationModel.predict_mask
        return (x.sum(dim=1) * 7).to(torch.uint8)

    out = predict_sharded(images, fake_predict)
    if rank == 0:
        q.put(bool(torch.equal(out, fake_predict(images))))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_predict_sharded_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    for B in (5, 4):  # uneven and even split
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
        for p in procs:
            p.start()
        ok = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert ok


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd.dist import allreduce_grads
    n = 1_000_003  # not a multiple of the bucket size
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    allreduce_grads(g, bucket_mb=1.0)  # 4 buckets, issued tail-first
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    if rank == 0:
        q.put(bool(torch.equal(g, expect)))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_grads_gloo_world2():
    """The training exchange step: bucketed sum of the flat gradient arena (backend nccl = RCCL on the GPUs)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok
