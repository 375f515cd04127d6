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


def test_grad_bucket_ranges_cover_the_arena_in_backward_order():
    """vitseg_grad_bucket_range (host-only call): head first, layers L-1 .. 0, embeddings last; the ranges tile
    the gradient arena exactly once and neighbours in issue order are adjacent in memory."""
    from visiontransformer_amd import _lib
    from visiontransformer_amd.config import vit_tiny16
    from visiontransformer_amd.dist import BucketReducer
    cfg = vit_tiny16(num_classes=2)
    ranges = _lib.grad_buckets(cfg)
    total = _lib.param_count(cfg)
    assert len(ranges) == cfg.num_hidden_layers + 2
    assert ranges[0][0] + ranges[0][1] == total                       # bucket 0 is the tail of the arena (head)
    assert ranges[-1][0] == 0                                         # the last bucket starts the arena (embeddings)
    for (o0, n0), (o1, n1) in zip(ranges, ranges[1:]):
        assert o1 + n1 == o0                                          # marching towards the front, no gaps
    assert sum(n for _, n in ranges) == total
    for l in range(cfg.num_hidden_layers):                            # layer l sits in bucket L - l, whole
        lo, n = ranges[cfg.num_hidden_layers - l]
        for t in (_lib.T_LN1_W, _lib.T_WQKV, _lib.T_W2, _lib.T_B2):
            off, cnt = _lib.param_offset(cfg, t, l)
            assert lo <= off and off + cnt <= lo + n
    for t in (_lib.T_LNF_W, _lib.T_HEAD0_W, _lib.T_HEAD2_B):
        off, cnt = _lib.param_offset(cfg, t, 0)
        assert ranges[0][0] <= off and off + cnt <= total
    for t in (_lib.T_CLS, _lib.T_POS, _lib.T_PATCH_W, _lib.T_PATCH_B):
        off, cnt = _lib.param_offset(cfg, t, 0)
        assert off + cnt <= ranges[-1][1]
    # merging keeps contiguity and order; a huge threshold gives one message, a tiny one L + 2
    assert len(BucketReducer(ranges, min_mb=1e9).groups) == 1
    assert BucketReducer(ranges, min_mb=1e9).groups[0][:2] == (0, total)
    assert len(BucketReducer(ranges, min_mb=0.0).groups) == len(ranges)
    seen = 0
    for lo, hi, last in BucketReducer(ranges, min_mb=4.0).groups:
        assert hi == total - seen
        seen += hi - lo
    assert seen == total


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd import _lib
    from visiontransformer_amd.config import vit_tiny16
    from visiontransformer_amd.dist import BucketReducer
    cfg = vit_tiny16(num_classes=2)
    total = _lib.param_count(cfg)
    g = (torch.arange(total, dtype=torch.float32) % 1000) * (rank + 1)
    red = BucketReducer(_lib.grad_buckets(cfg), min_mb=4.0)
    for w in red.reduce(g):
        w.wait()
    expect = (torch.arange(total, dtype=torch.float32) % 1000) * sum(r + 1 for r in range(world))
    if rank == 0:
        q.put((bool(torch.equal(g, expect)), len(red.groups)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_gloo_world2():
    """The overlapped exchange step without the GPU: per-bucket asynchronous sums over the backward-order ranges
    give the same arena as one flat all-reduce."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, ngroups = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok and ngroups > 1
