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


# ---------------------------------------------------------------- gradient accumulation under data parallelism
class _StubNet(torch.nn.Module):
    """Stands in for ViTSegmentationModel on the CPU: one flat `arena` parameter, the `no_sync()` / `_overlap_active()`
    protocol of visiontransformer_amd/model.py (the real methods are borrowed, not re-implemented)."""
    from visiontransformer_amd.model import ViTSegmentationModel as _M
    no_sync = _M.no_sync
    _overlap_active = _M._overlap_active

    def __init__(self):
        super().__init__()
        self.arena = torch.nn.Parameter(torch.linspace(-1, 1, 64))
        self.grad_sync, self._grads_reduced, self._require_sync = "overlap", False, True


class _StubLightning(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.model = _StubNet()
        self.overlap_seen = []

    def training_step(self, batch, idx):
        x, y = batch
        self.overlap_seen.append(self.model._overlap_active())   # what the real backward would consult
        return ((self.model.arena * x).sum(dim=1) - y).pow(2).mean()

    def validation_step(self, batch, idx):
        return self.training_step(batch, idx).detach()

    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), lr=0.1)

    def state_dict(self, *a, **k):
        return {"model.arena": self.model.arena.detach().clone()}

    def load_state_dict(self, sd, strict=True, assign=False):
        with torch.no_grad():
            self.model.arena.copy_(sd["model.arena"])


def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd import trainer
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)

    dist.all_reduce = counting
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(16, 2, 64, generator=g), torch.randn(16, 2, generator=g)     # 16 micro-batches of 2 samples
    mine = [(X[i], Y[i]) for i in range(rank, 16, world)]                           # 8 per rank
    lm = _StubLightning()
    w0 = lm.model.arena.detach().clone()
    trainer.fit(lm, mine, None, max_epochs=1, accumulate_grad_batches=4, device="cpu")
    grad_calls = [n for n in calls if n == 64]
    # reference: one process, the same 16 micro-batches, 2 optimizer steps of 8 micro-batches each (4 per rank x 2 ranks)
    ref = torch.nn.Parameter(w0.clone())
    opt = torch.optim.SGD([ref], lr=0.1)
    for s in range(2):
        opt.zero_grad()
        for r in range(world):
            for j in range(4):
                i = r + world * (4 * s + j)
                (((ref * X[i]).sum(dim=1) - Y[i]).pow(2).mean() / 4 / world).backward()
        opt.step()
    if rank == 0:
        q.put((len(grad_calls), bool(torch.allclose(lm.model.arena.detach(), ref.detach(), atol=1e-6)),
               lm.overlap_seen))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_reduces_once_per_optimizer_step_gloo_world2():
    """trainer.fit with accumulate_grad_batches=4 on 2 ranks: 8 micro-batches per rank = 2 optimizer steps -> exactly 2
    gradient all-reduces (not 8), micro-batches 1-3 of a step run under no_sync(), and the parameters equal the
    single-process run over the same 16 micro-batches."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n_calls, same, overlap_seen = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_calls == 2 and same
    # the overlapped (in-backward) reduce is never taken while an accumulated gradient is pending or inside no_sync():
    # only a step's first micro-batch could use it, and that one runs under no_sync() when it is not also the last
    assert overlap_seen == [False] * 8


# ---------------------------------------------------------------- trainer.fit end to end on two ranks
def _fit_data():
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 4, 64, generator=g), torch.randn(8, 4, generator=g)       # 8 global batches of 4 samples
    Xv, Yv = torch.randn(2, 4, 64, generator=g), torch.randn(2, 4, generator=g)
    return X, Y, Xv, Yv


def _fit_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd import trainer
    X, Y, Xv, Yv = _fit_data()
    # data parallelism: every global batch of 4 is split into one micro-batch of 2 per rank
    lo, hi = 2 * rank, 2 * rank + 2
    train = [(X[i, lo:hi], Y[i, lo:hi]) for i in range(8)]
    val = [(Xv[i, lo:hi], Yv[i, lo:hi]) for i in range(2)]
    lm = _StubLightning()
    rows = trainer.fit(lm, train, val, max_epochs=2, accumulate_grad_batches=2, device="cpu",
                       ckpt_dir=os.path.join(tmp, "ckpt"), log_dir=os.path.join(tmp, "logs"))
    if rank == 0:
        q.put((lm.model.arena.detach().tolist(), [r for r in rows if "valid_loss" in r]))   # plain lists: no shared storage
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_fit_two_epochs_gloo_world2_equals_single_process(tmp_path):
    """trainer.fit on 2 ranks over gloo -- two epochs, accumulate_grad_batches = 2, validation, checkpoints, metrics.csv --
    against the SAME fit in one process fed the global batches: identical parameters after 8 optimizer steps, identical
    validation losses, one checkpoint per epoch written by rank 0 only (the 8-GPU run changes the backend, not this code)."""
    from visiontransformer_amd import trainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    arena2, val_rows2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    X, Y, Xv, Yv = _fit_data()
    lm = _StubLightning()
    rows1 = trainer.fit(lm, [(X[i], Y[i]) for i in range(8)], [(Xv[i], Yv[i]) for i in range(2)], max_epochs=2,
                        accumulate_grad_batches=2, device="cpu")
    arena2 = torch.tensor(arena2)
    assert torch.allclose(arena2, lm.model.arena.detach(), atol=1e-6), (arena2 - lm.model.arena.detach()).abs().max()
    val_rows1 = [r for r in rows1 if "valid_loss" in r]
    assert len(val_rows2) == len(val_rows1) == 2
    for a, b in zip(val_rows2, val_rows1):
        assert a["step"] == b["step"] and abs(a["valid_loss"] - b["valid_loss"]) < 1e-6 * max(1.0, abs(b["valid_loss"]))
    ck = sorted(os.listdir(tmp_path / "ckpt"))
    assert ck == ["epoch=0-step=4.ckpt", "epoch=1-step=8.ckpt"], ck
    assert (tmp_path / "logs" / "metrics.csv").read_text().startswith("epoch,step,train_loss_step")


# ---------------------------------------------------------------- bench.py --gpus N without a launcher
_STUB_RANK = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
open(os.path.join(sys.argv[1], f"rank{rank}.pid"), "w").write(str(os.getpid()))
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[2]
if mode == "die" and rank == 1:
    time.sleep(0.5)
    sys.exit(3)
if mode == "die":
    time.sleep(600)          # a rank stuck in a collective: the launcher must not wait for it
print(json.dumps({"metric": "stub", "n_gpus": world, "rank": rank}), flush=True)
'''


def _alive(pid):
    try:
        os.kill(pid, 0)
        return True
    except OSError:
        return False


def test_bench_self_launch_starts_n_ranks_and_relays_rank0_once(tmp_path, capfd):
    """bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it) with a stub in place of a rank: N
    children with the rendezvous environment, rank 0's JSON line on stdout exactly once, exit code 0."""
    import json
    import sys
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB_RANK)
    for n in (2, 4):
        d = tmp_path / f"n{n}"
        d.mkdir()
        rc = bench.self_launch(n, child=[sys.executable, str(stub), str(d), "ok"])
        out = capfd.readouterr().out
        assert rc == 0
        assert sorted(os.listdir(d)) == [f"rank{r}.pid" for r in range(n)]
        lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        assert lines == [{"metric": "stub", "n_gpus": n, "rank": 0}], out


def test_bench_self_launch_fails_fast_and_leaves_no_orphan(tmp_path):
    """One rank dies (exit 3) while the others hang: the launcher returns non-zero within seconds and every child is
    gone afterwards."""
    import sys
    import time
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB_RANK)
    t0 = time.time()
    rc = bench.self_launch(4, child=[sys.executable, str(stub), str(tmp_path), "die"])
    assert rc != 0 and time.time() - t0 < 30
    pids = [int((tmp_path / f"rank{r}.pid").read_text()) for r in range(4)]
    assert not any(_alive(p) for p in pids), pids
