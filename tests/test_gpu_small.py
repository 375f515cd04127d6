"""GPU (-m gpu): the small-batch fp32 route (csrc/small.hpp: fewer than 2048 token rows per forward -- the reference's own
regime, batch 4 x 224x224 and the single-image worker call) -- each kernel through the C ABI against the oracle / fp64 on
seeded inputs, every tile variant, and the property the route is built around: an output's bits do not depend on M."""
import pytest
import torch

from oracle import vitseg_oracle as O
from visiontransformer_amd import _lib, synth
from visiontransformer_amd.config import ViTSegConfig
from visiontransformer_amd.model import ViTSegmentationModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NVARIANTS = 5       # tile variants of gemm_f32s_kernel (option small_variant 1..5)
NKW = 2             # + the one-image kernel's two tile widths (6, 7; taken where they apply, else the planner's choice)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def _linear_small(A, W, bias, epi):
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_linear_f32_small(A.data_ptr(), W.data_ptr(), bias.data_ptr(), C.data_ptr(), M, N, K, epi,
                                                     _stream()))
    return C


# (M, N, K): the QKV / fc1 shapes of the reference's widths at 1, 4 and 8 images of 224x224 (197 tokens), a ragged N (the
# Tiny/16 QKV: 576 = 4.5 x 128), one row, and a K of a single step
@pytest.mark.parametrize("M,N,K", [(197, 2304, 768), (788, 3072, 768), (1576, 1536, 512), (788, 576, 192), (1, 3072, 1024),
                                   (33, 100, 32), (785, 2304, 768)])
@pytest.mark.parametrize("epi", [0, 1])
def test_linear_small_direct_epilogues(M, N, K, epi):
    """C = A W^T + bias (and exact GELU) against fp64, identical bits from every tile variant and from the one-image kernel
    (the K pieces on four waves instead of one after the other)."""
    A, W, bias = _rand(M, K, seed=M).to(DEV), _rand(N, K, seed=N + 1, scale=0.05).to(DEV), _rand(N, seed=7, scale=0.1).to(DEV)
    ref = A.double() @ W.double().T + bias.double()
    if epi == 1:
        ref = O.gelu_erf(ref)
    outs = []
    for v in [0] + list(range(1, NVARIANTS + NKW + 1)):
        with _lib.option("small_variant", v):
            outs.append(_linear_small(A, W, bias, epi))
    assert (outs[0].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def _resln(A, W, bias, X, lnw, lnb, eps=1e-12):
    M, K = A.shape
    N = W.shape[0]
    S = _lib.lib().vitseg_small_splits(N, K)
    scratch = torch.full((S * M * N,), float("nan"), device=DEV)
    Xo, H = X.clone(), torch.full((M, N), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_linear_resln_f32_small(A.data_ptr(), W.data_ptr(), bias.data_ptr(), Xo.data_ptr(),
                                                           lnw.data_ptr(), lnb.data_ptr(), H.data_ptr(), scratch.data_ptr(),
                                                           scratch.numel(), M, N, K, eps, _stream()))
    return Xo, H, S


# o_proj / fc2 of the reference's three widths (chunked reductions: 3, 6, 2, 6, 4, 6 chunks) and the Tiny/16 o_proj (one)
@pytest.mark.parametrize("M,N,K", [(197, 768, 768), (788, 768, 3072), (788, 512, 512), (197, 512, 3072), (394, 1024, 1024),
                                   (788, 1024, 3072), (788, 192, 192)])
def test_linear_resln_small(M, N, K):
    """X += A W^T + bias; H = LayerNorm(X): fp64 reference, every tile variant identical, chunk count a function of (N, K)."""
    A, W, bias = _rand(M, K, seed=M + 3).to(DEV), _rand(N, K, seed=N + 5, scale=0.05).to(DEV), _rand(N, seed=9, scale=0.1).to(DEV)
    X, lnw, lnb = _rand(M, N, seed=13).to(DEV), (_rand(N, seed=15) * 0.1 + 1.0).to(DEV), _rand(N, seed=17, scale=0.1).to(DEV)
    xr = X.double() + (A.double() @ W.double().T + bias.double())
    hr = O.layer_norm(xr.cpu(), lnw.double().cpu(), lnb.double().cpu(), 1e-12)
    Xo, H, S = _resln(A, W, bias, X, lnw, lnb)
    assert S == {(768, 768): 3, (768, 3072): 6, (512, 512): 2, (512, 3072): 6, (1024, 1024): 4, (1024, 3072): 6, (192, 192): 1}[(N, K)]
    assert (Xo.double() - xr).abs().max().item() < 3e-5 * max(1.0, xr.abs().max().item())
    assert (H.double().cpu() - hr).abs().max().item() < 5e-5
    for v in range(1, NVARIANTS + 1):
        with _lib.option("small_variant", v):
            Xv, Hv, _ = _resln(A, W, bias, X, lnw, lnb)
        assert torch.equal(Xv, Xo) and torch.equal(Hv, H), v


@pytest.mark.parametrize("N,K,epi", [(2304, 768, 0), (3072, 768, 1), (768, 3072, 2), (768, 768, 2)])
def test_linear_small_rows_do_not_depend_on_the_batch(N, K, epi):
    """Rows 0..196 of an 8-image batch (1576 rows) = the same rows run alone (197 rows) and inside 4 images (788 rows), bit
    for bit -- tiles differ (small_plan picks per M), the summation order does not."""
    W, bias = _rand(N, K, seed=N + 1, scale=0.05).to(DEV), _rand(N, seed=7, scale=0.1).to(DEV)
    A = _rand(1576, K, seed=1).to(DEV)
    lnw, lnb = (_rand(N, seed=15) * 0.1 + 1.0).to(DEV), _rand(N, seed=17, scale=0.1).to(DEV)
    X = _rand(1576, N, seed=13).to(DEV)
    outs = []
    for M in (1576, 788, 197):
        if epi == 2:
            Xo, H, _ = _resln(A[:M].contiguous(), W, bias, X[:M].contiguous(), lnw, lnb)
            outs.append(torch.cat([Xo[:197], H[:197]]))
        else:
            outs.append(_linear_small(A[:M].contiguous(), W, bias, epi)[:197])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


# activation gradients of the four linears of ViT-B/16 at the reference's training batch (788 rows) and of Tiny/16
@pytest.mark.parametrize("M,Nd,Kd,epi", [(788, 768, 768, 0), (788, 2304, 768, 0), (788, 3072, 768, 0), (788, 768, 3072, 5),
                                         (197, 192, 3072, 5), (788, 576, 192, 0), (33, 192, 192, 0)])
def test_dgrad_small_t_form(M, Nd, Kd, epi):
    """dX = dY . W (W[Nd, Kd] as nn.Linear stores it; optionally x gelu'(R)) against fp64, identical from every tile variant."""
    dY, W = _rand(M, Nd, seed=M + 1).to(DEV), _rand(Nd, Kd, seed=Nd + 2, scale=0.05).to(DEV)
    R = _rand(M, Kd, seed=5, scale=1.5).to(DEV)
    ref = dY.double() @ W.double()
    if epi == 5:
        u = R.double()
        ref = ref * (0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5)
    S = _lib.lib().vitseg_small_splits(Kd, Nd)
    outs = []
    for v in range(0, NVARIANTS + 1):
        scratch = torch.full((max(S, 1) * M * Kd,), float("nan"), device=DEV)
        dX = torch.full((M, Kd), float("nan"), device=DEV)
        with _lib.option("small_variant", v):
            _lib.check(_lib.lib().vitseg_op_dgrad_f32_small(dY.data_ptr(), W.data_ptr(), R.data_ptr(), dX.data_ptr(), scratch.data_ptr(),
                                                            scratch.numel(), M, Nd, Kd, epi, _stream()))
        outs.append(dX)
    assert (outs[0].double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("M,Nd,Kd", [(788, 768, 3072), (788, 3072, 768), (392, 256, 6912), (784, 192, 768), (33, 64, 100), (197, 2304, 768)])
def test_wgrad_small_both_operands_token_major(M, Nd, Kd):
    """dW = dY^T X with the token rows as the reduction (a ragged last 32-row step) against fp64, every tile variant."""
    dY, X = _rand(M, Nd, seed=3).to(DEV), _rand(M, Kd, seed=4).to(DEV)
    ref = dY.double().T @ X.double()
    outs = []
    for v in range(0, NVARIANTS + 1):
        dW = torch.full((Nd, Kd), float("nan"), device=DEV)
        with _lib.option("small_variant", v):
            _lib.check(_lib.lib().vitseg_op_wgrad_f32_small(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), M, Nd, Kd, _stream()))
        outs.append(dW)
    assert (outs[0].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("f16", [0, 1])
@pytest.mark.parametrize("M,N,K,epi", [(788, 2304, 768, 0), (788, 3072, 768, 1), (788, 768, 768, 2), (788, 768, 3072, 2),
                                       (197, 2304, 768, 0), (197, 3072, 768, 1), (197, 768, 3072, 2), (74, 576, 192, 0),
                                       (1576, 1536, 512, 0), (3140, 3072, 1024, 1), (50, 1024, 1024, 2), (300, 192, 192, 2)])
def test_linear_small_16bit_operands(M, N, K, epi, f16):
    """The 16-bit form of the small-batch linears (the same kernels on v_mfma_f32_32x32x16_bf16 / _f16): bias -> fp32,
    bias + GELU -> 16-bit (the next GEMM's operand), chunk slabs -> fp32 -- against fp64 on the SAME 16-bit operand values;
    rows do not depend on M (every tile variant and the one-image kernel give the same bits)."""
    dt = torch.float16 if f16 else torch.bfloat16
    A = (_rand(M, K, seed=1) * 0.7).to(dt).to(DEV)
    W = (_rand(N, K, seed=2) * 0.05).to(dt).to(DEV)
    bias = _rand(N, seed=3).to(DEV)
    ref = A.double() @ W.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)

    def run(a, rows):
        C = torch.full((rows, N), float("nan"), device=DEV, dtype=dt if epi == 1 else torch.float32)
        S = _lib.lib().vitseg_small_splits(N, K)
        scratch = torch.empty((S + 1) * rows * N, device=DEV) if epi == 2 else None
        _lib.check(_lib.lib().vitseg_op_linear_h16_small(a.data_ptr(), W.data_ptr(), bias.data_ptr(), C.data_ptr(), rows, N, K, epi, f16,
                                                         scratch.data_ptr() if epi == 2 else None, scratch.numel() if epi == 2 else 0,
                                                         _stream()))
        return C

    outs = []
    for v in range(0, 8):   # the planner's choice, then every tile variant and the one-image kernels (skipped where they do not apply)
        with _lib.option("small_variant", v):
            outs.append(run(A, M))
    tol = 2e-5 * max(1.0, ref.abs().max().item()) if epi != 1 else (1e-3 if f16 else 8e-3) * max(1.0, ref.abs().max().item())
    assert (outs[0].double() - ref).abs().max().item() < tol
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    few = run(A[:5].contiguous(), 5)
    assert torch.equal(few, outs[0][:5])


def _attention_ref(qkv, B, Np, A):
    """fp64 softmax(q k^T / 8) v on the patches-first row layout (patch token t of image b in row b Np + t, CLS in row B Np + b)."""
    D = 64 * A
    out = torch.zeros(qkv.shape[0], D, dtype=torch.float64)
    q, k, v = qkv.double().split(D, dim=1)
    for b in range(B):
        rows = list(range(b * Np, (b + 1) * Np)) + [B * Np + b]
        for h in range(A):
            sl = slice(64 * h, 64 * h + 64)
            p = torch.softmax(q[rows, sl] @ k[rows, sl].T * 0.125, dim=-1)
            out[rows, sl] = p @ v[rows, sl]
    return out


@pytest.mark.parametrize("B,Np,A", [(1, 196, 12), (4, 196, 3), (2, 784, 2), (1, 1024, 2), (3, 16, 2), (1, 31, 1), (2, 127, 1)])
def test_attention_small(B, Np, A):
    rows, D = B * Np + B, 64 * A
    qkv = _rand(rows, 3 * D, seed=B * 1000 + Np, scale=1.5).to(DEV)
    ctx = torch.full((rows, D), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_f32_small(qkv.data_ptr(), ctx.data_ptr(), B, Np, A, _stream()))
    ref = _attention_ref(qkv.cpu(), B, Np, A)
    assert (ctx.double().cpu() - ref).abs().max().item() < 2e-5
    # the large-batch kernel on the same input: same function, different cut of the work
    big = torch.empty_like(ctx)
    _lib.check(_lib.lib().vitseg_op_attention_f32(qkv.data_ptr(), big.data_ptr(), B, Np, A, _stream()))
    assert (ctx - big).abs().max().item() < 2e-5


@pytest.mark.parametrize("f16", [0, 1])
@pytest.mark.parametrize("B,Np,A", [(1, 196, 12), (4, 196, 3), (2, 784, 2), (3, 16, 2), (2, 127, 1)])
def test_attention_small_16bit_form(B, Np, A, f16):
    """The key-split attention kernel with its products on the wide MFMA (q, k, P, v rounded to bf16 / fp16 in registers; fp32
    softmax and accumulation), as the 16-bit form of the route runs it: against fp64 attention on the fp32 inputs within the
    format's rounding, and batch-invariant bit for bit."""
    rows, D = B * Np + B, 64 * A
    qkv = _rand(rows, 3 * D, seed=B * 1000 + Np, scale=1.5).to(DEV)
    dt = torch.float16 if f16 else torch.bfloat16
    ctx = torch.full((rows, D), float("nan"), device=DEV, dtype=dt)
    _lib.check(_lib.lib().vitseg_op_attention_h16_small(qkv.data_ptr(), ctx.data_ptr(), B, Np, A, f16, _stream()))
    ref = _attention_ref(qkv.cpu(), B, Np, A)
    err = (ctx.double().cpu() - ref).abs().max().item()
    # |v| reaches ~6 here and the softmax is peaked (scores of std ~2): an output is close to ONE rounded v row, i.e. the budget
    # is a few units of the format's spacing at 6 (bf16: 2^-6 ... 2^-5 = 0.03; fp16: 2^-9 ... 2^-8 = 0.004) plus P's rounding
    assert err < (8e-3 if f16 else 6e-2), err
    one = torch.cat([qkv[(B - 1) * Np:B * Np], qkv[B * Np + B - 1:B * Np + B]]).contiguous()   # the last image alone
    c1 = torch.empty((Np + 1, D), device=DEV, dtype=dt)
    _lib.check(_lib.lib().vitseg_op_attention_h16_small(one.data_ptr(), c1.data_ptr(), 1, Np, A, f16, _stream()))
    assert torch.equal(c1[:Np], ctx[(B - 1) * Np:B * Np]) and torch.equal(c1[Np], ctx[B * Np + B - 1])


def test_attention_small_is_batch_invariant():
    """Image 1 of a batch of 4 = that image alone (its rows re-packed into the patches-first layout of a batch of 1)."""
    B, Np, A = 4, 196, 12
    D = 64 * A
    qkv = _rand(B * Np + B, 3 * D, seed=5, scale=1.5).to(DEV)
    ctx = torch.empty((B * Np + B, D), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_f32_small(qkv.data_ptr(), ctx.data_ptr(), B, Np, A, _stream()))
    one = torch.cat([qkv[Np:2 * Np], qkv[B * Np + 1:B * Np + 2]]).contiguous()
    c1 = torch.empty((Np + 1, D), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_f32_small(one.data_ptr(), c1.data_ptr(), 1, Np, A, _stream()))
    assert torch.equal(c1[:Np], ctx[Np:2 * Np]) and torch.equal(c1[Np], ctx[B * Np + 1])


@pytest.mark.parametrize("rows,D,splits,p", [(788, 768, 6, 0.1), (788, 768, 1, 0.0), (197, 192, 3, 0.25), (3140, 512, 6, 0.0),
                                             (50, 1024, 6, 0.1), (784, 768, 1, 0.1)])
def test_layernorm_backward_small_slabs_and_branch_outputs(rows, D, splits, p):
    """LayerNorm backward as the fp32 training step of the small-batch route runs it (backward.hip:layernorm_bwd_small_kernel):
    same bits as vitseg_op_layernorm_bwd_f32 on the chunk-order sum of the slabs; the next branch's dropped gradient equals
    the counter-based mask (tests/dropout_ref.py) times dres_out, and br_dbias its column sums."""
    from dropout_ref import Masks
    x = (_rand(rows, D, seed=1, scale=2.0) + 0.3).to(DEV)
    w = (_rand(D, seed=2) + 1.0).to(DEV)
    slabs = _rand(splits, rows, D, seed=4).to(DEV)
    dres = _rand(rows, D, seed=5).to(DEV)
    g = slabs[0].clone()
    for s_ in range(1, splits):
        g = g + slabs[s_]
    n_scr = _lib.lib().vitseg_op_layernorm_bwd_scratch_floats(rows, D)
    ref_out, ref_dw, ref_db = torch.empty(rows, D, device=DEV), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    scratch = torch.empty(n_scr, device=DEV)
    _lib.check(_lib.lib().vitseg_op_layernorm_bwd_f32(x.data_ptr(), w.data_ptr(), g.data_ptr(), dres.data_ptr(), ref_out.data_ptr(),
                                                      ref_dw.data_ptr(), ref_db.data_ptr(), scratch.data_ptr(), rows, D, 1e-12, _stream()))
    out, dw, db = torch.full((rows, D), float("nan"), device=DEV), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    br = torch.full((rows, D), float("nan"), device=DEV)
    dbias = torch.full((D,), float("nan"), device=DEV)
    seed, layer, site = 0x1234ABCD, 5, 2
    _lib.check(_lib.lib().vitseg_op_layernorm_bwd_f32_small(
        x.data_ptr(), w.data_ptr(), slabs.data_ptr(), rows * D, splits, dres.data_ptr(), out.data_ptr(), dw.data_ptr(), db.data_ptr(),
        scratch.data_ptr(), rows, D, 1e-12, br.data_ptr() if p else None, dbias.data_ptr(), p, seed, layer * 8 + site, _stream()))
    assert torch.equal(out, ref_out) and torch.equal(dw, ref_dw) and torch.equal(db, ref_db)
    if p:
        # Masks.rows works in the reference's [B, N, D] layout; one "image" of rows - 1 patches + CLS last = rows in kernel order
        mk = Masks(p, seed, 1, rows - 1, 1)
        m = mk.rows(layer, site, (1, rows, D))[0]                  # reference order: CLS (kernel row rows - 1) first
        m = torch.cat([m[1:], m[:1]]).to(DEV)
        assert torch.equal(br, out * m)
        want = br.double().sum(0)
    else:
        want = out.double().sum(0)
    assert (dbias.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    # without the branch outputs: nothing else changes
    out2 = torch.empty_like(out)
    _lib.check(_lib.lib().vitseg_op_layernorm_bwd_f32_small(
        x.data_ptr(), w.data_ptr(), slabs.data_ptr(), rows * D, splits, None, out2.data_ptr(), dw.data_ptr(), db.data_ptr(),
        scratch.data_ptr(), rows, D, 1e-12, None, None, 0.0, 0, 0, _stream()))
    assert torch.equal(dw, ref_dw) and torch.equal(db, ref_db) and (out2 - (ref_out - dres)).abs().max().item() < 1e-5


def _attention_small_fwd_bwd(qkv, dctx, B, Np, A, p=0.0, seed=0xBEEF1234, stream_id=3 * 8 + 1):
    Mt, D, N = B * Np + B, 64 * A, Np + 1
    ctx = torch.full((Mt, D), float("nan"), device=DEV)
    lse = torch.full((B * A * N,), float("nan"), device=DEV)
    dqkv = torch.full((Mt, 3 * D), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_bwd_f32_small(qkv.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), lse.data_ptr(),
                                                            dqkv.data_ptr(), B, Np, A, p, seed, stream_id, _stream()))
    return ctx, lse, dqkv


@pytest.mark.parametrize("B,Np,A,p", [(4, 196, 12, 0.0), (1, 196, 3, 0.0), (2, 49, 2, 0.0), (1, 30, 1, 0.0), (2, 399, 2, 0.0),
                                      (3, 100, 2, 0.0), (2, 196, 3, 0.1), (1, 127, 2, 0.25)])
def test_attention_small_forward_and_backward_for_training(B, Np, A, p):
    """The short-sequence attention pair of the fp32 training step (attention_small.hip with log-sum-exp + dropout,
    attention_bwd_small.hip: one launch, delta inside) against fp64 autograd of eager_attention_forward
    (modeling_vit.py:164-189) on the same inputs; dropout through the kernels' counter-based masks regenerated in numpy
    (tests/dropout_ref.py) and injected into the reference.  Also against the long-sequence kernel pair (p = 0)."""
    from dropout_ref import Masks
    D, Mt, N = 64 * A, B * Np + B, Np + 1
    qkv = _rand(Mt, 3 * D, seed=Np + A, scale=1.2)
    dctx = _rand(Mt, D, seed=9)
    seed, stream_id = 0xBEEF1234, 3 * 8 + 1
    mask = Masks(p, seed, B, Np, A).attn(3, (B, A, N, N)).double() if p else None   # reference token order (CLS first)
    x = qkv.double().requires_grad_(True)
    ctx_ref = torch.empty(Mt, D, dtype=torch.float64)
    lse_ref = torch.empty(B, A, N, dtype=torch.float64)
    outs = []
    for b in range(B):
        r = torch.cat([torch.tensor([B * Np + b]), torch.arange(b * Np, (b + 1) * Np)])
        q, k, v = [x[r][:, i * D:(i + 1) * D].reshape(N, A, 64).transpose(0, 1) for i in range(3)]
        sc = q @ k.transpose(-1, -2) * 0.125
        s = torch.softmax(sc, dim=-1)
        if mask is not None:
            s = s * mask[b]
        o = (s @ v).transpose(0, 1).reshape(N, D)
        ctx_ref[r] = o.detach()
        l2 = torch.logsumexp(sc.detach(), dim=-1) * 1.4426950408889634          # [A, N] reference order, log2 domain
        lse_ref[b] = torch.cat([l2[:, 1:], l2[:, :1]], dim=1)                    # kernel order: CLS last
        outs.append((o * dctx.double()[r]).sum())
    torch.stack(outs).sum().backward()
    qd, dd = qkv.to(DEV), dctx.to(DEV)
    ctx, lse, dqkv = _attention_small_fwd_bwd(qd, dd, B, Np, A, p, seed, stream_id)
    assert (ctx.double().cpu() - ctx_ref).abs().max().item() < 2e-5
    assert (lse.double().cpu().view(B, A, N) - lse_ref).abs().max().item() < 2e-5
    err = (dqkv.cpu().double() - x.grad).abs().max().item()
    assert err < 5e-5 * max(1.0, x.grad.abs().max().item()), err
    if not p:
        big_ctx, big_lse = torch.empty_like(ctx), torch.empty_like(lse)
        scr = torch.empty(B * A * N, device=DEV)
        big = torch.full((Mt, 3 * D), float("nan"), device=DEV)
        _lib.check(_lib.lib().vitseg_op_attention_bwd_f32(qd.data_ptr(), dd.data_ptr(), big_ctx.data_ptr(), big_lse.data_ptr(),
                                                          scr.data_ptr(), big.data_ptr(), B, Np, A, _stream()))
        assert (big - dqkv).abs().max().item() < 5e-5 * max(1.0, x.grad.abs().max().item())


def test_attention_small_backward_is_batch_invariant_and_reproducible():
    """Image 1 of a batch of 4 = that image alone, bit for bit (the key / query split is a function of N only, the partial sums
    are added in wave order); the same call twice gives the same bits (no atomics).  Dropout on: the masks are keyed by
    (image, head, query, key), so the lone image is run as image 1 of ITS batch by re-using rows, not re-keyed."""
    B, Np, A = 4, 196, 12
    D = 64 * A
    qkv = _rand(B * Np + B, 3 * D, seed=5, scale=1.2).to(DEV)
    dctx = _rand(B * Np + B, D, seed=6).to(DEV)
    c4, l4, g4 = _attention_small_fwd_bwd(qkv, dctx, B, Np, A)
    c4b, l4b, g4b = _attention_small_fwd_bwd(qkv, dctx, B, Np, A)
    assert torch.equal(g4, g4b) and torch.equal(c4, c4b) and torch.equal(l4, l4b)
    one = torch.cat([qkv[Np:2 * Np], qkv[B * Np + 1:B * Np + 2]]).contiguous()
    done = torch.cat([dctx[Np:2 * Np], dctx[B * Np + 1:B * Np + 2]]).contiguous()
    c1, l1, g1 = _attention_small_fwd_bwd(one, done, 1, Np, A)
    assert torch.equal(g1[:Np], g4[Np:2 * Np]) and torch.equal(g1[Np], g4[B * Np + 1])
    assert torch.equal(l1.view(A, Np + 1), l4.view(B, A, Np + 1)[1])
    # with dropout: twice the same bits
    a = _attention_small_fwd_bwd(qkv, dctx, B, Np, A, 0.1)
    b = _attention_small_fwd_bwd(qkv, dctx, B, Np, A, 0.1)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and not torch.equal(a[2], g4)


@pytest.mark.parametrize("P,D,L,A", [(16, 768, 2, 12), (16, 512, 2, 8), (8, 512, 1, 8), (16, 1024, 2, 16)])
def test_small_route_equals_large_route_within_rounding_and_oracle(P, D, L, A):
    """The whole forward at 2 x 224x224, 17 classes: small-batch route against the oracle (the fp32 gate: logits within 1e-3,
    masks identical wherever the measured logit error cannot flip them) and against the large-batch kernels on the same
    input (option no_small), which compute the same function with another summation order."""
    cfg = ViTSegConfig(17, P, D, L, A, image_size=224)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=3).items()}
    x = torch.from_numpy(synth.make_images(cfg, 2, seed=1))
    m = ViTSegmentationModel(17, P, D, L, A, image_size=224, device=DEV).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        mask, logits = m.predict_mask(x.to(DEV), return_logits=True)
        with _lib.option("no_small", 1):
            mask_l, logits_l = m.predict_mask(x.to(DEV), return_logits=True)
        ref = O.forward(x, sd, cfg)
    err = (logits.cpu() - ref).abs().max().item()
    assert err < 1e-3 and (logits - logits_l).abs().max().item() < 1e-3, err
    assert err < 2e-5, err   # measured: a few 1e-6
    stable = O.mask_stable(ref, 2.0 * err + 1e-7)
    assert int(((mask.cpu().long() != O.predict_mask(ref)) & stable).sum()) == 0 and float((~stable).float().mean()) < 2e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_small_route_batch_invariance_vit_base(precision):
    """ViT-B/16 (2 layers) at 224x224: images 1..2 of a batch of 8 (1576 rows) = the same two images as a batch of 2, and image
    0 = a batch of 1, bit for bit (logits and masks) -- in fp32 and in the route's 16-bit forms."""
    cfg = ViTSegConfig(17, 16, 768, 2, 12, image_size=224)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=3).items()}
    x = torch.from_numpy(synth.make_images(cfg, 8, seed=2)).to(DEV)
    m = ViTSegmentationModel(17, 16, 768, 2, 12, image_size=224, precision=precision, device=DEV).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        m8, l8 = m.predict_mask(x, return_logits=True)
        m2, l2 = m.predict_mask(x[1:3], return_logits=True)
        m1, l1 = m.predict_mask(x[:1], return_logits=True)
    assert torch.equal(l8[1:3], l2) and torch.equal(m8[1:3], m2)
    assert torch.equal(l8[:1], l1) and torch.equal(m8[:1], m1)
