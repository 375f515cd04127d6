"""GPU (-m gpu): each HIP kernel, called through the C ABI, against the oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import vitseg_oracle as O
from visiontransformer_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("rows,D", [(7, 192), (1025, 768), (33, 1024), (5, 2048), (9, 512)])
def test_layernorm(rows, D):
    x, w, b = _rand(rows, D, seed=1, scale=3.0) + 0.5, _rand(D, seed=2) + 1.0, _rand(D, seed=3)
    ref = O.layer_norm(x.double(), w.double(), b.double(), 1e-12)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.empty_like(xd)
    _lib.check(_lib.lib().vitseg_op_layernorm_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, D,
                                                  1e-12, _stream()))
    assert (y.cpu().double() - ref).abs().max().item() < 5e-6  # tolerance: fp32 rounding of O(1) outputs


@pytest.mark.parametrize("M,N,K,epi", [
    (128, 128, 32, 0), (257, 192, 64, 0), (788, 576, 192, 0), (300, 768, 768, 1), (1025, 768, 3072, 2),
    (130, 3072, 768, 1), (64, 256, 6912, 3), (33, 100, 48, 0), (2050, 2304, 768, 0)])
def test_linear_epilogues(M, N, K, epi):
    A, W, bias = _rand(M, K, seed=M), _rand(N, K, seed=N + 1, scale=0.05), _rand(N, seed=7, scale=0.1)
    R = _rand(M, N, seed=11)
    acc = A.double() @ W.double().T + bias.double()
    if epi == 1:
        ref = O.gelu_erf(acc)
    elif epi == 2:
        ref = R.double() + acc
    elif epi == 3:
        ref = torch.relu(acc)
    else:
        ref = acc
    Ad, Wd, bd, Rd = A.to(DEV), W.to(DEV), bias.to(DEV), R.to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    if epi == 2:  # in-place residual, as the forward uses it
        C.copy_(Rd)
        Rp = C.data_ptr()
    else:
        Rp = None
    _lib.check(_lib.lib().vitseg_op_linear_f32(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), Rp, C.data_ptr(), M, N, K,
                                               epi, _stream()))
    err = (C.cpu().double() - ref).abs().max().item()
    # fp32 fmaf chain over K terms of magnitude ~|a||w|: error ~ 1e-7 * sum|a w|
    bound = 4e-7 * (A.abs().double() @ W.abs().double().T).max().item() + 1e-6
    assert err < bound, (err, bound)


@pytest.mark.parametrize("M,N,K,epi,extra", [
    (32768, 768, 768, 2, ""),            # o_proj at the bench size: 3 tiles per block, in-place residual, inline epilogue
    (16484, 768, 256, 2, "drop"),        # ragged last row tile (masked stores) + hidden dropout (fp32 training forward)
    (12288, 2304, 768, 0, ""),           # QKV: 864 tiles
    (4196, 3072, 768, 1, "aux"),         # fc1: GELU + saved pre-activation, ragged (epilogue at the tile's end)
    (16384, 3072, 768, 1, ""),           # fc1: 1 536 tiles
    (8192, 768, 3072, 2, ""),            # fc2: long K (96 ring steps per tile), one tile per block
    (16384, 256, 96, 3, ""),             # ReLU, the minimum of 3 K steps (no inline epilogue below 4)
    (65536, 256, 128, 3, ""),            # ReLU, 4 K steps, 2 tiles per block
    (33024, 768, 64 * 5, 0, "nobias"),   # uneven tile counts per block, no bias
])
def test_linear_f32_persistent_kernel(M, N, K, epi, extra, monkeypatch):
    """csrc/gemm_f32p.hip (persistent 256x128 kernel of the fp32 linears) against a float64 product, and BITWISE against
    gemm.hip's tile kernel (same k order per dot product): modeling_vit.py:207-254."""
    L = _lib.lib()
    A, W = _rand(M, K, seed=M), _rand(N, K, seed=N + 1, scale=0.05)
    bias = None if "nobias" in extra else _rand(N, seed=7, scale=0.1)
    R = _rand(M, N, seed=11)
    p, seed, stream_id = (0.1, 0x1234ABCD, 13) if "drop" in extra else (0.0, 0, 0)
    acc = A.double() @ W.double().T + (bias.double() if bias is not None else 0.0)
    if epi == 1:
        ref = O.gelu_erf(acc)
    elif epi == 2:
        ref = R.double() + (acc * _drop_rows_np(M, N, p, seed, stream_id).double() if p else acc)
    elif epi == 3:
        ref = torch.relu(acc)
    else:
        ref = acc
    Ad, Wd, Rd = A.to(DEV), W.to(DEV), R.to(DEV)
    bd = bias.to(DEV) if bias is not None else None

    def run():
        C = torch.full((M, N), float("nan"), device=DEV)
        aux = torch.full((M, N), float("nan"), device=DEV) if "aux" in extra else None
        Rp = None
        if epi == 2:  # in-place residual, as the forward uses it
            C.copy_(Rd)
            Rp = C.data_ptr()
        _lib.check(L.vitseg_op_linear_f32_ex(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr() if bd is not None else None, Rp,
                                             C.data_ptr(), aux.data_ptr() if aux is not None else None, M, N, K, epi, p,
                                             seed, stream_id, _stream()))
        torch.cuda.synchronize()
        return C, aux

    C, aux = run()
    with _lib.option("no_f32p", 1):
        C_old, aux_old = run()
    err = (C.cpu().double() - ref).abs().max().item()
    bound = 4e-7 * (A.abs().double() @ W.abs().double().T).max().item() * (1.2 if p else 1.0) + 1e-6
    assert err < bound, (err, bound)
    assert torch.equal(C, C_old), (C - C_old).abs().max().item()
    if aux is not None:
        assert (aux.cpu().double() - acc).abs().max().item() < bound
        assert torch.equal(aux, aux_old)


def test_linear_matches_fp32_fmaf_semantics_exactly_small():
    """A = I (asymmetric W) must come out exactly: catches row/col swaps in the MFMA C layout."""
    M = N = K = 128
    A = torch.eye(M)
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1024.0
    C = torch.empty(M, N, device=DEV)
    Ad, Wd = A.to(DEV), W.to(DEV)  # keep the device tensors alive across the call
    _lib.check(_lib.lib().vitseg_op_linear_f32(Ad.data_ptr(), Wd.data_ptr(), None, None, C.data_ptr(),
                                               M, N, K, 0, _stream()))
    assert torch.equal(C.cpu(), W.T.contiguous())


@pytest.mark.parametrize("B,Np,A", [(2, 196, 3), (1, 1024, 2), (3, 784, 1), (1, 64, 1), (2, 128, 12), (1, 200, 2)])
def test_attention(B, Np, A):
    D = 64 * A
    Mt = B * Np + B
    qkv = _rand(Mt, 3 * D, seed=Np + A, scale=1.5)
    # spike one key against one query so the running max jumps mid-sequence (online-softmax rescale path)
    qkv[Np // 2, :64] *= 6.0
    qkv[(Np * 3) // 4, D:D + 64] = qkv[Np // 2, :64]

    def rows(b):  # oracle order: CLS first
        return torch.cat([torch.tensor([B * Np + b]), torch.arange(b * Np, (b + 1) * Np)])

    ref = torch.empty(Mt, D, dtype=torch.float64)
    x64 = qkv.double()
    for b in range(B):
        r = rows(b)
        q, k, v = [x64[r][:, i * D:(i + 1) * D].reshape(Np + 1, A, 64).transpose(0, 1) for i in range(3)]
        s = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
        ref[r] = (s @ v).transpose(0, 1).reshape(Np + 1, D)
    qd = qkv.to(DEV)
    ctx = torch.full((Mt, D), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_f32(qd.data_ptr(), ctx.data_ptr(), B, Np, A, _stream()))
    err = (ctx.cpu().double() - ref).abs().max().item()
    assert err < 2e-5, err  # fp32 exp2/softmax on O(1) values


@pytest.mark.parametrize("B,C,g,S", [(2, 2, 14, 224), (1, 17, 14, 224), (1, 2, 32, 512), (1, 3, 28, 224), (1, 1, 14, 112)])
def test_upsample_sigmoid_argmax_bit_exact(B, C, g, S):
    z = _rand(B, C, g, g, seed=g + C, scale=2.0)
    if C >= 3:
        z[:, 1] += 18.0  # saturate two classes: sigmoid -> 1.0f for both, first index must win
        z[:, 2] += 19.0
    ref_logits = O.upsample_bilinear(z, (S, S))
    assert torch.equal(ref_logits, torch.nn.functional.interpolate(z, size=(S, S), mode="bilinear", align_corners=False))
    ref_mask = O.predict_mask(ref_logits)
    zd = z.to(DEV)
    logits = torch.empty(B, C, S, S, device=DEV)
    mask = torch.empty(B, S, S, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_argmax(zd.data_ptr(), logits.data_ptr(), mask.data_ptr(), B, C, g, S,
                                                    _stream()))
    assert torch.equal(logits.cpu(), ref_logits)  # bit-exact: same fma placement as ATen's CPU kernel
    # every pixel, ties and saturated classes included: the kernel restates ATen's fp32 sigmoid (Sleef u10 exp, exact
    # add and divide) operation by operation, as the oracle does (pinned against torch.sigmoid on the CPU)
    assert torch.equal(mask.cpu().long(), ref_mask)
    # mask-only call (logits pointer NULL) gives the same mask
    mask2 = torch.empty_like(mask)
    _lib.check(_lib.lib().vitseg_op_upsample_argmax(zd.data_ptr(), None, mask2.data_ptr(), B, C, g, S, _stream()))
    assert torch.equal(mask, mask2)


@pytest.mark.parametrize("B,C,g,S", [(2, 2, 32, 512), (1, 17, 14, 224), (1, 3, 16, 64), (2, 2, 64, 1024), (1, 1, 12, 96),
                                     (1, 2, 9, 2304)])
def test_upsample_backward_is_the_adjoint(B, C, g, S):
    """grad of F.interpolate(z, (S, S), bilinear, align_corners=False) w.r.t. z (what autograd hands the reference's seg head,
    model/CE/classes.py:260) in fp64 on the CPU against the HIP kernel: one block per row of cells up to S = 2048, the
    one-wave-per-cell form beyond (last case); patch sizes 4 ... 256; twice the same bits (no atomics)."""
    gl = _rand(B, C, S, S, seed=S + g + C)
    z = torch.zeros(B, C, g, g, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.interpolate(z, size=(S, S), mode="bilinear", align_corners=False).backward(gl.double())
    gd = gl.to(DEV)
    out = torch.full((B, C, g, g), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_bwd(gd.data_ptr(), out.data_ptr(), B, C, g, S, _stream()))
    out2 = torch.full((B, C, g, g), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_bwd(gd.data_ptr(), out2.data_ptr(), B, C, g, S, _stream()))
    ref = z.grad
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * max(1.0, ref.abs().max().item()) * (S // g), (err, ref.abs().max().item())   # fp32 sums of (S/g)^2 ... 4 (S/g)^2 terms
    assert torch.equal(out, out2)


def test_upsample_backward_adjoint_identity_full_size():
    """BASELINE configs[2] size (B = 64, C = 2, 32 x 32 -> 512 x 512), where the fp64 autograd reference would take a while:
    <upsample(z), G> = <z, upsample_bwd(G)> with the HIP forward kernel (itself bit-exact against ATen) on the left."""
    B, C, g, S = 64, 2, 32, 512
    gen = torch.Generator(device="cpu").manual_seed(11)
    z = torch.randn(B, C, g, g, generator=gen).to(DEV)
    G = torch.randn(B, C, S, S, generator=gen).to(DEV)
    up = torch.empty(B, C, S, S, device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_argmax(z.data_ptr(), up.data_ptr(), None, B, C, g, S, _stream()))
    dz = torch.empty(B, C, g, g, device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_bwd(G.data_ptr(), dz.data_ptr(), B, C, g, S, _stream()))
    lhs = (up.double() * G.double()).sum().item()
    rhs = (z.double() * dz.double()).sum().item()
    scale = (up.double() * G.double()).abs().sum().item()
    assert abs(lhs - rhs) <= 1e-6 * scale, (lhs, rhs, scale)   # fp32 rounding of 33 M products, not a structural difference


@pytest.mark.parametrize("scale,delta,S,g", [(0.3, 1.0, 512, 32), (1.5, 1e-4, 512, 32), (6.0, 3e-3, 512, 32), (20.0, 1.0, 512, 32),
                                             (0.3, 0.0, 512, 32), (3.0, 1e-6, 1024, 64), (1.0, 0.5, 256, 16)])
def test_upsample_mask_only_two_classes(scale, delta, S, g):
    """The two-class mask-only kernel (upsample_mask2_kernel: the sign of the interpolated class DIFFERENCE wherever it
    clears the margin, the exact evaluation elsewhere) against the reference post-processing (bilinear -> sigmoid ->
    first-max argmax, testViTModel.py:122-126) on EVERY pixel: well separated classes, differences at and below the
    decision margins (every pixel ambiguous), exact ties (class 0 must win), logits beyond the range where any margin
    settles the fp32 sigmoid comparison -- and against the general kernel."""
    B = 3
    z0 = _rand(B, 1, g, g, seed=S + g, scale=scale)
    z = torch.cat([z0, z0 + delta * _rand(B, 1, g, g, seed=7)], dim=1).contiguous()
    ref_mask = O.predict_mask(O.upsample_bilinear(z, (S, S)))
    zd = z.to(DEV)
    m_fast = torch.full((B, S, S), 7, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().vitseg_op_upsample_argmax(zd.data_ptr(), None, m_fast.data_ptr(), B, 2, g, S, _stream()))
    with _lib.option("no_mask2", 1):
        m_gen = torch.full((B, S, S), 7, dtype=torch.uint8, device=DEV)
        _lib.check(_lib.lib().vitseg_op_upsample_argmax(zd.data_ptr(), None, m_gen.data_ptr(), B, 2, g, S, _stream()))
    assert torch.equal(m_fast.cpu().long(), ref_mask)
    assert torch.equal(m_fast, m_gen)
    if delta == 0.0:
        assert int(m_fast.sum()) == 0


# ---------------------------------------------------------------- 16-bit operand kernels (bf16 and IEEE half)
FMT = {"bf16": (torch.bfloat16, 2 ** -8, "vitseg_op_linear_bf16", "vitseg_op_attention_bf16"),
       "fp16": (torch.float16, 2 ** -11, "vitseg_op_linear_f16", "vitseg_op_attention_f16")}


@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 64, 0), (257, 192, 128, 0), (788, 576, 192, 1), (1025, 768, 3072, 2),
                                       (2050, 2304, 768, 0), (33, 96, 64, 2), (130, 3072, 768, 1),
                                       (4129, 768, 768, 0), (4608, 384, 192, 1), (5000, 200, 3072, 2), (8224, 2304, 768, 0),
                                       (16584, 2304, 768, 0),   # 585 tiles: three per block of gemm_h16p.hip, ragged last row tile
                                       (66048, 768, 512, 0)])   # 774 tiles of a short reduction (8 K steps per tile)
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_linear_bf16(M, N, K, epi, fmt):
    dt, ulp, fn, _ = FMT[fmt]
    A, W = _rand(M, K, seed=M).to(dt).float(), _rand(N, K, seed=N + 1, scale=0.05).to(dt).float()
    bias, R = _rand(N, seed=7, scale=0.1), _rand(M, N, seed=11)
    acc = A.double() @ W.double().T + bias.double()  # exact products of bf16 values, fp32-accumulated on device
    ref = O.gelu_erf(acc) if epi == 1 else (R.double() + acc if epi == 2 else acc)
    Ad, Wd = A.to(DEV).to(dt), W.to(DEV).to(dt)
    bd, Rd = bias.to(DEV), R.to(DEV)
    if epi == 2:
        C = Rd.clone()
        Rp = C.data_ptr()
    else:
        C = torch.zeros(M, N, device=DEV, dtype=dt)
        Rp = None
    _lib.check(getattr(_lib.lib(), fn)(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), Rp, C.data_ptr(), M, N, K, epi,
                                       _stream()))
    got = C.float().cpu().double()
    scale = (A.abs().double() @ W.abs().double().T).max().item()
    if epi == 2:   # fp32 output: only fp32 accumulation error
        assert (got - ref).abs().max().item() < 4e-7 * scale + 1e-5
    else:          # 16-bit output: one final rounding (2^-9 relative for bf16, 2^-12 for half)
        assert ((got - ref).abs() <= ulp * ref.abs() + 4e-7 * scale + 1e-6).all()


@pytest.mark.parametrize("B,Np,A", [(2, 196, 3), (1, 1024, 2), (3, 784, 1), (1, 64, 1), (2, 128, 12), (1, 200, 2)])
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_attention_bf16(B, Np, A, fmt):
    dt, ulp, _, fn = FMT[fmt]
    D = 64 * A
    Mt = B * Np + B
    qkv = _rand(Mt, 3 * D, seed=Np + A, scale=1.5).to(dt).float()
    qkv[Np // 2, :64] = (qkv[Np // 2, :64] * 6.0).to(dt).float()
    qkv[(Np * 3) // 4, D:D + 64] = qkv[Np // 2, :64]
    ref = torch.empty(Mt, D, dtype=torch.float64)
    x64 = qkv.double()
    for b in range(B):
        r = torch.cat([torch.tensor([B * Np + b]), torch.arange(b * Np, (b + 1) * Np)])
        q, k, v = [x64[r][:, i * D:(i + 1) * D].reshape(Np + 1, A, 64).transpose(0, 1) for i in range(3)]
        s = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
        ref[r] = (s @ v).transpose(0, 1).reshape(Np + 1, D)
    qd = qkv.to(DEV).to(dt)
    ctx = torch.zeros(Mt, D, device=DEV, dtype=dt)
    _lib.check(getattr(_lib.lib(), fn)(qd.data_ptr(), ctx.data_ptr(), B, Np, A, _stream()))
    err = (ctx.float().cpu().double() - ref).abs().max().item()
    # P is rounded to the operand format (2^-9 / 2^-12 relative) before P.V and the output is rounded too: |v| <= ~6
    assert err < 4e-2 * ulp * 2 ** 8, err
    assert (ctx.float().cpu().double() - ref).abs().mean().item() < 2e-3 * ulp * 2 ** 8


# ---------------------------------------------------------------- fp32 operands on the fp16 pipe (VITSEG_F32X3)
@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 64, 0), (257, 192, 96, 0), (788, 576, 192, 1), (1025, 768, 3072, 2),
                                       (2050, 2304, 768, 0), (33, 96, 32, 2), (130, 3072, 768, 1)])
def test_linear_f32x3_is_fp32_grade(M, N, K, epi):
    """Split-operand GEMM: a = hi + lo * 2^-11 in half precision, 3 MFMAs per product.  Each product is good to
    ~2^-21 relative (lo.lo' dropped, low parts rounded), so the result must sit within 1e-6 of the fp64 value relative
    to sum |a||w| -- the same budget class as true fp32 accumulation (4e-7), far from half precision (5e-4)."""
    A, W = _rand(M, K, seed=M), _rand(N, K, seed=N + 1, scale=0.05)
    A[0, :4] = torch.tensor([1e-6, -3e-5, 2.5e3, -7.0])        # tiny, subnormal-half and large magnitudes in one row
    bias, R = _rand(N, seed=7, scale=0.1), _rand(M, N, seed=11)
    acc = A.double() @ W.double().T + bias.double()
    ref = O.gelu_erf(acc) if epi == 1 else (R.double() + acc if epi == 2 else acc)
    Ad, Wd, bd, Rd = A.to(DEV), W.to(DEV), bias.to(DEV), R.to(DEV)
    C = Rd.clone() if epi == 2 else torch.zeros(M, N, device=DEV)
    _lib.check(_lib.lib().vitseg_op_linear_f32x3(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), C.data_ptr() if epi == 2 else None,
                                                 C.data_ptr(), M, N, K, epi, _stream()))
    scale = (A.abs().double() @ W.abs().double().T)
    err = ((C.cpu().double() - ref).abs() / (scale + 1e-3)).max().item()
    assert err < 1e-6, err
    # against the true-fp32 kernel: they agree to fp32 accumulation noise (which grows with sqrt(K) in that kernel's
    # sequential 32x32x2 chains; the 16-wide half MFMAs of the split path accumulate fewer, wider steps)
    C32 = Rd.clone() if epi == 2 else torch.zeros(M, N, device=DEV)
    _lib.check(_lib.lib().vitseg_op_linear_f32(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), C32.data_ptr() if epi == 2 else None,
                                               C32.data_ptr(), M, N, K, epi, _stream()))
    assert ((C - C32).abs().cpu().double() / (scale + 1e-3)).max().item() < 1e-5


@pytest.mark.parametrize("B,Np,A", [(2, 196, 3), (1, 1024, 2), (3, 784, 1), (1, 64, 1), (2, 128, 12), (1, 200, 2)])
def test_attention_f32x3_is_fp32_grade(B, Np, A):
    """Split-operand attention (attention_x3.hip): fp32 in / fp32 out, QK^T and PV as 3 half MFMAs per product.
    Same inputs and same tolerance as the exact-fp32 kernel's test."""
    D = 64 * A
    Mt = B * Np + B
    qkv = _rand(Mt, 3 * D, seed=Np + A, scale=1.5)
    qkv[Np // 2, :64] *= 6.0                       # a peaked softmax row
    qkv[(Np * 3) // 4, D:D + 64] = qkv[Np // 2, :64]
    ref = torch.empty(Mt, D, dtype=torch.float64)
    x64 = qkv.double()
    for b in range(B):
        r = torch.cat([torch.tensor([B * Np + b]), torch.arange(b * Np, (b + 1) * Np)])
        q, k, v = [x64[r][:, i * D:(i + 1) * D].reshape(Np + 1, A, 64).transpose(0, 1) for i in range(3)]
        s = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
        ref[r] = (s @ v).transpose(0, 1).reshape(Np + 1, D)
    qd = qkv.to(DEV)
    ctx = torch.full((Mt, D), float("nan"), device=DEV)
    _lib.check(_lib.lib().vitseg_op_attention_f32x3(qd.data_ptr(), ctx.data_ptr(), B, Np, A, _stream()))
    err = (ctx.cpu().double() - ref).abs().max().item()
    assert err < 2e-5, err
    ctx32 = torch.empty_like(ctx)
    _lib.check(_lib.lib().vitseg_op_attention_f32(qd.data_ptr(), ctx32.data_ptr(), B, Np, A, _stream()))
    print(f"attention x3 max err {err:.2e}, exact-fp32 kernel {(ctx32.cpu().double() - ref).abs().max().item():.2e}")


# ---------------------------------------------------------------- every shape-selected tile variant of the 16-bit GEMM
# csrc/gemm.hip picks the tile by shape: 256x256 for wide outputs (M >= 8192, N >= 2048: QKV, fc1 and the dgrad into
# them at the BASELINE batches), 256x128 for long K (M >= 4096, K >= 2048: fc2 and the dgrads with a long reduction),
# 128x128 otherwise, CLS rows optionally through the split-K side launch.  Each variant x epilogue the inference and
# training paths use is compared with the fp64 product here (BASELINE configs[2] runs all of them).
def _gelu_grad64(u):
    return 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * np.pi) ** 0.5


def _drop_rows_np(M, N, p, seed, stream):
    from dropout_ref import Masks
    mk = Masks(p, seed, 1, 1, 1)                       # only the hash is used: seed32 = seed ^ (seed >> 32)
    keep = mk._keep(stream, np.arange(M)[:, None], np.arange(N)[None, :])
    return torch.from_numpy(np.where(keep, mk.scale, np.float32(0)).astype(np.float32))


@pytest.mark.parametrize("M,N,K,epi,extra", [
    (8224, 2304, 768, 0, ""),            # 256x256, bias (QKV), ragged last row tile
    (8200, 3072, 768, 1, "aux"),         # 256x256, GELU + saved pre-activation (fc1 forward, training)
    (8200, 3072, 768, 1, ""),            # 256x256, GELU (fc1 forward, inference)
    (8200, 3072, 768, 5, ""),            # 256x256, dGELU (dgrad through fc2 into the MLP hidden)
    (8192, 2048, 256, 2, ""),            # 256x256, residual epilogue, fp32 out
    (8200, 2304, 768, 0, "thin"),        # 256x256 body + CLS rows by split-K (vitseg_train.hip: QKV forward)
    (4100, 768, 3072, 2, ""),            # 256x128 long-K, residual (fc2 forward)
    (4100, 768, 3072, 2, "drop"),        # ... with hidden dropout fused into the epilogue
    (4100, 768, 3072, 0, "thin"),        # 256x128 body + thin rows (dgrad through fc1, K = 3072)
    (4100, 768, 2304, 0, ""),            # 256x128, bias (dgrad through QKV)
    (4100, 512, 2048, 1, ""),            # 256x128, GELU
    (4100, 512, 2048, 5, ""),            # 256x128, dGELU
    (4100, 768, 768, 0, "thin"),         # 128x128 body + thin rows
    (8200, 768, 3072, 2, "thin drop"),   # persistent body + CLS rows by split-K with the dropout / residual epilogue (fc2, training)
    (8200, 3072, 768, 1, "thin aux"),    # ... with GELU + the saved derivative (fc1, training)
    (8200, 3072, 768, 5, "thin"),        # ... with the dGELU epilogue (dgrad through fc2)
    (8200, 768, 768, 2, "thin"),         # ... residual, no dropout (o_proj, evaluation)
    (1030, 3072, 768, 5, ""),            # 128x128, dGELU
    (1030, 3072, 768, 1, "aux"),         # 128x128, GELU + aux
    (16400, 768, 768, 2, "drop"),        # 128x128, residual + dropout (o_proj forward at training batch)
])
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_linear_h16_tile_variants(M, N, K, epi, extra, fmt, request):
    if fmt == "fp16" and (epi == 5 or "aux" in extra or "drop" in extra):
        pytest.skip("training epilogues exist for bf16 only (fp16 is an inference format)")
    if "thin" in extra:   # a ragged row tile that fits the persistent kernel's last round would ride along instead (gemm.hip)
        old = _lib.get_option("no_ragged_p8")
        request.addfinalizer(lambda: _lib.set_option("no_ragged_p8", old))
        _lib.set_option("no_ragged_p8", 1)
    dt, ulp, _, _ = FMT[fmt]
    A, W = _rand(M, K, seed=M).to(dt).float(), _rand(N, K, seed=N + 1, scale=0.05).to(dt).float()
    bias = _rand(N, seed=7, scale=0.1)
    acc = A.double() @ W.double().T
    scale = float((A.abs().double() @ W.abs().double().T).max())
    Ad, Wd, bd = A.to(DEV).to(dt), W.to(DEV).to(dt), bias.to(DEV)
    thin = M % 256 if "thin" in extra else 0             # the trailing "CLS" rows; the body is whole 256-row tiles
    scratch = torch.empty(16 * 64 * max(N, 3072), device=DEV) if thin else None
    p, seed, stream_id = (0.1, 0x1234ABCD, 13) if "drop" in extra else (0.0, 0, 0)
    aux = torch.zeros(M, N, device=DEV, dtype=dt) if "aux" in extra else None
    cs_out = cs_scr = None
    if epi == 5 and fmt == "bf16":   # every dGELU case also asks for the fused column sums (all dispatch paths)
        cs_out = torch.full((N,), float("nan"), device=DEV)
        cs_scr = torch.empty(_lib.lib().vitseg_op_colsum_scratch_floats(M, N), device=DEV)
    if epi == 2:
        R = _rand(M, N, seed=11)
        C = R.to(DEV)
        Rp = C.data_ptr()                                  # in place, as the forward uses it
        y = acc + bias.double()
        if p:
            y = y * _drop_rows_np(M, N, p, seed, stream_id).double()
        ref = R.double() + y
    elif epi == 5:
        U = _gelu_grad64(_rand(M, N, seed=12).double()).float().to(dt)   # the saved 16-bit gelu'(pre-activation)
        Ud = U.to(DEV)
        Rp = Ud.data_ptr()
        C = torch.zeros(M, N, device=DEV, dtype=dt)
        bd = None
        ref = acc * U.double()
    else:
        C = torch.zeros(M, N, device=DEV, dtype=dt)
        Rp = None
        ref = O.gelu_erf(acc + bias.double()) if epi == 1 else acc + bias.double()
    _lib.check(_lib.lib().vitseg_op_linear_h16_ex(
        Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr() if bd is not None else None, Rp, C.data_ptr(),
        aux.data_ptr() if aux is not None else None, M, N, K, epi, int(fmt == "fp16"), thin,
        scratch.data_ptr() if thin else None, scratch.numel() if thin else 0, p, seed, stream_id,
        cs_out.data_ptr() if cs_out is not None else None, cs_scr.data_ptr() if cs_out is not None else None, _stream()))
    got = C.float().cpu().double()
    if cs_out is not None:   # the bias gradient from the epilogue's per-tile partial sums (unrounded fp32 values), fixed order
        # (the paths without the 8-phase kernel sum the bf16-rounded C instead: independent roundings of 2^-9 relative)
        cref = ref.sum(dim=0)
        tol = 4 * 2.0 ** -9 * (ref ** 2).sum(dim=0).sqrt() + 4e-7 * scale * M ** 0.5
        assert ((cs_out.cpu().double() - cref).abs() <= tol).all()
    if epi == 2:      # fp32 output: fp32 accumulation error only
        assert (got - ref).abs().max().item() < 4e-7 * scale * (1.2 if p else 1.0) + 1e-5
    else:             # one rounding to the 16-bit format
        assert ((got - ref).abs() <= ulp * ref.abs() + 4e-7 * scale + 2e-6).all()
    if aux is not None:   # the forward saves gelu'(pre-activation) for the backward (A&S 7.1.28 erf + exp2: |err| <= 1e-6)
        dref = _gelu_grad64(acc + bias.double())
        assert ((aux.float().cpu().double() - dref).abs() <= ulp * dref.abs() + 4e-7 * scale + 2e-6).all()


@pytest.mark.parametrize("M,N,K", [
    (768, 3072, 8200),        # dW of fc2 at 8 images: 36 tiles of 256x256, sliced over the token rows (8-phase TT kernel)
    (3072, 768, 4100),        # dW of fc1; ragged last 64-token step
    (768, 768, 2050),         # dW of o_proj / patch embedding
    (2304, 768, 1025),        # dW of the fused QKV projection, one image
    (256, 6912, 2048),        # dW of seg_head.0 (im2col columns)
    (192, 576, 788),          # small (Tiny) shapes: the 128x128 TT kernel
    (128, 136, 70),
])
def test_wgrad_bf16_both_operands_token_major(M, N, K):
    """dW = dY^T X with both operands as they lie in memory ([tokens][columns]), no transposed copies: the MFMA
    operands are gathered with transposed LDS reads.  fp64 reference of the same bf16 values."""
    dY = _rand(K, M, seed=K).to(torch.bfloat16)
    X = _rand(K, N, seed=N + 3, scale=0.5).to(torch.bfloat16)
    ref = dY.double().T @ X.double()
    scale = float((dY.abs().double().T @ X.abs().double()).max())
    dYd, Xd = dY.to(DEV), X.to(DEV)
    dW = torch.full((M, N), float("nan"), device=DEV)
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    n = _lib.lib().vitseg_op_wgrad_bf16_scratch_floats(M, N, K)
    scratch = torch.empty(n, device=DEV)
    _lib.check(_lib.lib().vitseg_op_wgrad_bf16(dYd.data_ptr(), Xd.data_ptr(), dW.data_ptr(), scratch.data_ptr(),
                                               zeros.data_ptr(), M, N, K, _stream()))
    assert (dW.cpu().double() - ref).abs().max().item() < 4e-7 * scale + 1e-5
