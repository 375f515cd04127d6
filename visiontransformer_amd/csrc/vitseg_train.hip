// Training entry points of libvitseg (fp32): forward with saved activations, backward, fused Adam.
// Replaces what autograd + torch.optim.Adam run behind LightningViTModel.training_step /
// configure_optimizers (/root/reference/model/CE/classes.py:276-285, :296-297).  Dropout is not
// applied (p = 0): train-mode bitwise parity with torch's RNG stream is impossible anyway
//  and the parity tests run with dropout off.
#include "kernels.hpp"
#include "profile.hpp"
#include "plan.hpp"

using namespace vitseg;
using namespace vitseg::plan;

namespace {

// ---- training workspace: per-layer saved activations + backward temporaries (all fp32) ----
struct LayerBufs {
    size_t xin, h1, qkv, ctx, lse, xmid, h2, upre, uact;
    size_t dropw;  // bf16 path, Np % 128 == 0: keep-bit words of the attention dropout (written by the forward, read by the backward)
};
struct TrainPlan {
    size_t Mt, Mp;
    size_t layer0, layer_stride;  // per-layer block
    LayerBufs lb;                 // offsets inside a layer block
    size_t xfinal, hf, f, z;      // after the last layer
    size_t dxa, dxb, dh, dqkv, du, dctx, dvec, g, dz, df, t, wd, scratch, wscratch, ce_partial, total;
    size_t wscratch_floats;
    // bf16 training only: bf16 copy of the fp32 residual gradient, transposed weight for dgrad, zero page,
    // fp32 gradient of the final LayerNorm output
    size_t dxc, wt, zero, dhf, Kpad;
    size_t wt_all, wt_layer;  // every layer's four weights transposed (dgrad operands), elements per layer
    size_t dxm;  // fp32 path with dropout: masked copy of the residual gradient that enters a dropped branch
};

TrainPlan make_train_plan(const Shape& s, int B, int precision) {
    TrainPlan p{};
    p.Mp = (size_t)B * s.Np;
    p.Mt = p.Mp + B;
    const bool lp = precision == VITSEG_BF16;
    const size_t act = lp ? 2 : 4;  // bytes per element of the tensors that feed MFMAs
    const size_t MtD = p.Mt * s.D * 4, MtI = p.Mt * (size_t)s.I * 4;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += up(bytes, 256);
        return o;
    };
    // one layer block
    p.lb.xin = take(MtD);
    p.lb.h1 = take(MtD / 4 * act);
    p.lb.qkv = take(3 * MtD / 4 * act);
    p.lb.ctx = take(MtD / 4 * act);
    p.lb.lse = take((size_t)B * s.A * s.N * 4);
    p.lb.xmid = take(MtD);
    p.lb.h2 = take(MtD / 4 * act);
    p.lb.upre = take(MtI / 4 * act);
    p.lb.uact = take(MtI / 4 * act);
    p.lb.dropw = take(lp && s.Np % 128 == 0 ? attn_dropmask_words(B, s.Np, s.A) * 4 : 0);
    p.layer_stride = off;
    p.layer0 = 0;
    off = p.layer_stride * s.L;
    p.xfinal = take(MtD);
    p.hf = take(p.Mp * s.D * act);
    p.f = take(p.Mp * MID * 4);
    p.z = take((size_t)B * s.C * s.Np * 4);
    p.dxa = take(MtD);
    p.dxb = take(MtD);
    p.dh = take(MtD / 4 * act);
    p.dqkv = take(3 * MtD / 4 * act);
    p.du = take(MtI / 4 * act);
    p.dctx = take(MtD / 4 * act);
    p.dvec = take((size_t)B * s.A * s.N * 4);
    p.g = take((size_t)B * s.C * s.S * s.S * 4);
    p.dz = take((size_t)B * s.C * s.Np * 4);
    p.df = take(p.Mp * MID * 4);
    const size_t tcols = (size_t)(9 * s.D > s.Kp ? 9 * s.D : s.Kp);
    p.t = take(p.Mp * tcols * 4);
    p.wd = take((size_t)s.D * 9 * MID * 4);
    size_t sc = colsum_scratch_floats((int)p.Mt, s.I > 3 * s.D ? s.I : 3 * s.D);
    const size_t sc2 = layernorm_bwd_scratch_floats((int)p.Mt, s.D), sc3 = head1x1_bwd_scratch_floats(B, s.Np, s.C);
    sc = sc > sc2 ? sc : sc2;
    sc = sc > sc3 ? sc : sc3;
    p.scratch = take(sc * 4);
    {   // split-K partials of the weight-gradient GEMMs (largest over the shapes the backward uses)
        const int Mt = (int)p.Mt, Mp = (int)p.Mp;
        size_t w = wgrad_scratch_floats(s.D, s.I, Mt);
        auto mx = [&](size_t v) { w = v > w ? v : w; };
        mx(wgrad_scratch_floats(s.I, s.D, Mt));
        mx(wgrad_scratch_floats(s.D, s.D, Mt));
        mx(wgrad_scratch_floats(3 * s.D, s.D, Mt));
        mx(wgrad_scratch_floats(MID, 9 * s.D, Mp));
        mx(wgrad_scratch_floats(s.D, s.Kp, Mp));
        p.wscratch = take(w * 4);
        p.wscratch_floats = w;
    }
    p.ce_partial = take(ce_partial_count(B, s.S) * 8);
    p.dxm = take(MtD);
    if (lp) {
        p.Kpad = up(p.Mt, 64);
        size_t wide = (size_t)(s.I > 3 * s.D ? s.I : 3 * s.D);
        if (wide < 9 * (size_t)MID) wide = 9 * MID;   // also holds the bf16 copy of the rearranged seg_head.0 weight
        p.dxc = take(p.Mt * s.D * 2);
        p.dhf = take(p.Mp * s.D * 4);
        p.wt = take(wide * s.D * 2);
        p.wt_layer = (size_t)s.D * (3 * s.D + s.D + 2 * (size_t)s.I);
        p.wt_all = take(p.wt_layer * s.L * 2);
        p.zero = take(256);
        size_t w = wgrad_bf16_scratch_floats(s.D, s.I, (int)p.Kpad);
        auto mx = [&](size_t v) { w = v > w ? v : w; };
        mx(wgrad_bf16_scratch_floats(s.I, s.D, (int)p.Kpad));
        mx(wgrad_bf16_scratch_floats(s.D, s.D, (int)p.Kpad));
        mx(wgrad_bf16_scratch_floats(3 * s.D, s.D, (int)p.Kpad));
        mx(wgrad_scratch_floats(MID, 9 * s.D, (int)p.Mp));  // head and patch weight gradients: fp32 or bf16 slicing
        mx(wgrad_scratch_floats(s.D, s.Kp, (int)p.Mp));
        mx(wgrad_bf16_scratch_floats(MID, 9 * s.D, (int)p.Mp));
        mx(wgrad_bf16_scratch_floats(s.D, s.Kp, (int)p.Mp));
        mx(thin_scratch_floats(s.I > 3 * s.D ? s.I : 3 * s.D));   // also the split-K partials of the CLS rows (dgrad, qkv)
        p.wscratch = take(w * 4);
        p.wscratch_floats = w;
    }
    p.total = off;
    return p;
}

struct Ctx {
    void* const* events = nullptr;  // optional hipEvent_t per gradient bucket (vitseg_backward)
    // records bucket event i on the launch stream: every gradient of that bucket is final at this point
    int mark(int bucket) const {
        if (!events || !events[bucket]) return VITSEG_OK;
        hipError_t e = hipEventRecord((hipEvent_t)events[bucket], st);
        return e == hipSuccess ? VITSEG_OK : hip_fail(e, "hipEventRecord(grad bucket)");
    }
    Shape s;
    TrainPlan p;
    Layout lay;
    const float* params;
    char* ws;
    hipStream_t st;
    int B;
    float eps;
    bool lp;
    float drop_p = 0.f;
    unsigned drop_seed = 0;
    float loss_scale = 1.f;   // multiplies the gradient of the fused CE loss (not the reported loss value)
    // dropout sites: 0 embeddings, 1 attention probabilities, 2 attention output, 3 MLP output (modeling_vit.py:159,184,276,283)
    DropArgs dr(int layer, int site) const {
        DropArgs d{};
        if (drop_p > 0.f) {
            d.thresh = (unsigned)((double)drop_p * 65536.0 + 0.5);
            if (d.thresh == 0) d.thresh = 1;  // p below 2^-17 still drops something rather than switching dropout off
            d.seed = drop_seed;
            d.stream = (unsigned)(layer * 8 + site);
            d.scale = 1.0f / (1.0f - drop_p);
        }
        return d;
    }
    // keep-bit words of layer l's attention dropout, or null (dropout off / geometry without whole 128-token blocks)
    const unsigned* dropw(int l) const {
        return lp && drop_p > 0.f && s.Np % 128 == 0 && !getenv("VITSEG_NO_DROPMASK") ? (const unsigned*)LV(l, p.lb.dropw) : nullptr;
    }
    const unsigned short* params_lp;
    const unsigned short* WL(int t, int l = 0) const { return params_lp + tensor_offset(lay, t, l); }
    void* LV(int l, size_t off) const { return (void*)(ws + p.layer0 + (size_t)l * p.layer_stride + off); }
    void* TV(size_t off) const { return (void*)(ws + off); }
    const float* W(int t, int l = 0) const { return params + tensor_offset(lay, t, l); }
    float* L(int l, size_t off) const { return (float*)(ws + p.layer0 + (size_t)l * p.layer_stride + off); }
    float* T(size_t off) const { return (float*)(ws + off); }
};

int init_ctx(Ctx& c, const vitseg_config* cfg, const float* params, int B, int precision, void* ws, size_t ws_bytes,
             void* stream) {
    if (int rc = check_config(cfg, &c.s)) return rc;
    VITSEG_CHECK_ARG(precision == VITSEG_F32 || precision == VITSEG_BF16, VITSEG_EINVAL,
                     "training runs in VITSEG_F32 or VITSEG_BF16 (precision %d; fp16 is an inference format)", precision);
    VITSEG_CHECK_ARG(params && ws && B >= 1, VITSEG_EINVAL, "null pointer or batch < 1");
    VITSEG_CHECK_ARG(c.s.C <= 32, VITSEG_ESHAPE, "training supports at most 32 classes (got %d)", c.s.C);
    c.p = make_train_plan(c.s, B, precision);
    c.lp = precision == VITSEG_BF16;
    VITSEG_CHECK_ARG(ws_bytes >= c.p.total, VITSEG_EWORKSPACE, "training workspace %zu < required %zu", ws_bytes,
                     c.p.total);
    c.lay = make_layout(c.s);
    c.params = params;
    c.ws = (char*)ws;
    c.st = (hipStream_t)stream;
    c.B = B;
    c.eps = cfg->layer_norm_eps;
    return VITSEG_OK;
}

GemmArgs lin(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda,
             int ldc) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
    return g;
}


// algorithmic FLOPs of one layer's attention; products_x2 = 2 x the number of distinct N x N x hd matrix products:
// 4 for the forward (S, PV), 10 for the backward (S, dP, dV, dK, dQ -- the second kernel's recomputation of S and dP
// is not algorithmic work and is not counted)
static double attn_flops(const Ctx& c, int products_x2) {
    const double N = c.s.Np + 1;
    return (double)products_x2 * c.B * c.s.A * N * N * (c.s.D / c.s.A);
}

// =========================================================================================================
// Mixed-precision (bf16 operands, fp32 master weights / residual stream / gradients) training path.
// dgrad = N-form bf16 GEMM against a transposed bf16 copy of the (small) weight; wgrad = the T-form/T-form
// kernel (dY and X are read as they lie, MFMA operands gathered with transposed LDS reads), split-K, fp32 out.
// The head (1.7 % of the FLOPs) and the patch embedding stay on the fp32 kernels.
int forward_train_bf16(Ctx& c, const float* x, float* logits) {
    const Shape& s = c.s;
    const int Mt = (int)c.p.Mt, Mp = (int)c.p.Mp, D = s.D, I = s.I, batch = c.B;
    hipStream_t st = c.st;
    int rc;
    {
        float* X0 = c.L(0, c.p.lb.xin);
        GemmArgs g = lin(x, c.W(VITSEG_T_PATCH_W), c.W(VITSEG_T_PATCH_B), c.W(VITSEG_T_POS), X0, Mp, D, s.Kp, 0, D);
        g.S = s.S; g.P = s.P; g.g = s.g; g.Np = s.Np; g.Cin = s.Cin; g.D = D;
        if ((rc = launch_gemm_f32(g, A_PATCH, EPI_POS, st, 1))) return rc;
        if ((rc = launch_cls_rows(c.W(VITSEG_T_CLS), c.W(VITSEG_T_POS), X0, batch, s.Np, D, st))) return rc;
        if (c.drop_p > 0.f && (rc = launch_dropout_rows(X0, X0, 0, Mt, D, c.dr(0, 0), st))) return rc;
    }
    for (int l = 0; l < s.L; ++l) {
        float* Xin = c.L(l, c.p.lb.xin);
        float* Xmid = c.L(l, c.p.lb.xmid);
        float* Xout = l + 1 < s.L ? c.L(l + 1, c.p.lb.xin) : c.T(c.p.xfinal);
        void *H1 = c.LV(l, c.p.lb.h1), *QKV = c.LV(l, c.p.lb.qkv), *CTX = c.LV(l, c.p.lb.ctx), *H2 = c.LV(l, c.p.lb.h2);
        if ((rc = launch_layernorm(Xin, c.W(VITSEG_T_LN1_W, l), c.W(VITSEG_T_LN1_B, l), H1, Mt, D, c.eps, true, st)))
            return rc;
        // CLS rows as a split-K side launch with the same epilogue (whole-tile body, see GemmArgs::thin_rows)
        auto thin = [&](GemmArgs& t) {
            if (Mp % 256 == 0 && batch <= THIN_MAX_ROWS) {
                t.thin_rows = batch;
                t.thin_scratch = c.T(c.p.wscratch);
                t.thin_capacity = c.p.wscratch_floats;
            }
        };
        GemmArgs g = lin(H1, c.WL(VITSEG_T_WQKV, l), c.W(VITSEG_T_BQKV, l), nullptr, QKV, Mt, 3 * D, D, D, 3 * D);
        thin(g);
        {
            ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 2.0 * Mt * 3 * D * D, st);
            if ((rc = launch_gemm_bf16(g, A_PLAIN, EPI_BIAS, st))) return rc;
        }
        {
            ProfScope ps(VITSEG_K_TRAIN_ATTN_FWD, attn_flops(c, 4), st);
            const unsigned* mw = c.dropw(l);
            if (mw && (rc = launch_attn_dropmask((unsigned*)mw, batch, s.Np, s.A, c.dr(l, 1), st))) return rc;
            if ((rc = launch_attention_bf16(QKV, CTX, c.L(l, c.p.lb.lse), batch, s.Np, s.A, c.dr(l, 1), st, false, mw)))
                return rc;
        }
        g = lin(CTX, c.WL(VITSEG_T_WO, l), c.W(VITSEG_T_BO, l), Xin, Xmid, Mt, D, D, D, D);
        g.drop = c.dr(l, 2);
        thin(g);
        {
            ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 2.0 * Mt * D * D, st);
            if ((rc = launch_gemm_bf16(g, A_PLAIN, EPI_RESADD, st))) return rc;
        }
        if ((rc = launch_layernorm(Xmid, c.W(VITSEG_T_LN2_W, l), c.W(VITSEG_T_LN2_B, l), H2, Mt, D, c.eps, true, st)))
            return rc;
        g = lin(H2, c.WL(VITSEG_T_W1, l), c.W(VITSEG_T_B1, l), nullptr, c.LV(l, c.p.lb.uact), Mt, I, D, D, I);
        g.aux = c.LV(l, c.p.lb.upre);
        thin(g);
        ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 4.0 * Mt * D * I, st);
        if ((rc = launch_gemm_bf16_train(g, EPI_GELU, 0, nullptr, st))) return rc;
        g = lin(c.LV(l, c.p.lb.uact), c.WL(VITSEG_T_W2, l), c.W(VITSEG_T_B2, l), Xmid, Xout, Mt, D, I, I, D);
        g.drop = c.dr(l, 3);
        thin(g);
        if ((rc = launch_gemm_bf16(g, A_PLAIN, EPI_RESADD, st))) return rc;
    }
    void* Hf = c.TV(c.p.hf);
    float* F = c.T(c.p.f);
    float* Z = c.T(c.p.z);
    if ((rc = launch_layernorm(c.T(c.p.xfinal), c.W(VITSEG_T_LNF_W), c.W(VITSEG_T_LNF_B), Hf, Mp, D, c.eps, true, st)))
        return rc;
    {
        hipError_t e = hipMemsetAsync(c.ws + c.p.zero, 0, 256, st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(zero page)");
        GemmArgs g = lin(Hf, c.WL(VITSEG_T_HEAD0_W), c.W(VITSEG_T_HEAD0_B), nullptr, F, Mp, MID, 9 * D, 0, MID);
        g.g = s.g; g.Np = s.Np; g.D = D;
        g.zeros = c.ws + c.p.zero;
        if ((rc = launch_gemm_bf16(g, A_CONV3, EPI_RELU, st))) return rc;
        if ((rc = launch_head1x1(F, c.W(VITSEG_T_HEAD2_W), c.W(VITSEG_T_HEAD2_B), Z, batch, s.Np, s.C, st))) return rc;
    }
    if (logits) return launch_upsample(Z, logits, nullptr, batch, s.C, s.g, s.S, st);
    return VITSEG_OK;
}

int backward_bf16(Ctx& c, const float* x, const void* target, int target_is_u8, const float* grad_logits, float* grads,
                  float* loss) {
    const Shape& s = c.s;
    const int Mt = (int)c.p.Mt, Mp = (int)c.p.Mp, D = s.D, I = s.I, B = c.B, Kpad = (int)c.p.Kpad;
    hipStream_t st = c.st;
    int rc;
    auto G = [&](int t, int l = 0) { return grads + tensor_offset(c.lay, t, l); };
    float* scratch = c.T(c.p.scratch);
    float* wscr = c.T(c.p.wscratch);
    void *wT = c.TV(c.p.wt), *dXc = c.TV(c.p.dxc);
    {
        hipError_t e = hipMemsetAsync(grads, 0, c.lay.total * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(grads)");
    }
    // dW[Nd,Kd] = dY[Mt,Nd]^T . X[Mt,Kd]  (both bf16, row-major) -> transposes + N-form split-K GEMM, fp32 out
    auto wgrad = [&](const void* dY, const void* X, float* dW, int Nd, int Kd) {
        GemmArgs g = lin(dY, X, nullptr, nullptr, dW, Nd, Kd, Mt, Nd, Kd);  // both operands T-form, no copies
        g.ldw = Kd;
        g.zeros = c.ws + c.p.zero;
        ProfScope ps(VITSEG_K_TRAIN_WGRAD, 2.0 * Mt * Nd * Kd, st);
        return launch_wgrad_bf16_tt(g, wscr, st);
    };
    (void)Kpad;
    // dX[Mt,Kd] = dY[Mt,Nd] . W[Nd,Kd]  with W^T materialised as [Kd][Nd] bf16
    // bias_grad (EPI_DGELU): the column sums of dX -- the bias gradient of the layer dX is the pre-activation gradient of --
    // come out of the GEMM epilogue (GemmArgs::colsum_*) instead of a pass over dX
    // dgrad operands: W^T of the four linears of every layer, materialised by ONE launch (order: QKV, O, fc1, fc2)
    const size_t wt_off[4] = {0, (size_t)3 * D * D, (size_t)4 * D * D, (size_t)4 * D * D + (size_t)I * D};
    {
        const size_t src0[4] = {tensor_offset(c.lay, VITSEG_T_WQKV, 0), tensor_offset(c.lay, VITSEG_T_WO, 0),
                                tensor_offset(c.lay, VITSEG_T_W1, 0), tensor_offset(c.lay, VITSEG_T_W2, 0)};
        const int R4[4] = {3 * D, D, I, D}, C4[4] = {D, D, D, I};
        const size_t lstride = c.lay.layer_stride;
        ProfScope ps(VITSEG_K_TRAIN_DGRAD, 0.0, st);
        if ((rc = launch_transpose_layers_bf16(c.params_lp, c.TV(c.p.wt_all), src0, R4, C4, lstride, s.L, st))) return rc;
    }
    auto wT_of = [&](int l, int kind) {
        return (const void*)((const unsigned short*)c.TV(c.p.wt_all) + (size_t)l * c.p.wt_layer + wt_off[kind]);
    };
    auto dgrad = [&](const void* dY, const void* Wt, void* dX, int Nd, int Kd, int epi, const void* R,
                     float* bias_grad = nullptr) {
        ProfScope ps(VITSEG_K_TRAIN_DGRAD, 2.0 * Mt * Nd * Kd, st);
        GemmArgs g = lin(dY, Wt, nullptr, (const float*)R, dX, Mt, Kd, Nd, Nd, Kd);
        g.ldw = Nd;
        if (Mp % 256 == 0 && B <= THIN_MAX_ROWS) {  // CLS rows as a split-K side launch (whole-tile body, see GemmArgs)
            g.thin_rows = B;
            g.thin_scratch = wscr;                   // the weight-gradient partial buffer is idle here
            g.thin_capacity = c.p.wscratch_floats;
        }
        g.colsum_out = bias_grad;
        g.colsum_scratch = bias_grad ? scratch : nullptr;
        return launch_gemm_bf16_train(g, epi, 0, nullptr, st);
    };

    // ---- 1. loss -> d logits -> d low-res logits; 2. seg_head backward (fp32 kernels) ----
    float* dZ = c.T(c.p.dz);
    const float* Gfull = grad_logits;
    if (target) {
        if ((rc = launch_ce_loss(c.T(c.p.z), target, target_is_u8, c.T(c.p.g), (double*)(c.ws + c.p.ce_partial), loss, B,
                                 s.C, s.g, s.S, st, c.loss_scale)))
            return rc;
        Gfull = c.T(c.p.g);
    }
    if ((rc = launch_upsample_bwd(Gfull, dZ, B, s.C, s.g, s.S, st))) return rc;
    float* dF = c.T(c.p.df);
    float* dHf = c.T(c.p.dhf);
    if ((rc = launch_head1x1_bwd(dZ, c.T(c.p.f), c.W(VITSEG_T_HEAD2_W), dF, G(VITSEG_T_HEAD2_W), G(VITSEG_T_HEAD2_B),
                                 scratch, B, s.Np, s.C, st)))
        return rc;
    if ((rc = launch_colsum(dF, 0, G(VITSEG_T_HEAD0_B), scratch, Mp, MID, MID, st))) return rc;
    // seg_head.0 weight gradient on the bf16 pipe: dF cast to bf16 (into the not-yet-used dXc buffer), the 3x3 im2col of
    // the bf16 head input gathered as bf16 (half the bytes of the fp32 one), T-form x T-form GEMM over the patch rows
    if ((rc = launch_cast_bf16(dF, dXc, (size_t)Mp * MID, st))) return rc;
    if ((rc = launch_im2col3x3_bf16(c.TV(c.p.hf), c.TV(c.p.t), B, s.g, D, st))) return rc;
    {
        GemmArgs g = lin(dXc, c.TV(c.p.t), nullptr, nullptr, G(VITSEG_T_HEAD0_W), MID, 9 * D, Mp, MID, 9 * D);
        g.ldw = 9 * D;
        g.zeros = c.ws + c.p.zero;
        if ((rc = launch_wgrad_bf16_tt(g, wscr, st))) return rc;
    }
    if ((rc = launch_conv_dgrad_weight(c.W(VITSEG_T_HEAD0_W), c.T(c.p.wd), D, st))) return rc;
    {   // dgrad of the 3x3 conv on the bf16 pipe: the implicit-GEMM gather over the bf16 copy of dF (still in dXc)
        // against a bf16 copy of the rearranged weight (in the weight-transpose buffer, free until the layer loop)
        if ((rc = launch_cast_bf16(c.T(c.p.wd), wT, (size_t)D * 9 * MID, st))) return rc;
        GemmArgs g = lin(dXc, wT, nullptr, nullptr, dHf, Mp, D, 9 * MID, 0, D);
        g.g = s.g; g.Np = s.Np; g.D = MID;
        g.zeros = c.ws + c.p.zero;
        if ((rc = launch_gemm_bf16(g, A_CONV3, EPI_BIAS, st))) return rc;
    }
    // ---- 3. final LayerNorm backward ----
    float* dXa = c.T(c.p.dxa);
    float* dXb = c.T(c.p.dxb);
    {
        hipError_t e = hipMemsetAsync(dXa + (size_t)Mp * D, 0, (size_t)(Mt - Mp) * D * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dX cls rows)");
    }
    if ((rc = launch_layernorm_bwd(c.T(c.p.xfinal), c.W(VITSEG_T_LNF_W), dHf, 0, nullptr, dXa, G(VITSEG_T_LNF_W),
                                   G(VITSEG_T_LNF_B), scratch, Mp, D, c.eps, st)))
        return rc;
    if ((rc = c.mark(0))) return rc;  // bucket 0: final norm + seg_head
    // ---- 4. encoder layers ----
    void *dH = c.TV(c.p.dh), *dU = c.TV(c.p.du), *dQKV = c.TV(c.p.dqkv), *dCTX = c.TV(c.p.dctx);
    for (int l = s.L - 1; l >= 0; --l) {
        // The gradient entering a dropped residual branch (bf16, hidden-dropout mask applied) and the branch's bias
        // gradient are by-products of the LayerNorm backward that produced the residual-stream gradient (BR outputs);
        // only the first one of the walk, fed by the final norm (patch rows only), is a pass of its own.
        if (l == s.L - 1) {
            if (c.drop_p > 0.f) {
                if ((rc = launch_dropout_rows(dXa, dXc, 1, Mt, D, c.dr(l, 3), st))) return rc;
            } else {
                if ((rc = launch_cast_bf16(dXa, dXc, (size_t)Mt * D, st))) return rc;
            }
            if ((rc = launch_colsum(dXc, 1, G(VITSEG_T_B2, l), scratch, Mt, D, D, st))) return rc;
        }
        if ((rc = wgrad(dXc, c.LV(l, c.p.lb.uact), G(VITSEG_T_W2, l), D, I))) return rc;
        if ((rc = dgrad(dXc, wT_of(l, 3), dU, D, I, EPI_DGELU, c.LV(l, c.p.lb.upre), G(VITSEG_T_B1, l)))) return rc;
        if ((rc = wgrad(dU, c.LV(l, c.p.lb.h2), G(VITSEG_T_W1, l), I, D))) return rc;
        if ((rc = dgrad(dU, wT_of(l, 2), dH, I, D, EPI_BIAS, nullptr))) return rc;
        if ((rc = launch_layernorm_bwd(c.L(l, c.p.lb.xmid), c.W(VITSEG_T_LN2_W, l), dH, 1, dXa, dXb, G(VITSEG_T_LN2_W, l),
                                       G(VITSEG_T_LN2_B, l), scratch, Mt, D, c.eps, st, dXc, c.dr(l, 2), G(VITSEG_T_BO, l))))
            return rc;
        if ((rc = wgrad(dXc, c.LV(l, c.p.lb.ctx), G(VITSEG_T_WO, l), D, D))) return rc;
        if ((rc = dgrad(dXc, wT_of(l, 1), dCTX, D, D, EPI_BIAS, nullptr))) return rc;
        {
            ProfScope ps(VITSEG_K_TRAIN_ATTN_BWD, attn_flops(c, 10), st);
            if ((rc = launch_attention_bwd_bf16(c.LV(l, c.p.lb.qkv), c.LV(l, c.p.lb.ctx), dCTX, c.L(l, c.p.lb.lse),
                                                c.T(c.p.dvec), dQKV, B, s.Np, s.A, c.dr(l, 1), st, c.dropw(l))))
                return rc;
        }
        if ((rc = launch_colsum(dQKV, 1, G(VITSEG_T_BQKV, l), scratch, Mt, 3 * D, 3 * D, st))) return rc;
        if ((rc = wgrad(dQKV, c.LV(l, c.p.lb.h1), G(VITSEG_T_WQKV, l), 3 * D, D))) return rc;
        if ((rc = dgrad(dQKV, wT_of(l, 0), dH, 3 * D, D, EPI_BIAS, nullptr))) return rc;
        if (l > 0) {  // next branch of the walk: layer l-1's MLP output
            if ((rc = launch_layernorm_bwd(c.L(l, c.p.lb.xin), c.W(VITSEG_T_LN1_W, l), dH, 1, dXb, dXa, G(VITSEG_T_LN1_W, l),
                                           G(VITSEG_T_LN1_B, l), scratch, Mt, D, c.eps, st, dXc, c.dr(l - 1, 3),
                                           G(VITSEG_T_B2, l - 1))))
                return rc;
        } else if ((rc = launch_layernorm_bwd(c.L(l, c.p.lb.xin), c.W(VITSEG_T_LN1_W, l), dH, 1, dXb, dXa,
                                              G(VITSEG_T_LN1_W, l), G(VITSEG_T_LN1_B, l), scratch, Mt, D, c.eps, st))) {
            return rc;
        }
        if ((rc = c.mark(s.L - l))) return rc;  // bucket 1 + (L-1-l): layer l
    }
    // ---- 5. embeddings (fp32) ----
    if (c.drop_p > 0.f && (rc = launch_dropout_rows(dXa, dXa, 0, Mt, D, c.dr(0, 0), st))) return rc;
    if ((rc = launch_embed_bwd(dXa, G(VITSEG_T_POS), G(VITSEG_T_CLS), B, s.Np, D, st))) return rc;
    if ((rc = launch_colsum(dXa, 0, G(VITSEG_T_PATCH_B), scratch, Mp, D, D, st))) return rc;
    if (s.Kp % 8 == 0) {  // patch-embedding weight gradient on the bf16 pipe (bf16 patch rows x bf16 dX, fp32 accumulate)
        if ((rc = launch_cast_bf16(dXa, dXc, (size_t)Mp * D, st))) return rc;
        if ((rc = launch_im2col_patch_bf16(x, c.TV(c.p.t), B, s.Cin, s.S, s.P, st))) return rc;
        GemmArgs g = lin(dXc, c.TV(c.p.t), nullptr, nullptr, G(VITSEG_T_PATCH_W), D, s.Kp, Mp, D, s.Kp);
        g.ldw = s.Kp;
        g.zeros = c.ws + c.p.zero;
        if ((rc = launch_wgrad_bf16_tt(g, wscr, st))) return rc;
        return c.mark(s.L + 1);  // last bucket: embeddings
    }
    if ((rc = launch_im2col_patch(x, c.T(c.p.t), B, s.Cin, s.S, s.P, st))) return rc;
    GemmArgs g = lin(dXa, c.T(c.p.t), nullptr, nullptr, G(VITSEG_T_PATCH_W), D, s.Kp, Mp, D, s.Kp);
    g.ldw = s.Kp;
    if ((rc = launch_wgrad_f32(g, c.T(c.p.wscratch), st))) return rc;
    return c.mark(s.L + 1);  // last bucket: embeddings
}

}  // namespace

extern "C" {

int vitseg_train_workspace(const vitseg_config* cfg, int batch, int precision, size_t* bytes) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(batch >= 1 && bytes, VITSEG_EINVAL, "batch %d / null out pointer", batch);
    VITSEG_CHECK_ARG(precision == VITSEG_F32 || precision == VITSEG_BF16, VITSEG_EINVAL,
                     "training runs in VITSEG_F32 or VITSEG_BF16 (precision %d; fp16 is an inference format)", precision);
    *bytes = make_train_plan(s, batch, precision).total;
    return VITSEG_OK;
}

int vitseg_forward_train(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x,
                         int batch, int precision, float dropout_p, uint64_t dropout_seed, float* logits,
                         void* workspace, size_t workspace_bytes, void* stream) {
    Ctx c;
    if (int rc = init_ctx(c, cfg, params, batch, precision, workspace, workspace_bytes, stream)) return rc;
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "dropout_p %f", dropout_p);
    c.drop_p = dropout_p;
    c.drop_seed = (unsigned)(dropout_seed ^ (dropout_seed >> 32));
    VITSEG_CHECK_ARG(x, VITSEG_EINVAL, "x is null");
    VITSEG_CHECK_ARG(!c.lp || params_bf16, VITSEG_EINVAL, "bf16 training needs the bf16 arena");
    c.params_lp = (const unsigned short*)params_bf16;
    if (c.lp) return forward_train_bf16(c, x, logits);
    const Shape& s = c.s;
    const int Mt = (int)c.p.Mt, Mp = (int)c.p.Mp, D = s.D, I = s.I;
    hipStream_t st = c.st;
    int rc;
    // embeddings -> Xin[0]
    {
        float* X0 = c.L(0, c.p.lb.xin);
        GemmArgs g = lin(x, c.W(VITSEG_T_PATCH_W), c.W(VITSEG_T_PATCH_B), c.W(VITSEG_T_POS), X0, Mp, D, s.Kp, 0, D);
        g.S = s.S; g.P = s.P; g.g = s.g; g.Np = s.Np; g.Cin = s.Cin; g.D = D;
        if ((rc = launch_gemm_f32(g, A_PATCH, EPI_POS, st))) return rc;
        if ((rc = launch_cls_rows(c.W(VITSEG_T_CLS), c.W(VITSEG_T_POS), X0, batch, s.Np, D, st))) return rc;
        if (c.drop_p > 0.f && (rc = launch_dropout_rows(X0, X0, 0, Mt, D, c.dr(0, 0), st))) return rc;
    }
    for (int l = 0; l < s.L; ++l) {
        float* Xin = c.L(l, c.p.lb.xin);
        float* H1 = c.L(l, c.p.lb.h1);
        float* QKV = c.L(l, c.p.lb.qkv);
        float* CTX = c.L(l, c.p.lb.ctx);
        float* Xmid = c.L(l, c.p.lb.xmid);
        float* H2 = c.L(l, c.p.lb.h2);
        float* Xout = l + 1 < s.L ? c.L(l + 1, c.p.lb.xin) : c.T(c.p.xfinal);
        if ((rc = launch_layernorm(Xin, c.W(VITSEG_T_LN1_W, l), c.W(VITSEG_T_LN1_B, l), H1, Mt, D, c.eps, false, st)))
            return rc;
        GemmArgs g = lin(H1, c.W(VITSEG_T_WQKV, l), c.W(VITSEG_T_BQKV, l), nullptr, QKV, Mt, 3 * D, D, D, 3 * D);
        {
            ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 2.0 * Mt * 3 * D * D, st);
            if ((rc = launch_gemm_f32(g, A_PLAIN, EPI_BIAS, st))) return rc;
        }
        {
            ProfScope ps(VITSEG_K_TRAIN_ATTN_FWD, attn_flops(c, 4), st);
            if ((rc = launch_attention_f32(QKV, CTX, c.L(l, c.p.lb.lse), batch, s.Np, s.A, c.dr(l, 1), st))) return rc;
        }
        g = lin(CTX, c.W(VITSEG_T_WO, l), c.W(VITSEG_T_BO, l), Xin, Xmid, Mt, D, D, D, D);
        g.drop = c.dr(l, 2);
        {
            ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 2.0 * Mt * D * D, st);
            if ((rc = launch_gemm_f32(g, A_PLAIN, EPI_RESADD, st))) return rc;
        }
        if ((rc = launch_layernorm(Xmid, c.W(VITSEG_T_LN2_W, l), c.W(VITSEG_T_LN2_B, l), H2, Mt, D, c.eps, false, st)))
            return rc;
        g = lin(H2, c.W(VITSEG_T_W1, l), c.W(VITSEG_T_B1, l), nullptr, c.L(l, c.p.lb.uact), Mt, I, D, D, I);
        g.aux = c.L(l, c.p.lb.upre);
        ProfScope ps(VITSEG_K_TRAIN_GEMM_FWD, 4.0 * Mt * D * I, st);
        if ((rc = launch_gemm_f32(g, A_PLAIN, EPI_GELU, st))) return rc;
        g = lin(c.L(l, c.p.lb.uact), c.W(VITSEG_T_W2, l), c.W(VITSEG_T_B2, l), Xmid, Xout, Mt, D, I, I, D);
        g.drop = c.dr(l, 3);
        if ((rc = launch_gemm_f32(g, A_PLAIN, EPI_RESADD, st))) return rc;
    }
    float* Hf = c.T(c.p.hf);
    float* F = c.T(c.p.f);
    float* Z = c.T(c.p.z);
    if ((rc = launch_layernorm(c.T(c.p.xfinal), c.W(VITSEG_T_LNF_W), c.W(VITSEG_T_LNF_B), Hf, Mp, D, c.eps, false, st)))
        return rc;
    {
        GemmArgs g = lin(Hf, c.W(VITSEG_T_HEAD0_W), c.W(VITSEG_T_HEAD0_B), nullptr, F, Mp, MID, 9 * D, 0, MID);
        g.g = s.g; g.Np = s.Np; g.D = D;
        if ((rc = launch_gemm_f32(g, A_CONV3, EPI_RELU, st))) return rc;
        if ((rc = launch_head1x1(F, c.W(VITSEG_T_HEAD2_W), c.W(VITSEG_T_HEAD2_B), Z, batch, s.Np, s.C, st))) return rc;
    }
    if (logits) return launch_upsample(Z, logits, nullptr, batch, s.C, s.g, s.S, st);
    return VITSEG_OK;
}

int vitseg_backward(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x, int batch,
                    int precision, float dropout_p, uint64_t dropout_seed, const void* target, int target_is_u8,
                    const float* grad_logits, float* grads, float* loss, float loss_scale, void* const* bucket_events,
                    void* workspace, size_t workspace_bytes, void* stream) {
    Ctx c;
    c.events = bucket_events;
    c.loss_scale = loss_scale;
    if (int rc = init_ctx(c, cfg, params, batch, precision, workspace, workspace_bytes, stream)) return rc;
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "dropout_p %f", dropout_p);
    c.drop_p = dropout_p;
    c.drop_seed = (unsigned)(dropout_seed ^ (dropout_seed >> 32));
    VITSEG_CHECK_ARG(x && grads, VITSEG_EINVAL, "x / grads is null");
    VITSEG_CHECK_ARG(!c.lp || params_bf16, VITSEG_EINVAL, "bf16 training needs the bf16 arena");
    c.params_lp = (const unsigned short*)params_bf16;
    if (c.lp) return backward_bf16(c, x, target, target_is_u8, grad_logits, grads, loss);
    VITSEG_CHECK_ARG((target != nullptr) != (grad_logits != nullptr), VITSEG_EINVAL,
                     "pass exactly one of target (fused CE) and grad_logits");
    VITSEG_CHECK_ARG(!target || loss, VITSEG_EINVAL, "fused CE needs the loss output pointer");
    const Shape& s = c.s;
    const int Mt = (int)c.p.Mt, Mp = (int)c.p.Mp, D = s.D, I = s.I, B = batch;
    hipStream_t st = c.st;
    int rc;
    auto G = [&](int t, int l = 0) { return grads + tensor_offset(c.lay, t, l); };
    float* scratch = c.T(c.p.scratch);
    {
        hipError_t e = hipMemsetAsync(grads, 0, c.lay.total * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(grads)");
    }
    // dgrad: dX[M,Kd] = dY[M,Nd] . W[Nd,Kd]   (A N-form, W T-form)
    auto dgrad = [&](const float* dY, const float* Wt, float* dX, int M, int Nd, int Kd, int epi, const float* R) {
        GemmArgs g = lin(dY, Wt, nullptr, R, dX, M, Kd, Nd, Nd, Kd);
        g.ldw = Kd;
        ProfScope ps(VITSEG_K_TRAIN_DGRAD, 2.0 * M * Nd * Kd, st);
        return launch_gemm_f32_bwd(g, A_PLAIN, 0, 1, epi, st);
    };
    // wgrad: dW[Nd,Kd] = dY[M,Nd]^T . X[M,Kd]   (both T-form, reduction over the M token rows)
    auto wgrad = [&](const float* dY, const float* X, float* dW, int M, int Nd, int Kd) {
        GemmArgs g = lin(dY, X, nullptr, nullptr, dW, Nd, Kd, M, Nd, Kd);
        g.ldw = Kd;
        ProfScope ps(VITSEG_K_TRAIN_WGRAD, 2.0 * M * Nd * Kd, st);
        return launch_wgrad_f32(g, c.T(c.p.wscratch), st);
    };

    // ---- 1. loss -> d logits -> d low-res logits ----
    float* dZ = c.T(c.p.dz);
    const float* Gfull = grad_logits;
    if (target) {
        if ((rc = launch_ce_loss(c.T(c.p.z), target, target_is_u8, c.T(c.p.g), (double*)(c.ws + c.p.ce_partial), loss, B,
                                 s.C, s.g, s.S, st, c.loss_scale)))
            return rc;
        Gfull = c.T(c.p.g);
    }
    if ((rc = launch_upsample_bwd(Gfull, dZ, B, s.C, s.g, s.S, st))) return rc;

    // ---- 2. seg_head backward ----
    float* dF = c.T(c.p.df);
    float* Hf = c.T(c.p.hf);
    float* dH = c.T(c.p.dh);
    if ((rc = launch_head1x1_bwd(dZ, c.T(c.p.f), c.W(VITSEG_T_HEAD2_W), dF, G(VITSEG_T_HEAD2_W), G(VITSEG_T_HEAD2_B),
                                 scratch, B, s.Np, s.C, st)))
        return rc;
    if ((rc = launch_colsum(dF, 0, G(VITSEG_T_HEAD0_B), scratch, Mp, MID, MID, st))) return rc;
    if ((rc = launch_im2col3x3(Hf, 0, c.T(c.p.t), B, s.g, D, st))) return rc;
    if ((rc = wgrad(dF, c.T(c.p.t), G(VITSEG_T_HEAD0_W), Mp, MID, 9 * D))) return rc;
    if ((rc = launch_conv_dgrad_weight(c.W(VITSEG_T_HEAD0_W), c.T(c.p.wd), D, st))) return rc;
    {
        GemmArgs g = lin(dF, c.T(c.p.wd), nullptr, nullptr, dH, Mp, D, 9 * MID, 0, D);
        g.g = s.g; g.Np = s.Np; g.D = MID;
        if ((rc = launch_gemm_f32_bwd(g, A_CONV3, 0, 0, EPI_BIAS, st))) return rc;
    }
    // ---- 3. final LayerNorm backward (patch rows; the dropped CLS rows get zero gradient) ----
    float* dXa = c.T(c.p.dxa);
    float* dXb = c.T(c.p.dxb);
    {
        hipError_t e = hipMemsetAsync(dXa + (size_t)Mp * D, 0, (size_t)(Mt - Mp) * D * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dX cls rows)");
    }
    if ((rc = launch_layernorm_bwd(c.T(c.p.xfinal), c.W(VITSEG_T_LNF_W), dH, 0, nullptr, dXa, G(VITSEG_T_LNF_W),
                                   G(VITSEG_T_LNF_B), scratch, Mp, D, c.eps, st)))
        return rc;
    if ((rc = c.mark(0))) return rc;  // bucket 0: final norm + seg_head

    // ---- 4. encoder layers, last to first ----
    float* dU = c.T(c.p.du);
    float* dQKV = c.T(c.p.dqkv);
    float* dCTX = c.T(c.p.dctx);
    for (int l = s.L - 1; l >= 0; --l) {
        // MLP: Xout = Xmid + drop(fc2(gelu(fc1(H2))))
        const float* dB = dXa;  // gradient entering the (dropped) branch
        if (c.drop_p > 0.f) {
            if ((rc = launch_dropout_rows(dXa, c.T(c.p.dxm), 0, Mt, D, c.dr(l, 3), st))) return rc;
            dB = c.T(c.p.dxm);
        }
        if ((rc = launch_colsum(dB, 0, G(VITSEG_T_B2, l), scratch, Mt, D, D, st))) return rc;
        if ((rc = wgrad(dB, c.L(l, c.p.lb.uact), G(VITSEG_T_W2, l), Mt, D, I))) return rc;
        if ((rc = dgrad(dB, c.W(VITSEG_T_W2, l), dU, Mt, D, I, EPI_DGELU, c.L(l, c.p.lb.upre)))) return rc;
        if ((rc = launch_colsum(dU, 0, G(VITSEG_T_B1, l), scratch, Mt, I, I, st))) return rc;
        if ((rc = wgrad(dU, c.L(l, c.p.lb.h2), G(VITSEG_T_W1, l), Mt, I, D))) return rc;
        if ((rc = dgrad(dU, c.W(VITSEG_T_W1, l), dH, Mt, I, D, EPI_BIAS, nullptr))) return rc;
        if ((rc = launch_layernorm_bwd(c.L(l, c.p.lb.xmid), c.W(VITSEG_T_LN2_W, l), dH, 0, dXa, dXb, G(VITSEG_T_LN2_W, l),
                                       G(VITSEG_T_LN2_B, l), scratch, Mt, D, c.eps, st)))
            return rc;
        // attention: Xmid = Xin + drop(o_proj(attn(qkv(H1))))
        dB = dXb;
        if (c.drop_p > 0.f) {
            if ((rc = launch_dropout_rows(dXb, c.T(c.p.dxm), 0, Mt, D, c.dr(l, 2), st))) return rc;
            dB = c.T(c.p.dxm);
        }
        if ((rc = launch_colsum(dB, 0, G(VITSEG_T_BO, l), scratch, Mt, D, D, st))) return rc;
        if ((rc = wgrad(dB, c.L(l, c.p.lb.ctx), G(VITSEG_T_WO, l), Mt, D, D))) return rc;
        if ((rc = dgrad(dB, c.W(VITSEG_T_WO, l), dCTX, Mt, D, D, EPI_BIAS, nullptr))) return rc;
        {
            ProfScope ps(VITSEG_K_TRAIN_ATTN_BWD, attn_flops(c, 10), st);
            if ((rc = launch_attention_bwd_f32(c.L(l, c.p.lb.qkv), c.L(l, c.p.lb.ctx), dCTX, c.L(l, c.p.lb.lse),
                                               c.T(c.p.dvec), dQKV, B, s.Np, s.A, c.dr(l, 1), st)))
                return rc;
        }
        if ((rc = launch_colsum(dQKV, 0, G(VITSEG_T_BQKV, l), scratch, Mt, 3 * D, 3 * D, st))) return rc;
        if ((rc = wgrad(dQKV, c.L(l, c.p.lb.h1), G(VITSEG_T_WQKV, l), Mt, 3 * D, D))) return rc;
        if ((rc = dgrad(dQKV, c.W(VITSEG_T_WQKV, l), dH, Mt, 3 * D, D, EPI_BIAS, nullptr))) return rc;
        if ((rc = launch_layernorm_bwd(c.L(l, c.p.lb.xin), c.W(VITSEG_T_LN1_W, l), dH, 0, dXb, dXa, G(VITSEG_T_LN1_W, l),
                                       G(VITSEG_T_LN1_B, l), scratch, Mt, D, c.eps, st)))
            return rc;
        if ((rc = c.mark(s.L - l))) return rc;  // bucket 1 + (L-1-l): layer l
    }
    // ---- 5. embeddings ----
    if (c.drop_p > 0.f && (rc = launch_dropout_rows(dXa, dXa, 0, Mt, D, c.dr(0, 0), st))) return rc;
    if ((rc = launch_embed_bwd(dXa, G(VITSEG_T_POS), G(VITSEG_T_CLS), B, s.Np, D, st))) return rc;
    if ((rc = launch_colsum(dXa, 0, G(VITSEG_T_PATCH_B), scratch, Mp, D, D, st))) return rc;
    if ((rc = launch_im2col_patch(x, c.T(c.p.t), B, s.Cin, s.S, s.P, st))) return rc;
    if ((rc = wgrad(dXa, c.T(c.p.t), G(VITSEG_T_PATCH_W), Mp, D, s.Kp))) return rc;
    return c.mark(s.L + 1);  // last bucket: embeddings
}

int vitseg_grad_bucket_count(const vitseg_config* cfg) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    return s.L + 2;
}

int vitseg_grad_bucket_range(const vitseg_config* cfg, int bucket, size_t* offset_floats, size_t* n_floats) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(bucket >= 0 && bucket < s.L + 2 && offset_floats && n_floats, VITSEG_EINVAL,
                     "grad bucket %d out of [0, %d)", bucket, s.L + 2);
    const Layout lay = make_layout(s);
    const size_t post0 = lay.layer0 + lay.layer_stride * s.L;
    if (bucket == 0) {
        *offset_floats = post0;
        *n_floats = lay.total - post0;
    } else if (bucket <= s.L) {
        *offset_floats = lay.layer0 + (size_t)(s.L - bucket) * lay.layer_stride;
        *n_floats = lay.layer_stride;
    } else {
        *offset_floats = 0;
        *n_floats = lay.layer0;
    }
    return VITSEG_OK;
}

int vitseg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n_floats, float lr,
                     float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    VITSEG_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, VITSEG_EINVAL, "adam: null pointer");
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n_floats, lr, beta1, beta2, eps, step, grad_scale,
                       (hipStream_t)stream);
}

}  // extern "C"
