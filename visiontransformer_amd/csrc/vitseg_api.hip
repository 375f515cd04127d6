// C ABI of libvitseg.so (include/vitseg.h): parameter-arena layout, workspace planning and
// the forward orchestration of the ViT segmentation hot path
// (ViTSegmentationModel.forward, /root/reference/model/CE/classes.py:246-262).
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <strings.h>

#include <atomic>

#include <vector>

#include "kernels.hpp"
#include "profile.hpp"
#include "plan.hpp"
#include "small.hpp"

namespace vitseg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return VITSEG_EHIP;
}

// ---- dispatcher switches (common.hpp Opt) ----
static const char* const g_opt_names[OPT_COUNT] = {"no_f32p", "no_p8", "no_h16p", "no_ragged_p8", "no_dropmask",
                                                   "dropw_limit_mb", "upsample_global", "bf16_tiles", "f32p_noinl", "gn", "no_mask2", "no_small",
                                                   "small_variant", "small_max_rows", "conv_dma"};
static std::atomic<long> g_opts[OPT_COUNT];
static int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcasecmp(name, g_opt_names[i])) return i;
    return -1;
}
// environment form: a number is taken as is (VITSEG_NO_P8=0 leaves the switch OFF, like vitseg_set_option("no_p8", 0));
// an empty or non-numeric value of a boolean switch means 1 (VITSEG_NO_P8= , VITSEG_NO_P8=yes)
static long opt_parse(int id, const char* text) {
    if (id == OPT_BF16_TILES && (text[0] == 's' || text[0] == 'l' || text[0] == 'x')) return text[0] == 's' ? 1 : text[0] == 'l' ? 2 : 3;
    char* end = nullptr;
    const long v = strtol(text, &end, 10);
    if (end != text) return v;
    return (id == OPT_DROPW_LIMIT_MB || id == OPT_GN || id == OPT_SMALL_VARIANT || id == OPT_SMALL_MAX_ROWS || id == OPT_BF16_TILES) ? 0 : 1;
}
static const bool g_opts_loaded = [] {   // once, at library load
    for (int i = 0; i < OPT_COUNT; ++i) {
        char env[64] = "VITSEG_";
        for (size_t k = 0; g_opt_names[i][k] && k + 8 < sizeof(env); ++k) {
            env[7 + k] = (char)toupper((unsigned char)g_opt_names[i][k]);
            env[8 + k] = 0;
        }
        const char* e = getenv(env);
        g_opts[i].store(e ? opt_parse(i, e) : (i == OPT_DROPW_LIMIT_MB ? -1 : 0), std::memory_order_relaxed);
    }
    return true;
}();
long opt(int id) { return g_opts[id].load(std::memory_order_relaxed); }

namespace {

using namespace plan;

// ---- workspace plan -------------------------------------------------------------
struct Plan {
    size_t zero, x, h, qkv, u, f, z, thin, total;  // byte offsets
    size_t thin_floats;                // capacity of the split-K scratch
    size_t Mt, Mp;                     // total token rows, patch rows
};

Plan make_plan(const Shape& s, int B, int precision) {
    Plan p{};
    p.Mp = (size_t)B * s.Np;
    p.Mt = p.Mp + B;
    const size_t act = (precision == VITSEG_F32 || precision == VITSEG_F32X3) ? 4 : 2;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += up(bytes, 256);
        return o;
    };
    p.zero = take(256);                               // zero page (padding taps of the bf16 3x3 loader)
    p.x = take(p.Mt * s.D * 4);                       // fp32 residual stream
    p.h = take(p.Mt * s.D * 4);                       // LN output / attention context (fp32 sized)
    const bool small_rows = precision != VITSEG_F32X3 && (long)p.Mt < small_max_rows();   // (not the no_small switch: a workspace serves both routes)
    p.qkv = take(p.Mt * 3 * s.D * (small_rows ? 4 : act));   // q | k | v (fp32 on the small-batch route, also in its 16-bit form)
    p.u = take(p.Mt * (size_t)s.I * act);             // MLP hidden
    p.f = take(p.Mp * MID * 4);                       // seg_head.0 output (fp32)
    p.z = take((size_t)B * s.C * s.Np * 4);           // low-res logits
    const int widest = 3 * s.D > s.I ? 3 * s.D : s.I;
    p.thin_floats = thin_scratch_floats(widest);      // K-slice partials: CLS rows (GemmArgs::thin_scratch) ...
    const int shapes[4][2] = {{3 * s.D, s.D}, {s.D, s.D}, {s.I, s.D}, {s.D, s.I}};   // ... or whole small GEMMs (N, K)
    for (auto& nk : shapes) {
        const size_t need = (size_t)whole_split((int)p.Mt, nk[0], nk[1], 32) * p.Mt * nk[0];
        if (need > p.thin_floats) p.thin_floats = need;
    }
    if (small_rows) {   // K-chunk slabs of the small-batch route (small.hpp)
        const size_t need[4] = {(size_t)small_splits(s.D, s.D) * p.Mt * s.D, (size_t)small_splits(s.D, s.I) * p.Mt * s.D,
                                (size_t)small_splits(s.D, s.Kp) * p.Mt * s.D, (size_t)9 * p.Mp * MID};
        for (size_t n : need)
            if (n > p.thin_floats) p.thin_floats = n;
    }
    p.thin = take(p.thin_floats * 4);
    p.total = off;
    return p;
}

// ---- the small-batch fp32 forward (small.hpp): fewer than SMALL_MAX_ROWS token rows ------------------------------
// 7 launches per layer: QKV GEMM (+bias) | attention | o_proj chunks | chunk sum + bias + residual + LayerNorm |
// fc1 GEMM (+bias, GELU) | fc2 chunks | chunk sum + bias + residual + the next LayerNorm.
// precision VITSEG_BF16 / VITSEG_F16: the four linears of every block multiply 16-bit operands (weights from the 16-bit arena,
// LayerNorm output / attention context / MLP hidden written in that format by their producers) on the wide MFMA of the same
// kernels, the attention products too (q, k, P, v rounded in registers, fp32 softmax); the residual stream, q | k | v as stored,
// the patch embedding and the head stay fp32.
int forward_small(const vitseg_config* cfg, const Shape& s, const Layout& lay, const Plan& p, const float* params, const void* params_lp,
                  int precision, const float* x, int batch, float* logits, uint8_t* mask, char* ws, hipStream_t st) {
    auto W = [&](int t, int layer = 0) { return params + tensor_offset(lay, t, layer); };
    const int h16 = precision == VITSEG_BF16 ? 1 : precision == VITSEG_F16 ? 2 : 0;
    float* X = (float*)(ws + p.x);
    float* H = (float*)(ws + p.h);
    float* QKV = (float*)(ws + p.qkv);
    float* U = (float*)(ws + p.u);
    float* Z = (float*)(ws + p.z);
    float* part = (float*)(ws + p.thin);
    const int Mt = (int)p.Mt, Mp = (int)p.Mp, D = s.D;
    const size_t dstride = (size_t)Mt * D;   // slab stride of the D-wide chunk sums
    int rc;
    auto linear = [&](const float* A, int M, int K, int lda, int wt, int bt, int layer, float* C, int N, int epi, int kind) {
        SGemm g{};
        g.A = A; g.W = W(wt, layer); g.bias = W(bt, layer); g.C = C;
        if (h16) g.W = (const float*)((const unsigned short*)params_lp + tensor_offset(lay, wt, layer));   // (A is 16-bit too: its producer wrote it so)
        g.h16 = h16;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldc = N;
        g.splits = epi == SE_PARTIAL ? small_splits(N, K) : 1;
        g.split_stride = dstride;
        ProfScope ps(kind, 2.0 * M * N * K, st);
        return launch_gemm_f32s(g, epi, SA_PLAIN, st);
    };
    auto rows = [&](int splits, const float* bias, const float* lnw, const float* lnb, int ln_rows, bool embed, int h_fmt) {
        SRows r{};
        r.h_fmt = h_fmt;
        r.X = X; r.partial = part; r.split_stride = dstride; r.splits = splits; r.bias = bias;
        r.pos = W(VITSEG_T_POS); r.cls = W(VITSEG_T_CLS); r.lnw = lnw; r.lnb = lnb; r.H = H;
        r.rows = Mt; r.Mp = Mp; r.Np = s.Np; r.D = D; r.ln_rows = ln_rows; r.embed = embed ? 1 : 0;
        r.eps = cfg->layer_norm_eps;
        ProfScope ps(VITSEG_K_LAYERNORM, (double)Mt * D * 4 * (2 + splits) + (double)ln_rows * D * 4, st);
        return launch_resln(r, st);
    };
    // ---- embeddings (a2 + a3): patch projection chunks, then bias + position embedding + CLS rows + LayerNorm 1 of layer 0
    // (patch sizes the gathering DMA does not cover -- P = 4: 48 values per patch -- take the large-batch route's launches)
    const bool dma_patch = (s.P == 8 || s.P == 16 || s.P == 32) && s.Kp % 32 == 0;
    if (!dma_patch) {
        GemmArgs g{};
        g.A = x; g.W = W(VITSEG_T_PATCH_W); g.bias = W(VITSEG_T_PATCH_B); g.R = W(VITSEG_T_POS); g.C = X;
        g.M = Mp; g.N = D; g.K = s.Kp; g.lda = 0; g.ldc = D;
        g.S = s.S; g.P = s.P; g.g = s.g; g.Np = s.Np; g.Cin = s.Cin; g.D = D;
        {
            ProfScope ps(VITSEG_K_GEMM_PATCH, 2.0 * g.M * g.N * g.K, st);
            if ((rc = launch_gemm_f32(g, A_PATCH, EPI_POS, st, 0))) return rc;
        }
        if ((rc = launch_cls_rows(W(VITSEG_T_CLS), W(VITSEG_T_POS), X, batch, s.Np, D, st))) return rc;
        ProfScope ps(VITSEG_K_LAYERNORM, (double)Mt * D * 8, st);
        if ((rc = launch_layernorm(X, W(VITSEG_T_LN1_W, 0), W(VITSEG_T_LN1_B, 0), H, Mt, D, cfg->layer_norm_eps, h16, st))) return rc;
    } else {
        SGemm g{};
        g.A = x; g.W = W(VITSEG_T_PATCH_W); g.C = part;
        g.M = Mp; g.N = D; g.K = s.Kp; g.lda = 0; g.ldw = s.Kp; g.ldc = D;
        g.splits = small_splits(D, s.Kp); g.split_stride = dstride;
        g.g = s.g; g.Np = s.Np; g.S = s.S; g.P = s.P; g.Cin = s.Cin;
        {
            ProfScope ps(VITSEG_K_GEMM_PATCH, 2.0 * g.M * g.N * g.K, st);
            if ((rc = launch_gemm_f32s(g, SE_PARTIAL, SA_PATCH, st))) return rc;
        }
        if ((rc = rows(g.splits, W(VITSEG_T_PATCH_B), W(VITSEG_T_LN1_W, 0), W(VITSEG_T_LN1_B, 0), Mt, true, h16))) return rc;
    }
    for (int l = 0; l < s.L; ++l) {
        if ((rc = linear(H, Mt, D, D, VITSEG_T_WQKV, VITSEG_T_BQKV, l, QKV, 3 * D, SE_BIAS, VITSEG_K_GEMM_BIAS))) return rc;
        {
            ProfScope ps(VITSEG_K_ATTENTION, 4.0 * batch * s.A * (double)s.N * s.N * 64, st);
            // by the SHAPE only (a row's bits must not depend on the batch): patch counts that are whole 64-key tiles and long
            // enough to fill 128-query blocks (512x512: 1024) take attention_f32's tail-free loop, every other length the
            // key-split kernel (tools/attn_small_probe.py, profiles/r05_attn_small_probe.txt: 197 and 785 tokens 1.1-2.7x
            // faster at every batch; 1025 tokens faster only at batch 1; 3137 tokens 1.4x faster at batch 1, 1.06x at 2,
            // 0.92x at 4: the rows of such a forward end at batch 5)
            // (the 16-bit form of the route exists for the key-split kernel's lengths only: small_applies)
            rc = attn_small_infer(s.Np) ? launch_attention_small(QKV, H, batch, s.Np, s.A, st, nullptr, DropArgs{0, 0, 0, 1.f}, h16)
                                        : launch_attention_f32(QKV, H, nullptr, batch, s.Np, s.A, DropArgs{}, st);
            if (rc) return rc;
        }
        if ((rc = linear(H, Mt, D, D, VITSEG_T_WO, VITSEG_T_BO, l, part, D, SE_PARTIAL, VITSEG_K_GEMM_RESADD))) return rc;
        if ((rc = rows(small_splits(D, D), W(VITSEG_T_BO, l), W(VITSEG_T_LN2_W, l), W(VITSEG_T_LN2_B, l), Mt, false, h16))) return rc;
        if ((rc = linear(H, Mt, D, D, VITSEG_T_W1, VITSEG_T_B1, l, U, s.I, SE_GELU, VITSEG_K_GEMM_GELU))) return rc;
        if ((rc = linear(U, Mt, s.I, s.I, VITSEG_T_W2, VITSEG_T_B2, l, part, D, SE_PARTIAL, VITSEG_K_GEMM_RESADD))) return rc;
        const bool last = l + 1 == s.L;   // the final LayerNorm covers the patch rows only (CLS is dropped, classes.py:250)
        if ((rc = rows(small_splits(D, s.I), W(VITSEG_T_B2, l), last ? W(VITSEG_T_LNF_W) : W(VITSEG_T_LN1_W, l + 1),
                       last ? W(VITSEG_T_LNF_B) : W(VITSEG_T_LN1_B, l + 1), last ? Mp : Mt, false, last ? 0 : h16)))   // (the head reads fp32)
            return rc;
    }
    // ---- seg_head (a10 + a11): the 3x3 conv as nine shifted GEMMs (one tap per chunk), then ReLU + the 1x1 conv
    {
        SGemm g{};
        g.A = H; g.W = W(VITSEG_T_HEAD0_W); g.C = part;
        g.M = Mp; g.N = MID; g.K = D; g.lda = D; g.ldw = 9 * D; g.ldc = MID;
        g.splits = 9; g.split_stride = (size_t)Mp * MID;
        g.g = s.g; g.Np = s.Np;
        {
            ProfScope ps(VITSEG_K_GEMM_CONV3, 2.0 * g.M * g.N * 9 * g.K, st);
            if ((rc = launch_gemm_f32s(g, SE_PARTIAL, SA_CONV3, st))) return rc;
        }
        ProfScope ps(VITSEG_K_HEAD1X1, (double)Mp * MID * 4 * 9 + (double)batch * s.C * s.Np * 4, st);
        if ((rc = launch_headfin(part, g.split_stride, W(VITSEG_T_HEAD0_B), W(VITSEG_T_HEAD2_W), W(VITSEG_T_HEAD2_B), Z, batch, s.Np, s.C, st)))
            return rc;
    }
    const double px = (double)batch * s.S * s.S;
    ProfScope ps(VITSEG_K_UPSAMPLE, (logits ? px * s.C * 4 : 0.0) + (mask ? px : 0.0) + (double)batch * s.C * s.Np * 4, st);
    return launch_upsample(Z, logits, mask, batch, s.C, s.g, s.S, st);
}

}  // namespace

Profiler& profiler() {
    static Profiler p;
    return p;
}

long small_max_rows() {
    const long v = opt(OPT_SMALL_MAX_ROWS);
    return v > 0 ? v : SMALL_MAX_ROWS;
}

bool small_applies(const vitseg_config* cfg, int batch, int precision) {
    Shape s;
    if (precision == VITSEG_F32X3 || opt(OPT_NO_SMALL) || check_config(cfg, &s)) return false;
    const long rows = (long)batch * s.N;
    if (!(rows < small_max_rows() && s.D % 64 == 0 && s.I % 32 == 0 && s.I > s.D && s.S % 4 == 0)) return false;
    // 16-bit operands (inference): whole 64-value K steps per chunk, the sequence lengths of the key-split attention kernel (it
    // writes the 16-bit context), and fewer rows than fp32 -- the large-batch 16-bit kernels (256 x 256 tiles) catch up at
    // batch 16 of 197 tokens and at batch 4 of 785 (profiles/r05_h16_route_probe.txt)
    const long lim16 = opt(OPT_SMALL_MAX_ROWS) > 0 ? opt(OPT_SMALL_MAX_ROWS) : small_max_rows_16(s.N);   // (the option: probes of the limit)
    if (precision != VITSEG_F32 && !(s.I % 64 == 0 && attn_small_infer(s.Np) && rows < lim16)) return false;
    return true;
}
}  // namespace vitseg

using namespace vitseg;

extern "C" {

int vitseg_version(void) { return VITSEG_VERSION; }
const char* vitseg_last_error(void) { return g_err; }

int vitseg_set_option(const char* name, long long value) {
    const int id = opt_index(name);
    VITSEG_CHECK_ARG(id >= 0, VITSEG_EINVAL, "vitseg_set_option: unknown option '%s'", name ? name : "(null)");
    g_opts[id].store((long)value, std::memory_order_relaxed);
    return VITSEG_OK;
}
int vitseg_get_option(const char* name, long long* value) {
    const int id = opt_index(name);
    VITSEG_CHECK_ARG(id >= 0 && value, VITSEG_EINVAL, "vitseg_get_option: unknown option '%s'", name ? name : "(null)");
    *value = opt(id);
    return VITSEG_OK;
}

int vitseg_param_count(const vitseg_config* cfg, size_t* n_floats) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(n_floats, VITSEG_EINVAL, "n_floats is null");
    *n_floats = make_layout(s).total;
    return VITSEG_OK;
}

int vitseg_param_offset(const vitseg_config* cfg, int tensor, int layer, size_t* offset_floats, size_t* numel) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(tensor >= 0 && tensor < VITSEG_T_COUNT, VITSEG_EINVAL, "tensor id %d", tensor);
    VITSEG_CHECK_ARG(!per_layer(tensor) || (layer >= 0 && layer < s.L), VITSEG_EINVAL, "layer %d out of range", layer);
    const Layout l = make_layout(s);
    if (offset_floats) *offset_floats = tensor_offset(l, tensor, layer);
    if (numel) *numel = tensor_numel(s, tensor);
    return VITSEG_OK;
}

int vitseg_cast_params_bf16(const float* params, void* params_bf16, size_t n_floats, void* stream) {
    VITSEG_CHECK_ARG(params && params_bf16, VITSEG_EINVAL, "null arena");
    return launch_cast_bf16(params, params_bf16, n_floats, (hipStream_t)stream);
}

int vitseg_cast_params_split(const float* params, void* params_split, size_t n_floats, void* stream) {
    VITSEG_CHECK_ARG(params && params_split, VITSEG_EINVAL, "null arena");
    return launch_cast_split(params, params_split, n_floats, (hipStream_t)stream);
}

int vitseg_cast_params_f16(const float* params, void* params_f16, size_t n_floats, void* stream) {
    VITSEG_CHECK_ARG(params && params_f16, VITSEG_EINVAL, "null arena");
    return launch_cast_bf16(params, params_f16, n_floats, (hipStream_t)stream, true);
}

// which kernels vitseg_forward takes for this call: 1 = the small-batch route (small.hpp), 0 = the large-batch kernels; < 0: error
int vitseg_forward_route(const vitseg_config* cfg, int batch, int precision) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(batch >= 1 && precision >= VITSEG_F32 && precision <= VITSEG_F32X3, VITSEG_EINVAL, "forward_route: batch %d precision %d", batch, precision);
    return small_applies(cfg, batch, precision) ? 1 : 0;
}

int vitseg_query_workspace(const vitseg_config* cfg, int batch, int precision, size_t* bytes) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(batch >= 1 && bytes, VITSEG_EINVAL, "batch %d / null out pointer", batch);
    VITSEG_CHECK_ARG(precision >= VITSEG_F32 && precision <= VITSEG_F32X3, VITSEG_EINVAL, "precision %d", precision);
    *bytes = make_plan(s, batch, precision).total;
    return VITSEG_OK;
}

int vitseg_workspace_offset(const vitseg_config* cfg, int batch, int precision, int buffer, size_t* offset_bytes,
                            size_t* bytes) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(batch >= 1, VITSEG_EINVAL, "batch %d", batch);
    const Plan p = make_plan(s, batch, precision);
    size_t o = 0, n = 0;
    switch (buffer) {
        case VITSEG_BUF_TOKENS: o = p.x; n = p.Mt * s.D * 4; break;
        case VITSEG_BUF_LOWRES: o = p.z; n = (size_t)batch * s.C * s.Np * 4; break;
        default: set_error("buffer id %d", buffer); return VITSEG_EINVAL;
    }
    if (offset_bytes) *offset_bytes = o;
    if (bytes) *bytes = n;
    return VITSEG_OK;
}

int vitseg_forward(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x, int batch,
                   int precision, float* logits, uint8_t* mask, void* workspace, size_t workspace_bytes,
                   void* stream_) {
    Shape s;
    if (int rc = check_config(cfg, &s)) return rc;
    VITSEG_CHECK_ARG(params && x && workspace && batch >= 1, VITSEG_EINVAL, "null pointer or batch < 1");
    VITSEG_CHECK_ARG(logits || mask, VITSEG_EINVAL, "both outputs are null");
    VITSEG_CHECK_ARG(precision >= VITSEG_F32 && precision <= VITSEG_F32X3, VITSEG_EINVAL, "precision %d", precision);
    VITSEG_CHECK_ARG(precision == VITSEG_F32 || precision == VITSEG_F32X3 || params_bf16, VITSEG_EINVAL,
                     "16-bit forward needs the 16-bit arena");
    VITSEG_CHECK_ARG(((uintptr_t)params | (uintptr_t)x | (uintptr_t)workspace | (uintptr_t)logits) % 16 == 0,
                     VITSEG_EINVAL, "pointers must be 16-byte aligned");
    // fp32 storage, GEMMs on the fp16 pipe with split operands; 2 = the weights come pre-split (params_bf16 slot)
    const int x3 = precision == VITSEG_F32X3 ? (params_bf16 ? 2 : 1) : 0;
    const bool lp = precision == VITSEG_BF16 || precision == VITSEG_F16, f16 = precision == VITSEG_F16;
    const Plan p = make_plan(s, batch, precision);
    VITSEG_CHECK_ARG(workspace_bytes >= p.total, VITSEG_EWORKSPACE, "workspace %zu < required %zu", workspace_bytes,
                     p.total);
    hipStream_t st = (hipStream_t)stream_;
    const Layout lay = make_layout(s);
    auto W = [&](int t, int layer = 0) { return params + tensor_offset(lay, t, layer); };
    // weight operand of a GEMM: fp32 arena or its bf16 shadow (same element offsets)
    auto WG = [&](int t, int layer = 0) -> const void* {
        const size_t off = tensor_offset(lay, t, layer);
        if (x3 == 2) return (const void*)((const float*)params_bf16 + off);  // split arena: same float offsets
        return lp ? (const void*)((const unsigned short*)params_bf16 + off) : (const void*)(params + off);
    };
    char* ws = (char*)workspace;
    if (small_applies(cfg, batch, precision)) return forward_small(cfg, s, lay, p, params, params_bf16, precision, x, batch, logits, mask, ws, st);
    float* X = (float*)(ws + p.x);
    void* H = (void*)(ws + p.h);      // fp32 or bf16 by precision
    void* QKV = (void*)(ws + p.qkv);
    void* U = (void*)(ws + p.u);
    float* F = (float*)(ws + p.f);
    float* Z = (float*)(ws + p.z);
    const int Mt = (int)p.Mt, Mp = (int)p.Mp, D = s.D;
    int rc;

    // ---- embeddings (a2 + a3): patch GEMM gathers straight from the NCHW image ----
    {
        GemmArgs g{};
        g.A = x; g.W = x3 == 2 ? WG(VITSEG_T_PATCH_W) : (const void*)W(VITSEG_T_PATCH_W);  // fp32 (or pre-split) weights
        g.bias = W(VITSEG_T_PATCH_B); g.R = W(VITSEG_T_POS); g.C = X;
        g.M = Mp; g.N = D; g.K = s.Kp; g.lda = 0; g.ldc = D;
        g.S = s.S; g.P = s.P; g.g = s.g; g.Np = s.Np; g.Cin = s.Cin; g.D = D;
        {
            ProfScope ps(VITSEG_K_GEMM_PATCH, 2.0 * g.M * g.N * g.K, st);
            // 16-bit modes: the image and the patch weights are fp32 either way; the split-operand kernel (fp32-grade
            // products on the fp16 pipe) does this GEMM in 0.16 ms instead of 0.36 ms
            if ((rc = launch_gemm_f32(g, A_PATCH, EPI_POS, st, lp ? 1 : x3))) return rc;
        }
        if ((rc = launch_cls_rows(W(VITSEG_T_CLS), W(VITSEG_T_POS), X, batch, s.Np, D, st))) return rc;
    }
    // ---- encoder layers (a4..a8) ----
    const double ln_bytes = 2.0 * Mt * D * 4;
    // when the patch rows are a whole number of row tiles, the trailing CLS rows of the linear layers go
    // through the split-K side launch (bitwise batch-invariant: a CLS row takes that path at every batch size)
    const int thin_rows = (p.Mp % 256 == 0 && batch <= THIN_MAX_ROWS) ? batch : 0;
    auto gemm = [&](GemmArgs g, int epi, int kind) {
        ProfScope ps(kind, 2.0 * g.M * g.N * g.K, st);
        g.thin_rows = thin_rows;
        g.thin_scratch = (float*)(ws + p.thin);
        g.thin_capacity = p.thin_floats;
        return lp ? launch_gemm_bf16(g, A_PLAIN, epi, st, f16) : launch_gemm_f32(g, A_PLAIN, epi, st, x3);
    };
    auto lnorm = [&](const float* w, const float* b, int rows) {
        ProfScope ps(VITSEG_K_LAYERNORM, (double)rows * D * (lp ? 6 : 8), st);
        return launch_layernorm(X, w, b, H, rows, D, cfg->layer_norm_eps, lp ? (f16 ? 2 : 1) : 0, st);
    };
    (void)ln_bytes;
    for (int l = 0; l < s.L; ++l) {
        if ((rc = lnorm(W(VITSEG_T_LN1_W, l), W(VITSEG_T_LN1_B, l), Mt))) return rc;
        GemmArgs g{};
        g.A = H; g.W = WG(VITSEG_T_WQKV, l); g.bias = W(VITSEG_T_BQKV, l); g.C = QKV;
        g.M = Mt; g.N = 3 * D; g.K = D; g.lda = D; g.ldc = 3 * D;
        if ((rc = gemm(g, EPI_BIAS, VITSEG_K_GEMM_BIAS))) return rc;
        {
            ProfScope ps(VITSEG_K_ATTENTION, 4.0 * batch * s.A * (double)s.N * s.N * 64, st);
            rc = lp ? launch_attention_bf16(QKV, H, nullptr, batch, s.Np, s.A, DropArgs{}, st, f16)
                    : launch_attention_f32((const float*)QKV, (float*)H, nullptr, batch, s.Np, s.A, DropArgs{}, st, x3 != 0);
            if (rc) return rc;
        }
        g = GemmArgs{};
        g.A = H; g.W = WG(VITSEG_T_WO, l); g.bias = W(VITSEG_T_BO, l); g.R = X; g.C = X;
        g.M = Mt; g.N = D; g.K = D; g.lda = D; g.ldc = D;
        if ((rc = gemm(g, EPI_RESADD, VITSEG_K_GEMM_RESADD))) return rc;
        if ((rc = lnorm(W(VITSEG_T_LN2_W, l), W(VITSEG_T_LN2_B, l), Mt))) return rc;
        g = GemmArgs{};
        g.A = H; g.W = WG(VITSEG_T_W1, l); g.bias = W(VITSEG_T_B1, l); g.C = U;
        g.M = Mt; g.N = s.I; g.K = D; g.lda = D; g.ldc = s.I;
        if ((rc = gemm(g, EPI_GELU, VITSEG_K_GEMM_GELU))) return rc;
        g = GemmArgs{};
        g.A = U; g.W = WG(VITSEG_T_W2, l); g.bias = W(VITSEG_T_B2, l); g.R = X; g.C = X;
        g.M = Mt; g.N = D; g.K = s.I; g.lda = s.I; g.ldc = D;
        if ((rc = gemm(g, EPI_RESADD, VITSEG_K_GEMM_RESADD))) return rc;
    }
    // ---- final LayerNorm on the patch rows only (CLS is dropped, classes.py:250) ----
    if ((rc = lnorm(W(VITSEG_T_LNF_W), W(VITSEG_T_LNF_B), Mp))) return rc;
    // ---- seg_head (a10 + a11): 3x3 conv as implicit GEMM over the token-major map ----
    {
        GemmArgs g{};
        g.A = H; g.W = WG(VITSEG_T_HEAD0_W); g.bias = W(VITSEG_T_HEAD0_B); g.C = F;
        g.M = Mp; g.N = MID; g.K = 9 * D; g.lda = 0; g.ldc = MID;
        g.g = s.g; g.Np = s.Np; g.D = D;
        g.zeros = ws + p.zero;
        if (lp) {
            hipError_t e = hipMemsetAsync(ws + p.zero, 0, 256, st);
            if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(zero page)");
        }
        {
            ProfScope ps(VITSEG_K_GEMM_CONV3, 2.0 * g.M * g.N * g.K, st);
            if (!lp && !x3 && D % 32 == 0 && opt(OPT_CONV_DMA) && (size_t)(Mp + 128) * D * 4 < 0x7fffffffull) {
                // (switch conv_dma, off by default) fp32: the same fmaf chain per output (k = (ky, kx, d): bit-identical to the
                // implicit GEMM) with the operands through the LDS-DMA ring of gemm_f32s -- taps outside the image are out-of-range
                // offsets that read as zeros.  Measured at the headline size (profiles/r05_notes.md): 0.94 -> 0.85 ms per forward,
                // but 1 794 MB of L2 misses per launch against 978 MB: nine taps x two 128-column tiles re-read the map through
                // the Infinity Cache.  Not the default: more traffic for 0.2 % of the step.
                SGemm c{};
                c.A = (const float*)H; c.W = W(VITSEG_T_HEAD0_W); c.bias = W(VITSEG_T_HEAD0_B); c.C = (float*)F;
                c.M = Mp; c.N = MID; c.K = D; c.lda = D; c.ldw = 9 * D; c.ldc = MID; c.splits = 1;
                c.g = s.g; c.Np = s.Np;
                rc = launch_gemm_f32s(c, SE_RELU, SA_CONV3_ALL, st);
            } else {
                rc = lp ? launch_gemm_bf16(g, A_CONV3, EPI_RELU, st, f16) : launch_gemm_f32(g, A_CONV3, EPI_RELU, st, x3);
            }
            if (rc) return rc;
        }
        ProfScope ps(VITSEG_K_HEAD1X1, (double)Mp * MID * 4 + (double)batch * s.C * s.Np * 4, st);
        if ((rc = launch_head1x1(F, W(VITSEG_T_HEAD2_W), W(VITSEG_T_HEAD2_B), Z, batch, s.Np, s.C, st))) return rc;
    }
    // ---- bilinear upsample (+ sigmoid -> argmax) (a12 + a14) ----
    const double px = (double)batch * s.S * s.S;
    ProfScope ps(VITSEG_K_UPSAMPLE, (logits ? px * s.C * 4 : 0.0) + (mask ? px : 0.0) + (double)batch * s.C * s.Np * 4,
                 st);
    return launch_upsample(Z, logits, mask, batch, s.C, s.g, s.S, st);
}

size_t vitseg_ce_scratch_bytes(int batch, int S) { return ce_partial_count(batch, S) * sizeof(double); }

int vitseg_ce_loss(const float* lowres, const void* target, int target_is_u8, float* grad_logits, void* scratch,
                   float* loss, int batch, int C, int g, int S, void* stream) {
    VITSEG_CHECK_ARG(batch >= 1 && C >= 1 && g >= 1 && S >= g, VITSEG_EINVAL, "ce_loss: bad shape");
    return launch_ce_loss(lowres, target, target_is_u8, grad_logits, (double*)scratch, loss, batch, C, g, S,
                          (hipStream_t)stream);
}

int vitseg_profile_enable(int on) {
    profiler().clear();
    profiler().on = on != 0;
    return VITSEG_OK;
}

int vitseg_profile_collect(int kind, double* total_ms, int64_t* launches, double* work) {
    VITSEG_CHECK_ARG(kind >= 0 && kind < VITSEG_K_COUNT, VITSEG_EINVAL, "kernel kind %d", kind);
    double ms = 0, w = 0;
    int64_t n = 0;
    for (auto& r : profiler().recs) {
        if (r.kind != kind) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        ms += t;
        w += r.work;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (work) *work = w;
    return VITSEG_OK;
}

// ---- single-operator entry points ---------------------------------------------------
int vitseg_op_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps,
                            void* stream) {
    return launch_layernorm(x, w, b, y, rows, D, eps, false, (hipStream_t)stream);
}

int vitseg_op_linear_f32(const float* A, const float* Wt, const float* bias, const float* R, float* C, int M, int N,
                         int K, int epilogue, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 3, VITSEG_EINVAL, "linear: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(epilogue != EPI_RESADD || R, VITSEG_EINVAL, "linear: residual epilogue needs R");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    return launch_gemm_f32(g, A_PLAIN, epilogue, (hipStream_t)stream);
}

int vitseg_op_linear_f32_ex(const float* A, const float* Wt, const float* bias, const float* R, float* C, float* aux,
                            int M, int N, int K, int epilogue, float dropout_p, uint32_t dropout_seed,
                            uint32_t dropout_stream, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear_f32_ex: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 3, VITSEG_EINVAL, "linear_f32_ex: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(epilogue != EPI_RESADD || R, VITSEG_EINVAL, "linear_f32_ex: residual epilogue needs R");
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "linear_f32_ex: dropout_p %f", dropout_p);
    VITSEG_CHECK_ARG(!aux || epilogue == EPI_GELU, VITSEG_EINVAL, "linear_f32_ex: aux belongs to the GELU epilogue");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = R; g.C = C; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    if (dropout_p > 0.f && epilogue == EPI_RESADD) {
        g.drop.thresh = (unsigned)((double)dropout_p * 65536.0 + 0.5);
        if (g.drop.thresh == 0) g.drop.thresh = 1;
        g.drop.seed = dropout_seed;
        g.drop.stream = dropout_stream;
        g.drop.scale = 1.0f / (1.0f - dropout_p);
    }
    return launch_gemm_f32(g, A_PLAIN, epilogue, (hipStream_t)stream);
}

int vitseg_op_linear_bf16(const void* A, const void* Wt, const float* bias, const float* R, void* C, int M, int N,
                          int K, int epilogue, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VITSEG_EINVAL, "linear_bf16: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(epilogue != EPI_RESADD || R, VITSEG_EINVAL, "linear: residual epilogue needs R");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    return launch_gemm_bf16(g, A_PLAIN, epilogue, (hipStream_t)stream);
}

int vitseg_op_linear_f32x3(const float* A, const float* Wt, const float* bias, const float* R, float* C, int M, int N,
                           int K, int epilogue, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VITSEG_EINVAL, "linear_f32x3: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(epilogue != EPI_RESADD || R, VITSEG_EINVAL, "linear: residual epilogue needs R");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    return launch_gemm_f32(g, A_PLAIN, epilogue, (hipStream_t)stream, true);
}

int vitseg_op_linear_f16(const void* A, const void* Wt, const float* bias, const float* R, void* C, int M, int N, int K,
                         int epilogue, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VITSEG_EINVAL, "linear_f16: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(epilogue != EPI_RESADD || R, VITSEG_EINVAL, "linear: residual epilogue needs R");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    return launch_gemm_bf16(g, A_PLAIN, epilogue, (hipStream_t)stream, true);
}

int vitseg_op_linear_h16_ex(const void* A, const void* Wt, const float* bias, const void* R, void* C, void* aux, int M,
                            int N, int K, int epilogue, int f16, int thin_rows, float* scratch, size_t scratch_floats,
                            float dropout_p, uint32_t dropout_seed, uint32_t dropout_stream, float* colsum_out,
                            float* colsum_scratch, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "linear_h16_ex: null pointer");
    VITSEG_CHECK_ARG(epilogue == EPI_BIAS || epilogue == EPI_GELU || epilogue == EPI_RESADD || epilogue == EPI_DGELU,
                     VITSEG_EINVAL, "linear_h16_ex: epilogue %d", epilogue);
    VITSEG_CHECK_ARG((epilogue != EPI_RESADD && epilogue != EPI_DGELU) || R, VITSEG_EINVAL, "linear_h16_ex: epilogue needs R");
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "linear_h16_ex: dropout_p %f", dropout_p);
    GemmArgs g{};
    g.A = A; g.W = Wt; g.bias = bias; g.R = (const float*)R; g.C = C; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    if (thin_rows > 0) {
        g.thin_rows = thin_rows;
        g.thin_scratch = scratch;
        g.thin_capacity = scratch_floats;
    }
    if (dropout_p > 0.f) {
        g.drop.thresh = (unsigned)((double)dropout_p * 65536.0 + 0.5);
        if (g.drop.thresh == 0) g.drop.thresh = 1;
        g.drop.seed = dropout_seed;
        g.drop.stream = dropout_stream;
        g.drop.scale = 1.0f / (1.0f - dropout_p);
    }
    VITSEG_CHECK_ARG(!colsum_out || (colsum_scratch && epilogue == EPI_DGELU && !f16), VITSEG_EINVAL,
                     "linear_h16_ex: column sums come with the bf16 dGELU epilogue and need scratch");
    g.colsum_out = colsum_out;
    g.colsum_scratch = colsum_out ? colsum_scratch : nullptr;
    return launch_gemm_bf16(g, A_PLAIN, epilogue, (hipStream_t)stream, f16 != 0);
}

size_t vitseg_op_colsum_scratch_floats(int M, int N) { return colsum_scratch_floats(M, N); }

size_t vitseg_op_wgrad_bf16_scratch_floats(int M, int N, int K) { return wgrad_bf16_scratch_floats(M, N, K); }

int vitseg_op_wgrad_bf16(const void* dY, const void* X, float* dW, float* scratch, const void* zeros, int M, int N, int K,
                         void* stream) {
    VITSEG_CHECK_ARG(dY && X && dW && zeros, VITSEG_EINVAL, "wgrad_bf16: null pointer");
    GemmArgs g{};
    g.A = dY; g.W = X; g.C = dW;
    g.M = M; g.N = N; g.K = K; g.lda = M; g.ldw = N; g.ldc = N;
    g.zeros = zeros;
    return launch_wgrad_bf16_tt(g, scratch, (hipStream_t)stream);
}

int vitseg_op_gemm_f32(const float* A, const float* Wt, const float* R, float* C, int M, int N, int K, int ta, int tb,
                       int epilogue, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && C, VITSEG_EINVAL, "gemm: null pointer");
    GemmArgs g{};
    g.A = A; g.W = Wt; g.R = R; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = ta ? M : K; g.ldw = tb ? N : K; g.ldc = N;
    if (!ta && !tb) return launch_gemm_f32(g, A_PLAIN, epilogue, (hipStream_t)stream);
    return launch_gemm_f32_bwd(g, A_PLAIN, ta, tb, epilogue, (hipStream_t)stream);
}

// forward (saving the log-sum-exp) followed by the backward of the attention core; scratch: B*A*(Np+1) floats
int vitseg_op_attention_bwd_f32(const float* qkv, const float* dctx, float* ctx_out, float* lse_out, float* scratch,
                                float* dqkv, int batch, int num_patches, int num_heads, void* stream) {
    VITSEG_CHECK_ARG(qkv && dctx && ctx_out && lse_out && scratch && dqkv, VITSEG_EINVAL, "attention_bwd: null pointer");
    if (int rc = launch_attention_f32(qkv, ctx_out, lse_out, batch, num_patches, num_heads, DropArgs{}, (hipStream_t)stream)) return rc;
    return launch_attention_bwd_f32(qkv, ctx_out, dctx, lse_out, scratch, dqkv, batch, num_patches, num_heads, DropArgs{},
                                    (hipStream_t)stream);
}

// the same pair for short sequences (small.hpp): key-split forward saving the log-sum-exp, one-launch backward
int vitseg_op_attention_bwd_f32_small(const float* qkv, const float* dctx, float* ctx_out, float* lse_out, float* dqkv,
                                      int batch, int num_patches, int num_heads, float dropout_p, uint32_t dropout_seed,
                                      uint32_t dropout_stream, void* stream) {
    VITSEG_CHECK_ARG(qkv && dctx && ctx_out && lse_out && dqkv, VITSEG_EINVAL, "attention_bwd_small: null pointer");
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "attention_bwd_small: dropout_p %f", dropout_p);
    DropArgs d{0, 0, 0, 1.f};
    if (dropout_p > 0.f) {
        d.thresh = (unsigned)((double)dropout_p * 65536.0 + 0.5);
        if (d.thresh == 0) d.thresh = 1;
        d.seed = dropout_seed;
        d.stream = dropout_stream;
        d.scale = (float)(1.0 / (1.0 - (double)dropout_p));
    }
    if (int rc = launch_attention_small(qkv, ctx_out, batch, num_patches, num_heads, (hipStream_t)stream, lse_out, d)) return rc;
    return launch_attention_bwd_small(qkv, ctx_out, dctx, lse_out, dqkv, batch, num_patches, num_heads, d, (hipStream_t)stream);
}

// bf16 forward (saving the log-sum-exp, optional attention-probability dropout) + backward of the attention core
size_t vitseg_attention_dropmask_bytes(int batch, int num_patches, int num_heads) {
    return batch > 0 && num_heads > 0 && num_patches > 0 && num_patches % 128 == 0
               ? attn_dropmask_words(batch, num_patches, num_heads) * sizeof(unsigned) : 0;
}

size_t vitseg_attention_bwd_scratch_floats(int batch, int num_patches, int num_heads) {
    return batch > 0 && num_heads > 0 && num_patches > 0 ? attention_bwd_bf16_scratch_floats(batch, num_patches, num_heads) : 0;
}

int vitseg_op_attention_bwd_bf16(const void* qkv, const void* dctx, void* ctx_out, float* lse_out, float* scratch,
                                 void* dqkv, int batch, int num_patches, int num_heads, float dropout_p,
                                 uint32_t dropout_seed, uint32_t dropout_stream, void* dropmask_words, float* dbias_qkv,
                                 void* stream) {
    VITSEG_CHECK_ARG(qkv && dctx && ctx_out && lse_out && scratch && dqkv, VITSEG_EINVAL, "attention_bwd: null pointer");
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "attention_bwd: dropout_p %f", dropout_p);
    DropArgs d{};
    if (dropout_p > 0.f) {
        d.thresh = (unsigned)((double)dropout_p * 65536.0 + 0.5);
        if (d.thresh == 0) d.thresh = 1;
        d.seed = dropout_seed;
        d.stream = dropout_stream;
        d.scale = 1.0f / (1.0f - dropout_p);
    }
    const unsigned* mw = nullptr;
    if (dropmask_words) {
        VITSEG_CHECK_ARG(d.thresh && num_patches % 128 == 0, VITSEG_EINVAL,
                         "attention_bwd: mask words need dropout_p > 0 and num_patches %% 128 == 0");
        if (int rc = launch_attn_dropmask((unsigned*)dropmask_words, batch, num_patches, num_heads, d, (hipStream_t)stream))
            return rc;
        mw = (const unsigned*)dropmask_words;
    }
    if (int rc = launch_attention_bf16(qkv, ctx_out, lse_out, batch, num_patches, num_heads, d, (hipStream_t)stream, false, mw))
        return rc;
    return launch_attention_bwd_bf16(qkv, ctx_out, dctx, lse_out, scratch, dqkv, batch, num_patches, num_heads, d,
                                     (hipStream_t)stream, mw, dbias_qkv);
}

size_t vitseg_op_layernorm_bwd_scratch_floats(int rows, int D) { return rows > 0 && D > 0 ? layernorm_bwd_scratch_floats(rows, D) : 0; }

int vitseg_op_layernorm_bwd_f32(const float* x, const float* w, const float* g, const float* dres_in, float* dres_out,
                                float* dw, float* db, float* scratch, int rows, int D, float eps, void* stream) {
    return launch_layernorm_bwd(x, w, g, 0, dres_in, dres_out, dw, db, scratch, rows, D, eps, (hipStream_t)stream);
}
// the small-batch training step's form: g as K-chunk slabs, the next branch's dropped gradient + bias gradient (small.hpp)
int vitseg_op_layernorm_bwd_f32_small(const float* x, const float* w, const float* g, size_t g_stride, int g_splits,
                                      const float* dres_in, float* dres_out, float* dw, float* db, float* scratch, int rows,
                                      int D, float eps, float* br_out, float* br_dbias, float dropout_p, uint32_t dropout_seed,
                                      uint32_t dropout_stream, void* stream) {
    VITSEG_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, VITSEG_EINVAL, "layernorm_bwd_small: dropout_p %f", dropout_p);
    DropArgs d{0, 0, 0, 1.f};
    if (dropout_p > 0.f) {
        d.thresh = (unsigned)((double)dropout_p * 65536.0 + 0.5);
        if (d.thresh == 0) d.thresh = 1;
        d.seed = dropout_seed;
        d.stream = dropout_stream;
        d.scale = 1.0f / (1.0f - dropout_p);
    }
    return launch_layernorm_bwd_small(x, w, g, g_stride, g_splits, dres_in, dres_out, dw, db, scratch, rows, D, eps,
                                      (hipStream_t)stream, br_out, d, br_dbias);
}

int vitseg_op_attention_bf16(const void* qkv, void* ctx, int batch, int num_patches, int num_heads, void* stream) {
    return launch_attention_bf16(qkv, ctx, nullptr, batch, num_patches, num_heads, DropArgs{}, (hipStream_t)stream);
}

int vitseg_op_attention_f16(const void* qkv, void* ctx, int batch, int num_patches, int num_heads, void* stream) {
    return launch_attention_bf16(qkv, ctx, nullptr, batch, num_patches, num_heads, DropArgs{}, (hipStream_t)stream,
                                 true);
}

int vitseg_op_attention_f32x3(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream) {
    return launch_attention_f32(qkv, ctx, nullptr, batch, num_patches, num_heads, DropArgs{}, (hipStream_t)stream, true);
}

int vitseg_small_splits(int N, int K) { return small_splits(N, K); }

// diagnostics (tools/small_stamps.py): the direct-epilogue small GEMM with per-block time stamps (8 words per block,
// small.hpp SGemm::stamps; `stamps` holds >= 8 * blocks words, blocks <= M / 32 * N / 64 + ...: the caller sizes generously)
int vitseg_dbg_linear_f32_small(const float* A, const float* Wt, const float* bias, float* C, int M, int N, int K, int epilogue,
                                unsigned long long* stamps, int lds_pad, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && bias && C && stamps, VITSEG_EINVAL, "dbg_linear_f32_small: null pointer");
    SGemm g{};
    g.A = A; g.W = Wt; g.bias = bias; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.splits = 1;
    g.stamps = stamps; g.lds_pad = lds_pad;
    return launch_gemm_f32s(g, epilogue == EPI_GELU ? SE_GELU : SE_BIAS, SA_PLAIN, (hipStream_t)stream);
}

int vitseg_op_linear_f32_small(const float* A, const float* Wt, const float* bias, float* C, int M, int N, int K, int epilogue,
                               void* stream) {
    VITSEG_CHECK_ARG(A && Wt && bias && C, VITSEG_EINVAL, "linear_f32_small: null pointer");
    VITSEG_CHECK_ARG(epilogue == EPI_BIAS || epilogue == EPI_GELU, VITSEG_EINVAL, "linear_f32_small: epilogue %d", epilogue);
    VITSEG_CHECK_ARG(small_splits(N, K) == 1, VITSEG_ESHAPE, "linear_f32_small: N=%d K=%d is a chunked shape (use linear_resln)", N, K);
    SGemm g{};
    g.A = A; g.W = Wt; g.bias = bias; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.splits = 1;
    return launch_gemm_f32s(g, epilogue == EPI_GELU ? SE_GELU : SE_BIAS, SA_PLAIN, (hipStream_t)stream);
}

// the 16-bit form of the route's linears (operands as raw bf16 / fp16 bits): epilogue 0 bias -> fp32 C, 1 bias + GELU -> 16-bit C,
// 2 (chunked shapes) the K-chunk slabs into scratch, then C (fp32) = chunk sums in chunk order + bias
int vitseg_op_linear_h16_small(const void* A, const void* Wt, const float* bias, void* C, int M, int N, int K, int epilogue, int f16,
                               float* scratch, size_t scratch_floats, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && bias && C, VITSEG_EINVAL, "linear_h16_small: null pointer");
    VITSEG_CHECK_ARG(epilogue >= 0 && epilogue <= 2, VITSEG_EINVAL, "linear_h16_small: epilogue %d", epilogue);
    SGemm g{};
    g.A = (const float*)A; g.W = (const float*)Wt; g.bias = bias; g.C = (float*)C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.splits = 1;
    g.h16 = f16 ? 2 : 1;
    if (epilogue != 2) {
        VITSEG_CHECK_ARG(small_splits(N, K) == 1, VITSEG_ESHAPE, "linear_h16_small: N=%d K=%d is a chunked shape (epilogue 2)", N, K);
        return launch_gemm_f32s(g, epilogue == 1 ? SE_GELU : SE_BIAS, SA_PLAIN, (hipStream_t)stream);
    }
    g.splits = small_splits(N, K);
    g.split_stride = (size_t)M * N;
    VITSEG_CHECK_ARG(scratch && scratch_floats >= (g.splits + 1) * g.split_stride, VITSEG_EWORKSPACE,
                     "linear_h16_small: scratch %zu < %zu floats", scratch_floats, (g.splits + 1) * g.split_stride);
    g.C = scratch;
    if (int rc = launch_gemm_f32s(g, SE_PARTIAL, SA_PLAIN, (hipStream_t)stream)) return rc;
    // chunk sums + bias through the rows kernel's own arithmetic: X = 0 + (sum + bias), its LayerNorm output discarded
    float* zero = scratch + g.splits * g.split_stride;
    hipError_t e = hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(linear_h16_small)");
    SRows r{};
    r.X = (float*)C; r.partial = scratch; r.split_stride = g.split_stride; r.splits = g.splits; r.bias = bias;
    r.lnw = bias; r.lnb = bias; r.H = zero; r.rows = M; r.Mp = M; r.Np = M; r.D = N; r.ln_rows = 0; r.embed = 0; r.eps = 1e-12f;
    return launch_resln(r, (hipStream_t)stream);
}

int vitseg_op_linear_resln_f32_small(const float* A, const float* Wt, const float* bias, float* X, const float* lnw,
                                     const float* lnb, float* H, float* scratch, size_t scratch_floats, int M, int N, int K,
                                     float eps, void* stream) {
    VITSEG_CHECK_ARG(A && Wt && bias && X && lnw && lnb && H && scratch, VITSEG_EINVAL, "linear_resln_f32_small: null pointer");
    SGemm g{};
    g.A = A; g.W = Wt; g.C = scratch;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    g.splits = small_splits(N, K);
    g.split_stride = (size_t)M * N;
    VITSEG_CHECK_ARG(scratch_floats >= g.splits * g.split_stride, VITSEG_EWORKSPACE, "linear_resln_f32_small: scratch %zu < %zu floats",
                     scratch_floats, g.splits * g.split_stride);
    if (int rc = launch_gemm_f32s(g, SE_PARTIAL, SA_PLAIN, (hipStream_t)stream)) return rc;
    SRows r{};
    r.X = X; r.partial = scratch; r.split_stride = g.split_stride; r.splits = g.splits; r.bias = bias;
    r.lnw = lnw; r.lnb = lnb; r.H = H; r.rows = M; r.Mp = M; r.Np = M; r.D = N; r.ln_rows = M; r.embed = 0; r.eps = eps;
    return launch_resln(r, (hipStream_t)stream);
}

int vitseg_op_dgrad_f32_small(const float* dY, const float* Wt, const float* R, float* dX, float* scratch, size_t scratch_floats,
                              int M, int Nd, int Kd, int epilogue, void* stream) {
    VITSEG_CHECK_ARG(dY && Wt && dX, VITSEG_EINVAL, "dgrad_f32_small: null pointer");
    VITSEG_CHECK_ARG(epilogue == EPI_BIAS || (epilogue == EPI_DGELU && R), VITSEG_EINVAL, "dgrad_f32_small: epilogue %d", epilogue);
    SGemm g{};
    g.A = dY; g.W = Wt; g.M = M; g.N = Kd; g.K = Nd; g.lda = Nd; g.ldw = Kd; g.ldc = Kd;
    if (epilogue == EPI_DGELU) {
        VITSEG_CHECK_ARG(small_splits(Kd, Nd) == 1, VITSEG_ESHAPE, "dgrad_f32_small: the dGELU form is the wide one (Kd > Nd)");
        g.C = dX; g.R = R; g.splits = 1;
        return launch_gemm_f32s(g, SE_DGELU, SA_PLAIN_WT, (hipStream_t)stream);
    }
    g.splits = small_splits(Kd, Nd);
    g.split_stride = (size_t)M * Kd;
    VITSEG_CHECK_ARG(g.splits == 1 || (scratch && scratch_floats >= g.splits * g.split_stride), VITSEG_EWORKSPACE, "dgrad_f32_small: scratch");
    g.C = g.splits > 1 ? scratch : dX;
    if (int rc = launch_gemm_f32s(g, SE_PARTIAL, SA_PLAIN_WT, (hipStream_t)stream)) return rc;
    return g.splits > 1 ? launch_slabsum(scratch, g.split_stride, g.splits, dX, (size_t)M * Kd, (hipStream_t)stream) : VITSEG_OK;
}

int vitseg_op_wgrad_f32_small(const float* dY, const float* X, float* dW, int M, int Nd, int Kd, void* stream) {
    VITSEG_CHECK_ARG(dY && X && dW && M > 0, VITSEG_EINVAL, "wgrad_f32_small: null pointer");
    SGemm g{};
    g.A = dY; g.W = X; g.C = dW; g.M = Nd; g.N = Kd; g.K = (M + 31) / 32 * 32; g.kvalid = M;
    g.lda = Nd; g.ldw = Kd; g.ldc = Kd; g.splits = 1;
    return launch_gemm_f32s(g, SE_PARTIAL, SA_TT, (hipStream_t)stream);
}

int vitseg_op_attention_f32_small(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream) {
    return launch_attention_small(qkv, ctx, batch, num_patches, num_heads, (hipStream_t)stream);
}
// its 16-bit form (the small-batch route under VITSEG_BF16 / VITSEG_F16): fp32 q | k | v in, operands rounded in registers,
// products on the wide MFMA, fp32 softmax, context written as 16-bit values
int vitseg_op_attention_h16_small(const float* qkv, void* ctx16, int batch, int num_patches, int num_heads, int f16, void* stream) {
    VITSEG_CHECK_ARG(qkv && ctx16, VITSEG_EINVAL, "attention_h16_small: null pointer");
    return launch_attention_small(qkv, (float*)ctx16, batch, num_patches, num_heads, (hipStream_t)stream, nullptr,
                                  DropArgs{0, 0, 0, 1.f}, f16 ? 2 : 1);
}

int vitseg_op_attention_f32(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream) {
    return launch_attention_f32(qkv, ctx, nullptr, batch, num_patches, num_heads, DropArgs{}, (hipStream_t)stream);
}

int vitseg_op_upsample_argmax(const float* lowres, float* logits, uint8_t* mask, int batch, int C, int g, int S,
                              void* stream) {
    return launch_upsample(lowres, logits, mask, batch, C, g, S, (hipStream_t)stream);
}

int vitseg_op_upsample_bwd(const float* grad_logits, float* grad_lowres, int batch, int C, int g, int S, void* stream) {
    VITSEG_CHECK_ARG(grad_logits && grad_lowres && batch > 0 && C > 0 && g > 0 && S >= g && S % g == 0, VITSEG_EINVAL,
                     "upsample_bwd: null pointer or bad geometry (S must be a multiple of g)");
    return launch_upsample_bwd(grad_logits, grad_lowres, batch, C, g, S, (hipStream_t)stream);
}

}  // extern "C"
