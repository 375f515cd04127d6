// Host-side launchers of the gfx950 kernels (one .hip file each).  All launchers only
// enqueue work on `stream`; none allocates or synchronises.
#pragma once
#include "common.hpp"

namespace vitseg {

// ---- how the A operand of a GEMM is addressed ---------------------------------
enum AMode {
    A_PLAIN = 0,  // A[m*lda + k]
    A_PATCH = 1,  // im2col of the NCHW image: row = (b, gy, gx), k = (c, py, px)   (a2)
    A_CONV3 = 2   // im2col of the token-major (NHWC) feature map, 3x3 zero pad 1: k = (ky, kx, d)  (a11)
};
// ---- what happens to the accumulator ------------------------------------------
enum Epi {
    EPI_BIAS = 0,    // C = acc + bias
    EPI_GELU = 1,    // C = gelu_erf(acc + bias)                      (a7 fc1)
    EPI_RESADD = 2,  // C = R + acc + bias   (R may alias C)          (a8 residual adds)
    EPI_RELU = 3,    // C = max(acc + bias, 0)                        (a11 conv3x3)
    EPI_POS = 4,     // C = acc + bias + pos[1 + m % Np]              (a3 position embeddings)
    EPI_DGELU = 5    // C = acc * gelu'(R)                            (backward through the MLP activation)
};

struct GemmArgs {
    const void* A;
    const void* W;      // [N, K] row-major (nn.Linear weight layout)
    const float* bias;  // [N]
    const float* R;     // residual [M, ldc] (EPI_RESADD) or position table [Np+1, N] (EPI_POS)
    void* C;
    int M, N, K;
    int lda, ldc;
    int ldw;            // leading dimension of W (0 = dense: K for N-form, N for T-form)
    void* aux;          // EPI_GELU: optional second output, the pre-activation (saved for backward)
    // EPI_DGELU (16-bit): also produce the column sums of the output rows (the bias gradient of the layer whose
    // pre-activation gradient C is) without re-reading C: the 8-phase kernel writes per-tile partial sums to
    // colsum_scratch (colsum_scratch_floats(M, N) floats), the dispatcher adds the rows that kernel did not cover and
    // reduces in a fixed order into colsum_out[N]
    float* colsum_out;
    float* colsum_scratch;
    // geometry for A_PATCH / A_CONV3 / EPI_POS
    int S, P, g, Np, Cin, D;
    const void* zeros;  // >= 128 zero bytes (bf16 A_CONV3: source of the padding taps)
    int splitk;           // > 1: grid.y slices of the K range, each writing C + y * split_stride
    size_t split_stride;
    // CLS rows by split-K (fp32 / x3 forward, A_PLAIN): when the last `thin_rows` rows (<= 64) follow a whole number of
    // 128-row tiles and the GEMM has few column tiles, they are computed by a second launch cut into K slices (partials in
    // `thin_scratch`, >= thin_scratch_floats(N) floats) and a fixed-order reduce + epilogue, instead of one extra
    // quarter-cost tile per column that prolongs the whole launch by a partial round.
    float* thin_scratch;
    size_t thin_capacity;  // floats available in thin_scratch
    int thin_rows;
    int gn;               // column-group width of the tile order (0 = pick from K; VITSEG_GN overrides for experiments)
    DropArgs drop;        // EPI_RESADD: C = R + dropout(acc + bias)   (hidden dropout, modeling_vit.py:276,283)
    int row_base;         // added to the row index of the dropout hash when a launch covers rows [row_base, row_base + M)
};

// x3: fp32 operands split into half pairs while staged, 3 fp16 MFMAs per product (fp32-grade results, gemm.hip X3)
constexpr int THIN_MAX_SPLITS = 16, THIN_MAX_ROWS = 64;
inline size_t thin_scratch_floats(int max_n) { return (size_t)THIN_MAX_SPLITS * THIN_MAX_ROWS * max_n; }
// Small batches: a GEMM with at most 128 output tiles leaves most of the 512 block slots idle while each tile walks its whole K
// range (a ViT-B/16 forward of ONE 224x224 image took 5.9 ms in fp32).  Such a GEMM is cut into K slices as a whole:
// returns the slice count (0 = do not split); kstep = elements per staged K step (32 fp32, 64 for 16-bit operands).
inline int whole_split(int M, int N, int K, int kstep) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles > 128 || K < 512) return 0;
    int s = K / kstep / 4;
    if (s > 8) s = 8;
    return s >= 2 ? s : 0;
}
int launch_gemm_f32(const GemmArgs& a, int amode, int epi, hipStream_t s, int x3 = 0);  // 1: split A and W, 2: W pre-split
int launch_gemm_f32_bwd(const GemmArgs& a, int amode, int ta, int tb, int epi, hipStream_t s);
size_t wgrad_scratch_floats(int M, int N, int K);
int launch_wgrad_f32(GemmArgs a, float* scratch, hipStream_t s);
int launch_wgrad_bf16_tt(GemmArgs a, float* scratch, hipStream_t s);
int launch_gemm_bf16_train(GemmArgs a, int epi, int out_f32, float* scratch, hipStream_t s);
size_t wgrad_bf16_scratch_floats(int M, int N, int K);
// 16-bit operands, fp32 accumulate; f16 selects IEEE half instead of bf16 (inference formats, common.hpp H16<>)
int launch_gemm_bf16(const GemmArgs& a, int amode, int epi, hipStream_t s, bool f16 = false);
// persistent 8-phase 256x256 kernel for the large plain linear layers (gemm_p8.hip); `applies` = shape / alignment test
int gemm_p8_rounds(int M, int N);
// compute units of the current device (cached per device; 256 when the query fails): grid size of the persistent kernels
int device_num_cus();
bool gemm_p8_applies(const GemmArgs& a, int epi);
int launch_gemm_p8(const GemmArgs& a, int epi, hipStream_t s, bool f16);
// 16-bit persistent kernel for the bias epilogue with a short reduction (gemm_h16p.hip): one wave per SIMD, 128 x 128 per
// wave, a finished tile leaves under the next tile's first K step.
bool gemm_h16p_applies(const GemmArgs& a, int epi);
int launch_gemm_h16p(const GemmArgs& a, hipStream_t s, bool f16);
// persistent 256x128 fp32 kernel for the large plain linear layers of the parity path (gemm_f32p.hip)
bool gemm_f32p_applies(const GemmArgs& a, int epi);
int launch_gemm_f32p(const GemmArgs& a, int epi, hipStream_t s);
// the same kernel in its T-form x T-form guise (weight gradients, split over the token rows)
bool wgrad_p8_applies(const GemmArgs& a);
int wgrad_p8_splits(int M, int N, int K);
int launch_wgrad_p8(GemmArgs a, float* scratch, hipStream_t s);
int launch_splitk_reduce(const float* partial, float* out, size_t n4, int splits, hipStream_t s);

// LayerNorm over the last dim (a4); out_fmt: 0 fp32 output, 1 bf16, 2 IEEE half.
int launch_layernorm(const float* x, const float* w, const float* b, void* y, int rows, int D, float eps,
                     int out_fmt, hipStream_t s);

// Multi-head self-attention core (a6) on the patches-first row layout.
// lse (optional): fp32 [B, A, Np+1] log2-domain log-sum-exp per query (CLS last), saved for the backward
int launch_attention_f32(const float* qkv, float* ctx, float* lse, int B, int Np, int A, DropArgs dr, hipStream_t s,
                         bool x3 = false);
int launch_attention_x3_main(const float* qkv, float* ctx, int B, int Np, int A, hipStream_t s);
// dqkv[Mt,3D] from dctx[Mt,D]; dvec: scratch fp32 [B, A, Np+1]
int launch_attention_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dvec,
                             float* dqkv, int B, int Np, int A, DropArgs dr, hipStream_t s);
// maskw: optional precomputed keep-bit words of this layer's attention dropout (launch_attn_dropmask; common.hpp
// attn_dropmask_words) -- used when dropout is on and Np % 128 == 0, otherwise the kernels hash per element
int launch_attention_bf16(const void* qkv, void* ctx, float* lse, int B, int Np, int A, DropArgs dr, hipStream_t s,
                          bool f16 = false, const unsigned* maskw = nullptr);
// dvec: scratch of attention_bwd_bf16_scratch_floats(B, Np, A) floats (delta + the per-block partials of the CLS token's
// gradients and of the column sums); dbias (optional): [3 D] fp32, the column sums of dqkv over all B (Np + 1) rows = the
// gradient of the fused QKV bias, formed from the kernels' fp32 accumulators instead of by a pass over dqkv
size_t attention_bwd_bf16_scratch_floats(int B, int Np, int A);
int launch_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dvec,
                              void* dqkv, int B, int Np, int A, DropArgs dr, hipStream_t s,
                              const unsigned* maskw = nullptr, float* dbias = nullptr);
int launch_attn_dropmask(unsigned* W, int B, int Np, int A, DropArgs dr, hipStream_t s);

// seg_head.2 (1x1 conv) on the ReLU'd mid features -> NCHW low-res logits (a11)
int launch_head1x1(const float* F, const float* W2, const float* b2, float* Z, int B, int Np, int C, hipStream_t s);
// bilinear upsample + optional sigmoid->argmax mask (a12 + a14)
int launch_upsample(const float* Z, float* logits, uint8_t* mask, int B, int C, int g, int S, hipStream_t s);

// CE loss on the (virtually) upsampled logits (a13); optional full-resolution gradient G
size_t ce_partial_count(int B, int S);
int launch_ce_loss(const float* Z, const void* target, int target_is_u8, float* G, double* partial, float* loss, int B,
                   int C, int g, int S, hipStream_t s, float gscale = 1.0f);

// CLS token rows of the embedding output: X[B*Np + b] = cls + pos[0]  (a3)
int launch_cls_rows(const float* cls, const float* pos, float* X, int B, int Np, int D, hipStream_t s);

int launch_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s, bool f16 = false);
// fp32 -> per 4 values: 4 hi halves | 4 scaled lo halves (same 16 bytes, same offsets): the W operand of VITSEG_F32X3
int launch_cast_split(const float* src, void* dst, size_t n, hipStream_t s);
// dropout on a row-major [rows][cols] fp32 tensor: dst = keep ? src * scale : 0 (dst may alias src; dst_bf16 selects
// a bf16 destination).  Used for the embedding dropout and for masking branch gradients in the backward.
int launch_dropout_rows(const float* src, void* dst, int dst_bf16, int rows, int cols, DropArgs d, hipStream_t s);

// ---- backward pass (backward.hip) ----
size_t colsum_scratch_floats(int M, int N);
int launch_colsum_finish_fused(const void* tail_rows, int tail, int chunk0, float* out, float* scratch, int N, int ld,
                               hipStream_t s);
int launch_colsum(const void* X, int x_is_bf16, float* out, float* scratch, int M, int N, int ld, hipStream_t s);
// the four weight matrices (R[k] x C[k], at element offsets src0[k] + layer * src_stride of the 16-bit shadow arena) of
// every layer, transposed into out[layer][k] (dense, layer stride = sum R C)
int launch_transpose_layers_bf16(const void* arena_lp, void* out, const size_t src0[4], const int R[4], const int C[4],
                                 size_t src_stride, int layers, hipStream_t s);
int launch_transpose_bf16(const void* in, void* out, int R, int C, int ldin, int Rpad, hipStream_t s);
size_t layernorm_bwd_scratch_floats(int rows, int D);
int launch_layernorm_bwd(const float* x, const float* w, const void* g, int g_is_bf16, const float* dres_in,
                         float* dres_out, float* dw, float* db, float* scratch, int rows, int D, float eps,
                         hipStream_t s, void* br_out = nullptr, DropArgs br_drop = DropArgs{}, float* br_dbias = nullptr);
int launch_upsample_bwd(const float* G, float* dZ, int B, int C, int g, int S, hipStream_t s);
size_t head1x1_bwd_scratch_floats(int B, int Np, int C);
int launch_head1x1_bwd(const float* dZ, const float* F, const float* W2, float* dFpre, float* dW2, float* db2,
                       float* scratch, int B, int Np, int C, hipStream_t s);
int launch_im2col3x3(const void* H, int h_is_bf16, float* T, int B, int g, int D, hipStream_t s);
int launch_im2col3x3_bf16(const void* H, void* T, int B, int g, int D, hipStream_t s);
int launch_im2col_patch_bf16(const float* img, void* T, int B, int Cin, int S, int P, hipStream_t s);
int launch_im2col_patch(const float* img, float* T, int B, int Cin, int S, int P, hipStream_t s);
int launch_conv_dgrad_weight(const float* W0, float* Wd, int D, hipStream_t s);
int launch_embed_bwd(const float* dX, float* dpos, float* dcls, int B, int Np, int D, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
                float grad_scale, hipStream_t s, float weight_decay = 0.f);

}  // namespace vitseg
