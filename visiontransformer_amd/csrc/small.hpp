// The SMALL-BATCH regime of the fp32 path: fewer than SMALL_MAX_ROWS token rows per forward -- the regime the reference
// itself runs and publishes in (batch 4 x 224x224: 788 rows at P = 16, 3 140 at P = 8, 12 548 at P = 4;
// model/CE/datasetTestViTmodel.py:97-109,174-186) and the worker's single-image call (197 rows;
// model/CE/testViTModel.py:92-126).  At these sizes the GEMMs of the large-batch path are a fraction of one round (or a
// few ragged rounds) of its 256x128 / 128x128 tiles; the kernels declared here cut the same arithmetic differently
// (gemm_f32s.hip, rows_small.hip, attention_small.hip) and vitseg_api.hip:forward_small strings them together.
// Where the boundary sits is a measurement (profiles/r05_ref_grid_threshold.txt): the route wins up to the largest
// published configuration (12 548 rows: 0.74 of the fp32 peak against 0.72) and loses to the persistent kernel at the
// headline's 32 800 rows (0.85).
//
// The route has a 16-bit form (VITSEG_BF16 / VITSEG_F16 inference below small_max_rows_16 token rows): the SAME kernels with the
// four linears of a block and the attention products on v_mfma_f32_32x32x16_* (SGemm::h16, SRows::h_fmt, attn_small_kernel<1|2>),
// the residual stream, the stored q | k | v, the patch embedding and the head as in fp32.
//
// Summation order (what makes the results independent of the batch size inside the regime): a linear layer's reduction is
// cut into small_splits(N, K) chunks -- a function of the layer's shape only, never of M -- each chunk is one fp32 fmaf
// chain in the k order of gemm.hip's kernels (k = 32 kt + 8 j + 4 h + e), and the chunk sums are added in chunk order,
// then the bias, then the residual.  Tile shapes are chosen per M (small_plan) and do not enter the order.
#pragma once
#include "kernels.hpp"

namespace vitseg {

constexpr int SMALL_MAX_ROWS = 16384;

enum SmallEpi {
    SE_PARTIAL = 0,   // C[s] = chunk sum s (no bias): consumed by resln_kernel / headfin_kernel
    SE_BIAS = 1,      // C = acc + bias            (fused QKV projection; one chunk)
    SE_GELU = 2,      // C = gelu_erf(acc + bias)  (fc1; one chunk); SGemm::aux (optional) receives acc + bias (saved for the backward)
    SE_DGELU = 3,     // C = acc * gelu'(R)        (backward through the MLP activation; one chunk, no bias)
    SE_RELU = 4       // C = max(acc + bias, 0)    (the 3x3 head conv as ONE chain over its nine taps, SA_CONV3_ALL)
};
enum SmallAMode {
    SA_PLAIN = 0,   // A[m * lda + k]
    SA_CONV3 = 1,   // chunk s = tap (ky, kx) of the 3x3 head conv: rows of the token-major map shifted by the tap, zero outside
    SA_PATCH = 2,   // im2col of the NCHW image: row (b, gy, gx), k = (c, py, px)
    SA_PLAIN_WT = 3,// A plain, W in T-form: W[k * ldw + n] (the activation-gradient GEMMs read nn.Linear weights as they lie)
    SA_CONV3_ALL = 5,// the whole 3x3 conv in one launch and ONE fmaf chain per output, k = (ky, kx, d) as gemm.hip's implicit GEMM
                    // walks it (bit-identical to that kernel; K = D per tap, 9 D in all): the training forward's head conv, whose
                    // ReLU mask must not depend on a summation order (profiles/r05_notes.md)
    SA_TT = 4       // both in T-form: C[i][j] = sum_r A[r * lda + i] W[r * ldw + j], r < kvalid (weight gradients dW = dY^T X: the
                    // reduction runs over the token rows of both operands; K = kvalid rounded up to 32, the tail reads as zeros)
};

struct SGemm {
    const float* A;
    const float* W;      // [N, ldw] row-major (nn.Linear layout); conv3: (out, ky, kx, in) = [256, 9 D]
    const float* bias;   // [N] (SE_BIAS / SE_GELU)
    float* C;            // [M, ldc], SE_PARTIAL: slab s at C + s * split_stride
    float* aux;          // SE_GELU: optional [M, ldc] pre-activation output
    const float* R;      // SE_DGELU: [M, ldc] saved pre-activation
    int M, N, K;         // K: the whole reduction (conv3: per tap = D)
    int lda, ldw, ldc;
    int splits;          // chunks of K / splits values each (conv3: 9 taps)
    size_t split_stride; // floats
    int tiles_m, tiles_n, variant;   // filled by the launcher (small_plan)
    int kvalid;                      // SA_TT: token rows present (0: K)
    int kh;                          // direct epilogues: equal pieces the reduction is summed in (small_pieces(K)); filled by the launcher
    int g, Np, S, P, Cin;            // geometry of SA_CONV3 / SA_PATCH
    // diagnostics (tools/small_stamps.py; null in the product path): 8 words per block -- s_memrealtime at entry and exit,
    // s_memtime at entry / after the prologue / after the K loop / at exit, HW_ID, XCC_ID
    unsigned long long* stamps;
    int lds_pad;                     // extra dynamic LDS bytes per block (experiments: forces fewer blocks per CU)
    int h16;                         // 0: fp32 operands; 1 / 2: A and W hold bf16 / fp16 values (K, lda, ldw count VALUES; SA_PLAIN; partial /
                                     // bias epilogues write fp32, the GELU epilogue writes the 16-bit format: the next GEMM's operand)
};

// chunks of a reduction of length K feeding N outputs per row (shape-only rule, see above)
inline int small_splits(int N, int K) {
    if (N > K || K % 32) return 1;             // wide outputs (QKV, fc1): enough tiles without a split, and fc1 needs GELU in the epilogue
    // chunk = the largest whole number of 32-value K steps up to 256 values (K <= 1024) / 512 values that divides K:
    // o_proj (K = 768) 3 chunks, fc2 (K = 3072) 6, the QKV activation gradient (K = 2304) 6 chunks of 384
    const int steps = K / 32, cmax = K <= 1024 ? 8 : 16;
    int c = cmax < steps ? cmax : steps;
    while (steps % c) --c;
    return steps / c;
}
// pieces of a direct-epilogue reduction (one chunk): out = ((h0 + h1) + h2) + h3 + bias, h_i = the fmaf chain over the i-th
// quarter of K -- a shape-only rule like small_splits (the one-image kernel computes the quarters on four waves at once)
inline int small_pieces(int K) { return K % 128 == 0 ? 4 : (K % 64 == 0 ? 2 : 1); }
int launch_gemm_f32s(SGemm a, int epi, int amode, hipStream_t s);

// Rows kernel between the GEMMs: x = residual + (sum of the chunk slabs in order + bias), LayerNorm of the new row.
//   embed != 0: the residual is the position embedding (rows < Mp: pos[1 + row % Np]); rows >= Mp are the CLS rows
//   (cls + pos[0], no slabs).   ln_rows: rows [0, ln_rows) get their LayerNorm written to H (the final norm skips CLS).
struct SRows {
    float* X;                 // [rows, D] residual stream (read unless embed, written)
    const float* partial;     // splits slabs of [rows, D]
    size_t split_stride;
    int splits;
    const float* bias;        // [D]
    const float* pos;         // embed: [Np + 1, D]
    const float* cls;         // embed: [D]
    const float* lnw;
    const float* lnb;
    float* H;                 // [rows, D]
    int rows, Mp, Np, D, ln_rows, embed;
    float eps;
    const float* Xres;        // residual source when it is not X itself (the training forward keeps every block input); null: X
    DropArgs drop;            // training: x = residual + dropout(chunk sum + bias)  (hidden dropout, modeling_vit.py:276,283)
    int h_fmt;                // H as 0 fp32, 1 bf16, 2 fp16 (the 16-bit route: the next GEMM's operand), [rows, D] of that type
};
int launch_resln(const SRows& a, hipStream_t s);
// out[i] = slab 0 [i] + slab 1 [i] + ... (chunk order): the activation gradients that a LayerNorm / attention backward reads
int launch_slabsum(const float* partial, size_t split_stride, int splits, float* out, size_t n, hipStream_t s);

// seg_head tail: F = relu(sum of the 9 tap slabs + b0) (256 mid channels), Z[b, c, y, x] = W2[c] . F + b2[c]
// F_out (optional): the 256 ReLU'd mid features per pixel, kept for the backward
int launch_headfin(const float* partial, size_t split_stride, const float* b0, const float* W2, const float* b2, float* Z,
                   int B, int Np, int C, hipStream_t s, float* F_out = nullptr);

// softmax(q k^T / 8) v for short sequences: one block per 32 queries of one (image, head), the four waves split the keys
// lse (optional): [B, A, Np + 1] log2-domain log-sum-exp per query, saved for the backward; dr: dropout of the probabilities
// ctx_fmt: ctx as 0 fp32, 1 bf16, 2 fp16 values (the 16-bit route: o_proj's operand)
int launch_attention_small(const float* qkv, float* ctx, int B, int Np, int A, hipStream_t s, float* lse = nullptr,
                           DropArgs dr = DropArgs{0, 0, 0, 1.f}, int ctx_fmt = 0);
// its backward (attention_bwd_small.hip): dqkv = (dq | dk | dv) from dctx, the saved qkv / ctx / lse; ONE launch, the
// blocks of the first half produce dk / dv (32 keys each, the four waves split the queries), those of the second dq
// (32 queries each, the waves split the keys); delta = rowsum(dctx o ctx) is formed inside
int launch_attention_bwd_small(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dqkv, int B,
                               int Np, int A, DropArgs dr, hipStream_t s);
// LayerNorm backward of the small-batch training step (backward.hip): layernorm_bwd's arithmetic with g taken as g_splits K-chunk
// slabs (summed in chunk order), and -- br_dbias != null -- the gradient entering the next dropped residual branch,
// br_out = mask * dres_out (written only when br_drop.thresh != 0; otherwise the branch reads dres_out), with its column sums
// br_dbias = that branch's bias gradient.  scratch: layernorm_bwd_scratch_floats(rows, D).
int launch_layernorm_bwd_small(const float* x, const float* w, const float* g, size_t g_stride, int g_splits,
                               const float* dres_in, float* dres_out, float* dw, float* db, float* scratch, int rows, int D,
                               float eps, hipStream_t s, float* br_out = nullptr, DropArgs br_drop = DropArgs{0, 0, 0, 1.f},
                               float* br_dbias = nullptr);
// the sequence lengths attn_small_kernel takes in the forward of the route (by the SHAPE only: a row's bits must not depend on the
// batch): every length but the whole-64-key-tile ones that fill 128-query blocks (512x512 at P = 16: 1025 tokens), which take
// attention_f32's tail-free loop (tools/attn_small_probe.py, profiles/r05_attn_small_probe.txt)
inline bool attn_small_infer(int Np) { return !(Np % 64 == 0 && Np + 1 > 400 && Np + 1 <= 2048); }
// row limit of the route's 16-bit form (bf16 / fp16 operands in the four linears, everything else as in fp32): by measurement
inline long small_max_rows_16(int N) { return N <= 400 ? 3200 : 2400; }   // profiles/r05_h16_route_probe.txt: 197 tokens even at batch 16, 785 at batch 4
// the sequence lengths the short-sequence attention kernels take in the TRAINING step (a function of the shape only)
inline bool attn_small_train(int Np) { return Np + 1 <= 400; }

// true when vitseg_forward takes the small-batch route (fewer than small_max_rows() token rows: SMALL_MAX_ROWS unless the
// small_max_rows option says otherwise)
long small_max_rows();
bool small_applies(const vitseg_config* cfg, int batch, int precision);

}  // namespace vitseg
