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
    EPI_POS = 4      // C = acc + bias + pos[1 + m % Np]              (a3 position embeddings)
};

struct GemmArgs {
    const void* A;
    const void* W;      // [N, K] row-major (nn.Linear weight layout)
    const float* bias;  // [N]
    const float* R;     // residual [M, ldc] (EPI_RESADD) or position table [Np+1, N] (EPI_POS)
    void* C;
    int M, N, K;
    int lda, ldc;
    // geometry for A_PATCH / A_CONV3 / EPI_POS
    int S, P, g, Np, Cin, D;
    const void* zeros;  // >= 128 zero bytes (bf16 A_CONV3: source of the padding taps)
};

int launch_gemm_f32(const GemmArgs& a, int amode, int epi, hipStream_t s);
int launch_gemm_bf16(const GemmArgs& a, int amode, int epi, hipStream_t s);

// LayerNorm over the last dim (a4); out_bf16 selects the bf16-output variant.
int launch_layernorm(const float* x, const float* w, const float* b, void* y, int rows, int D, float eps,
                     bool out_bf16, hipStream_t s);

// Multi-head self-attention core (a6) on the patches-first row layout.
int launch_attention_f32(const float* qkv, float* ctx, int B, int Np, int A, hipStream_t s);
int launch_attention_bf16(const void* qkv, void* ctx, int B, int Np, int A, hipStream_t s);

// seg_head.2 (1x1 conv) on the ReLU'd mid features -> NCHW low-res logits (a11)
int launch_head1x1(const float* F, const float* W2, const float* b2, float* Z, int B, int Np, int C, hipStream_t s);
// bilinear upsample + optional sigmoid->argmax mask (a12 + a14)
int launch_upsample(const float* Z, float* logits, uint8_t* mask, int B, int C, int g, int S, hipStream_t s);

// CE loss on the (virtually) upsampled logits (a13); optional full-resolution gradient G
size_t ce_partial_count(int B, int S);
int launch_ce_loss(const float* Z, const void* target, int target_is_u8, float* G, double* partial, float* loss, int B,
                   int C, int g, int S, hipStream_t s);

// CLS token rows of the embedding output: X[B*Np + b] = cls + pos[0]  (a3)
int launch_cls_rows(const float* cls, const float* pos, float* X, int B, int Np, int D, hipStream_t s);

int launch_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s);

}  // namespace vitseg
