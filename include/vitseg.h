/*
 * vitseg.h -- C ABI of libvitseg.so: the MI355X (gfx950) implementation of the
 * ViT-encoder + conv-segmentation-head hot path of mtumalan/VisionTransformer.
 *
 * The reference has no FFI of its own (it is pure Python on PyTorch); its
 * boundary for this path is the Python class surface
 *     ViTSegmentationModel.__init__/forward      /root/reference/model/CE/classes.py:221-262
 *     LightningViTModel.training_step/...        /root/reference/model/CE/classes.py:264-297
 *     inference post-processing                  /root/reference/model/CE/testViTModel.py:119-126
 * Every entry point below replaces the arithmetic behind one of those calls;
 * the Python mirror of the class surface lives in visiontransformer_amd/ and
 * binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain C types only; no exceptions cross the boundary.
 *  - every function returns 0 on success or a negative VITSEG_E* code; a
 *    human-readable message is kept per thread (vitseg_last_error()).
 *  - the caller owns ALL device memory (parameters, activations, workspace).
 *    No hipMalloc/hipFree/synchronise happens inside a call; all work is
 *    enqueued on the `stream` argument (a hipStream_t passed as void*).
 *  - parameters live in ONE fp32 arena whose layout this library defines
 *    (vitseg_param_offset); data pointers must be 16-byte aligned.
 *  - token rows inside the workspace are laid out "patches first":
 *    row b*Np + t for patch token t of image b, row B*Np + b for the CLS token.
 */
#ifndef VITSEG_H
#define VITSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITSEG_VERSION 110 /* 0.1.1: vitseg_op_attention_bwd_bf16 gained dbias_qkv and a larger scratch (round 4); the
                              small-batch ops (round 5).  Bindings should compare vitseg_version() with the header they were
                              written against. */

enum vitseg_status {
    VITSEG_OK = 0,
    VITSEG_EINVAL = -1,     /* bad argument / null pointer / misalignment */
    VITSEG_ESHAPE = -2,     /* shape the reference would reject (ValueError) or this build cannot run */
    VITSEG_EWORKSPACE = -3, /* workspace too small */
    VITSEG_EHIP = -4        /* a HIP runtime call / launch failed */
};

/* Constructor arguments of ViTSegmentationModel (classes.py:222) + the values
 * the reference hard-codes in its ViTConfig (classes.py:224-235). */
typedef struct vitseg_config {
    int32_t num_classes;       /* C */
    int32_t patch_size;        /* P  (multiple of 4) */
    int32_t hidden_size;       /* D  (multiple of 4) */
    int32_t num_layers;        /* L */
    int32_t num_heads;         /* A;  D / A must be 64 in this build */
    int32_t image_size;        /* S  (reference: 224) */
    int32_t intermediate_size; /* I  (reference: 3072) */
    int32_t num_channels;      /* 3 */
    float layer_norm_eps;      /* 1e-12, configuration_vit.py:58 */
} vitseg_config;

/* Parameter tensors, in arena order.  Per-layer tensors take a layer index. */
enum vitseg_tensor {
    VITSEG_T_CLS = 0,   /* [D]            backbone.embeddings.cls_token */
    VITSEG_T_POS,       /* [N, D]         backbone.embeddings.position_embeddings (row 0 = CLS) */
    VITSEG_T_PATCH_W,   /* [D, 3*P*P]     ...patch_embeddings.projection.weight, K order (c,py,px) */
    VITSEG_T_PATCH_B,   /* [D] */
    VITSEG_T_LN1_W,     /* [D]            layers.i.layernorm_before */
    VITSEG_T_LN1_B,
    VITSEG_T_WQKV,      /* [3D, D]        rows: q_proj, k_proj, v_proj weights stacked */
    VITSEG_T_BQKV,      /* [3D] */
    VITSEG_T_WO,        /* [D, D]         attention.o_proj */
    VITSEG_T_BO,
    VITSEG_T_LN2_W,     /* [D]            layernorm_after */
    VITSEG_T_LN2_B,
    VITSEG_T_W1,        /* [I, D]         mlp.fc1 */
    VITSEG_T_B1,
    VITSEG_T_W2,        /* [D, I]         mlp.fc2 */
    VITSEG_T_B2,
    VITSEG_T_LNF_W,     /* [D]            backbone.layernorm */
    VITSEG_T_LNF_B,
    VITSEG_T_HEAD0_W,   /* [256, 3, 3, D] seg_head.0.weight permuted (out, ky, kx, in) */
    VITSEG_T_HEAD0_B,   /* [256] */
    VITSEG_T_HEAD2_W,   /* [C, 256]       seg_head.2.weight */
    VITSEG_T_HEAD2_B,   /* [C] */
    VITSEG_T_COUNT
};

/* Precision of the encoder arithmetic. */
enum vitseg_precision {
    VITSEG_F32 = 0, /* fp32 storage, fp32-input MFMA (exact fmaf chains): the parity path */
    VITSEG_BF16 = 1, /* bf16 operands / fp32 accumulate MFMA, fp32 residual stream and softmax */
    VITSEG_F16 = 2,  /* IEEE-half operands, otherwise as VITSEG_BF16; inference only (BASELINE configs[4]) */
    VITSEG_F32X3 = 3 /* fp32 storage everywhere; GEMM operands split into (hi, lo) half pairs while staged and multiplied
                        with 3 fp16 MFMAs per product (22-bit operand significands, fp32 accumulate); attention,
                        LayerNorm, softmax as VITSEG_F32.  Inference only.  Domain: GEMM / attention operand values
                        must lie inside the half range (|x| < 65504), larger ones become inf and surface as NaN. */
};

/* Workspace buffers whose contents are defined after vitseg_forward returns
 * (used by the parity tests to read intermediate stages). */
enum vitseg_buffer {
    VITSEG_BUF_TOKENS = 0, /* fp32 [B*Np + B, D] residual stream after the last layer */
    VITSEG_BUF_LOWRES,     /* fp32 [B, C, g, g]  low-resolution logits (seg_head output) */
    VITSEG_BUF_COUNT
};

int vitseg_version(void);
const char* vitseg_last_error(void);
/* Dispatcher switches for A/B measurements and tests (which kernel family takes a GEMM, precomputed dropout words on or
 * off, ...): process-wide, read by the launch path with one atomic load.  Names (case-insensitive): no_f32p, no_p8,
 * no_h16p, no_ragged_p8, no_dropmask, dropw_limit_mb, upsample_global, bf16_tiles, f32p_noinl, gn, no_mask2, no_small,
 * small_variant, small_max_rows, conv_dma.  Each starts from the environment variable VITSEG_<NAME> as it was when the library was loaded (the launch
 * path itself never calls getenv): a numeric value is taken as is (VITSEG_NO_P8=0 leaves the switch off), an empty or
 * non-numeric value of an on/off switch means 1.
 * The reference has no counterpart: its dispatch is ATen's. */
int vitseg_set_option(const char* name, long long value);
int vitseg_get_option(const char* name, long long* value);

/* ---- parameter arena (replaces nn.Module parameter storage, classes.py:222-244) ---- */
int vitseg_param_count(const vitseg_config* cfg, size_t* n_floats);
int vitseg_param_offset(const vitseg_config* cfg, int tensor, int layer, size_t* offset_floats, size_t* numel);

/* fp32 arena -> bf16 shadow arena (same offsets, 2 bytes/elt); needed before a VITSEG_BF16 forward. */
int vitseg_cast_params_bf16(const float* params, void* params_bf16, size_t n_floats, void* stream);
/* fp32 arena -> split arena for VITSEG_F32X3 (optional; same size and offsets as the fp32 arena: every 4 values become
 * 4 hi halves | 4 scaled lo halves).  Passed in the params_bf16 slot it spares the GEMMs the weight half of the
 * operand splitting; with NULL there they split the fp32 weights themselves. */
int vitseg_cast_params_split(const float* params, void* params_split, size_t n_floats, void* stream);
/* same for IEEE half (VITSEG_F16); the shadow arena is passed in the params_bf16 slot of vitseg_forward */
int vitseg_cast_params_f16(const float* params, void* params_f16, size_t n_floats, void* stream);

/* Which kernels vitseg_forward takes for a call of this shape (host arithmetic, no GPU work): 1 = the small-batch route (fp32 below
 * 16 384 token rows: the reference's batch 4 x 224x224 and the worker's single image; bf16 / fp16 below 3 200 rows at up to 400
 * tokens, 2 400 at the other key-split lengths), 0 = the large-batch kernels; negative: a vitseg_status.  Results are bit-identical
 * for every batch size INSIDE one route; the two routes sum in different orders (both within the parity gate). */
int vitseg_forward_route(const vitseg_config* cfg, int batch, int precision);
/* ---- forward (replaces ViTSegmentationModel.forward, classes.py:246-262, and the
 *      sigmoid->argmax post-processing of testViTModel.py:122-126) ---- */
int vitseg_query_workspace(const vitseg_config* cfg, int batch, int precision, size_t* bytes);
int vitseg_workspace_offset(const vitseg_config* cfg, int batch, int precision, int buffer, size_t* offset_bytes,
                            size_t* bytes);

/* x: fp32 NCHW [batch, 3, S, S] on device.  logits (fp32 [batch, C, S, S]) and mask
 * (uint8 [batch, S, S], = argmax_c sigmoid(logits), first maximal index) may each be NULL.
 * params_bf16 (the 16-bit shadow arena in the format of `precision`) is only read when precision != VITSEG_F32. */
int vitseg_forward(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x, int batch,
                   int precision, float* logits, uint8_t* mask, void* workspace, size_t workspace_bytes, void* stream);

/* ---- single-operator entry points (same kernels the forward uses; exported so each
 *      stage can be checked against the oracle in isolation) ---- */
int vitseg_op_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps,
                            void* stream);
/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias); epi: 0 none, 1 erf-GELU, 2 + R[M,N] (R may alias C), 3 ReLU */
int vitseg_op_linear_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N,
                         int K, int epilogue, void* stream);
/* the same with the training forms of the fp32 forward: aux (optional, epilogue 1) receives the pre-activation
 * A.W^T + bias; dropout_p > 0 (epilogue 2): C = R + dropout(A.W^T + bias) with the counter-based mask of
 * (dropout_seed, dropout_stream, row, column), as vitseg_forward_train applies it (reference: hidden dropout,
 * transformers/models/vit/modeling_vit.py:276,283). */
int vitseg_op_linear_f32_ex(const float* A, const float* W, const float* bias, const float* R, float* C, float* aux,
                            int M, int N, int K, int epilogue, float dropout_p, uint32_t dropout_seed,
                            uint32_t dropout_stream, void* stream);
/* qkv: [B*Np + B, 3*A*64] rows as in the workspace; ctx: [B*Np + B, A*64] */
int vitseg_op_attention_f32(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream);
/* ---- the small-batch fp32 route (fewer than 2048 token rows per forward; vitseg_forward takes it by itself) ----
 * One linear layer as that route computes it.  The reduction is cut into vitseg_small_splits(N, K) chunks -- a function of
 * the layer's shape only, so an output's bits do not depend on M -- each chunk one fp32 fmaf chain, the chunk sums added in
 * chunk order, then the bias.
 * epilogue 0 / 1 (bias / bias + exact GELU; needs vitseg_small_splits(N, K) == 1): C[M, N] written directly.
 * vitseg_op_linear_resln_f32_small (the o_proj / fc2 step of a pre-LN block, modeling_vit.py:266-286): chunk slabs into
 * `scratch` (>= vitseg_small_splits(N, K) * M * N floats), then ONE row kernel: X += chunk sums + bias (in place),
 * H = LayerNorm(X; lnw, lnb, eps). */
int vitseg_small_splits(int N, int K);
int vitseg_op_linear_f32_small(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int epilogue,
                               void* stream);
int vitseg_op_linear_resln_f32_small(const float* A, const float* W, const float* bias, float* X, const float* lnw,
                                     const float* lnb, float* H, float* scratch, size_t scratch_floats, int M, int N, int K,
                                     float eps, void* stream);
/* the activation gradient of a linear layer as the small-batch training step computes it (the reference trains at batch 4 x
 * 224x224, model/CE/trainCurrentViTmodel.py:57):  dX[M, Kd] = dY[M, Nd] . W[Nd, Kd]  with the nn.Linear weight read as it lies
 * (the reduction runs down its rows).  epilogue 0: plain (K-chunk slabs in `scratch`, >= vitseg_small_splits(Kd, Nd) * M * Kd
 * floats, summed in chunk order); epilogue 5: dX *= gelu'(R) with R[M, Kd] the saved pre-activation (the wide form). */
int vitseg_op_dgrad_f32_small(const float* dY, const float* W, const float* R, float* dX, float* scratch, size_t scratch_floats,
                              int M, int Nd, int Kd, int epilogue, void* stream);
/* ... and its weight gradient  dW[Nd, Kd] = dY[M, Nd]^T . X[M, Kd]  (both operands token-major as they lie; the M token rows are
 * the reduction, one fp32 fmaf chain in row order) */
int vitseg_op_wgrad_f32_small(const float* dY, const float* X, float* dW, int M, int Nd, int Kd, void* stream);
/* diagnostics: vitseg_op_linear_f32_small with per-block time stamps written by the kernel (8 words per block: s_memrealtime
 * at entry / exit, s_memtime at entry / after the prologue / after the K loop / at exit, HW_ID, XCC_ID; `stamps` must hold
 * 8 words per launched block) and `lds_pad` extra bytes of LDS per block (limits the blocks per CU).  tools/small_stamps.py. */
int vitseg_dbg_linear_f32_small(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int epilogue,
                                unsigned long long* stamps, int lds_pad, void* stream);
/* the 16-bit form of that route's linears (vitseg_forward takes it for VITSEG_BF16 / VITSEG_F16 below 16 384 token rows when the
 * sequence length is one of the key-split attention kernel's): A [M, K] and W [N, K] as raw bf16 (f16 = 0) or IEEE half bits, the
 * same kernels on v_mfma_f32_32x32x16_*, fp32 accumulate.  epilogue 0: C fp32 = acc + bias; 1: C 16-bit = gelu(acc + bias) (the
 * next GEMM's operand); 2 (any shape): the vitseg_small_splits(N, K) chunk slabs into scratch, C fp32 = chunk sums in chunk order
 * + bias (scratch: (splits + 1) * M * N floats). */
int vitseg_op_linear_h16_small(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int epilogue, int f16,
                               float* scratch, size_t scratch_floats, void* stream);
/* attention core for short sequences (same arguments and layout as vitseg_op_attention_f32) */
int vitseg_op_attention_f32_small(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream);
/* its 16-bit form (what the route runs under VITSEG_BF16 / VITSEG_F16): fp32 q | k | v in; q, k, the probabilities and v rounded
 * to bf16 (f16 = 0) or IEEE half in registers and multiplied on v_mfma_f32_32x32x16_*, fp32 accumulate and softmax; ctx16
 * [Mt, D] written as 16-bit values (o_proj's operand) */
int vitseg_op_attention_h16_small(const float* qkv, void* ctx16, int batch, int num_patches, int num_heads, int f16, void* stream);
/* bf16 operands (A, W as raw bf16 bits), fp32 accumulate; bias and R fp32.  C is bf16 for epilogues 0/1
 * (tensors that feed the next MFMA) and fp32 for epilogue 2 (the residual stream). */
int vitseg_op_linear_bf16(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N,
                          int K, int epilogue, void* stream);
int vitseg_op_attention_bf16(const void* qkv, void* ctx, int batch, int num_patches, int num_heads, void* stream);
/* fp32 in / fp32 out through the split-operand fp16 MFMA path of VITSEG_F32X3 (same arguments as linear_f32) */
int vitseg_op_linear_f32x3(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N,
                           int K, int epilogue, void* stream);
int vitseg_op_attention_f32x3(const float* qkv, float* ctx, int batch, int num_patches, int num_heads, void* stream);
/* IEEE-half variants of the two above (operands as raw fp16 bits) */
int vitseg_op_linear_f16(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K,
                         int epilogue, void* stream);
int vitseg_op_attention_f16(const void* qkv, void* ctx, int batch, int num_patches, int num_heads, void* stream);
/* The 16-bit linear layer with every epilogue / dispatch option the training path uses (exported so the parity
 * tests can reach each shape-selected tile variant): epilogue 0 bias, 1 bias+GELU (aux != NULL: also stores
 * gelu'(pre-activation) in the 16-bit format -- what the backward multiplies by), 2 residual R fp32 [M,N] +
 * dropout(acc + bias) with fp32 output, 5 acc * R with R = that saved 16-bit derivative [M,N].  f16: IEEE half instead of bf16.  thin_rows > 0: the last thin_rows rows (the CLS rows of the
 * patches-first layout) go through the split-K side launch (scratch: fp32 partials).  dropout_p > 0 (epilogue 2):
 * keep(seed, stream, row, col) of csrc/common.hpp, i.e. hidden dropout of modeling_vit.py:276,283. */
int vitseg_op_linear_h16_ex(const void* A, const void* W, const float* bias, const void* R, void* C, void* aux, int M,
                            int N, int K, int epilogue, int f16, int thin_rows, float* scratch, size_t scratch_floats,
                            float dropout_p, uint32_t dropout_seed, uint32_t dropout_stream, float* colsum_out,
                            float* colsum_scratch, void* stream);
/* colsum_out (epilogue 5, bf16; optional): the column sums of C [N] -- the bias gradient the backward needs next -- produced
 * from the GEMM epilogue's per-tile partial sums; colsum_scratch: vitseg_op_colsum_scratch_floats(M, N) floats. */
size_t vitseg_op_colsum_scratch_floats(int M, int N);
/* bf16 weight gradient dW[M,N] (fp32) = dY^T X with dY = [K tokens][M], X = [K tokens][N] bf16 row-major (the form
 * autograd's linear backward meets: both operands lie token-major, the reduction runs over the token rows); split over
 * the token rows, fp32 partials in `scratch` (>= vitseg_op_wgrad_bf16_scratch_floats floats), fixed-order reduce.
 * zeros: >= 256 zero bytes on the device. */
size_t vitseg_op_wgrad_bf16_scratch_floats(int M, int N, int K);
int vitseg_op_wgrad_bf16(const void* dY, const void* X, float* dW, float* scratch, const void* zeros, int M, int N, int K,
                         void* stream);
/* lowres fp32 [B, C, g, g] -> logits fp32 [B, C, S, S] and/or mask uint8 [B, S, S] */
int vitseg_op_upsample_argmax(const float* lowres, float* logits, uint8_t* mask, int batch, int C, int g, int S,
                              void* stream);
/* the adjoint (what autograd derives for F.interpolate(..., mode="bilinear", align_corners=False), classes.py:260):
 * grad_logits fp32 [B, C, S, S] -> grad_lowres fp32 [B, C, g, g]; S a multiple of g; deterministic */
int vitseg_op_upsample_bwd(const float* grad_logits, float* grad_lowres, int batch, int C, int g, int S, void* stream);

/* ---- loss (replaces nn.CrossEntropyLoss()(logits, y), classes.py:268,280) ----
 * lowres: fp32 [B, C, g, g] low-resolution logits (VITSEG_BUF_LOWRES after vitseg_forward); target: class
 * indices [B, S, S], int64 (torch.long, as the reference passes them) or uint8.  Writes the mean loss to
 * *loss (device fp32).  scratch: >= vitseg_ce_scratch_bytes() device bytes.  grad_logits (optional, fp32
 * [B, C, S, S]) receives d loss / d logits. */
size_t vitseg_ce_scratch_bytes(int batch, int S);
int vitseg_ce_loss(const float* lowres, const void* target, int target_is_u8, float* grad_logits, void* scratch,
                   float* loss, int batch, int C, int g, int S, void* stream);

/* ---- training (replaces autograd behind LightningViTModel.training_step, classes.py:276-285, and
 *      torch.optim.Adam(lr=1e-5).step(), classes.py:296-297).  dropout_p (the reference trains with 0.1,
 *      classes.py:233-234) is applied at the four sites of HF ViT (embeddings, attention probabilities,
 *      attention output, MLP output) with a counter-based generator: the mask is a pure function of
 *      (dropout_seed, layer, site, element), so the SAME (p, seed) must be passed to vitseg_forward_train
 *      and vitseg_backward of one step; torch's RNG stream cannot be matched.  VITSEG_BF16 = mixed precision:
 *      bf16 MFMA operands (params_bf16 shadow arena, bf16 saved activations), fp32 master parameters,
 *      residual stream, LayerNorm / softmax statistics and gradients.
 * vitseg_forward_train saves every activation the backward needs inside `workspace`
 * (vitseg_train_workspace bytes; it must stay untouched until vitseg_backward has run) and optionally
 * writes the fp32 logits.  vitseg_backward takes EITHER integer targets (fused CE: writes the mean loss to
 * *loss) OR the gradient of an arbitrary loss w.r.t. the logits (fp32 [B, C, S, S]) and fills `grads`, an
 * arena-shaped fp32 buffer (same offsets as the parameters).  loss_scale multiplies the gradient of the fused CE loss
 * at its source (1 / accumulate_grad_batches in a gradient-accumulation loop, what Lightning applies to every micro-batch
 * loss); the value written to *loss is not scaled. */
int vitseg_train_workspace(const vitseg_config* cfg, int batch, int precision, size_t* bytes);
int vitseg_forward_train(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x,
                         int batch, int precision, float dropout_p, uint64_t dropout_seed, float* logits,
                         void* workspace, size_t workspace_bytes, void* stream);
int vitseg_backward(const vitseg_config* cfg, const float* params, const void* params_bf16, const float* x, int batch,
                    int precision, float dropout_p, uint64_t dropout_seed, const void* target, int target_is_u8,
                    const float* grad_logits, float* grads, float* loss, float loss_scale, void* const* bucket_events,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Gradient buckets for overlapping the data-parallel all-reduce with the backward (the reference
 * trains single-process, this replaces what DDP would do behind trainer.fit, trainCurrentViTmodel.py:97-101).
 * The gradient arena splits into vitseg_grad_bucket_count() = L + 2 contiguous ranges in the order the backward
 * finishes them: 0 = final norm + seg_head, 1 .. L = encoder layers L-1 .. 0, L + 1 = embeddings.
 * vitseg_backward records bucket_events[i] (hipEvent_t, created by the caller; NULL array or NULL entries = skip)
 * on `stream` right after the last kernel that writes bucket i, so a communication stream can wait on it and
 * reduce that range while the rest of the backward still runs. */
int vitseg_grad_bucket_count(const vitseg_config* cfg);
int vitseg_grad_bucket_range(const vitseg_config* cfg, int bucket, size_t* offset_floats, size_t* n_floats);
/* ---- pre-processing (replaces transforms.Resize((S, S)) + transforms.ToTensor() on the PIL image,
 *      trainCurrentViTmodel.py:48-51 / testViTModel.py:92-97, and the mask side Resize(NEAREST) + value->class
 *      remap + F.interpolate(nearest), classes.py:76-83, 273-274).  Bit-exact with Pillow's 8-bit two-pass
 *      antialiased bilinear resampling (libImaging/Resample.c) and with its NEAREST (Geometry.c).
 * Host side (no GPU touched): vitseg_resize_taps = taps per output sample of one axis; vitseg_resize_coeffs fills
 * bounds[out][2] = (first source index, tap count) and kk[out][taps] (22-bit fixed point) for that axis;
 * vitseg_nearest_index fills the source index of every destination sample (mode 0 = Image.resize(NEAREST),
 * 1 = F.interpolate(mode='nearest')).  The caller uploads the tables once per (source size, S).
 * Device side: vitseg_preprocess_u8 turns n RGB images uint8 [n, H, W, 3] into fp32 [n, 3, S, S] in [0, 1];
 * x* tables (and scratch >= n * rows * S * 3 bytes) are needed when W != S, y* tables when H != S;
 * [row_first, row_first + rows) = the source rows the vertical pass touches (ybounds[0][0] .. last bound).
 * vitseg_resize_nearest_u8 gathers uint8 [n, H, W] -> [n, out_h, out_w] through the index tables and an optional
 * 256-entry LUT, writing uint8 or int64 (torch.long targets). */
int vitseg_resize_taps(int in_size, int out_size);
int vitseg_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk);
int vitseg_nearest_index(int in_size, int out_size, int mode, int32_t* idx);
int vitseg_preprocess_u8(const uint8_t* img, int n, int H, int W, int S, const int32_t* xbounds, const int32_t* xk, int xksize,
                         const int32_t* ybounds, const int32_t* yk, int yksize, int row_first, int rows, uint8_t* scratch,
                         float* out, void* stream);
int vitseg_resize_nearest_u8(const uint8_t* src, int n, int H, int W, const int32_t* yidx, const int32_t* xidx, int out_h,
                             int out_w, const uint8_t* lut, int out_is_i64, void* out, void* stream);
/* the same gather for int64 class-index maps (torch.long targets): LightningViTModel._resize_target,
 * model/CE/classes.py:273-274 = F.interpolate(y[:, None].float(), size, mode='nearest').long() with the mode-1 tables;
 * writes int64 (the reference's dtype) or uint8 (what vitseg_backward / vitseg_ce_loss read at a quarter of the bytes) */
int vitseg_resize_nearest_i64(const int64_t* src, int n, int H, int W, const int32_t* yidx, const int32_t* xidx, int out_h,
                              int out_w, int out_is_i64, void* out, void* stream);

/* ---- soft PAED loss for C classes (replaces softmax + one_hot + paed_loss_multiclass_soft and its autograd in the
 *      17-class LightningViTModel of model/PAED, classes.py:336-369, 449-478).  logits: fp32 [B, C, H, W]; target:
 *      class indices [B, H, W] (int64 or uint8); sigma = 3 and class_penalty = 1 are the reference's defaults.  Writes
 *      the scalar loss and, when grad_logits != NULL, d loss / d logits (fp32 [B, C, H, W]).  scratch:
 *      vitseg_paed_scratch_bytes() device bytes. */
size_t vitseg_paed_scratch_bytes(int batch, int C, int H, int W);
int vitseg_paed_multiclass_loss(const float* logits, const void* target, int target_is_u8, int batch, int C, int H, int W,
                                float sigma, int class_penalty, void* scratch, float* loss, float* grad_logits, void* stream);

/* ---- loss tail of the binary PAED trainer (replaces sigmoid + F.binary_cross_entropy + dice_loss + paed_loss_soft
 *      and their autograd in PAEDTrainer._forward_step_paed, model/PAED/classes.py:608-701).  logits fp32 [B, 1, H, W];
 *      mask fp32 0/1 [B, H, W] (already resized to the prediction); sdf_ext / sdf_int fp32 [B, sdf_h, sdf_w] (resized
 *      bilinearly on the fly).  out8 (device, 8 floats): loss = bce + 0.1 dice + 5 |paed|, bce, dice, paed, then the
 *      counts tp, fp, fn and #((p > 0.5) == mask) for the trainer's logged metrics.  grad_logits (optional) receives
 *      d loss / d logits.  scratch: vitseg_paed_binary_scratch_bytes() device bytes. */
size_t vitseg_paed_binary_scratch_bytes(int batch, int H, int W);
int vitseg_paed_binary_loss(const float* logits, const float* mask, const float* sdf_ext, const float* sdf_int, int sdf_h,
                            int sdf_w, int batch, int H, int W, void* scratch, float* out8, float* grad_logits,
                            void* stream);

/* ---- evaluation statistics (replaces the per-image numpy loops of datasetTestViTmodel.py:193-219) ----
 * pred: uint8 [n, S, S] class masks; gt: uint8 [n, gt_h, gt_w] label maps, nearest-resized on the fly through
 * yidx/xidx (device tables of S entries each; NULL when the sizes already match).  counts: int64 [n, 3, 256] =
 * per label value |gt == v & pred == v|, |gt == v|, |pred == v|; accuracy, IoU, Dice and the class sets follow from
 * these integers exactly. */
int vitseg_eval_counts(const uint8_t* pred, const uint8_t* gt, int n, int S, int gt_h, int gt_w, const int32_t* yidx,
                       const int32_t* xidx, int64_t* counts, void* stream);

/* one Adam step over a flat fp32 buffer (torch.optim.Adam semantics, weight_decay 0, amsgrad off);
 * step is 1-based; gradients are multiplied by grad_scale first (1/world for summed all-reduce). */
int vitseg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n_floats, float lr,
                     float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* the same with torch.optim.AdamW's decoupled weight decay (params *= 1 - lr * weight_decay in front of the update): what
 * PAEDTrainer.configure_optimizers builds (model/PAED/classes.py:536-548, AdamW(lr=1e-4), weight_decay 1e-2 by default) */
int vitseg_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n_floats, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* fp32 GEMM with explicit operand forms, exported for the parity tests of the backward GEMMs:
 * C[M,N] = A . W with A N-form [M][K] (ta = 0) or T-form [K][M] (ta = 1), W N-form [N][K] (tb = 0) or
 * T-form [K][N] (tb = 1); epilogue 0 = plain, 5 = multiply by gelu'(R[M,N]). */
int vitseg_op_gemm_f32(const float* A, const float* W, const float* R, float* C, int M, int N, int K, int ta, int tb,
                       int epilogue, void* stream);
int vitseg_op_attention_bwd_f32(const float* qkv, const float* dctx, float* ctx_out, float* lse_out, float* scratch,
                                float* dqkv, int batch, int num_patches, int num_heads, void* stream);
/* the same pair for short sequences, as the fp32 training step of the small-batch route runs it (num_patches + 1 <= 400:
 * the reference's 197 tokens): key-split forward saving the log-sum-exp, then ONE backward launch (dq blocks | dk, dv blocks,
 * delta formed inside; no scratch).  dropout_p > 0: the counter-based mask of csrc/common.hpp on the probabilities. */
int vitseg_op_attention_bwd_f32_small(const float* qkv, const float* dctx, float* ctx_out, float* lse_out, float* dqkv,
                                      int batch, int num_patches, int num_heads, float dropout_p, uint32_t dropout_seed,
                                      uint32_t dropout_stream, void* stream);
/* bf16 attention core forward + backward (dropout_p > 0: the counter-based mask of csrc/common.hpp on the attention
 * probabilities, stream id = layer * 8 + 1); ctx_out bf16 [Mt, D], lse_out fp32 [B, A, Np + 1], dqkv bf16 [Mt, 3D].
 * dropmask_words (optional, vitseg_attention_dropmask_bytes; needs num_patches % 128 == 0 and dropout_p > 0): the keep
 * bits are generated once into this buffer and read by the three kernels -- the path vitseg_forward_train /
 * vitseg_backward take -- instead of being hashed per element in each of them; the masks are the same bits.
 * scratch: vitseg_attention_bwd_scratch_floats() floats (delta [B, A, Np + 1] + the per-block partial sums of the CLS
 * token's own gradients and of the column sums).
 * dbias_qkv (optional, fp32 [3 D]): the column sums of dqkv over all rows = the gradient of the fused q|k|v bias
 * (modeling_vit.py:207-222, qkv_bias=True), taken from the kernels' fp32 accumulators (patch rows) and the stored CLS rows. */
size_t vitseg_attention_dropmask_bytes(int batch, int num_patches, int num_heads);
size_t vitseg_attention_bwd_scratch_floats(int batch, int num_patches, int num_heads);
int vitseg_op_attention_bwd_bf16(const void* qkv, const void* dctx, void* ctx_out, float* lse_out, float* scratch,
                                 void* dqkv, int batch, int num_patches, int num_heads, float dropout_p,
                                 uint32_t dropout_seed, uint32_t dropout_stream, void* dropmask_words, float* dbias_qkv,
                                 void* stream);
/* scratch of vitseg_op_layernorm_bwd_f32: per-block partial sums of dw / db (the block count depends on rows and on the device) */
size_t vitseg_op_layernorm_bwd_scratch_floats(int rows, int D);
int vitseg_op_layernorm_bwd_f32(const float* x, const float* w, const float* g, const float* dres_in, float* dres_out,
                                float* dw, float* db, float* scratch, int rows, int D, float eps, void* stream);
/* the form the fp32 training step of the small-batch route uses (same arithmetic and bits for dres_out / dw / db): g arrives as
 * g_splits K-chunk slabs, g = slab 0 + slab 1 + ... (slab s at g + s * g_stride floats; 1 = plain); br_dbias != NULL: also the
 * gradient entering the NEXT dropped residual branch of the backward walk -- br_out = mask * dres_out (written only when
 * dropout_p > 0; hidden dropout of csrc/common.hpp with the given stream id), br_dbias [D] = its column sums (= column sums of
 * dres_out when dropout_p == 0): that branch's bias gradient.  scratch: vitseg_op_layernorm_bwd_scratch_floats(rows, D). */
int vitseg_op_layernorm_bwd_f32_small(const float* x, const float* w, const float* g, size_t g_stride, int g_splits,
                                      const float* dres_in, float* dres_out, float* dw, float* db, float* scratch, int rows,
                                      int D, float eps, float* br_out, float* br_dbias, float dropout_p, uint32_t dropout_seed,
                                      uint32_t dropout_stream, void* stream);

/* ---- measurement hooks (bench.py's roofline object) ----
 * While enabled, vitseg_forward brackets every kernel launch of the hot path with a pair of
 * hipEvents on the launch stream; vitseg_forward_train / vitseg_backward bracket the GEMMs and
 * the attention kernels of the encoder layers (the VITSEG_K_TRAIN_* kinds).  vitseg_profile_collect synchronises those events (the only
 * call in this library that blocks) and returns, for one kernel kind, the summed device time,
 * the number of launches and the algorithmic work of those launches (FLOPs for the MFMA kinds,
 * HBM bytes for the bandwidth-bound kinds).  Process-global and not re-entrant: a debugging
 * facility, off by default. */
enum vitseg_kernel_kind {
    VITSEG_K_GEMM_BIAS = 0, /* gemm kernel, plain A, bias epilogue      (QKV projection) */
    VITSEG_K_GEMM_GELU,     /* gemm kernel, plain A, bias+GELU          (mlp.fc1) */
    VITSEG_K_GEMM_RESADD,   /* gemm kernel, plain A, bias+residual      (o_proj, mlp.fc2) */
    VITSEG_K_GEMM_PATCH,    /* gemm kernel, patchify loader, +pos       (patch embedding) */
    VITSEG_K_GEMM_CONV3,    /* gemm kernel, 3x3 im2col loader, ReLU     (seg_head.0) */
    VITSEG_K_ATTENTION,     /* flash attention core (patch queries) + CLS-query kernel */
    VITSEG_K_LAYERNORM,     /* bytes */
    VITSEG_K_HEAD1X1,       /* bytes */
    VITSEG_K_UPSAMPLE,      /* bytes */
    /* training step (vitseg_forward_train / vitseg_backward), encoder layers only: FLOPs */
    VITSEG_K_TRAIN_GEMM_FWD, /* the four forward linears of a layer */
    VITSEG_K_TRAIN_DGRAD,    /* activation-gradient GEMMs (incl. the weight transposes they consume) */
    VITSEG_K_TRAIN_WGRAD,    /* weight-gradient GEMMs (incl. their split-K reduction) */
    VITSEG_K_TRAIN_ATTN_FWD, /* attention forward (patch + CLS query kernels) */
    VITSEG_K_TRAIN_ATTN_BWD, /* attention backward (delta, dQ, dK/dV, CLS kernels) */
    VITSEG_K_COUNT
};
int vitseg_profile_enable(int on);
int vitseg_profile_collect(int kind, double* total_ms, int64_t* launches, double* work);

#ifdef __cplusplus
}
#endif
#endif /* VITSEG_H */
