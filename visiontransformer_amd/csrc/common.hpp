// Shared device/host helpers for libvitseg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitseg.h"

namespace vitseg {

constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// thread-local error message (vitseg_last_error)
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// Dispatcher switches (A/B experiments, tests): a process-wide table read with one relaxed atomic load -- no getenv() on
// the launch path.  Initial values come from the environment variable of the same name ("VITSEG_" + upper case) ONCE, when
// the library is loaded; vitseg_set_option() (include/vitseg.h) changes them afterwards.
enum Opt {
    OPT_NO_F32P = 0,       // fp32 linears on gemm.hip's tile kernel instead of gemm_f32p.hip
    OPT_NO_P8,             // 16-bit linears / weight gradients on the round-1 kernels instead of gemm_p8.hip
    OPT_NO_H16P,           // bias-epilogue 16-bit linears on gemm_p8.hip instead of gemm_h16p.hip
    OPT_NO_RAGGED_P8,      // a ragged last row tile never rides in the persistent kernel's last round
    OPT_NO_DROPMASK,       // attention dropout hashed per element instead of read from precomputed keep-bit words
    OPT_DROPW_LIMIT_MB,    // keep-bit words are kept per layer up to this many MiB in all (-1: the built-in limit)
    OPT_UPSAMPLE_GLOBAL,   // upsample kernel without the LDS-staged source rows
    OPT_BF16_TILES,        // round-1 16-bit tile choice: 0 by shape, 1 small (128x128), 2 large (256x128), 3 xl (256x256)
    OPT_F32P_NOINL,        // gemm_f32p: every epilogue at its tile's end
    OPT_GN,                // column-group width of the tile order (0: by shape)
    OPT_NO_MASK2,          // two-class mask-only upsample through the general kernel instead of upsample_mask2_kernel
    OPT_NO_SMALL,          // fp32 forwards of fewer than 2048 token rows on the large-batch kernels instead of the small-batch route (small.hpp)
    OPT_SMALL_VARIANT,     // gemm_f32s tile variant 1..5 for every launch (0: small_plan picks)
    OPT_SMALL_MAX_ROWS,    // fp32 forwards below this many token rows take the small-batch route (0: the built-in SMALL_MAX_ROWS)
    OPT_CONV_DMA,          // the large-batch fp32 3x3 head conv on the LDS-DMA kernel (gemm_f32s SA_CONV3_ALL: same bits, -10 % time, 1.8x the L2-miss bytes) instead of gemm.hip's implicit GEMM
    OPT_COUNT
};
long opt(int id);

#define VITSEG_CHECK_ARG(cond, code, ...) \
    do {                                  \
        if (!(cond)) {                    \
            ::vitseg::set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

#define VITSEG_LAUNCH_CHECK(what)                               \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return ::vitseg::hip_fail(e__, what); \
    } while (0)

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give each
// XCD a contiguous run of the logical tile list so neighbouring tiles (which share
// operand panels) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Attention grids are 1-D: block -> (row tile, head, image) with all row tiles of one (image, head) pair on ONE XCD and
// resident together, so its K/V (or Q/dO) panels are fetched into a single L2 once.  (With a 3-D grid the hardware
// deals the tiles of a pair round-robin over the 8 XCDs: measured 5.7x over-fetch, profiles/r01_traffic_f32.json.)
struct AttnTile {
    int rt, head, b;
};
__device__ __forceinline__ AttnTile attn_tile(int ntiles, int A) {
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = t / ntiles;
    AttnTile r;
    r.rt = t - bh * ntiles;
    r.b = bh / A;
    r.head = bh - r.b * A;
    return r;
}

// ---- dropout: counter-based keep/drop decision (no state, identical in forward and backward) ----
// keep(seed, stream, major, minor) = half[minor & 1](mix(minor >> 1, key)) >= thresh,  key = fmix32(seed ^ stream*C1 ^ major*C2),
// thresh = round(p * 2^16): the low / high 16 bits of one hash serve the even / odd element of a pair.
// `stream` = layer * 8 + site (0 embeddings, 1 attention probabilities, 2 attention output, 3 MLP output);
// (major, minor) = (row, column) of the tensor, or (bh * N + query, key) for attention probabilities.
// The same function is restated in numpy by the tests (tests/dropout_ref.py) to inject identical masks
// into the oracle.
__host__ __device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ unsigned drop_key(unsigned seed, unsigned stream, unsigned major) {
    return fmix32(seed ^ (stream * 0x9E3779B1u) ^ (major * 0x85EBCA77u));
}
// low 24 bits of a times the 24-bit constant k, plus c (one v_mad_u32_u24: full rate, unlike the 32-bit multiply)
__host__ __device__ __forceinline__ unsigned mad24(unsigned a, unsigned k, unsigned c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, k) + c;
#else
    return (unsigned)(((unsigned long long)(a & 0xffffffu) * (k & 0xffffffu) + c) & 0xffffffffull);
#endif
}
// One hash decides TWO neighbouring elements (minor = 2j, 2j + 1): 16 bits each against thresh = round(p * 2^16).
// The per-row key is a full fmix32 (once per row); the per-pair mixer is one 24-bit multiply-add, one xor-shift and one
// 32-bit multiply (4 VALU ops; v_mul_lo_u32 issues at full rate on gfx950, tools/probes/imul_rate.hip) -- it is
// evaluated for every element of the N x N attention probabilities.  Statistics (keep rate, pair / neighbour / row /
// diagonal / layer correlations at the 1e-3 level over 12 M samples, column and row means) are those of independent
// bits (tests/test_host_cpu.py).
__host__ __device__ __forceinline__ unsigned drop_pair_hash(unsigned key, unsigned pair) {
    unsigned h = mad24(pair, 0x9E3779u, key);
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    return h;  // low half decides minor = 2 pair, high half minor = 2 pair + 1
}
__host__ __device__ __forceinline__ bool drop_keep(unsigned key, unsigned minor, unsigned thresh) {
    const unsigned h = drop_pair_hash(key, minor >> 1);
    return ((minor & 1u) ? (h >> 16) : (h & 0xffffu)) >= thresh;
}
struct DropArgs {
    unsigned thresh;  // round(p * 65536); 0 = dropout off
    unsigned seed;
    unsigned stream;
    float scale;      // 1 / (1 - p)
};

// Attention-probability keep bits as precomputed words (attention_dropmask.hip), patch queries x patch keys of one
// (image, head) pair bh, Np a multiple of 128, nb = Np / 32:
//   W[((bh * nb + qg) * nb + kblk) * 32 + 2 r + h]   bit j = keep(query 32 qg + j, key 32 kblk + kappa(r, h)),
//   kappa(r, h) = (r & 3) + 8 (r >> 2) + 4 h   (the key an MFMA 32x32 accumulator register r holds on lane half h).
// The 64-bit pair (2 r, 2 r + 1) is therefore the lane mask of accumulator register r in the kernels that put a QUERY
// on each lane (forward, dQ): a scalar load and one v_cndmask per element replace the hash.  The dK/dV kernel (a KEY on
// each lane) reads its key's word per 32-query group and tests bit = query.
inline size_t attn_dropmask_words(int B, int Np, int A) { return (size_t)B * A * (Np / 32) * Np; }
// The four 512-bit scalar loads of one 64-key tile's lane masks (2 blocks of 32 keys x 16 accumulator registers), issued
// by hand at the END of the previous loop iteration: SMEM shares lgkmcnt with the LDS and returns out of order, so a
// scalar load in flight turns every LDS wait behind it into a full drain -- placed before the tile-end barrier it has
// the whole staging wait to land.  mask_wait() (top of the iteration) is the only wait the consumer needs.
typedef unsigned long u64x8 __attribute__((ext_vector_type(8)));
struct TileMasks {
    u64x8 v[4];  // v[2 kb + (r >> 3)][r & 7] = lane mask of register r of key block kb
    __device__ __forceinline__ void load(const unsigned long* p) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile(
            "s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\t"
            "s_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0"
            : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3])
            : "s"(p));
#endif
    }
    __device__ __forceinline__ void wait() {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]));
#endif
    }
    __device__ __forceinline__ unsigned long reg(int kb, int r) const { return v[2 * kb + (r >> 3)][r & 7]; }
};
__device__ __forceinline__ float mask_select(unsigned long lanes, float v) {  // lanes: wave-uniform 64-bit mask
#if defined(__HIP_DEVICE_COMPILE__)
    float o;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(lanes));
    return o;
#else
    return v;
#endif
}

// Reductions over the 64 lanes of a wave, the result on every lane (wave-uniform).  Six DPP steps inside the vector ALU
// -- lane ^ 1, lane ^ 2 (quad permutes), mirror within 8 and within 16 lanes (every lane then holds its row's value), the
// row totals chained by row_bcast:15 / row_bcast:31 into lane 63 -- and one v_readlane.  (The __shfl_xor butterfly this
// replaces in round 4 compiles to six DEPENDENT ds_bpermute_b32 per reduction, an LDS round trip each: LayerNorm forward
// does two of them per row, its backward four.)
#if defined(__HIP_DEVICE_COMPILE__)
#define VITSEG_DPP(x, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
    v += VITSEG_DPP(v, 0xb1, 0xf);    // quad_perm [1, 0, 3, 2]
    v += VITSEG_DPP(v, 0x4e, 0xf);    // quad_perm [2, 3, 0, 1]
    v += VITSEG_DPP(v, 0x141, 0xf);   // row_half_mirror
    v += VITSEG_DPP(v, 0x140, 0xf);   // row_mirror
    v += VITSEG_DPP(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3 (the other rows add the 0 passed as `old`)
    v += VITSEG_DPP(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), on every lane of the row: the first four steps of wave_sum
__device__ __forceinline__ float row16_sum(float v) {
    v += VITSEG_DPP(v, 0xb1, 0xf);
    v += VITSEG_DPP(v, 0x4e, 0xf);
    v += VITSEG_DPP(v, 0x141, 0xf);
    v += VITSEG_DPP(v, 0x140, 0xf);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    // (a disabled row keeps its own value: max(v, v))
#define VITSEG_DPP_KEEP(x, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (x)), __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0xb1, 0xf));
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0x4e, 0xf));
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0x141, 0xf));
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0x140, 0xf));
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0x142, 0xa));
    v = fmaxf(v, VITSEG_DPP_KEEP(v, 0x143, 0xc));
#undef VITSEG_DPP_KEEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef VITSEG_DPP
#else
__device__ __forceinline__ float wave_sum(float v) { return v; }
__device__ __forceinline__ float row16_sum(float v) { return v; }
__device__ __forceinline__ float wave_max(float v) { return v; }
#endif

// exact (erf) GELU, activations.py:78-83:  gelu(u) = u Phi(u),  Phi(u) = (1 + erf(u / sqrt 2)) / 2.
// Evaluated through the COMPLEMENTARY error function of a = |u| / sqrt 2,  erfc(a) = 2^(-a Q(a)):
//     Phi(u) = 1 - erfc(a) / 2  (u >= 0),   erfc(a) / 2  (u < 0),
// Q = an 8th-degree polynomial fitted (weighted minimax on [0, 5], tools/fit_gelu_erfc.py) to -log2(erfc(a)) / a.
// Why not libm's erff (what rounds 1-3 used, through a branch-free restatement): 1 + erf(x) cancels for x < 0, so that
// form is good to 6e-8 ABSOLUTE in Phi -- 8e-6 of the result at u = -3 -- and costs ~34 vector instructions (two
// polynomials, an emulated expf, a select); erfc keeps the tail's RELATIVE precision and is one polynomial, one v_exp_f32
// and a select: 18 instructions.  On the fp32 path every one of them is time added to the fp32 MFMAs.
// Error against fp64 over ALL 2^32 inputs (tools/probes/gelu_check.hip, profiles/r04_gelu_check_all_2e32_inputs.txt):
// max |error| / max(|gelu|, 1e-2) = 8.1e-7 (up to ~4400 ulps of the tiny results near u = -5, where the libm form is off by
// 8e-6 of the same scale).  Once a q exceeds 126 (|u| > ~11.4, gelu < 1e-36) v_exp_f32 flushes the product to 0 instead of
// returning a denormal: gelu is then exactly -0 or u.  gelu(+inf) = +inf, gelu(-huge) = -0, NaN stays NaN.
__device__ __forceinline__ float gelu_phi(float u) {   // Phi(u) = (1 + erf(u / sqrt 2)) / 2
#pragma clang fp contract(off)   // every fused multiply-add is spelled out: all kernels must agree bit for bit
    const float a = fabsf(u) * 0.70710678118654752440f;
    const float c = fminf(a, 5.0f);   // the fit's range; beyond it Q stays at Q(5) = 7.86 and erfc decays as 2^(-7.86 a)
    float q = __uint_as_float(0xb6e5811du);   // (erfc(5) = 1.5e-12: what is left of Phi there is below every fp32 ulp that matters)
    q = fmaf(c, q, __uint_as_float(0x38d55d79u));
    q = fmaf(c, q, __uint_as_float(0xba2329bbu));
    q = fmaf(c, q, __uint_as_float(0x3ae4efa2u));
    q = fmaf(c, q, __uint_as_float(0x3a233c25u));
    q = fmaf(c, q, __uint_as_float(0xbce796e6u));
    q = fmaf(c, q, __uint_as_float(0x3e181a64u));
    q = fmaf(c, q, __uint_as_float(0x3f6b1c07u));
    q = fmaf(c, q, __uint_as_float(0x3fd05f5fu));
    const float e = __builtin_amdgcn_exp2f(-(a * q));                  // erfc(a); 0 once a q > 149
    return u >= 0.f ? fmaf(e, -0.5f, 1.0f) : 0.5f * e;                 // a NaN takes the second side and stays one
}
__device__ __forceinline__ float gelu_erf(float u) {
#pragma clang fp contract(off)
    return u * gelu_phi(u);
}

// d/du [u * Phi(u)] = Phi(u) + u * phi(u), Phi from the SAME erfc evaluation as gelu_erf (the fp32 training path's forward
// and its derivative are one function; the 1 + erff form this replaces cancels for u < 0: 8e-6 of the tail)
__device__ __forceinline__ float gelu_erf_grad(float u) {
#pragma clang fp contract(off)
    const float pdf = __builtin_amdgcn_exp2f((u * u) * -0.72134752044448170368f);   // exp(-u^2 / 2)
    return fmaf(u * pdf, 0.39894228040143267794f, gelu_phi(u));
}

// GELU alone for tensors that are rounded to 16 bits right away (inference): erf by Abramowitz-Stegun 7.1.28,
// erf x = 1 - (1 + a1 x + ... + a6 x^6)^-16, |abs error| <= 3e-7 -- three orders below the rounding of the result -- at 15
// plain vector instructions.  (The erfc form below needs 16 for GELU alone and measured 5 % slower in fc1's epilogue,
// 358-363 vs 343-345 us at B = 64; it wins when the derivative is wanted too.)
__device__ __forceinline__ float gelu_erf_fast(float u) {
    const float x = fabsf(u) * 0.70710678118654752440f;
    float t = fmaf(x, 0.0000430638f, 0.0002765672f);
    t = fmaf(x, t, 0.0001520143f);
    t = fmaf(x, t, 0.0092705272f);
    t = fmaf(x, t, 0.0422820123f);
    t = fmaf(x, t, 0.0705230784f);
    t = fmaf(x, t, 1.0f);
    t = t * t;
    t = t * t;
    t = t * t;
    t = t * t;
    const float erf_abs = 1.0f - __builtin_amdgcn_rcpf(t);  // t -> inf gives erf = 1
    return 0.5f * u + 0.5f * fabsf(u) * erf_abs;              // u * Phi(u), erf odd
}
// GELU AND its derivative (training: fc1's epilogue saves gelu'(u) for the backward), both rounded to 16 bits right away:
// the erfc form of gelu_erf above with a degree-5 Q fitted on [0, 4.5] (tools/fit_gelu_erfc.py: fit(4.5, 5, 1e-5, 1e-2)):
// |error| <= 1.9e-6 of max(|gelu|, 1e-2) and 6.8e-7 absolute in gelu' -- two orders below the rounding of the results.
// Phi is shared: 22 vector instructions for the pair against 27 with two Abramowitz-Stegun evaluations (rounds 1-3);
// fc1 + GELU + saved derivative at B = 64 on one box: 413-420 vs 434-436 us (the epilogue is VALU-bound).
struct GeluPair {
    float g, d;   // gelu(u), gelu'(u)
};
__device__ __forceinline__ GeluPair gelu_erf_pair_fast(float u) {
    const float a = fabsf(u) * 0.70710678118654752440f;
    const float c = fminf(a, 4.5f);
    float q = __uint_as_float(0xb98ef51du);
    q = fmaf(c, q, __uint_as_float(0x3b8f4085u));
    q = fmaf(c, q, __uint_as_float(0xbd04359fu));
    q = fmaf(c, q, __uint_as_float(0x3e1a8172u));
    q = fmaf(c, q, __uint_as_float(0x3f6af0d8u));
    q = fmaf(c, q, __uint_as_float(0x3fd0616eu));
    const float e = __builtin_amdgcn_exp2f(-(a * q));                      // erfc(a)
    const float phi = u >= 0.f ? fmaf(e, -0.5f, 1.0f) : 0.5f * e;          // Phi(u) = (1 + erf(u / sqrt 2)) / 2
    const float pdf = __builtin_amdgcn_exp2f((u * u) * -0.72134752044448170368f);   // exp(-u^2 / 2)
    return GeluPair{u * phi, fmaf(u * pdf, 0.39894228040143267794f, phi)};   // u Phi(u), Phi(u) + u phi(u)
}

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// two fp32 -> packed bf16x2 (round-to-nearest-even, NaN preserved): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
    typedef float f32v2_t __attribute__((ext_vector_type(2)));
    const f32v2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16v2_t));
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// ---- the two 16-bit operand formats of the MFMA paths ----
// bf16 (VITSEG_BF16: inference and mixed-precision training) is carried as raw bits; IEEE half (VITSEG_F16:
// inference, BASELINE configs[4]) as _Float16 -- a distinct type, so one kernel template is instantiated per format
// and H16<T> supplies the conversions and the 32x32x16 MFMA of that format.  Both round to nearest even.
typedef unsigned short bf16_t;
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <typename H> struct H16;
template <> struct H16<bf16_t> {
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
    static __device__ __forceinline__ unsigned short bits(float f) { return f32_to_bf16(f); }
    static __device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct H16<f16_t> {
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        f16x2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ unsigned short bits(float f) {
        return __builtin_bit_cast(unsigned short, (_Float16)f);
    }
    static __device__ __forceinline__ float lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
    static __device__ __forceinline__ float hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    }
};

}  // namespace vitseg
