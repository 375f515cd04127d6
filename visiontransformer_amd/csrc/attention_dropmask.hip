// Keep bits of the attention-probability dropout (modeling_vit.py:184, p = 0.1 in training) as words, generated ONCE
// per layer and step and read by the forward and both backward kernels (layout: common.hpp, attn_dropmask_words).
// The bits are exactly drop_keep(drop_key(seed, stream, bh (Np + 1) + q), k, thresh) -- the definition the hashing
// kernels and tests/dropout_ref.py use -- so the two paths are interchangeable element for element.
//
// A thread owns one hash pair (keys 2 i, 2 i + 1) of a (bh, 32-query group) and walks the 32 queries from the last to
// the first: the compare result enters its word as the carry of w + w + carry (v_addc), i.e. one instruction per bit;
// the 16-bit fields are compared in place (v_cmp_ge_u16 on the low half, a 32-bit compare against thresh << 16 for the
// high half).  ~5.5 VALU ops per element, once, instead of 3.5 - 7 hash ops per element in each of three kernels.
// HBM-light (4 B per 32 elements), VALU-bound: ~0.11 ms per layer at B = 64, Np = 1024, A = 12.
#include "kernels.hpp"

namespace vitseg {
namespace {

__global__ __launch_bounds__(256) void attn_dropmask_kernel(unsigned* __restrict__ W, int Np, int A, DropArgs dr) {
    const int nb = Np >> 5;
    const int qg = blockIdx.x % nb, bh = blockIdx.x / nb;
    const int lane = threadIdx.x & 63;
    // lane j (mod 32) hashes query j's row key; the loop below broadcasts them with v_readlane
    const unsigned myk = drop_key(dr.seed, dr.stream, (unsigned)(bh * (Np + 1) + qg * 32 + (lane & 31)));
    const unsigned thr_lo = dr.thresh, thr_hi = dr.thresh << 16;
    unsigned* out = W + ((size_t)(bh * nb + qg) * nb) * 32;
    // the 32 row keys as scalars, once (inside the pair loop they cost a v_readlane and -- one scalar operand per VOP3 -- a
    // v_mov per hash: the pair's 24-bit product is formed once and the key ADDED, the same bits as drop_pair_hash)
    unsigned kj[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) kj[j] = __builtin_amdgcn_readlane(myk, j);
    for (int pi = threadIdx.x; pi < (Np >> 1); pi += 256) {
        unsigned w0 = 0, w1 = 0;
        const unsigned prod = __umul24((unsigned)pi, 0x9E3779u);
#pragma unroll
        for (int j = 31; j >= 0; --j) {
            unsigned h = prod + kj[j];
            h ^= h >> 15;
            h *= 0x85EBCA6Bu;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("v_cmp_ge_u16 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w0) : "v"(h), "v"(thr_lo) : "vcc");
            asm volatile("v_cmp_ge_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w1) : "v"(h), "v"(thr_hi) : "vcc");
#else
            w0 = (w0 << 1) | ((h & 0xffffu) >= thr_lo);
            w1 = (w1 << 1) | (h >= thr_hi);
#endif
        }
        const int k0 = 2 * pi, kk = k0 & 31;                 // even key: register r = (kk & 3) + 4 (kk >> 3), half (kk >> 2) & 1
        const int pos = 2 * ((kk & 3) + 4 * (kk >> 3)) + ((kk >> 2) & 1);
        out[(k0 >> 5) * 32 + pos] = w0;
        out[(k0 >> 5) * 32 + pos + 2] = w1;                  // key k0 + 1 = register r + 1, same half
    }
}

}  // namespace

int launch_attn_dropmask(unsigned* W, int B, int Np, int A, DropArgs dr, hipStream_t s) {
    VITSEG_CHECK_ARG(W && B > 0 && A > 0 && Np > 0 && Np % 128 == 0 && dr.thresh > 0 && dr.thresh < 65536, VITSEG_EINVAL,
                     "attn_dropmask: Np %% 128 != 0 or dropout off");
    hipLaunchKernelGGL(attn_dropmask_kernel, dim3((unsigned)(B * A * (Np / 32))), dim3(256), 0, s, W, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_dropmask");
    return VITSEG_OK;
}

}  // namespace vitseg
