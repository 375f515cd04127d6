// fp32 GEMM on the fp32-input matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias)
//
// Replaces aten::addmm / mkldnn_convolution behind nn.Linear / Conv2d in the reference
// (56-65 % of the CPU profile): q/k/v/o projections and MLP
// (transformers/models/vit/modeling_vit.py:207-254), patch embedding (:62-69) and
// seg_head.0 (model/CE/classes.py:241) through the gathering A loaders.
//
// v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain (no TF32 on gfx950), 64 cycles per
// issue per SIMD, so the kernel is matrix-pipe bound by a wide margin: per 128x128x32
// block step a wave issues 64 MFMAs (4096 cycles) against 16 ds_read_b128 and 8 global
// 16-byte loads.  Roofline: 157.3 TFLOP/s (fp32 matrix peak); HBM traffic is irrelevant.
//
// Tiling: block 128x128, BK = 32 floats (one 128-B line per operand row), 4 waves as
// 2(M) x 2(N), each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).
// The k index inside a 32x32x2 MFMA is arbitrary as long as A and B agree, so lane half
// h consumes k = 8j + 4h + e (j = 0..3, e = 0..3): each lane's four k-steps are ONE
// 16-byte LDS read from a row-major [row][32] tile.  Rows are 128 B, so the 16-B chunk
// index is XOR-swizzled with (row >> 1) & 7 to make ds_read_b128 conflict-free.
// Global->LDS goes through registers (the gathering loaders need per-chunk predicates),
// double-buffered in LDS with one barrier per K step.
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int BM = 128, BN = 128, BK = 32;

template <int AMODE>
struct ARow {
    // per-row state of the A loader, computed once (row is fixed for a thread)
    const float* base;  // row base pointer (A_PLAIN / A_PATCH: image base of (b, gy, gx))
    int y, x;           // A_CONV3: pixel coordinates
    bool valid;
};

template <int AMODE>
__device__ __forceinline__ ARow<AMODE> make_arow(const GemmArgs& p, int m) {
    ARow<AMODE> r;
    r.valid = m < p.M;
    r.y = r.x = 0;
    const float* A = (const float*)p.A;
    if (!r.valid) {
        r.base = A;
        return r;
    }
    if (AMODE == A_PLAIN) {
        r.base = A + (size_t)m * p.lda;
    } else if (AMODE == A_PATCH) {
        const int b = m / p.Np, t = m - b * p.Np;
        const int gy = t / p.g, gx = t - gy * p.g;
        r.base = A + ((size_t)b * p.Cin * p.S + (size_t)gy * p.P) * p.S + (size_t)gx * p.P;
    } else {
        const int b = m / p.Np, t = m - b * p.Np;
        r.y = t / p.g;
        r.x = t - r.y * p.g;
        r.base = A + (size_t)m * p.D;  // centre pixel's token row
    }
    return r;
}

template <int AMODE>
__device__ __forceinline__ f32x4 load_a(const GemmArgs& p, const ARow<AMODE>& r, int k) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (!r.valid || k >= p.K) return z;
    if (AMODE == A_PLAIN) {
        return *(const f32x4*)(r.base + k);
    } else if (AMODE == A_PATCH) {
        const int pp = p.P * p.P;
        const int c = k / pp, rem = k - c * pp;
        const int py = rem / p.P, px = rem - py * p.P;
        return *(const f32x4*)(r.base + ((size_t)c * p.S + py) * p.S + px);
    } else {
        const int tap = k / p.D, d = k - tap * p.D;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int yy = r.y + ky - 1, xx = r.x + kx - 1;
        if ((unsigned)yy >= (unsigned)p.g || (unsigned)xx >= (unsigned)p.g) return z;
        return *(const f32x4*)(r.base + ((ptrdiff_t)(ky - 1) * p.g + (kx - 1)) * p.D + d);
    }
}

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BM * BK];  // [buffer][A|W][row*32 + swizzled chunk]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- global -> register staging: thread owns chunk lc of rows lr + 32 i ----
    const int lc = tid & 7, lr = tid >> 3;
    ARow<AMODE> arow[4];
    const float* wrow[4];
    bool wvalid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        arow[i] = make_arow<AMODE>(p, m0 + lr + 32 * i);
        const int n = n0 + lr + 32 * i;
        wvalid[i] = n < p.N;
        wrow[i] = (const float*)p.W + (size_t)(wvalid[i] ? n : 0) * p.K;
    }
    f32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k = kt * BK + lc * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = load_a<AMODE>(p, arow[i], k);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[i] = (wvalid[i] && k < p.K) ? *(const f32x4*)(wrow[i] + k) : z;
        }
    };
    const int wpos = lr * BK + ((lc ^ ((lr >> 1) & 7)) << 2);  // + 32*i rows -> same swizzle term
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4*)&lds[buf][0][wpos + 32 * i * BK] = ra[i];
            *(f32x4*)&lds[buf][1][wpos + 32 * i * BK] = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    const int a_off = (wm * 64 + li) * BK, b_off = (wn * 64 + li) * BK;

    const int KT = (p.K + BK - 1) / BK;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
        const float* As = lds[buf][0];
        const float* Bs = lds[buf][1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = (((2 * j + lh) ^ sw) << 2);
            f32x4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)&As[a_off + mi * 32 * BK + ch];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const f32x4*)&Bs[b_off + ni * 32 * BK + ch];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < KT) swrite(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc reg r of lane (li, lh) = C[row (r&3) + 8 (r>>2) + 4 lh][col li] ----
    float* C = (float*)p.C;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + li;
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r] + bias;
                if (EPI == EPI_GELU) v = gelu_erf(v);
                if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                if (EPI == EPI_RESADD) v = p.R[(size_t)row * p.ldc + col] + v;
                if (EPI == EPI_POS) v += p.R[(size_t)(1 + row % p.Np) * p.N + col];
                C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

template <int AMODE, int EPI>
int launch_one(const GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_f32_kernel<AMODE, EPI>), dim3(tiles), dim3(256), 0, s, a);
    VITSEG_LAUNCH_CHECK("gemm_f32");
    return VITSEG_OK;
}

}  // namespace

int launch_gemm_f32(const GemmArgs& a, int amode, int epi, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 4 == 0, VITSEG_EINVAL, "gemm_f32: bad M/N/K %d %d %d", a.M,
                     a.N, a.K);
    if (amode == A_PLAIN) {
        VITSEG_CHECK_ARG(a.lda % 4 == 0, VITSEG_EINVAL, "gemm_f32: lda %% 4");
        switch (epi) {
            case EPI_BIAS: return launch_one<A_PLAIN, EPI_BIAS>(a, s);
            case EPI_GELU: return launch_one<A_PLAIN, EPI_GELU>(a, s);
            case EPI_RESADD: return launch_one<A_PLAIN, EPI_RESADD>(a, s);
            case EPI_RELU: return launch_one<A_PLAIN, EPI_RELU>(a, s);
        }
    } else if (amode == A_PATCH && epi == EPI_POS) {
        VITSEG_CHECK_ARG(a.P % 4 == 0, VITSEG_ESHAPE, "patch size must be a multiple of 4");
        return launch_one<A_PATCH, EPI_POS>(a, s);
    } else if (amode == A_CONV3 && epi == EPI_RELU) {
        VITSEG_CHECK_ARG(a.D % 4 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 4");
        return launch_one<A_CONV3, EPI_RELU>(a, s);
    }
    set_error("gemm_f32: unsupported amode/epilogue %d/%d", amode, epi);
    return VITSEG_EINVAL;
}

}  // namespace vitseg
