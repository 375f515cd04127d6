// GEMM on the matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias),  fp32 or bf16 operands.
//
// Replaces aten::addmm / mkldnn_convolution behind nn.Linear / Conv2d in the reference
// (56-65 % of the CPU profile): q/k/v/o projections and MLP
// (transformers/models/vit/modeling_vit.py:207-254), patch embedding (:62-69) and
// seg_head.0 (model/CE/classes.py:241) through the gathering A loaders.
//
// T = float : v_mfma_f32_32x32x2_f32, an exact fp32 fmaf chain (no TF32 on gfx950), 64 cycles per
//             issue per SIMD -> matrix-pipe bound by a wide margin (roofline 157.3 TFLOP/s).
// T = bf16  : v_mfma_f32_32x32x16_bf16, fp32 accumulate, 32 cycles per issue (roofline 2.5 PFLOP/s);
//             8x the FLOPs per staged byte, so this tile shape leans on L2 bandwidth.
//
// Both element types share one structure because a staged operand row is 128 bytes either way:
// block 128x128, BK = 128 B of K per row (32 floats / 64 bf16), 4 waves as 2(M) x 2(N), each wave
// 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Lane half h consumes the 16-byte
// chunks 2j+h (j = 0..3) of a row: for bf16 that chunk IS the 32x32x16 operand (k = 8h..8h+7 of
// k-step j); for fp32 the k index inside a 32x32x2 MFMA is arbitrary as long as A and B agree, so
// the chunk's four floats feed four consecutive MFMAs.  Either way a fragment is ONE ds_read_b128
// from a row-major tile whose chunk index is XOR-swizzled with (row >> 1) & 7 (conflict-free).
// Global->LDS goes through registers (the gathering loaders need per-chunk zero-fill),
// double-buffered in LDS with one barrier per K step; fragments are double-buffered in registers.
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int BM = 128, BN = 128;
constexpr int BKF = 32;    // row length in 4-byte LDS words

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int CE = 4, BKE = 32; };           // elements per chunk / per row
template <> struct Elem<unsigned short> { static constexpr int CE = 8, BKE = 64; };  // bf16 bits
typedef unsigned short bf16_t;

template <typename T>
struct ARow {
    // per-row state of the A loader, computed once (row is fixed for a thread)
    const T* base;  // row base pointer (A_PLAIN / A_PATCH: image base of (b, gy, gx))
    int y, x;           // A_CONV3: pixel coordinates
    bool valid;
};

template <typename T, int AMODE>
__device__ __forceinline__ ARow<T> make_arow(const GemmArgs& p, int m) {
    ARow<T> r;
    r.valid = m < p.M;
    r.y = r.x = 0;
    const T* A = (const T*)p.A;
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

// Branch-free: out-of-range chunks read a safe in-bounds address and are zeroed by a select, so
// the whole K step stays one basic block and the scheduler can spread the loads between MFMAs.
template <typename T, int AMODE>
__device__ __forceinline__ f32x4 load_a(const GemmArgs& p, const ARow<T>& r, int k, bool& ok) {
    ok = r.valid && k < p.K;
    const int kc = min(k, p.K - Elem<T>::CE);
    const T* ptr;
    if (AMODE == A_PLAIN) {
        ptr = r.base + kc;
    } else if (AMODE == A_PATCH) {
        const int pp = p.P * p.P;
        const int c = kc / pp, rem = kc - c * pp;
        const int py = rem / p.P, px = rem - py * p.P;
        ptr = r.base + ((size_t)c * p.S + py) * p.S + px;
    } else {
        const int tap = kc / p.D, d = kc - tap * p.D;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int yy = r.y + ky - 1, xx = r.x + kx - 1;
        const bool in = (unsigned)yy < (unsigned)p.g && (unsigned)xx < (unsigned)p.g;
        ok = ok && in;
        ptr = r.base + (in ? ((ptrdiff_t)(ky - 1) * p.g + (kx - 1)) * p.D : 0) + d;
    }
    return *(const f32x4*)ptr;  // zeroed by the caller when !ok, at LDS-write time (keeps the load in flight)
}

// MI x NI MFMA tiles (32x32) per wave; rows start at a_row0, columns at b_col0 inside the block tile.
//   <2,2>: the regular 2(M) x 2(N) wave grid, 64x64 per wave.
//   <1,1>/<2,1>: "thin" row tiles (<= 32 / <= 64 valid rows: the CLS rows that follow the B*Np patch
//   rows); the 4 waves split the 128 columns so such a tile costs 1/4 (1/2) of a regular one and is
//   scheduled first, instead of adding a whole extra round of blocks to the launch.
template <typename T, typename OutT, int AMODE, int EPI, int MI, int NI>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, float (*lds)[2][BM * BKF], int m0, int n0, int a_row0,
                                          int b_col0) {
    constexpr int CE = Elem<T>::CE, BKE = Elem<T>::BKE;
    constexpr int BK = BKF;  // LDS words per row
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- global -> register staging: thread owns chunk lc of rows lr + 32 i ----
    const int lc = tid & 7, lr = tid >> 3;
    ARow<T> arow[4];
    const T* wrow[4];
    bool wvalid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        arow[i] = make_arow<T, AMODE>(p, m0 + lr + 32 * i);
        const int n = n0 + lr + 32 * i;
        wvalid[i] = n < p.N;
        wrow[i] = (const T*)p.W + (size_t)(wvalid[i] ? n : 0) * p.K;
    }
    f32x4 ra[4], rb[4];
    bool oka[4], okb[4];
    auto gload = [&](int kt) {
        const int k = kt * BKE + lc * CE;
        const int kc = min(k, p.K - CE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = load_a<T, AMODE>(p, arow[i], k, oka[i]);
            rb[i] = *(const f32x4*)(wrow[i] + kc);
            okb[i] = wvalid[i] && k < p.K;
        }
    };
    const int wpos = lr * BK + ((lc ^ ((lr >> 1) & 7)) << 2);  // + 32*i rows -> same swizzle term
    auto swrite = [&](int buf) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4*)&lds[buf][0][wpos + 32 * i * BK] = oka[i] ? ra[i] : z;
            *(f32x4*)&lds[buf][1][wpos + 32 * i * BK] = okb[i] ? rb[i] : z;
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    const int a_off = (a_row0 + li) * BK, b_off = (b_col0 + li) * BK;

    // Fragment registers are double-buffered one MFMA group (16 MFMAs = 1024 matrix-pipe cycles)
    // ahead, so no LDS latency is exposed: group j+1's ds_reads are issued before group j's MFMAs.
    // The loop is rotated around the barrier: group 3 of tile kt runs AFTER the barrier that
    // publishes tile kt+1, with tile kt+1's group-0 fragments already being read.
    f32x4 a[2][MI], b[2][NI];
    auto lfrag = [&](int buf, int j, int slot) {
        const int ch = (((2 * j + lh) ^ sw) << 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[slot][mi] = *(const f32x4*)&lds[buf][0][a_off + mi * 32 * BK + ch];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[slot][ni] = *(const f32x4*)&lds[buf][1][b_off + ni * 32 * BK + ch];
    };
    // one group = the MFMAs fed by one 16-byte chunk per operand: 4 k-steps of 32x32x2 (fp32, quarter
    // q = one float of the chunk) or 1 k-step of 32x32x16 (bf16, issued with quarter 0)
    auto mfma_group = [&](int slot, int e0, int e1) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = e0; e < e1; ++e)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][mi][e], b[slot][ni][e],
                                                                           acc[mi][ni], 0, 0, 0);
        } else {
            if (e0 == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[slot][mi]), __builtin_bit_cast(bf16x8, b[slot][ni]),
                            acc[mi][ni], 0, 0, 0);
            }
        }
    };

    const int KT = (p.K + BKE - 1) / BKE;
    gload(0);
    swrite(0);
    __syncthreads();
    lfrag(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const int kn = min(kt + 1, KT - 1);  // the last step re-stages its own tile: keeps the body branch-free
        // group 0: next tile's global loads are issued here and stay in flight for ~2 groups
        gload(kn);
        lfrag(buf, 1, 1);
        __builtin_amdgcn_sched_barrier(0);  // pin: hipcc otherwise sinks the loads down to their use
        mfma_group(0, 0, 4);
        // group 1
        lfrag(buf, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, 0, 4);
        // group 2: the staged tile is zero-masked and written to the idle LDS buffer mid-group
        lfrag(buf, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        swrite(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 2, 4);
        // group 3: one barrier hands the buffers over, then the next tile's first fragments are read
        __syncthreads();
        lfrag(buf ^ 1, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, 0, 4);
    }

    // ---- epilogue: acc reg r of lane (li, lh) = C[row (r&3) + 8 (r>>2) + 4 lh][col li] ----
    OutT* C = (OutT*)p.C;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + b_col0 + ni * 32 + li;
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + a_row0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r] + bias;
                if (EPI == EPI_GELU) v = gelu_erf(v);
                if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                if (EPI == EPI_RESADD) v = p.R[(size_t)row * p.ldc + col] + v;
                if (EPI == EPI_POS) v += p.R[(size_t)(1 + row % p.Np) * p.N + col];
                if constexpr (sizeof(OutT) == 4)
                    C[(size_t)row * p.ldc + col] = v;
                else
                    C[(size_t)row * p.ldc + col] = f32_to_bf16(v);
            }
        }
    }
}

template <typename T, typename OutT, int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BM * BKF];  // [buffer][A|W][row*32 + swizzled chunk]

    const int wave = threadIdx.x >> 6;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    // the last (possibly thin) row tile is placed first in the logical order
    const int tm = t / tiles_n, tile_n = t - tm * tiles_n;
    const int tile_m = tm == 0 ? tiles_m - 1 : tm - 1;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int rows_valid = p.M - m0;
    if (rows_valid <= 32)
        gemm_tile<T, OutT, AMODE, EPI, 1, 1>(p, lds, m0, n0, 0, wave * 32);
    else if (rows_valid <= 64)
        gemm_tile<T, OutT, AMODE, EPI, 2, 1>(p, lds, m0, n0, 0, wave * 32);
    else
        gemm_tile<T, OutT, AMODE, EPI, 2, 2>(p, lds, m0, n0, (wave >> 1) * 64, (wave & 1) * 64);
}

template <typename T, typename OutT, int AMODE, int EPI>
int launch_one(const GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_kernel<T, OutT, AMODE, EPI>), dim3(tiles), dim3(256), 0, s, a);
    VITSEG_LAUNCH_CHECK("gemm");
    return VITSEG_OK;
}

}  // namespace

int launch_gemm_f32(const GemmArgs& a, int amode, int epi, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 4 == 0, VITSEG_EINVAL, "gemm_f32: bad M/N/K %d %d %d", a.M,
                     a.N, a.K);
    if (amode == A_PLAIN) {
        VITSEG_CHECK_ARG(a.lda % 4 == 0, VITSEG_EINVAL, "gemm_f32: lda %% 4");
        switch (epi) {
            case EPI_BIAS: return launch_one<float, float, A_PLAIN, EPI_BIAS>(a, s);
            case EPI_GELU: return launch_one<float, float, A_PLAIN, EPI_GELU>(a, s);
            case EPI_RESADD: return launch_one<float, float, A_PLAIN, EPI_RESADD>(a, s);
            case EPI_RELU: return launch_one<float, float, A_PLAIN, EPI_RELU>(a, s);
        }
    } else if (amode == A_PATCH && epi == EPI_POS) {
        VITSEG_CHECK_ARG(a.P % 4 == 0, VITSEG_ESHAPE, "patch size must be a multiple of 4");
        return launch_one<float, float, A_PATCH, EPI_POS>(a, s);
    } else if (amode == A_CONV3 && epi == EPI_RELU) {
        VITSEG_CHECK_ARG(a.D % 4 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 4");
        return launch_one<float, float, A_CONV3, EPI_RELU>(a, s);
    }
    set_error("gemm_f32: unsupported amode/epilogue %d/%d", amode, epi);
    return VITSEG_EINVAL;
}

// bf16 operands (A and W), fp32 accumulate.  Output type follows the consumer: bf16 for tensors
// that feed the next MFMA (q|k|v, MLP hidden), fp32 for the residual stream and the head features.
int launch_gemm_bf16(const GemmArgs& a, int amode, int epi, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 8 == 0, VITSEG_EINVAL, "gemm_bf16: bad M/N/K %d %d %d", a.M,
                     a.N, a.K);
    if (amode == A_PLAIN) {
        VITSEG_CHECK_ARG(a.lda % 8 == 0, VITSEG_EINVAL, "gemm_bf16: lda %% 8");
        switch (epi) {
            case EPI_BIAS: return launch_one<bf16_t, bf16_t, A_PLAIN, EPI_BIAS>(a, s);
            case EPI_GELU: return launch_one<bf16_t, bf16_t, A_PLAIN, EPI_GELU>(a, s);
            case EPI_RESADD: return launch_one<bf16_t, float, A_PLAIN, EPI_RESADD>(a, s);
        }
    } else if (amode == A_CONV3 && epi == EPI_RELU) {
        VITSEG_CHECK_ARG(a.D % 8 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 8");
        return launch_one<bf16_t, float, A_CONV3, EPI_RELU>(a, s);
    }
    set_error("gemm_bf16: unsupported amode/epilogue %d/%d", amode, epi);
    return VITSEG_EINVAL;
}

}  // namespace vitseg
