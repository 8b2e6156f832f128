// NT bf16 GEMM on MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate, fused epilogues.
//
//   C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]      A: activations, B: weights ([out,in] as
//   torch.nn.Linear stores them) - both K-contiguous, which is exactly the MFMA fragment
//   order, so neither operand is ever transposed in memory.
//
// Tile: BM x BN x 64, 256 threads = 4 waves as 2x2, each wave (BM/2)x(BN/2) in 16x16 MFMA
// tiles.  Staging: global_load_dwordx4 -> registers -> ds_write_b128 into a double-buffered
// LDS image whose 16-byte chunks are XOR-swizzled by (row & 7) (128-byte rows would
// otherwise put every ds_read_b128 lane group on two 16-byte slots of the 256-byte bank row).
// The MFMA is issued with the weight fragment as the "A" operand so each lane ends up owning
// 4 CONSECUTIVE output columns of one row (D[i=n][j=m]: j = lane&15, i = 4*(lane>>4)+reg):
// the epilogue stores 8-byte (bf16) / 16-byte (f32) pieces instead of 2-byte scatters.
//
// Epilogues (what the reference does between two Linears, fused):
//   EPI_BF16   C_bf16 = alpha*acc + bias
//   EPI_F32    C_f32  = alpha*acc + bias
//   EPI_RESID  C_f32 += acc + bias                         (x = f(x) + x, residual stream in fp32)
//   EPI_GEGLU  C_bf16[m][c] = (acc_x + b_x) * gelu_erf(acc_g + b_g)   weights pre-packed so that
//              packed rows [32t,32t+16) are the 'x' half and [32t+16,32t+32) the 'gate' half of
//              output columns [16t,16t+16)  (models_radar_generation.py:93-95, models_ae.py:52-54)
#include "common.h"
#include "kernels.h"

namespace rald {

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
    constexpr int BK = 64;
    constexpr int MT = BM / 32;   // 16-row m-tiles per wave
    constexpr int NT = BN / 32;   // 16-col n-tiles per wave
    constexpr int PA = BM / 32;   // staging passes (32 rows x 128 B per pass)
    constexpr int PB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
    bf16x8* sA = reinterpret_cast<bf16x8*>(smem);                       // [2][BM][8 chunks]
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem + 2 * BM * BK * 2);     // [2][BN][8 chunks]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int bz = blockIdx.z;
    const bf16* A = a.A + (int64_t)bz * a.strideA;
    const bf16* B = a.B + (int64_t)bz * a.strideB;

    // staging coordinates: thread -> (row within pass, 16-byte chunk)
    const int srow = tid >> 3, schunk = tid & 7;
    const bf16* gA[PA];
    const bf16* gB[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int r = m0 + srow + 32 * p;
        r = r < a.M ? r : a.M - 1;                       // clamp: tail rows read valid memory
        gA[p] = A + (int64_t)r * a.lda + schunk * 8;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        int r = n0 + srow + 32 * p;
        r = r < a.N ? r : a.N - 1;
        gB[p] = B + (int64_t)r * a.ldb + schunk * 8;
    }
    bf16x8 rA[PA], rB[PB];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int p = 0; p < PA; ++p) rA[p] = *reinterpret_cast<const bf16x8*>(gA[p] + kt * BK);
#pragma unroll
        for (int p = 0; p < PB; ++p) rB[p] = *reinterpret_cast<const bf16x8*>(gB[p] + kt * BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int r = srow + 32 * p;
            sA[(buf * BM + r) * 8 + (schunk ^ (r & 7))] = rA[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            int r = srow + 32 * p;
            sB[(buf * BN + r) * 8 + (schunk ^ (r & 7))] = rB[p];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MT], fb[NT];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int r = wm * (BM / 2) + i * 16 + fr;
                fa[i] = sA[(buf * BM + r) * 8 + (chunk ^ (r & 7))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int r = wn * (BN / 2) + j * 16 + fr;
                fb[j] = sB[(buf * BN + r) * 8 + (chunk ^ (r & 7))];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns row m = ..+fr, 4 consecutive columns n = ..+4*fq+{0..3} -----
    const int mb = m0 + wm * (BM / 2);
    const int nb = n0 + wn * (BN / 2);
    if constexpr (EPI == EPI_GEGLU) {
        bf16* C = reinterpret_cast<bf16*>(a.C) + (int64_t)bz * a.strideC;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = mb + i * 16 + fr;
            if (m >= a.M) continue;
#pragma unroll
            for (int p = 0; p < NT / 2; ++p) {
                const int nx = nb + 32 * p + 4 * fq;           // packed row of the 'x' half
                const int ng = nx + 16;                        // packed row of the gate half
                float4 bx = *reinterpret_cast<const float4*>(a.bias + nx);
                float4 bg = *reinterpret_cast<const float4*>(a.bias + ng);
                f32x4 x = acc[i][2 * p], g = acc[i][2 * p + 1];
                bf16x4 o = pack4((x[0] + bx.x) * gelu_erf(g[0] + bg.x), (x[1] + bx.y) * gelu_erf(g[1] + bg.y),
                                 (x[2] + bx.z) * gelu_erf(g[2] + bg.z), (x[3] + bx.w) * gelu_erf(g[3] + bg.w));
                const int c = nb / 2 + 16 * p + 4 * fq;
                *reinterpret_cast<bf16x4*>(C + (int64_t)m * a.ldc + c) = o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = mb + i * 16 + fr;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = nb + j * 16 + 4 * fq;
                if (n >= a.N) continue;                        // N is a multiple of 4
                f32x4 v = acc[i][j];
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + n);
                if constexpr (EPI == EPI_BF16) {
                    bf16* C = reinterpret_cast<bf16*>(a.C) + (int64_t)bz * a.strideC;
                    *reinterpret_cast<bf16x4*>(C + (int64_t)m * a.ldc + n) =
                        pack4(a.alpha * v[0] + b.x, a.alpha * v[1] + b.y, a.alpha * v[2] + b.z, a.alpha * v[3] + b.w);
                } else if constexpr (EPI == EPI_F32) {
                    float* C = reinterpret_cast<float*>(a.C) + (int64_t)bz * a.strideC;
                    *reinterpret_cast<float4*>(C + (int64_t)m * a.ldc + n) =
                        make_float4(a.alpha * v[0] + b.x, a.alpha * v[1] + b.y, a.alpha * v[2] + b.z, a.alpha * v[3] + b.w);
                } else {  // EPI_RESID
                    float* C = reinterpret_cast<float*>(a.C) + (int64_t)bz * a.strideC;
                    float4* p = reinterpret_cast<float4*>(C + (int64_t)m * a.ldc + n);
                    float4 r = *p;
                    *p = make_float4(r.x + v[0] + b.x, r.y + v[1] + b.y, r.z + v[2] + b.z, r.w + v[3] + b.w);
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_tile(const GemmArgs& a, int epi, hipStream_t st) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.batch);
    switch (epi) {
        case EPI_BF16:  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI_BF16>), grid, dim3(256), 0, st, a); break;
        case EPI_F32:   hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI_F32>), grid, dim3(256), 0, st, a); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI_RESID>), grid, dim3(256), 0, st, a); break;
        case EPI_GEGLU: hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI_GEGLU>), grid, dim3(256), 0, st, a); break;
        default: set_error("gemm: bad epilogue"); return 1;
    }
    RALD_HIP(hipGetLastError());
    return 0;
}

// Host-side shape contract is checked here, before any launch (an out-of-bounds MFMA tile
// can take the whole node down, so nothing is left to the kernel).
int gemm_nt(const GemmArgs& a, int epi, hipStream_t st) {
    RALD_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm: empty problem");
    RALD_CHECK(a.K % 64 == 0, "gemm: K must be a multiple of 64 (pad with zeros)");
    RALD_CHECK(a.N % 4 == 0, "gemm: N must be a multiple of 4");
    RALD_CHECK(a.lda % 8 == 0 && a.ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements (16-byte rows)");
    RALD_CHECK(a.lda >= a.K && a.ldb >= a.K, "gemm: leading dimension smaller than K");
    RALD_CHECK(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && ((uintptr_t)a.C % 16 == 0), "gemm: pointers must be 16-byte aligned");
    RALD_CHECK(a.ldc % 4 == 0, "gemm: ldc must be a multiple of 4");
    if (epi == EPI_GEGLU) {
        RALD_CHECK(a.N % 128 == 0 && a.bias != nullptr, "gemm: GEGLU needs N % 128 == 0 and a packed bias");
        RALD_CHECK(a.ldc >= a.N / 2, "gemm: GEGLU ldc < N/2");
    } else {
        RALD_CHECK(a.ldc >= a.N, "gemm: ldc < N");
    }
    // 128x128 tiles when they fill the chip; 64x64 tiles for the small-M (batch-1) regime.
    const int64_t wg128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.batch;
    if (wg128 >= 192) return launch_tile<128, 128>(a, epi, st);
    return launch_tile<64, 64>(a, epi, st);
}

}  // namespace rald
