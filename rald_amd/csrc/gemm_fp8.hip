// MXFP8 GEMM (BASELINE config #5: "fp8 MFMA QKV/proj path"):  C[M,N] = A[M,K] . W[N,K]^T with both
// operands in the OCP microscaling format - e4m3 elements plus one e8m0 (power-of-two) scale per 32
// consecutive K elements of a row - on v_mfma_scale_f32_16x16x128_f8f6f4, which applies the block
// scales in hardware and runs at twice the bf16 MFMA rate.
//
// Operand layout of that instruction, measured on MI355X (tools/probe/mfma_fp8*.hip; not in the guides):
//   lane l (r = l & 15, g = l >> 4) holds row r, K = 16g .. 16g+15 in VGPRs 0-3 and K = 64+16g .. 64+16g+15
//   in VGPRs 4-7 (two 16-byte chunks, g and g+4, of the row's 128-byte k-step);
//   byte[opsel] of lane l's scale operand is the e8m0 scale of row r, K-block g (K = 32g .. 32g+31);
//   C/D as every 16x16 MFMA: col = l & 15, row = 4*(l >> 4) + reg.
//
// Same engine as gemm_nt_glds_kernel (gemm.hip): a k-step is 128 BYTES per row in both, so the LDS-DMA
// staging, the XOR swizzle, the XCD-aware strip walk and the LDS-staged epilogue are byte-for-byte the
// same; only the fragment reads (chunks g and g+4) and the MFMA differ.  Scales are one byte per lane
// per tile row per k-step, fetched a k-step ahead with plain byte loads (they stay in L2).
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"
#include "mx8.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void8;
typedef const __attribute__((address_space(1))) void glb_void8;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mx8_kernel(Mx8Args a) {
    constexpr int WAVES = WM * WN;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    constexpr int CA = BM / 8 / WAVES;       // 1-KiB DMA pieces (8 rows x 128 B) per wave
    constexpr int CB = BN / 8 / WAVES;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][A tile | B tile]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware strip walk: see gemm_nt_glds_kernel
    const int ntn = gridDim.x, ntm = gridDim.y, nt = ntn * ntm;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, q = nt >> 3, rr = nt & 7;
    const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lin >> 3);
    constexpr int GN = 8;
    int tm, tn;
    if (ntn % GN == 0) {
        const int strip = tile / (ntm * GN), within = tile % (ntm * GN);
        tm = within / GN;
        tn = strip * GN + within % GN;
    } else {
        tm = tile / ntn;
        tn = tile % ntn;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;                // (the MXFP8 engine has no inner batch: batch2 = 1)
    const GemmArgs& g = a.g;
    const int64_t coff = (int64_t)bz * g.strideC;
    const unsigned char* A = a.A8 + (int64_t)bz * g.strideA;
    const unsigned char* B = a.B8 + (int64_t)bz * g.strideB;
    const unsigned char* SA = a.SA + (int64_t)bz * a.strideSA;
    const unsigned char* SB = a.SB + (int64_t)bz * a.strideSB;
    const int kb = g.K / 32;                 // scale bytes per row

    const int lr = lane >> 3;
    const int lc = (lane & 7) ^ lr;
    const unsigned char* gA[CA];
    const unsigned char* gB[CB];
#pragma unroll
    for (int p = 0; p < CA; ++p) {
        int r = m0 + 8 * (wave + WAVES * p) + lr;
        r = r < g.M ? r : g.M - 1;
        gA[p] = A + (int64_t)r * g.lda + lc * 16;
    }
#pragma unroll
    for (int p = 0; p < CB; ++p) {
        int r = n0 + 8 * (wave + WAVES * p) + lr;
        r = r < g.N ? r : g.N - 1;
        gB[p] = B + (int64_t)r * g.ldb + lc * 16;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < CA; ++p)
            __builtin_amdgcn_global_load_lds((glb_void8*)(gA[p] + kt * 128), (lds_void8*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < CB; ++p)
            __builtin_amdgcn_global_load_lds((glb_void8*)(gB[p] + kt * 128), (lds_void8*)(base + BM * 128 + (wave + WAVES * p) * 1024), 16, 0, 0);
    };

    const int fr = lane & 15, fq = lane >> 4;
    // scale byte of this lane's (tile row, K-block fq) for k-step kt: S[row * kb + kt * 4 + fq]
    int offA[MT], offB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = m0 + wm * (BM / WM) + i * 16 + fr;
        r = r < g.M ? r : g.M - 1;
        offA[i] = r * kb + fq;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int r = n0 + wn * (BN / WN) + j * 16 + fr;
        r = r < g.N ? r : g.N - 1;
        offB[j] = r * kb + fq;
    }
    int sa_next[MT], sb_next[NT];
    auto load_scales = [&](int kt) {
#pragma unroll
        for (int i = 0; i < MT; ++i) sa_next[i] = SA[offA[i] + kt * 4];
#pragma unroll
        for (int j = 0; j < NT; ++j) sb_next[j] = SB[offB[j] + kt * 4];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / 128;
    auto read_frag = [&](const unsigned char* tile_base, int row) -> i32x8 {
        const i32x4* s = reinterpret_cast<const i32x4*>(tile_base) + row * 8;
        const i32x4 lo = s[fq ^ (row & 7)], hi = s[(fq + 4) ^ (row & 7)];
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    stage(0, 0);
    load_scales(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt and its scales have landed
        __builtin_amdgcn_s_barrier();                          // ... for every wave; buffer cur^1 is free
        asm volatile("" ::: "memory");
        int sa[MT], sb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) sa[i] = sa_next[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) sb[j] = sb_next[j];
        if (kt + 1 < nk) {
            stage(kt + 1, cur ^ 1);
            load_scales(kt + 1);
        }
        const unsigned char* tA = smem + cur * STAGE_BYTES;
        const unsigned char* tB = tA + BM * 128;
        i32x8 fa[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = read_frag(tA, wm * (BM / WM) + i * 16 + fr);
        // weight fragments one n-tile ahead of the MFMAs that consume them (an LDS round trip per n-tile otherwise)
        i32x8 fb = read_frag(tB, wn * (BN / WN) + fr);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            i32x8 fb_next = fb;
            if (j + 1 < NT) fb_next = read_frag(tB, wn * (BN / WN) + (j + 1) * 16 + fr);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[i], acc[i][j], 0, 0, 0, sb[j], 0, sa[i]);
            fb = fb_next;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();             // staging buffers become the epilogue patches
    asm volatile("" ::: "memory");
    gemm_epilogue_lds<MT, NT, EPI>(acc, g, m0 + wm * (BM / WM), n0 + wn * (BN / WN), coff, lane, smem + wave * 8704);
}

template <int BM, int BN, int WM, int WN, int EPI>
static int launch_mx8_epi(const Mx8Args& a, hipStream_t st) {
    constexpr int WAVES = WM * WN;
    constexpr int smem = 2 * (BM + BN) * 128;
    static_assert(smem >= WAVES * 8704, "epilogue patches must fit in the staging buffers");
    static bool attr_set = false;
    auto kern = gemm_mx8_kernel<BM, BN, WM, WN, EPI>;
    if (!attr_set && smem > 64 * 1024) {
        RALD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid(cdiv(a.g.N, BN), cdiv(a.g.M, BM), a.g.batch);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}
template <int BM, int BN, int WM, int WN>
static int launch_mx8(const Mx8Args& a, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_BF16:  return launch_mx8_epi<BM, BN, WM, WN, EPI_BF16>(a, st);
        case EPI_F32:   return launch_mx8_epi<BM, BN, WM, WN, EPI_F32>(a, st);
        case EPI_RESID: return launch_mx8_epi<BM, BN, WM, WN, EPI_RESID>(a, st);
        case EPI_GEGLU: return launch_mx8_epi<BM, BN, WM, WN, EPI_GEGLU>(a, st);
        default: set_error("gemm_mx8: bad epilogue"); return 1;
    }
}

// Shape contract checked on the host before any launch.
int gemm_mx8(const Mx8Args& a0, int epi, hipStream_t st) {
    Mx8Args a = a0;
    const GemmArgs& g = a.g;
    RALD_CHECK(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, "gemm_mx8: empty problem");
    RALD_CHECK(g.K % 128 == 0, "gemm_mx8: K must be a multiple of 128 (one MFMA k-step)");
    RALD_CHECK(g.N % 4 == 0, "gemm_mx8: N must be a multiple of 4");
    RALD_CHECK(g.lda % 16 == 0 && g.ldb % 16 == 0 && g.lda >= g.K && g.ldb >= g.K, "gemm_mx8: lda/ldb must be >= K and multiples of 16 bytes");
    RALD_CHECK(a.A8 && a.B8 && a.SA && a.SB && g.C, "gemm_mx8: null operand");
    RALD_CHECK(((uintptr_t)a.A8 % 16 == 0) && ((uintptr_t)a.B8 % 16 == 0) && ((uintptr_t)g.C % 16 == 0), "gemm_mx8: pointers must be 16-byte aligned");
    if (epi == EPI_GEGLU) RALD_CHECK(g.N % 128 == 0 && g.bias != nullptr && g.ldc % 4 == 0 && g.ldc >= g.N / 2, "gemm_mx8: GEGLU needs N % 128 == 0, a packed bias and ldc >= N/2");
    else RALD_CHECK(g.ldc % 4 == 0 && g.ldc >= g.N, "gemm_mx8: ldc must be >= N and a multiple of 4");
    RALD_CHECK((int64_t)g.M * (g.K / 32) < ((int64_t)1 << 31) && (int64_t)g.N * (g.K / 32) < ((int64_t)1 << 31), "gemm_mx8: scale index overflow");
    a.g.ablate = 64;                           // streamed (non-temporal) bf16 output, as in gemm_nt
    if (g.out8) RALD_CHECK(epi == EPI_GEGLU && g.outs && g.batch == 1 && g.M % 256 == 0 && g.N % 256 == 0 && (int64_t)(g.M / 256) * (g.N / 256) >= 256,
                           "gemm_mx8: the MXFP8 output form needs the GEGLU epilogue on full 256x256 tiles");
    const int64_t wg256 = (int64_t)(g.M / 256) * (g.N / 256) * g.batch;
    if (g.M % 256 == 0 && g.N % 256 == 0 && wg256 >= 256) return launch_mx8<256, 256, 4, 2>(a, epi, st);
    return launch_mx8<128, 128, 2, 2>(a, epi, st);
}

// =================================================================================================
// Quantisers: rows of fp32 / bf16 -> MXFP8 (e4m3 elements + e8m0 scale per 32 elements, OCP MX v1.0
// format; shared exponent = the smallest power of two with amax / scale <= 448).
// One wave per row pass of 512 elements: lane l owns 8 consecutive elements, 4 lanes form a block.
// =================================================================================================

template <typename T>
__global__ __launch_bounds__(256) void quantize_mx8_kernel(const T* __restrict__ in, int64_t ld_in, unsigned char* __restrict__ q, int64_t ld_q,
                                                           unsigned char* __restrict__ s, int64_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    for (int c0 = 0; c0 < K; c0 += 512) {
        const int c = c0 + lane * 8;
        if (c >= K) break;                                          // K % 32 == 0: whole blocks (4 lanes) drop out together
        float v[8];
        if constexpr (sizeof(T) == 4) {
            const float4 a = *reinterpret_cast<const float4*>(in + row * ld_in + c), b = *reinterpret_cast<const float4*>(in + row * ld_in + c + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(in + row * ld_in + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
        }
        mx8_block(v, q + row * ld_q + c, s + row * (K / 32) + c / 32, (lane & 3) == 0);
    }
}

int quantize_mx8(const void* in, int in_is_bf16, int64_t ld_in, unsigned char* q, int64_t ld_q, unsigned char* scales, int64_t rows, int K,
                 hipStream_t st) {
    RALD_CHECK(rows >= 0 && K > 0 && K % 32 == 0, "quantize_mx8: K must be a positive multiple of 32");
    RALD_CHECK(ld_in >= K && ld_q >= K && ld_in % 8 == 0 && ld_q % 8 == 0, "quantize_mx8: leading dimensions must be >= K and multiples of 8");
    RALD_CHECK(((uintptr_t)in % 16 == 0) && ((uintptr_t)q % 8 == 0), "quantize_mx8: misaligned pointer");
    if (rows == 0) return 0;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (in_is_bf16) hipLaunchKernelGGL(quantize_mx8_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, ld_in, q, ld_q, scales, rows, K);
    else hipLaunchKernelGGL(quantize_mx8_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ld_in, q, ld_q, scales, rows, K);
    RALD_HIP(hipGetLastError());
    return 0;
}

// AdaLayerNorm (models_radar_generation.py:119-131) with an MXFP8 result: h = LN(x) * (1 + scale) + shift,
// D = 512 (one wave per row, the row stays in registers), quantised as above.
__global__ __launch_bounds__(256) void layernorm_mod_mx8_kernel(const float* __restrict__ x, unsigned char* __restrict__ q, unsigned char* __restrict__ s,
                                                                const float* __restrict__ gam, const float* __restrict__ bet, int64_t gstride,
                                                                int rows_per_group, int64_t rows, float add_one, float eps) {
    constexpr int D = 512;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 a = *reinterpret_cast<const float4*>(x + row * D + lane * 8), b = *reinterpret_cast<const float4*>(x + row * D + lane * 8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
    const float mean = wave_sum(sum) * (1.0f / D);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] -= mean; var += v[i] * v[i]; }
    const float rstd = rsqrtf(wave_sum(var) * (1.0f / D) + eps);
    const int64_t grp = row / rows_per_group;
    const float* gp = gam + grp * gstride + lane * 8;
    const float* bp = bet + grp * gstride + lane * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * rstd * (add_one + gp[i]) + bp[i];
    mx8_block(v, q + row * D + lane * 8, s + row * (D / 32) + lane / 4, (lane & 3) == 0);
}

int layernorm_mod_mx8(const float* x, unsigned char* q, unsigned char* scales, int64_t rows, int D, const float* gam, const float* bet,
                      int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st) {
    RALD_CHECK(D == 512, "layernorm_mod_mx8: D must be 512");
    RALD_CHECK(rows >= 0 && rows_per_group > 0, "layernorm_mod_mx8: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(layernorm_mod_mx8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, q, scales, gam, bet, gstride, rows_per_group,
                       rows, add_one, eps);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
