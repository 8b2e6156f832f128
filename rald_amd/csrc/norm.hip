// Row-wise kernels over the fp32 residual stream x[M][D] (HBM-bound: one wave per row, every
// global access a fully coalesced 16-byte-per-lane instruction).
//
//   layernorm_mod   LN(x) * (add_one + g) + b -> bf16   : AdaLayerNorm (models_radar_generation.py
//                   :127-131, g = scale, b = shift, add_one = 1, per-sample or shared modulation
//                   rows) and plain affine LayerNorm (models_ae.py:38-42, g = weight, b = bias).
//   proj_in         x = c_in * xin @ W^T                : :221 fused with EDM c_in (:424,:427)
//   final_norm_proj D = c_skip*xin + c_out*(LN_affine(x) @ Wout^T) : :230-232 fused with :429
#include "common.h"
#include "kernels.h"

namespace rald {

// Lane l of the row's wave owns float4 chunks l, l+64, ... (VPL/4 chunks).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const float* __restrict__ x, bf16* __restrict__ out,
                                                            int M, const float* __restrict__ g, const float* __restrict__ b,
                                                            int64_t gstride, int rows_per_group, float add_one, float eps) {
    constexpr int NC = VPL / 4;
    constexpr int D = VPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = xr[lane + 64 * c];
        s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);     // biased variance, as torch LN
    const int64_t goff = (int64_t)(row / rows_per_group) * gstride;
    const float4* gr = reinterpret_cast<const float4*>(g + goff);
    const float4* br = reinterpret_cast<const float4*>(b + goff);
    bf16x4* orow = reinterpret_cast<bf16x4*>(out + (int64_t)row * D);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float4 gg = gr[lane + 64 * c], bb = br[lane + 64 * c];
        orow[lane + 64 * c] = pack4((v[c].x - mean) * rstd * (add_one + gg.x) + bb.x,
                                    (v[c].y - mean) * rstd * (add_one + gg.y) + bb.y,
                                    (v[c].z - mean) * rstd * (add_one + gg.z) + bb.z,
                                    (v[c].w - mean) * rstd * (add_one + gg.w) + bb.w);
    }
}

int layernorm_mod(const float* x, bf16* out, int M, int D, const float* g, const float* b,
                  int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st) {
    RALD_CHECK(M > 0 && rows_per_group > 0, "layernorm: empty");
    RALD_CHECK(D == 256 || D == 512 || D == 1024, "layernorm: D must be 256, 512 or 1024");
    dim3 grid(cdiv(M, 4)), block(256);
    if (D == 256) hipLaunchKernelGGL((layernorm_mod_kernel<4>), grid, block, 0, st, x, out, M, g, b, gstride, rows_per_group, add_one, eps);
    else if (D == 512) hipLaunchKernelGGL((layernorm_mod_kernel<8>), grid, block, 0, st, x, out, M, g, b, gstride, rows_per_group, add_one, eps);
    else hipLaunchKernelGGL((layernorm_mod_kernel<16>), grid, block, 0, st, x, out, M, g, b, gstride, rows_per_group, add_one, eps);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- split-K tail for small-M residual GEMMs: x += bias + sum_s part[s]; h = LN(x)*(add_one+g)+b (optional) ---------------
// At M <= 2048 rows a [M,512] x K = 2048 product has only a few dozen 64x64 tiles, each walking the whole K; splitting K over
// the batch dimension fills the chip, and this kernel is the (deterministic) reduction, fused with the residual add and the
// following LayerNorm - one launch instead of LayerNorm's own.  D = 512, one wave per row.
// SC: the slab count when it is 4 or 8 (compile time: all 2*SC slab loads of a row are in flight together; with the runtime trip count the
// loop waited for each slab in turn - 8 dependent L2 round trips, 5 us per launch at 512 rows), 0 = any count.
template <int SC, bool F16 = false>
__global__ __launch_bounds__(256) void reduce_resid_ln_kernel(const float* __restrict__ part, int S, int64_t part_stride, const float* __restrict__ bias,
                                                              float* __restrict__ x, bf16* __restrict__ h, int M, const float* __restrict__ g,
                                                              const float* __restrict__ b, int64_t gstride, int rows_per_group, float add_one, float eps) {
    constexpr int D = 512;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[8];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(x + (int64_t)row * D + lane * 8), a1 = *reinterpret_cast<const float4*>(x + (int64_t)row * D + lane * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bias + lane * 8), b1 = *reinterpret_cast<const float4*>(bias + lane * 8 + 4);
        v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
        v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
    }
    if constexpr (SC > 0) {
        float4 p0[SC], p1[SC];
        if constexpr (F16) {
            static_assert(SC > 0, "fp16 slabs: compile-time slab count");
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 hp[SC];
#pragma unroll
            for (int s = 0; s < SC; ++s) hp[s] = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(part) + s * part_stride + (int64_t)row * D + lane * 8);
#pragma unroll
            for (int s = 0; s < SC; ++s) {
                p0[s] = make_float4((float)hp[s][0] * 64.f, (float)hp[s][1] * 64.f, (float)hp[s][2] * 64.f, (float)hp[s][3] * 64.f);
                p1[s] = make_float4((float)hp[s][4] * 64.f, (float)hp[s][5] * 64.f, (float)hp[s][6] * 64.f, (float)hp[s][7] * 64.f);
            }
        } else {
#pragma unroll
        for (int s = 0; s < SC; ++s) {
            const float* p = part + s * part_stride + (int64_t)row * D + lane * 8;
            p0[s] = *reinterpret_cast<const float4*>(p); p1[s] = *reinterpret_cast<const float4*>(p + 4);
        }
        }
#pragma unroll
        for (int s = 0; s < SC; ++s) {                                  // same order of additions as the loop below
            v[0] += p0[s].x; v[1] += p0[s].y; v[2] += p0[s].z; v[3] += p0[s].w; v[4] += p1[s].x; v[5] += p1[s].y; v[6] += p1[s].z; v[7] += p1[s].w;
        }
    } else {
    for (int s = 0; s < S; ++s) {
        const float* p = part + s * part_stride + (int64_t)row * D + lane * 8;
        const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + 4);
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    }
    }
    *reinterpret_cast<float4*>(x + (int64_t)row * D + lane * 8) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(x + (int64_t)row * D + lane * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    if (!h) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
    const float mean = wave_sum(sum) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    const int64_t goff = (int64_t)(row / rows_per_group) * gstride;
    const float4 g0 = *reinterpret_cast<const float4*>(g + goff + lane * 8), g1 = *reinterpret_cast<const float4*>(g + goff + lane * 8 + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(b + goff + lane * 8), c1 = *reinterpret_cast<const float4*>(b + goff + lane * 8 + 4);
    bf16x4* o = reinterpret_cast<bf16x4*>(h + (int64_t)row * D + lane * 8);
    o[0] = pack4(v[0] * rstd * (add_one + g0.x) + c0.x, v[1] * rstd * (add_one + g0.y) + c0.y, v[2] * rstd * (add_one + g0.z) + c0.z,
                 v[3] * rstd * (add_one + g0.w) + c0.w);
    o[1] = pack4(v[4] * rstd * (add_one + g1.x) + c1.x, v[5] * rstd * (add_one + g1.y) + c1.y, v[6] * rstd * (add_one + g1.z) + c1.z,
                 v[7] * rstd * (add_one + g1.w) + c1.w);
}

int reduce_resid_ln(const float* part, int S, int64_t part_stride, const float* bias, float* x, bf16* h, int M, const float* g, const float* b,
                    int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st, bool part_f16) {
    RALD_CHECK(!part_f16 || S == 8 || S == 4, "reduce_resid_ln: fp16 slabs come in eights (one per head) or fours (split-K)");
    if (part_f16) {
        if (S == 8) hipLaunchKernelGGL((reduce_resid_ln_kernel<8, true>), dim3(cdiv(M, 4)), dim3(256), 0, st, part, S, part_stride, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps);
        else hipLaunchKernelGGL((reduce_resid_ln_kernel<4, true>), dim3(cdiv(M, 4)), dim3(256), 0, st, part, S, part_stride, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps);
        RALD_HIP(hipGetLastError());
        return 0;
    }
    RALD_CHECK(part && bias && x && S >= 1 && S <= 64 && M >= 1, "reduce_resid_ln: bad arguments");
    RALD_CHECK(!h || (g && b && rows_per_group > 0), "reduce_resid_ln: LayerNorm parameters missing");
    if (S == 8) hipLaunchKernelGGL(reduce_resid_ln_kernel<8>, dim3(cdiv(M, 4)), dim3(256), 0, st, part, S, part_stride, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps);
    else if (S == 4) hipLaunchKernelGGL(reduce_resid_ln_kernel<4>, dim3(cdiv(M, 4)), dim3(256), 0, st, part, S, part_stride, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps);
    else hipLaunchKernelGGL(reduce_resid_ln_kernel<0>, dim3(cdiv(M, 4)), dim3(256), 0, st, part, S, part_stride, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps);
    RALD_HIP(hipGetLastError());
    return 0;
}

// x[M][512] += A[M][K].W[512][K]^T + bias (K split over `splits` batch entries into `scratch` [splits][M][512] f32), then the LayerNorm
int resid_splitk_ln(const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, float* x, bf16* h, const float* g, const float* b,
                    int64_t gstride, int rows_per_group, float add_one, float eps, int M, int K, int splits, float* scratch, hipStream_t st) {
    RALD_CHECK(splits >= 1 && splits <= 16 && K % (splits * 64) == 0 && scratch && bias && x, "resid_splitk_ln: bad arguments");
    RALD_CHECK(!h || (g && b && rows_per_group > 0), "resid_splitk_ln: LayerNorm parameters missing");
    GemmArgs p = gemm_args(A, lda, W, ldw, scratch, 512, nullptr, M, 512, K / splits);
    p.batch = splits; p.strideA = K / splits; p.strideB = K / splits; p.strideC = (int64_t)M * 512;
    // where an LDS-DMA engine runs the product, the four slabs travel as fp16 x 2^-6 like the per-head slabs of attn_small.hip: half the bytes
    const bool f16 = splits == 4 && gemm_f16s_ok(M, 512, splits);
    RALD_TRY(gemm_nt(p, f16 ? EPI_F16S : EPI_F32, st));
    return reduce_resid_ln(scratch, splits, (int64_t)M * 512, bias, x, h, M, g, b, gstride, rows_per_group, add_one, eps, st, f16);
}

// ---- proj_in: K = C (latent channels, <= 64) is far too small for MFMA and stays fp32.
// Small M: 8 rows per workgroup, the weight read through L1/L2 (few workgroups, nothing to amortise a staging pass over).
__global__ __launch_bounds__(256) void proj_in_small_kernel(const float* __restrict__ xin, const float* __restrict__ W,
                                                            float* __restrict__ x, int M, int C, int D,
                                                            const float* __restrict__ coef, int coef_stride, int rows_per_group) {
    __shared__ float sx[8][64];
    const int m0 = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * C; i += 256) {
        int r = i / C, c = i % C, m = m0 + r;
        float v = 0.f;
        if (m < M) v = xin[(int64_t)m * C + c] * coef[(int64_t)(m / rows_per_group) * coef_stride + 0];
        sx[r][c] = v;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < D; n += 256) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float* w = W + (int64_t)n * C;
        for (int c = 0; c < C; ++c) {
            float wv = w[c];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += sx[r][c] * wv;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (m0 + r < M) x[(int64_t)(m0 + r) * D + n] = acc[r];
    }
}
// Large M: 32 rows x 256 output columns per workgroup; that half of the weight sits in LDS as [n][C + 1] (odd row stride:
// lane n reads its own row conflict-free; the small-M form reads it from global memory at a 128-byte lane stride, 64 cache
// lines per load instruction - 45 us for the 67 MB x at B = 64), the input rows (pre-multiplied by c_in) as broadcasts.
// Output rows leave fully coalesced.
template <int C>
__global__ __launch_bounds__(256) void proj_in_kernel(const float* __restrict__ xin, const float* __restrict__ W,
                                                      float* __restrict__ x, int M, int D,
                                                      const float* __restrict__ coef, int coef_stride, int rows_per_group) {
    constexpr int R = 32, LDW = C + 1;
    __shared__ float sw[256 * LDW];
    __shared__ float sx[R * C];
    const int m0 = blockIdx.x * R, n0 = blockIdx.y * 256;
    for (int i = threadIdx.x; i < 256 * C; i += 256) sw[(i / C) * LDW + i % C] = W[(int64_t)n0 * C + i];
    for (int i = threadIdx.x; i < R * C; i += 256) {
        const int m = m0 + i / C;
        sx[i] = m < M ? xin[(int64_t)m0 * C + i] * coef[(int64_t)(m / rows_per_group) * coef_stride + 0] : 0.f;
    }
    __syncthreads();
    const int n = threadIdx.x;
    const float* w = sw + n * LDW;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float wv = w[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(sx[(16 * half + r) * C + c], wv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 16 * half + r;
            if (m < M) x[(int64_t)m * D + n0 + n] = acc[r];
        }
    }
}

template <int C>
static int launch_proj_in(const float* xin, const float* W, float* x, int M, int D, const float* coef, int coef_stride, int rows_per_group,
                          hipStream_t st) {
    hipLaunchKernelGGL(proj_in_kernel<C>, dim3(cdiv(M, 32), D / 256), dim3(256), 0, st, xin, W, x, M, D, coef, coef_stride, rows_per_group);
    RALD_HIP(hipGetLastError());
    return 0;
}
int proj_in(const float* xin, const float* W, float* x, int M, int C, int D, const float* coef,
            int coef_stride, int rows_per_group, hipStream_t st) {
    RALD_CHECK(C >= 1 && C <= 64, "proj_in: latent channels must be in [1,64]");
    RALD_CHECK(D % 256 == 0, "proj_in: D must be a multiple of 256");
    if (M > 4096) {
        switch (C) {                                   // the reference's factories: 4, 8, 16, 32 latent channels
            case 4: return launch_proj_in<4>(xin, W, x, M, D, coef, coef_stride, rows_per_group, st);
            case 8: return launch_proj_in<8>(xin, W, x, M, D, coef, coef_stride, rows_per_group, st);
            case 16: return launch_proj_in<16>(xin, W, x, M, D, coef, coef_stride, rows_per_group, st);
            case 32: return launch_proj_in<32>(xin, W, x, M, D, coef, coef_stride, rows_per_group, st);
            default: break;
        }
    }
    hipLaunchKernelGGL(proj_in_small_kernel, dim3(cdiv(M, 8)), dim3(256), 0, st, xin, W, x, M, C, D, coef, coef_stride, rows_per_group);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- final LayerNorm(affine) + proj_out (D -> C, no bias) + EDM skip/out scaling, all fp32:
// this is the network's output layer, so nothing here is rounded to bf16.
template <int VPL>
__global__ __launch_bounds__(256) void final_norm_proj_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ Wout,
                                                              const float* __restrict__ xin, float* __restrict__ out, int M, int C,
                                                              const float* __restrict__ coef, int coef_stride, int rows_per_group) {
    constexpr int NC = VPL / 4;
    constexpr int D = VPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = xr[lane + 64 * c];
        s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float4 gg = reinterpret_cast<const float4*>(gamma)[lane + 64 * c];
        float4 bb = reinterpret_cast<const float4*>(beta)[lane + 64 * c];
        v[c].x = (v[c].x - mean) * rstd * gg.x + bb.x;
        v[c].y = (v[c].y - mean) * rstd * gg.y + bb.y;
        v[c].z = (v[c].z - mean) * rstd * gg.z + bb.z;
        v[c].w = (v[c].w - mean) * rstd * gg.w + bb.w;
    }
    const float* cf = coef + (int64_t)(row / rows_per_group) * coef_stride;
    const float c_skip = cf[1], c_out = cf[2];
    // 32 output channels at a time: every lane forms its partial dot products for all 32, then a reduce-scatter over the
    // lanes (xor 32, 16, 8, 4, 2 - each step halves the channels a lane still carries - and one last add over xor 1) leaves
    // channel c0 + (lane >> 1) on the lane: 32 cross-lane adds per row instead of 32 full 6-step reductions (192).
    for (int c0 = 0; c0 < C; c0 += 32) {
        float p[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int cc = c0 + c < C ? c0 + c : C - 1;
            const float4* w = reinterpret_cast<const float4*>(Wout + (int64_t)cc * D);
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float4 ww = w[lane + 64 * k];
                a += v[k].x * ww.x + v[k].y * ww.y + v[k].z * ww.z + v[k].w * ww.w;
            }
            p[c] = a;
        }
#define RALD_RS_STEP(N, OFF)                                                                          \
        _Pragma("unroll") for (int i = 0; i < N; ++i) {                                              \
            const bool hi = (lane & OFF) != 0;                                                        \
            const float keep = hi ? p[i + N] : p[i], send = hi ? p[i] : p[i + N];                     \
            p[i] = keep + __shfl_xor(send, OFF, 64);                                                  \
        }
        RALD_RS_STEP(16, 32) RALD_RS_STEP(8, 16) RALD_RS_STEP(4, 8) RALD_RS_STEP(2, 4) RALD_RS_STEP(1, 2)
#undef RALD_RS_STEP
        const float tot = p[0] + __shfl_xor(p[0], 1, 64);
        const int ch = c0 + (lane >> 1);
        if ((lane & 1) == 0 && ch < C) {
            const int64_t o = (int64_t)row * C + ch;
            out[o] = c_skip * xin[o] + c_out * tot;
        }
    }
}

// Large-M form on the matrix cores.  The one-row kernel above is VALU-bound (1.07 GFLOP of fp32 FMAs with their operand shuffles: 79 us at
// 32 768 rows for a pass whose HBM traffic takes 10); here a workgroup normalises 32 rows into LDS as bf16 hi + lo (x = hi + lo to ~2^-17)
// and multiplies them by the likewise split weight on v_mfma_f32_16x16x32_bf16 - hi.hi + lo.hi + hi.lo, fp32 accumulation - so the
// output layer keeps fp32-grade accuracy (measured against fp64: see tests) while the multiply costs ~nothing.
__global__ __launch_bounds__(256) void final_norm_proj_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ Wout,
                                                                   const float* __restrict__ xin, float* __restrict__ out, int M,
                                                                   const float* __restrict__ coef, int coef_stride, int rows_per_group,
                                                                   const bf16* __restrict__ Whl) {      // optional: the weight already split, [hi | lo][32][512]
    constexpr int D = 512, C = 32, R = 32;
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * R * D * 2];         // [hi | lo][32 rows][1 KiB], 16-byte chunk c of row n at c ^ (n & 15)
    unsigned char* const thi = sm;
    unsigned char* const tlo = sm + R * D * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * R;
    const float4 g0 = reinterpret_cast<const float4*>(gamma)[lane], g1 = reinterpret_cast<const float4*>(gamma)[lane + 64];
    const float4 b0 = reinterpret_cast<const float4*>(beta)[lane], b1 = reinterpret_cast<const float4*>(beta)[lane + 64];
    // ---- phase 1: LayerNorm(affine) of 8 rows per wave, split into bf16 hi + lo
    float4 v[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = row0 + 8 * wave + r < M ? row0 + 8 * wave + r : M - 1;
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
        v[r][0] = xr[lane]; v[r][1] = xr[lane + 64];
    }
    // (the 8 rows' reductions run side by side: 8 independent shuffle chains instead of 16 dependent ones in a row)
    float mean8[8], rstd8[8];
    {
        float s8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            s8[r] = 0.f;
            s8[r] += v[r][0].x + v[r][0].y + v[r][0].z + v[r][0].w;
            s8[r] += v[r][1].x + v[r][1].y + v[r][1].z + v[r][1].w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 8; ++r) s8[r] += __shfl_xor(s8[r], o, 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            mean8[r] = s8[r] * (1.0f / D);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float dx = v[r][c].x - mean8[r], dy = v[r][c].y - mean8[r], dz = v[r][c].z - mean8[r], dw = v[r][c].w - mean8[r];
                q += dx * dx + dy * dy + dz * dz + dw * dw;
            }
            s8[r] = q;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 8; ++r) s8[r] += __shfl_xor(s8[r], o, 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) rstd8[r] = rsqrtf(s8[r] * (1.0f / D) + 1e-5f);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float mean = mean8[r], rstd = rstd8[r];
        const int rl = 8 * wave + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 gg = c ? g1 : g0, bb = c ? b1 : b0;
            const float y0 = (v[r][c].x - mean) * rstd * gg.x + bb.x, y1 = (v[r][c].y - mean) * rstd * gg.y + bb.y;
            const float y2 = (v[r][c].z - mean) * rstd * gg.z + bb.z, y3 = (v[r][c].w - mean) * rstd * gg.w + bb.w;
            const bf16 h0 = (bf16)y0, h1 = (bf16)y1, h2 = (bf16)y2, h3 = (bf16)y3;
            const int col = 256 * c + 4 * lane;                       // 4 consecutive columns = half a 16-byte chunk
            const int off = rl * 1024 + ((((col >> 3) ^ (rl & 15))) << 4) + ((col >> 2) & 1) * 8;
            bf16x4 hv, lv;
            hv[0] = h0; hv[1] = h1; hv[2] = h2; hv[3] = h3;
            lv[0] = (bf16)(y0 - (float)h0); lv[1] = (bf16)(y1 - (float)h1); lv[2] = (bf16)(y2 - (float)h2); lv[3] = (bf16)(y3 - (float)h3);
            *reinterpret_cast<bf16x4*>(thi + off) = hv;
            *reinterpret_cast<bf16x4*>(tlo + off) = lv;
        }
    }
    __syncthreads();
    // ---- phase 2: out tile [16 rows rt][16 channels ct] per wave
    const int rt = wave >> 1, ct = wave & 1, i = lane & 15, kq = lane >> 4;
    const int arow = 16 * rt + i;
    const float* wp = Wout + (int64_t)(16 * ct + i) * D + 8 * kq;
    const bf16* wh = Whl ? Whl + (int64_t)(16 * ct + i) * D + 8 * kq : nullptr;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {                              // (fully unrolled: all 32 weight fragments of the wave in flight at once)
        bf16x8 bh, bl;
        if (wh) {
            bh = *reinterpret_cast<const bf16x8*>(wh + 32 * ks);
            bl = *reinterpret_cast<const bf16x8*>(wh + C * D + 32 * ks);
        } else {
            const float4 w0 = *reinterpret_cast<const float4*>(wp + 32 * ks), w1 = *reinterpret_cast<const float4*>(wp + 32 * ks + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { bh[e] = (bf16)wv[e]; bl[e] = (bf16)(wv[e] - (float)bh[e]); }
        }
        const int aoff = arow * 1024 + (((4 * ks + kq) ^ (arow & 15)) << 4);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(thi + aoff), al = *reinterpret_cast<const bf16x8*>(tlo + aoff);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    }
    // acc[e] = F[row0 + 16 rt + 4 kq + e][16 ct + i]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * rt + 4 * kq + e;
        if (row < M) {
            const float* cf = coef + (int64_t)(row / rows_per_group) * coef_stride;
            const int64_t o = (int64_t)row * C + 16 * ct + i;
            out[o] = cf[1] * xin[o] + cf[2] * acc[e];
        }
    }
}

// Wout [C][512] fp32 -> bf16 [hi | lo][C][512] (x = hi + lo to ~2^-17): the large-M kernel's B operand, once per weight load
__global__ void split_hi_lo_kernel(const float* __restrict__ w, bf16* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bf16 h = (bf16)w[i];
    out[i] = h;
    out[n + i] = (bf16)(w[i] - (float)h);
}
int split_hi_lo(const float* w, bf16* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(split_hi_lo_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, out, n);
    RALD_HIP(hipGetLastError());
    return 0;
}

int final_norm_proj(const float* x, const float* gamma, const float* beta, const float* Wout,
                    const float* xin, float* out, int M, int D, int C, const float* coef,
                    int coef_stride, int rows_per_group, hipStream_t st, const bf16* Whl) {
    RALD_CHECK(C >= 1 && C <= 64, "final_norm_proj: output channels must be in [1,64]");
    RALD_CHECK(D == 512, "final_norm_proj: D must be 512");
    if (C == 32 && M >= 8192) {
        hipLaunchKernelGGL(final_norm_proj_mfma_kernel, dim3(cdiv(M, 32)), dim3(256), 0, st, x, gamma, beta, Wout, xin, out, M, coef, coef_stride,
                           rows_per_group, Whl);
        RALD_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((final_norm_proj_kernel<8>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, Wout, xin, out, M, C,
                       coef, coef_stride, rows_per_group);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
