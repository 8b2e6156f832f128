// LDS-staged GEMM epilogue shared by the bf16 (gemm.hip) and MXFP8 (gemm_fp8.hip) LDS-DMA engines.
#pragma once
#include "common.h"
#include "kernels.h"
#include "mx8.h"

namespace rald {

// Diagnostic counter of the fp16-slab epilogue (EPI_F16S): lanes that clamped a value (|v| >= 65504 * 64).  One copy per translation unit
// that includes this header; gemm.hip's is the one EPI_F16S launches write and f16_saturation_gemm() reads.
static __device__ unsigned g_f16_sat_gemm;

// ---- LDS-staged epilogue (LDS-DMA engine): the MFMA accumulator layout gives every lane 4 columns
// of 16 different rows, so direct stores touch 32-64 B per row per instruction (measured: 40 % of
// the FF1 kernel).  Instead each wave transposes one 16-row m-tile at a time through a private LDS
// patch (row stride padded by 16 B) and writes it back as whole rows, 16 B per lane: full 128-B
// lines.  Wave-private, so no workgroup barrier; LDS ops of one wave execute in order.
// lds_bias (GEGLU only): the wave's NT*16 bias values already in LDS (persistent engine: an ordinary global load next to
// in-flight LDS-DMA makes hipcc wait vmcnt(0), draining the next tile's prefetch - guide section 5, trap (b))
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue_lds(f32x4 (&acc)[MT][NT], const GemmArgs& a, int mb, int nb, int64_t coff, int lane,
                                                  unsigned char* patch, const float* lds_bias = nullptr) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr bool F32OUT = (EPI == EPI_F32 || EPI == EPI_RESID);
    constexpr int OC = (EPI == EPI_GEGLU) ? NT * 8 : NT * 16;            // output columns of this wave
    constexpr int ROWB = OC * (F32OUT ? 4 : 2);                          // bytes per output row
    constexpr int STRIDE = ROWB + 16;
    constexpr int LPR = ROWB / 16;                                       // lanes per row (16 B each)
    constexpr int RPI = (64 / LPR) > 16 ? 16 : (64 / LPR);               // rows per store instruction (narrow tiles: lanes >= 16*LPR idle)
    static_assert(ROWB % 16 == 0 && 64 % LPR == 0 && 16 % RPI == 0 && 16 * STRIDE <= 8704, "epilogue tiling");
    const int oc0 = (EPI == EPI_GEGLU) ? nb / 2 : nb;                     // first output column
    const int ncols = (EPI == EPI_GEGLU) ? a.N / 2 : a.N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // ---- lane-owned values -> LDS patch [16 rows][OC]
        if constexpr (EPI == EPI_GEGLU) {
#pragma unroll
            for (int p = 0; p < NT / 2; ++p) {
                const int nx = nb + 32 * p + 4 * fq;
                const float4 bx = lds_bias ? *reinterpret_cast<const float4*>(lds_bias + 32 * p + 4 * fq) : *reinterpret_cast<const float4*>(a.bias + nx);
                const float4 bg = lds_bias ? *reinterpret_cast<const float4*>(lds_bias + 32 * p + 4 * fq + 16) : *reinterpret_cast<const float4*>(a.bias + nx + 16);
                const f32x4 x = acc[i][2 * p], g = acc[i][2 * p + 1];
                const f32x2 g01 = gelu_poly2(f32x2{g[0] + bg.x, g[1] + bg.y});
                const f32x2 g23 = gelu_poly2(f32x2{g[2] + bg.z, g[3] + bg.w});
                const f32x2 o01 = f32x2{x[0] + bx.x, x[1] + bx.y} * g01;
                const f32x2 o23 = f32x2{x[2] + bx.z, x[3] + bx.w} * g23;
                *reinterpret_cast<bf16x4*>(patch + fr * STRIDE + (16 * p + 4 * fq) * 2) = pack4(o01[0], o01[1], o23[0], o23[1]);
            }
        } else {
            if constexpr (EPI == EPI_SOFTMAX64) {
                // row fr's 64 columns of a group = 4 n-tiles x the 4 lanes fr, fr+16, fr+32, fr+48 (4 columns each): 16 values in
                // the lane, two cross-lane steps for the maximum and two for the sum
                static_assert(NT % 4 == 0, "softmax groups of 64 columns");
#pragma unroll
                for (int g = 0; g < NT / 4; ++g) {
                    float e[16];
                    float mx = -3.0e38f;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int c = 0; c < 4; ++c) { e[4 * t + c] = a.alpha * acc[i][4 * g + t][c]; mx = fmaxf(mx, e[4 * t + c]); }
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    float sm = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { e[q] = __builtin_amdgcn_exp2f(e[q] - mx); sm += e[q]; }
                    sm += __shfl_xor(sm, 16, 64);
                    sm += __shfl_xor(sm, 32, 64);
                    const float inv = 1.0f / sm;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        *reinterpret_cast<bf16x4*>(patch + fr * STRIDE + (16 * (4 * g + t) + 4 * fq) * 2) =
                            pack4(e[4 * t] * inv, e[4 * t + 1] * inv, e[4 * t + 2] * inv, e[4 * t + 3] * inv);
                }
            } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = nb + j * 16 + 4 * fq;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n < a.N) b = *reinterpret_cast<const float4*>(a.bias + n);
                const f32x4 v = acc[i][j];
                const float al = n < a.alpha_ncols ? a.alpha : 1.0f;
                const float o0 = al * v[0] + b.x, o1 = al * v[1] + b.y, o2 = al * v[2] + b.z, o3 = al * v[3] + b.w;
                if constexpr (F32OUT) *reinterpret_cast<float4*>(patch + fr * STRIDE + (16 * j + 4 * fq) * 4) = make_float4(o0, o1, o2, o3);
                else if constexpr (EPI == EPI_F16S) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    constexpr float SC = 0.015625f, LIM = 65504.f;
                    h4 hv;
                    hv[0] = (_Float16)fminf(fmaxf(o0 * SC, -LIM), LIM); hv[1] = (_Float16)fminf(fmaxf(o1 * SC, -LIM), LIM);
                    hv[2] = (_Float16)fminf(fmaxf(o2 * SC, -LIM), LIM); hv[3] = (_Float16)fminf(fmaxf(o3 * SC, -LIM), LIM);
                    if (fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3))) * SC >= LIM) atomicAdd(&g_f16_sat_gemm, 1u);
                    *reinterpret_cast<h4*>(patch + fr * STRIDE + (16 * j + 4 * fq) * 2) = hv;
                } else *reinterpret_cast<bf16x4*>(patch + fr * STRIDE + (16 * j + 4 * fq) * 2) = pack4(o0, o1, o2, o3);
            }
            }
        }
        // ---- whole rows back out: lane -> (row lane/LPR, 16-byte piece lane%LPR)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RPI) {
            const int r = r0 + lane / LPR, pc = lane % LPR;
            if (lane / LPR >= RPI) continue;
            const int m = mb + i * 16 + r;
            const uint4 v = *reinterpret_cast<const uint4*>(patch + r * STRIDE + pc * 16);
            constexpr int EPP = F32OUT ? 4 : 8;                           // elements per 16-byte piece
            const int c = oc0 + pc * EPP;
            if constexpr (!F32OUT) {
                if (a.out8) {
                    // MXFP8 output (host contract: every lane of the wave is active here and the tile is full): this
                    // lane's piece is 8 consecutive columns, 4 consecutive lanes = one 32-column block
                    const bf16x8 hv = *reinterpret_cast<const bf16x8*>(&v);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)hv[e];
                    unsigned char q8[8] __attribute__((aligned(8)));
                    unsigned char sc;
                    mx8_block(f, q8, &sc, true);
                    if (m < a.M && c < ncols) {
                        *reinterpret_cast<uint2*>(a.out8 + (int64_t)m * a.ldc + c) = *reinterpret_cast<const uint2*>(q8);
                        if ((pc & 3) == 0) a.outs[(int64_t)m * (ncols / 32) + c / 32] = sc;
                    }
                    continue;
                }
            }
            if (m < a.M && c < ncols && !RALD_ABLATED(a.ablate, 16)) {      // 16: probe builds, no global stores
                if constexpr (EPI == EPI_RESID) {
                    float* C = reinterpret_cast<float*>(a.C) + coff + (int64_t)m * a.ldc + c;
                    float4 x = *reinterpret_cast<float4*>(C);
                    const float4 d = *reinterpret_cast<const float4*>(&v);
                    x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
                    *reinterpret_cast<float4*>(C) = x;
                } else if constexpr (EPI == EPI_F32) {
                    float* C = reinterpret_cast<float*>(a.C) + coff + (int64_t)m * a.ldc + c;
                    *reinterpret_cast<uint4*>(C) = v;
                } else {
                    bf16* C = reinterpret_cast<bf16*>(a.C) + coff + (int64_t)m * a.ldc + c;
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    if (c + 8 <= ncols) {
                        // streamed output: system-scope, non-temporal stores write through L2 without allocating.  With plain
                        // stores - and, inside the NFE, with the compiler's `nt`-only non-temporal store as well - the output
                        // lines displace the weight / activation panels: FETCH_SIZE of the FF1 GEMM in situ 193 MB per launch
                        // against 80 MB with this policy (= with no stores at all; profiles/traffic.json)
                        const u32x4 vv = u32x4{v.x, v.y, v.z, v.w};
                        if (RALD_ABLATED(a.ablate, 512)) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(C), "v"(vv) : "memory");     // probe: write-through without the nt hint
                        else if (RALD_ABLATED(a.ablate, 1024)) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(C), "v"(vv) : "memory");   // probe: sc1 only
#ifdef RALD_STORE_SC01        // A/B builds (tools/build_variant.sh): write-through without the nt hint
                        else if (a.ablate & 64) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(C), "v"(vv) : "memory");
#endif
                        else if (a.ablate & 64) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(C), "v"(vv) : "memory");
                        else *reinterpret_cast<uint4*>(C) = v;
                    } else *reinterpret_cast<uint2*>(C) = make_uint2(v.x, v.y);   // N % 8 == 4 tail
                }
            }
        }
    }
}

}  // namespace rald
