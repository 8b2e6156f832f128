// Flash-style multi-head attention for head dim 64 on v_mfma_f32_32x32x16_bf16
// (CrossAttention, models_radar_generation.py:66-75; Attention, models_ae.py:91-104).
//
// One wave owns 32 queries of one (batch, head) and streams the keys in tiles of 32 with an
// online softmax; nothing of the [nq x nk] score matrix ever reaches memory (the reference
// materialises it in fp32).  CDNA4-specific structure:
//   * S^T = K.Q^T (keys on the accumulator ROWS, the query on the LANE): every lane holds 16 of
//     its query's 32 scores, its partner lane (lane^32) the other 16, so the row max / row sum
//     are 15 in-register ops + ONE cross-half exchange, and the rescale factor is lane-local.
//   * The S^T accumulator is then directly the B operand of O^T = V^T.P^T (sum over the
//     accumulator's row index) - no LDS round trip, no lane movement.  The k order inside a
//     16-key step is permuted (element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)),
//     so the V^T fragment is gathered as two 8-byte pieces in that same order.
//   * V arrives pre-transposed (Vt[b][h*64+d][key], keys contiguous): the producing GEMM is
//     simply issued with the operand roles swapped, so no transpose pass exists anywhere.
#include "common.h"
#include "kernels.h"

namespace rald {

__global__ __launch_bounds__(256) void attention_d64_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= a.nq) return;                                  // no barriers below: early exit is safe
    const int h = blockIdx.y, b = blockIdx.z;
    const bf16* Q = a.Q + (int64_t)b * a.strideQ + (int64_t)(q0 + r) * a.ldq + h * 64 + 8 * hf;
    const bf16* K = a.K + (int64_t)b * a.strideK + (int64_t)r * a.ldk + h * 64 + 8 * hf;
    const bf16* Vt = a.Vt + (int64_t)b * a.strideVt + (int64_t)(h * 64 + r) * a.ldvt + 4 * hf;

    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Q + 16 * s);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -1e30f, l = 0.f;
    const float sc = a.scale * 1.4426950408889634f;          // exp(x) = exp2(x*log2e)

    for (int j0 = 0; j0 < a.nk; j0 += 32) {
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(K + (int64_t)j0 * a.ldk + 16 * s);
        // V^T fragments for this tile: [d-tile][k-step], element j <-> key j0 + 16s + 8(j>>2) + 4hf + (j&3)
        bf16x8 vf[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16* p = Vt + (int64_t)(32 * dt) * a.ldvt + j0 + 16 * s;
                bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
                bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 8);
                vf[dt][s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], st, 0, 0, 0);

        // st[i] = score(key j0 + (i&3) + 8*(i>>2) + 4*hf, query q0 + r)
        float mx = -1e30f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = j0 + (i & 3) + 8 * (i >> 2) + 4 * hf;
            st[i] = key < a.nk ? st[i] * sc : -1e30f;
            mx = fmaxf(mx, st[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            st[i] = exp2f(st[i] - mn);
            ps += st[i];
        }
        l = l * alpha + ps;                                   // per-half partial; halves summed at the end
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        bf16x8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s][j] = (bf16)st[8 * s + j];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][s], pf[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][s], pf[s], o1, 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // o{dt}[i] = O^T[d = 32dt + (i&3) + 8(i>>2) + 4hf][query q0+r]: 4 consecutive d per register group
    bf16* O = a.O + (int64_t)b * a.strideO + (int64_t)(q0 + r) * a.ldo + h * 64 + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<bf16x4*>(O + 8 * g) = pack4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<bf16x4*>(O + 32 + 8 * g) = pack4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
}

int attention_d64(const AttnArgs& a, hipStream_t st) {
    RALD_CHECK(a.nq > 0 && a.nk > 0 && a.heads > 0 && a.batch > 0, "attention: empty problem");
    RALD_CHECK(a.nq % 32 == 0, "attention: nq must be a multiple of 32");
    RALD_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 4 == 0 && a.ldo % 4 == 0, "attention: leading dimensions must keep 16/8-byte alignment");
    RALD_CHECK(a.k_rows >= round_up(a.nk, 32), "attention: K must have rows allocated up to a multiple of 32 keys");
    RALD_CHECK(a.ldvt >= round_up(a.nk, 32), "attention: Vt rows must be padded (zero-filled) to a multiple of 32 keys");
    RALD_CHECK(((uintptr_t)a.Q % 16 == 0) && ((uintptr_t)a.K % 16 == 0) && ((uintptr_t)a.Vt % 8 == 0) && ((uintptr_t)a.O % 8 == 0), "attention: pointer alignment");
    dim3 grid(cdiv(a.nq, 128), a.heads, a.batch);
    hipLaunchKernelGGL(attention_d64_kernel, grid, dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
