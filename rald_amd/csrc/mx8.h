// MXFP8 block quantisation (OCP microscaling: e4m3 elements + one e8m0 scale per 32 consecutive elements), shared by
// the stand-alone quantisers (gemm_fp8.hip) and the fused LayerNorm epilogue (gemm_ln.hip).
#pragma once
#include "common.h"

namespace rald {

// Each lane holds 8 consecutive elements of a row; 4 consecutive lanes form a 32-element block.
__device__ __forceinline__ void mx8_block(const float (&v)[8], unsigned char* q_out, unsigned char* s_out, bool write_scale) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const unsigned ab = __float_as_uint(amax);
    // biased floor(log2(amax)) (0 for zero / subnormal), one more when the mantissa exceeds 1.75 so that
    // amax / scale <= 448 and nothing saturates (the spec's plain floor rule clips up to 12.5 % off the
    // largest element of such a block)
    const int e = (int)((ab >> 23) & 0xFF) + ((ab & 0x7FFFFFu) > 0x600000u ? 1 : 0);
    const int sb = e > 8 ? e - 8 : 0;                                // e8m0 scale byte: 2^(sb - 127)
    const float inv = __uint_as_float((unsigned)(254 - sb) << 23);   // 2^(127 - sb)
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = fminf(fmaxf(v[i] * inv, -448.f), 448.f);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(s[0], s[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(s[2], s[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(s[4], s[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(s[6], s[7], w1, true);
    *reinterpret_cast<uint2*>(q_out) = make_uint2((unsigned)w0, (unsigned)w1);
    if (write_scale) *s_out = (unsigned char)sb;
}

}  // namespace rald
