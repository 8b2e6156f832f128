// Small-batch (<= 2048 rows: the reference's eval_batch_size = 1) forms of the two attention
// sub-blocks of BasicTransformerBlock (model/models_radar_generation.py:166-167; CrossAttention :55-76).
//
// At 512 rows a transformer block is a chain of ~12 dependent launches of a few microseconds each, and every launch pays a
// ~1.5 us boundary plus its own ramp: the chain, not the arithmetic, is the cost (1.6 ms per NFE for 132 GFLOP).  These
// two kernels cut the chain by fusing everything that one (head, 32-query block) workgroup can do on its own:
//
//   attn_self_proj    softmax(q k^T) v for one head and 32 queries - the 512 keys split over the workgroup's 4 waves and
//                     merged through LDS, so 128 workgroups run instead of 32 - and that head's slice of to_out:
//                     part[h] = O_h . Wo[:, 64h:64h+64]^T                                  (K = 64 partial of the out-projection)
//   xattn_q2_proj     q = to_q(h) for one head and 32 rows (K = 512 split over the 4 waves), the 64-key radar cross-attention
//                     against the cached K / V^T of the condition, and that head's slice of to_out, likewise as a partial.
//
// The 8 per-head partials are summed, added to the fp32 residual stream with the bias and normalised for the next sub-block
// by reduce_resid_ln_kernel (norm.hip) - the one seam that needs whole rows.  Per block: 8 launches instead of 12.
//
// MFMA operand plumbing (v_mfma_f32_32x32x16_bf16; guide section 3, "an accumulator tile as the next MFMA's operand"):
// every product is oriented so that the next one sums over the previous accumulator's ROW index - S^T = K.Q^T puts keys on
// rows, O^T = V^T.P^T consumes them; O^T has d on rows, part = (O^T)^T.Wo_h^T consumes them as the A operand - so no
// accumulator ever crosses lanes; the other operand is read in the permuted k order (two 8-byte pieces per fragment).
#include "common.h"
#include "kernels.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// fragment of a row-major [rows][K] bf16 matrix in the PERMUTED k order of an accumulator-fed partner operand:
// element j of lane half hf <-> k = k0 + 8*(j>>2) + 4*hf + (j&3)
__device__ __forceinline__ bf16x8 load_perm_frag(const bf16* row_ptr, int k0, int hf) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row_ptr + k0 + 4 * hf);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row_ptr + k0 + 8 + 4 * hf);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// accumulator rows 8s..8s+7 of a 32x32 tile as a bf16 operand fragment (scaled)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s, float scale) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)(x[8 * s + j] * scale);
    return f;
}

// part[q][n0 + lane&31] = acc (rows = queries on the registers, column on the lane): two 128-byte row segments per store
__device__ __forceinline__ void store_part_tile(float* part_row0, int n, const f32x16& acc, int hf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hf;
        part_row0[(int64_t)row * 512 + n] = acc[i];
    }
}

// =================================================================================================================
// self-attention + out-projection partial
// =================================================================================================================
struct SelfProjArgs {
    const bf16* qkv; int64_t ld;      // [batch*NL][ld]: q (pre-scaled by scale*log2e) | k | v at column offsets 0, D, 2D
    const bf16* Wo;                   // [512][512] to_out weight, row = output column n, K-contiguous
    float* part;                      // [heads][batch*NL][512]
    int NL, heads, batch, D;
};

template <int NTW>     // key tiles (64 keys) per wave: NL = 256 * NTW
__global__ __launch_bounds__(256) void attn_self_proj_kernel(SelfProjArgs a) {
    constexpr int VT = 64 * 128;                                   // one V tile: 64 keys x 128 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const bf16* base = a.qkv + (int64_t)b * a.NL * a.ld;
    const bf16* Kb = base + a.D + h * 64;
    const bf16* Vb = base + 2 * a.D + h * 64;

    // ---- V tiles of this wave -> wave-private LDS by LDS-DMA (row-major [key][d], read transposed below)
    unsigned char* sV = smem + wave * NTW * VT;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int j0 = 64 * (wave + 4 * t);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = 8 * p + (lane >> 3);
            const int lcv = (lane & 7) ^ ((row & 2) << 1);
            __builtin_amdgcn_global_load_lds((glb_void*)(Vb + (int64_t)(j0 + row) * a.ld + lcv * 8), (lds_void*)(sV + t * VT + p * 1024), 16, 0, 0);
        }
    }
    // ---- Q fragments (B operand, natural k order) and all K fragments of this wave's tiles (A operand)
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(base + (int64_t)(q0 + r) * a.ld + h * 64 + 8 * hf + 16 * s);
    bf16x8 kf[NTW][2][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                kf[t][u][s] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)(64 * (wave + 4 * t) + 32 * u + r) * a.ld + 8 * hf + 16 * s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // my own DMA pieces have landed (wave-private buffer: no barrier needed)

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        f32x16 st[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[u][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t][u][s], qf[s], st[u], 0, 0, 0);
        }
        float mx = st[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);      // first tile: exp2(-inf) = 0 on zeros
        m = mn;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                st[u][i] = __builtin_amdgcn_exp2f(st[u][i] - m);
                ps += st[u][i];
            }
        l += ps;                                                   // per-half partial of the row sum
        const unsigned char* tV = sV + t * VT;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = acc_frag(st[u], s, 1.0f);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // ds_read_b64_tr_b16 (see attention.hip): group g of 16 lanes reads a 4-key x 16-d block transposed
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const int gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
                    const int krow = 32 * u + 16 * s + 4 * (gq >> 1) + qq;
                    const int lch = 4 * dt + 2 * (gq & 1) + (pp >> 1);
                    const unsigned char* p_lo = tV + krow * 128 + ((lch ^ ((krow & 2) << 1)) << 4) + 8 * (pp & 1);
                    const unsigned char* p_hi = tV + (krow + 8) * 128 + ((lch ^ (((krow + 8) & 2) << 1)) << 4) + 8 * (pp & 1);
                    const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_lo);
                    const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_hi);
                    const bf16x8 vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, l4), __builtin_bit_cast(bf16x4, h4), 0, 1, 2, 3, 4, 5, 6, 7);
                    if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
                }
            }
    }
    l += __shfl_xor(l, 32, 64);

    // ---- merge the four key ranges: M = max m_w; O = sum_w 2^(m_w - M) O_w / sum_w 2^(m_w - M) l_w
    __syncthreads();                                               // every wave is done with its V tiles: LDS is reused
    float* tab = reinterpret_cast<float*>(smem);                   // [4][32][2] = {m, l}
    float4* ex = reinterpret_cast<float4*>(smem + 1024);           // [4 waves][8 groups][64 lanes] float4
    if (hf == 0) *reinterpret_cast<float2*>(tab + (wave * 32 + r) * 2) = make_float2(m, l);
    __syncthreads();
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, tab[(w * 32 + r) * 2]);
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) L += __builtin_amdgcn_exp2f(tab[(w * 32 + r) * 2] - M) * tab[(w * 32 + r) * 2 + 1];
    const float f = __builtin_amdgcn_exp2f(m - M) / L;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ex[(wave * 8 + g) * 64 + lane] = make_float4(o0[4 * g] * f, o0[4 * g + 1] * f, o0[4 * g + 2] * f, o0[4 * g + 3] * f);
        ex[(wave * 8 + 4 + g) * 64 + lane] = make_float4(o1[4 * g] * f, o1[4 * g + 1] * f, o1[4 * g + 2] * f, o1[4 * g + 3] * f);
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 x0 = ex[(w * 8 + g) * 64 + lane], x1 = ex[(w * 8 + 4 + g) * 64 + lane];
            s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
            s1.x += x1.x; s1.y += x1.y; s1.z += x1.z; s1.w += x1.w;
        }
        o0[4 * g] = s0.x; o0[4 * g + 1] = s0.y; o0[4 * g + 2] = s0.z; o0[4 * g + 3] = s0.w;
        o1[4 * g] = s1.x; o1[4 * g + 1] = s1.y; o1[4 * g + 2] = s1.z; o1[4 * g + 3] = s1.w;
    }
    // ---- this head's slice of the out-projection: part[q][n] = sum_d O[q][d] Wo[n][64h + d]; wave w takes n in [128w, 128w+128)
    bf16x8 af[4];
    af[0] = acc_frag(o0, 0, 1.0f); af[1] = acc_frag(o0, 1, 1.0f); af[2] = acc_frag(o1, 0, 1.0f); af[3] = acc_frag(o1, 1, 1.0f);
    float* prow = a.part + ((int64_t)h * a.batch * a.NL + (int64_t)b * a.NL + q0) * 512;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = wave * 128 + nt * 32 + r;
        const bf16* wrow = a.Wo + (int64_t)n * a.D + h * 64;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], load_perm_frag(wrow, 16 * ks, hf), acc, 0, 0, 0);
        store_part_tile(prow, n, acc, hf);
    }
}

int attn_self_proj(const bf16* qkv, int64_t ld, const bf16* Wo, float* part, int NL, int heads, int batch, hipStream_t st) {
    RALD_CHECK(qkv && Wo && part && batch >= 1 && batch <= 65535, "attn_self_proj: bad arguments");
    RALD_CHECK(heads == 8 && ld >= 3 * 512 && ld % 8 == 0, "attn_self_proj: 8 heads x 64 and a fused q|k|v buffer expected");
    RALD_CHECK(NL == 256 || NL == 512, "attn_self_proj: 256 or 512 latents (key tiles split evenly over 4 waves)");
    RALD_CHECK((uintptr_t)qkv % 16 == 0 && (uintptr_t)Wo % 16 == 0 && (uintptr_t)part % 16 == 0, "attn_self_proj: 16-byte alignment");
    SelfProjArgs a;
    a.qkv = qkv; a.ld = ld; a.Wo = Wo; a.part = part; a.NL = NL; a.heads = heads; a.batch = batch; a.D = 512;
    dim3 grid(NL / 32, heads, batch);
    if (NL == 512) {
        static bool attr_set = false;
        if (!attr_set) { RALD_HIP(hipFuncSetAttribute((const void*)attn_self_proj_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 8192)); attr_set = true; }
        hipLaunchKernelGGL(attn_self_proj_kernel<2>, grid, dim3(256), 4 * 2 * 8192, st, a);
    } else {
        hipLaunchKernelGGL(attn_self_proj_kernel<1>, grid, dim3(256), 4 * 8192 + 8192 + 1024, st, a);
    }
    RALD_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================================
// q-projection + 64-key cross-attention + out-projection partial
// =================================================================================================================
struct CrossProjArgs {
    const bf16* hin;                  // [M][512] AdaLN output (A operand of to_q)
    const bf16* Wq;                   // [512][512] attn2.to_q weight
    const bf16* Kc; int64_t ldk, strideK;     // cached condition keys: Kc[b*strideK + key*ldk + 64h + d]
    const bf16* Vt; int64_t ldvt, strideVt;   // cached condition values, transposed: Vt[b*strideVt + (64h + d)*ldvt + key]
    const bf16* Wo;                   // [512][512] attn2.to_out weight
    float* part;                      // [heads][M][512]
    int M, NL;                        // rows, rows per sample
    float qscale;                     // softmax scale * log2(e)
};

__global__ __launch_bounds__(256) void xattn_q2_proj_kernel(CrossProjArgs a) {
    __shared__ __attribute__((aligned(16))) float4 ex[4 * 8 * 64];         // Q^T partials of the 4 K-slices
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int m0 = blockIdx.x * 32, h = blockIdx.y;
    const int b = m0 / a.NL;
    // ---- Q^T[d][row] = sum_c Wq[64h + d][c] h[row][c], this wave's quarter of c (128 columns = 8 k-steps)
    const bf16* wq = a.Wq + (int64_t)(h * 64 + r) * 512 + wave * 128 + 8 * hf;
    const bf16* hr = a.hin + (int64_t)(m0 + r) * 512 + wave * 128 + 8 * hf;
    bf16x8 wa[2][8], hb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        hb[s] = *reinterpret_cast<const bf16x8*>(hr + 16 * s);
        wa[0][s] = *reinterpret_cast<const bf16x8*>(wq + 16 * s);
        wa[1][s] = *reinterpret_cast<const bf16x8*>(wq + 32 * 512 + 16 * s);
    }
    // condition K (A operand of S^T = K.Q^T in the permuted k order) and V^T (A operand of O^T = V^T.P^T): 64 keys
    const bf16* kc = a.Kc + (int64_t)b * a.strideK + h * 64;
    const bf16* vt = a.Vt + (int64_t)b * a.strideVt + (int64_t)(h * 64) * a.ldvt;
    bf16x8 kf[2][4], vf[2][2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[u][s] = load_perm_frag(kc + (int64_t)(32 * u + r) * a.ldk, 16 * s, hf);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) vf[dt][u][s] = load_perm_frag(vt + (int64_t)(32 * dt + r) * a.ldvt, 32 * u + 16 * s, hf);
    f32x16 qt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) qt[t][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) qt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[t][s], hb[s], qt[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            ex[(wave * 8 + 4 * t + g) * 64 + lane] = make_float4(qt[t][4 * g], qt[t][4 * g + 1], qt[t][4 * g + 2], qt[t][4 * g + 3]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float4 x0 = ex[(w * 8 + 4 * t + g) * 64 + lane];
                s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
            }
            qt[t][4 * g] = s0.x; qt[t][4 * g + 1] = s0.y; qt[t][4 * g + 2] = s0.z; qt[t][4 * g + 3] = s0.w;
        }
    // (every wave now holds the whole Q^T [64 d][32 rows]: the rest is small enough to be done redundantly per wave)
    bf16x8 qf[4];
    qf[0] = acc_frag(qt[0], 0, a.qscale); qf[1] = acc_frag(qt[0], 1, a.qscale);
    qf[2] = acc_frag(qt[1], 0, a.qscale); qf[3] = acc_frag(qt[1], 1, a.qscale);
    // ---- S^T[key][row] over the 64 condition tokens, softmax over the keys (lane-local + one cross-half exchange)
    f32x16 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[u][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[u][s], qf[s], st[u], 0, 0, 0);
    }
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            st[u][i] = __builtin_amdgcn_exp2f(st[u][i] - mx);
            l += st[u][i];
        }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // ---- O^T[d][row] = sum_key V^T[d][key] P^T[key][row]
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][u][s], acc_frag(st[u], s, 1.0f), o[dt], 0, 0, 0);
    }
    // ---- this head's slice of to_out as a partial; wave w takes output columns [128w, 128w + 128)
    bf16x8 af[4];
    af[0] = acc_frag(o[0], 0, inv); af[1] = acc_frag(o[0], 1, inv); af[2] = acc_frag(o[1], 0, inv); af[3] = acc_frag(o[1], 1, inv);
    float* prow = a.part + ((int64_t)h * a.M + m0) * 512;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = wave * 128 + nt * 32 + r;
        const bf16* wrow = a.Wo + (int64_t)n * 512 + h * 64;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], load_perm_frag(wrow, 16 * ks, hf), acc, 0, 0, 0);
        store_part_tile(prow, n, acc, hf);
    }
}

int xattn_q2_proj(const bf16* hin, const bf16* Wq, const bf16* Kc, int64_t ldk, int64_t strideK, const bf16* Vt, int64_t ldvt, int64_t strideVt,
                  const bf16* Wo, float* part, int M, int NL, int heads, int n_keys, float qscale, hipStream_t st) {
    RALD_CHECK(hin && Wq && Kc && Vt && Wo && part && M >= 32, "xattn_q2_proj: bad arguments");
    RALD_CHECK(heads == 8 && n_keys == 64, "xattn_q2_proj: 8 heads x 64 and 64 condition tokens expected");
    RALD_CHECK(M % 32 == 0 && NL % 32 == 0 && M % NL == 0, "xattn_q2_proj: rows must come in whole 32-row blocks of one sample");
    RALD_CHECK(ldk % 4 == 0 && ldvt % 4 == 0 && strideK % 4 == 0 && strideVt % 4 == 0 && (uintptr_t)Kc % 8 == 0 && (uintptr_t)Vt % 8 == 0 &&
               (uintptr_t)hin % 16 == 0 && (uintptr_t)Wq % 16 == 0 && (uintptr_t)Wo % 16 == 0, "xattn_q2_proj: alignment");
    CrossProjArgs a;
    a.hin = hin; a.Wq = Wq; a.Kc = Kc; a.ldk = ldk; a.strideK = strideK; a.Vt = Vt; a.ldvt = ldvt; a.strideVt = strideVt; a.Wo = Wo; a.part = part;
    a.M = M; a.NL = NL; a.qscale = qscale;
    hipLaunchKernelGGL(xattn_q2_proj_kernel, dim3(M / 32, heads), dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
