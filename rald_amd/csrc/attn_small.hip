// Small-batch (<= 2048 rows: the reference's eval_batch_size = 1) forms of the two attention
// sub-blocks of BasicTransformerBlock (model/models_radar_generation.py:166-167; CrossAttention :55-76).
//
// At 512 rows a transformer block is a chain of ~12 dependent launches of a few microseconds each, and every launch pays a
// ~1.5 us boundary plus its own ramp: the chain, not the arithmetic, is the cost (1.6 ms per NFE for 132 GFLOP).  These
// two kernels cut the chain by fusing everything that one (head, 32-query block) workgroup can do on its own:
//
//   attn_self_proj    softmax(q k^T) v for one head and 32 queries - one 64-key tile per wave, merged through LDS, so 128
//                     workgroups run instead of 32 - and that head's slice of to_out:
//                     part[h] = O_h . Wo[:, 64h:64h+64]^T                                  (K = 64 partial of the out-projection)
//   xattn_q2_proj     q = to_q(h) for one head and 32 rows (K = 512 split over the 4 waves), the 64-key radar cross-attention
//                     against the cached K / V^T of the condition, and that head's slice of to_out, likewise as a partial.
//
// The 8 per-head partials are summed, added to the fp32 residual stream with the bias and normalised for the next sub-block
// by reduce_resid_ln_kernel (norm.hip) - the one seam that needs whole rows.  Per block: 8 launches instead of 12.
//
// EVERY operand reaches the MFMAs through LDS in whole 128-byte lines (LDS-DMA, XOR-swizzled via the source address).  The
// first version loaded the fragments straight from global memory: 16- and 8-byte pieces of 32 different rows per instruction,
// ~100 such instructions per wave, and the address unit needed 60-120 cycles for each - 5 700 / 11 800 shader clocks just to
// ISSUE the loads of the two kernels (tools/probe/small_stamps.hip), 12 / 14 us per launch.
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

// tools/probe/small_stamps.hip only (never defined in the library): shader-clock stamps of one wave at the phase boundaries
#ifdef RALD_SMALL_STAMPS
__device__ long long g_small_stamps[2][16];
#define RALD_STAMP(K, I) do { if (blockIdx.x == 3 && blockIdx.y == 2 && threadIdx.x == 0) g_small_stamps[K][I] = clock64(); } while (0)
#else
#define RALD_STAMP(K, I) do { } while (0)
#endif

// ---- LDS images -------------------------------------------------------------------------------------------------------
// "line image": rows of 128 bytes (64 bf16), 16-byte chunk c of row n stored at chunk c ^ ((n >> 1) & 7): conflict-free for
// ds_read_b128 fragment reads of 16 consecutive rows at one column (two rows share a 256-byte bank row) - attention.hip.
// One DMA piece = 8 rows x 128 B = one wave instruction; src = address of row 0 column 0, ld in elements.
__device__ __forceinline__ void dma_line_piece(const bf16* src, int64_t ld, int row0, unsigned char* img, int lane) {
    const int row = row0 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    __builtin_amdgcn_global_load_lds((glb_void*)(src + (int64_t)row * ld + lc * 8), (lds_void*)(img + row0 * 128), 16, 0, 0);
}
// natural-order fragment: elements k0 + 8*hf + 0..7 of row n (k0 a multiple of 16)
__device__ __forceinline__ bf16x8 line_frag(const unsigned char* img, int n, int k0, int hf) {
    return *reinterpret_cast<const bf16x8*>(img + n * 128 + ((((k0 >> 3) + hf) ^ ((n >> 1) & 7)) << 4));
}
// permuted-k fragment for a partner operand that comes out of an accumulator: element j of lane half hf <-> k = k0 + 8*(j>>2) + 4*hf + (j&3)
__device__ __forceinline__ bf16x8 line_perm_frag(const unsigned char* img, int n, int k0, int hf) {
    const int sw = (n >> 1) & 7;
    const int c0 = k0 >> 3, o = 8 * hf;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(img + n * 128 + ((c0 ^ sw) << 4) + o);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(img + n * 128 + (((c0 + 1) ^ sw) << 4) + o);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// "wide image": rows of 1 KiB (512 bf16), chunk c of row n at chunk c ^ (n & 15) (low 4 bits of the 64 chunk indices): ds_read_b128
// lane groups read 16 different rows at one column -> 16 different 16-byte slots of the 256-byte bank row.  One DMA piece = one row.
__device__ __forceinline__ void dma_wide_row(const bf16* src_row, int n, unsigned char* img, int lane) {
    __builtin_amdgcn_global_load_lds((glb_void*)(src_row + ((lane ^ (n & 15)) * 8)), (lds_void*)(img + n * 1024), 16, 0, 0);
}
__device__ __forceinline__ bf16x8 wide_frag(const unsigned char* img, int n, int chunk) {
    return *reinterpret_cast<const bf16x8*>(img + n * 1024 + ((chunk ^ (n & 15)) << 4));
}

// accumulator rows 8s..8s+7 of a 32x32 tile as a bf16 operand fragment (scaled)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s, float scale) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)(x[8 * s + j] * scale);
    return f;
}

// part[q][n0 + (lane & 31)] = acc (rows = queries on the registers, column on the lane) through a wave-private 4-KiB LDS patch:
// 4 stores of 16 bytes per lane (8 whole 128-byte row segments each) instead of 16 four-byte stores - the store tail of these
// kernels is issue-bound (~45 clocks per store instruction, guide T21)
// f16 = the slabs are fp16, scaled by 2^-6 and saturated (range +-4.2e6, 11 mantissa bits: below the bf16 rounding of the operands that
// produced them; halves the 8 MB of slab traffic per sub-block: NFE -3 % at B = 1, -6.5 % at B = 2); part_row0 then points at halves.
// Diagnostic: elements whose fp16 slab value was clamped (|v| >= 65504 * 64 = 4.19e6).  Never expected; the stress tests assert 0.
__device__ unsigned g_f16_sat_attn;
int f16_saturation_attn(unsigned* count, bool reset) {
    RALD_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_f16_sat_attn), sizeof(unsigned)));
    if (reset) { const unsigned z = 0; RALD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_f16_sat_attn), &z, sizeof(unsigned))); }
    return 0;
}
__device__ __forceinline__ void store_part_tile(float* part_row0, int n0, const f32x16& acc, float* patch, int lane, bool f16) {
    const int c = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) patch[((i & 3) + 8 * (i >> 2) + 4 * hf) * 32 + c] = acc[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3), ch = lane & 7;
        const float4 v = *reinterpret_cast<const float4*>(patch + row * 32 + 4 * ch);
        if (f16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            constexpr float SC = 0.015625f, LIM = 65504.f;
            h4 hv;
            hv[0] = (_Float16)fminf(fmaxf(v.x * SC, -LIM), LIM); hv[1] = (_Float16)fminf(fmaxf(v.y * SC, -LIM), LIM);
            hv[2] = (_Float16)fminf(fmaxf(v.z * SC, -LIM), LIM); hv[3] = (_Float16)fminf(fmaxf(v.w * SC, -LIM), LIM);
            if (fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) * SC >= LIM) atomicAdd(&g_f16_sat_attn, 1u);
            *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(part_row0) + (int64_t)row * 512 + n0 + 4 * ch) = hv;
        } else *reinterpret_cast<float4*>(part_row0 + (int64_t)row * 512 + n0 + 4 * ch) = v;
    }
}

// =================================================================================================================
// self-attention + out-projection partial (512 latents: 8 waves, one 64-key tile each)
// =================================================================================================================
struct SelfProjArgs {
    const bf16* qkv; int64_t ld;      // [batch*NL][ld]: q (pre-scaled by scale*log2e) | k | v at column offsets 0, D, 2D
    const bf16* Wo;                   // [512][512] to_out weight, row = output column n, K-contiguous
    float* part;                      // [heads][batch*NL][512] fp32, or fp16 x 2^-6 when part_f16
    int NL, heads, batch, D, part_f16;
};

// LDS: [ Q tile 4 KiB | (m, l) table 2 KiB + pad | merged O (bf16) 4 KiB | per wave: K tile 8 KiB, V tile 8 KiB ] = 140 KiB
//   * the wave's K tile is dead after its 8 QK^T MFMAs: its 8 KiB then receive the wave's share of the to_out slice
//     (64 output columns x 64 k), wave-private, in flight under the softmax / PV / merge phases;
//   * the wave's V tile is dead after its PV MFMAs: its 8 KiB then hold the wave's partial O for the merge, later its store patch.
__global__ __launch_bounds__(512) void attn_self_proj_kernel(SelfProjArgs a) {
    constexpr int NW = 8, TILE = 64 * 128, HEAD = 12288;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const s_q = smem;                               // 32 rows x 128 B
    float* const tab = reinterpret_cast<float*>(smem + 4096);      // [8 waves][32 queries] {m, l}
    bf16x4* const obuf = reinterpret_cast<bf16x4*>(smem + 8192);   // [8 groups][64 lanes] x 8 B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const s_k = smem + HEAD + wave * 2 * TILE;
    unsigned char* const s_v = s_k + TILE;
    const int r = lane & 31, hf = lane >> 5;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const bf16* base = a.qkv + (int64_t)b * a.NL * a.ld;
#ifdef RALD_SMALL_STAMPS
    for (int rep = 0; rep < 3; ++rep) {                            // probe only: does a second pass over the same code run faster?
    __syncthreads();
    RALD_STAMP(0, 8 + rep);
#endif
    RALD_STAMP(0, 0);
    // ---- this wave's key tile: K (line image) and V (row-major [key][d], chunk ^ 4 on rows with bit 1 set: read transposed below)
    {
        const bf16* Kt = base + a.D + h * 64 + (int64_t)(64 * wave) * a.ld;
        const bf16* Vt = base + 2 * a.D + h * 64 + (int64_t)(64 * wave) * a.ld;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma_line_piece(Kt, a.ld, 8 * p, s_k, lane);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = 8 * p + (lane >> 3);
            const int lcv = (lane & 7) ^ ((row & 2) << 1);
            __builtin_amdgcn_global_load_lds((glb_void*)(Vt + (int64_t)row * a.ld + lcv * 8), (lds_void*)(s_v + p * 1024), 16, 0, 0);
        }
        if (wave < 4) dma_line_piece(base + (int64_t)q0 * a.ld + h * 64, a.ld, 8 * wave, s_q, lane);
    }
    RALD_STAMP(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                               // the Q tile is shared
    RALD_STAMP(0, 2);
    // ---- S^T = K.Q^T for the two 32-key halves of the tile
    f32x16 st[2];
    {
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = line_frag(s_q, r, 16 * s, hf);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[u][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(line_frag(s_k, 32 * u + r, 16 * s, hf), qf[s], st[u], 0, 0, 0);
        }
    }
    // the K tile is dead: fetch this wave's share of the to_out slice into it (rows n = 64*wave .. +63, columns 64h .. 64h+63)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // every K fragment read has returned
    {
        const bf16* Wn = a.Wo + (int64_t)(64 * wave) * a.D + h * 64;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma_line_piece(Wn, a.D, 8 * p, s_k, lane);
    }
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m = mx;
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            st[u][i] = __builtin_amdgcn_exp2f(st[u][i] - m);
            l += st[u][i];
        }
    l += __shfl_xor(l, 32, 64);
    // ---- O^T = V^T.P^T (unnormalised, relative to this wave's own maximum)
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
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
                const unsigned char* p_lo = s_v + krow * 128 + ((lch ^ ((krow & 2) << 1)) << 4) + 8 * (pp & 1);
                const unsigned char* p_hi = s_v + (krow + 8) * 128 + ((lch ^ (((krow + 8) & 2) << 1)) << 4) + 8 * (pp & 1);
                const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_lo);
                const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_hi);
                const bf16x8 vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, l4), __builtin_bit_cast(bf16x4, h4), 0, 1, 2, 3, 4, 5, 6, 7);
                if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
                else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
            }
        }
    RALD_STAMP(0, 3);

    // ---- merge the 8 key tiles: M = max m_w; O = sum_w 2^(m_w - M) O_w / sum_w 2^(m_w - M) l_w
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // my V reads have returned: the tile's space is mine to overwrite
    {
        // unscaled partial O of this wave -> its V space as [8 groups][64 lanes] float4 (group g = registers 4g..4g+3 of o0, then of o1)
        float4* ex = reinterpret_cast<float4*>(s_v);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ex[g * 64 + lane] = make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
            ex[(4 + g) * 64 + lane] = make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
        }
        if (hf == 0) *reinterpret_cast<float2*>(tab + (wave * 32 + r) * 2) = make_float2(m, l);
    }
    __syncthreads();
    // reduce-scatter: wave w sums group w over the 8 waves, each partial weighted by 2^(m_w' - M) / L, and publishes the group in
    // bf16 (the next MFMA's operand precision)
    {
        float mw[NW], M = -INFINITY, L = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { mw[w] = tab[(w * 32 + r) * 2]; M = fmaxf(M, mw[w]); }
#pragma unroll
        for (int w = 0; w < NW; ++w) { mw[w] = __builtin_amdgcn_exp2f(mw[w] - M); L += mw[w] * tab[(w * 32 + r) * 2 + 1]; }
        const float invL = 1.0f / L;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float4 x0 = reinterpret_cast<const float4*>(smem + HEAD + w * 2 * TILE + TILE)[wave * 64 + lane];
            s0.x = fmaf(x0.x, mw[w], s0.x); s0.y = fmaf(x0.y, mw[w], s0.y); s0.z = fmaf(x0.z, mw[w], s0.z); s0.w = fmaf(x0.w, mw[w], s0.w);
        }
        obuf[wave * 64 + lane] = pack4(s0.x * invL, s0.y * invL, s0.z * invL, s0.w * invL);
    }
    __syncthreads();
    // every wave: the whole normalised O^T as A-operand fragments (k-step ks = 2dt + s <-> groups 2ks, 2ks + 1)
    bf16x8 af[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x4 lo = obuf[(2 * ks) * 64 + lane], hi = obuf[(2 * ks + 1) * 64 + lane];
        af[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    RALD_STAMP(0, 4);
    // ---- this head's slice of the out-projection: part[q][n] = sum_d O[q][d] Wo[n][64h + d]; wave w takes n in [64w, 64w + 64)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my share of the to_out slice has landed (wave-private space)
    const int64_t prow_el = ((int64_t)h * a.batch * a.NL + (int64_t)b * a.NL + q0) * 512;
    float* prow = a.part_f16 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(a.part) + prow_el) : a.part + prow_el;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], line_perm_frag(s_k, 32 * nt + r, 16 * ks, hf), acc, 0, 0, 0);
        store_part_tile(prow, 64 * wave + 32 * nt, acc, reinterpret_cast<float*>(s_v) + nt * 1024, lane, a.part_f16 != 0);     // (every read of the exchange is behind the barrier above)
    }
    RALD_STAMP(0, 5);
#ifdef RALD_SMALL_STAMPS
    RALD_STAMP(0, 12 + rep);
    }
#endif
}

int attn_self_proj(const bf16* qkv, int64_t ld, const bf16* Wo, float* part, int NL, int heads, int batch, hipStream_t st, bool part_f16) {
    RALD_CHECK(qkv && Wo && part && batch >= 1 && batch <= 65535, "attn_self_proj: bad arguments");
    RALD_CHECK(heads == 8 && ld >= 3 * 512 && ld % 8 == 0, "attn_self_proj: 8 heads x 64 and a fused q|k|v buffer expected");
    RALD_CHECK(NL == 512, "attn_self_proj: 512 latents (one 64-key tile per wave of an 8-wave workgroup)");
    RALD_CHECK((uintptr_t)qkv % 16 == 0 && (uintptr_t)Wo % 16 == 0 && (uintptr_t)part % 16 == 0, "attn_self_proj: 16-byte alignment");
    SelfProjArgs a;
    a.qkv = qkv; a.ld = ld; a.Wo = Wo; a.part = part; a.NL = NL; a.heads = heads; a.batch = batch; a.D = 512; a.part_f16 = part_f16 ? 1 : 0;
    constexpr int LDS = 12288 + 8 * 16384;
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)attn_self_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_self_proj_kernel, dim3(NL / 32, heads, batch), dim3(512), LDS, st, a);
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
    float* part;                      // [heads][M][512] fp32, or fp16 x 2^-6 when part_f16
    int part_f16;
    int M, NL;                        // rows, rows per sample
    float qscale;                     // softmax scale * log2(e)
};

// LDS: A [64 KiB]: this head's 64 rows of to_q (wide image) -> later the to_out slice (line image, 512 rows)
//      C [32 KiB]: the 32 input rows (wide image)             -> later the Q^T partials of the 4 K-slices
//      D [16 KiB]: condition K tile [64 keys][64 d] and V^T tile [64 d][64 keys] (line images)
// 8 waves: all of them issue the DMA pieces (the issue cost of ~70 clocks per piece is per wave) and take 64 output columns of the
// to_out slice each; waves 0-3 split the K = 512 of the q-projection; the 64-key attention is small enough for every wave to
// repeat it (each needs the whole O^T as its A operand).
__global__ __launch_bounds__(512) void xattn_q2_proj_kernel(CrossProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const s_a = smem;
    unsigned char* const s_c = smem + 65536;
    unsigned char* const s_kc = smem + 65536 + 32768;
    unsigned char* const s_vt = s_kc + 8192;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int m0 = blockIdx.x * 32, h = blockIdx.y;
    const int b = m0 / a.NL;
    RALD_STAMP(1, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = wave + 8 * i;
        dma_wide_row(a.Wq + (int64_t)(h * 64 + n) * 512, n, s_a, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = wave + 8 * i;
        dma_wide_row(a.hin + (int64_t)(m0 + n) * 512, n, s_c, lane);
    }
    dma_line_piece(a.Kc + (int64_t)b * a.strideK + h * 64, a.ldk, 8 * wave, s_kc, lane);
    dma_line_piece(a.Vt + (int64_t)b * a.strideVt + (int64_t)(h * 64) * a.ldvt, a.ldvt, 8 * wave, s_vt, lane);
    RALD_STAMP(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RALD_STAMP(1, 2);
    // ---- Q^T[d][row] = sum_c Wq[64h + d][c] h[row][c] over this wave's 128 input columns (8 k-steps), waves 0-3
    f32x16 qt[2];
    if (wave < 4) {
        bf16x8 hb[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) hb[s] = wide_frag(s_c, r, wave * 16 + 2 * s + hf);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) qt[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) qt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wide_frag(s_a, 32 * t + r, wave * 16 + 2 * s + hf), hb[s], qt[t], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                               // everyone is done reading to_q and the input rows
    // the to_out slice (rows n, columns 64h..64h+63) replaces to_q; it is needed last and lands under the attention below
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_line_piece(a.Wo + h * 64, 512, 8 * (wave + 8 * i), s_a, lane);
    float4* ex = reinterpret_cast<float4*>(s_c);
    if (wave < 4) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                ex[(wave * 8 + 4 * t + g) * 64 + lane] = make_float4(qt[t][4 * g], qt[t][4 * g + 1], qt[t][4 * g + 2], qt[t][4 * g + 3]);
    }
    __syncthreads();
    RALD_STAMP(1, 3);
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
    RALD_STAMP(1, 4);
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
        for (int s = 0; s < 4; ++s) st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(line_perm_frag(s_kc, 32 * u + r, 16 * s, hf), qf[s], st[u], 0, 0, 0);
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
            for (int s = 0; s < 2; ++s)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(line_perm_frag(s_vt, 32 * dt + r, 32 * u + 16 * s, hf), acc_frag(st[u], s, 1.0f), o[dt], 0, 0, 0);
    }
    RALD_STAMP(1, 5);
    // ---- this head's slice of to_out as a partial; wave w takes output columns [64w, 64w + 64)
    bf16x8 af[4];
    af[0] = acc_frag(o[0], 0, inv); af[1] = acc_frag(o[0], 1, inv); af[2] = acc_frag(o[1], 0, inv); af[3] = acc_frag(o[1], 1, inv);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my pieces of the to_out slice have landed ...
    __syncthreads();                                               // ... and everyone's; nobody reads the Q^T partials any more
    const int64_t prow_el = ((int64_t)h * a.M + m0) * 512;
    float* prow = a.part_f16 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(a.part) + prow_el) : a.part + prow_el;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n0 = wave * 64 + nt * 32;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], line_perm_frag(s_a, n0 + r, 16 * ks, hf), acc, 0, 0, 0);
        store_part_tile(prow, n0, acc, reinterpret_cast<float*>(s_c) + wave * 1024, lane, a.part_f16 != 0);
    }
    RALD_STAMP(1, 6);
}

int xattn_q2_proj(const bf16* hin, const bf16* Wq, const bf16* Kc, int64_t ldk, int64_t strideK, const bf16* Vt, int64_t ldvt, int64_t strideVt,
                  const bf16* Wo, float* part, int M, int NL, int heads, int n_keys, float qscale, hipStream_t st, bool part_f16) {
    RALD_CHECK(hin && Wq && Kc && Vt && Wo && part && M >= 32, "xattn_q2_proj: bad arguments");
    RALD_CHECK(heads == 8 && n_keys == 64, "xattn_q2_proj: 8 heads x 64 and 64 condition tokens expected");
    RALD_CHECK(M % 32 == 0 && NL % 32 == 0 && M % NL == 0, "xattn_q2_proj: rows must come in whole 32-row blocks of one sample");
    RALD_CHECK(ldk % 8 == 0 && ldvt % 8 == 0 && strideK % 8 == 0 && strideVt % 8 == 0 && (uintptr_t)Kc % 16 == 0 && (uintptr_t)Vt % 16 == 0 &&
               (uintptr_t)hin % 16 == 0 && (uintptr_t)Wq % 16 == 0 && (uintptr_t)Wo % 16 == 0, "xattn_q2_proj: 16-byte alignment of every row");
    CrossProjArgs a;
    a.hin = hin; a.Wq = Wq; a.Kc = Kc; a.ldk = ldk; a.strideK = strideK; a.Vt = Vt; a.ldvt = ldvt; a.strideVt = strideVt; a.Wo = Wo; a.part = part;
    a.M = M; a.NL = NL; a.qscale = qscale; a.part_f16 = part_f16 ? 1 : 0;
    constexpr int LDS = 65536 + 32768 + 16384;
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)xattn_q2_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(xattn_q2_proj_kernel, dim3(M / 32, heads), dim3(512), LDS, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
