// Flash-style multi-head attention for head dim 64 on v_mfma_f32_32x32x16_bf16
// (CrossAttention, models_radar_generation.py:66-75; Attention, models_ae.py:91-104).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and the
// whole workgroup streams the keys in tiles of 64 with an online softmax.  Nothing of the
// [nq x nk] score matrix reaches memory (the reference materialises it in fp32).  CDNA4 structure:
//   * K and V^T tiles (8 KB each) are written straight into LDS by LDS-DMA (global_load_lds_dwordx4),
//     double-buffered, one barrier per tile; the 16-byte chunks of the 128-byte LDS rows are
//     XOR-swizzled via the DMA SOURCE address so the fragment reads are bank-conflict free.
//   * S^T = K.Q^T (keys on the accumulator ROWS, the query on the LANE): every lane holds 16 of its
//     query's 32 scores per sub-tile, its partner lane (lane^32) the other 16, so the row max / row
//     sum are in-register ops + ONE cross-half exchange, and the rescale factor is lane-local.
//   * The S^T accumulator is directly the B operand of O^T = V^T.P^T (sum over the accumulator's row
//     index): no LDS round trip for P.  The k order inside a 16-key step is permuted (element j of
//     lane half h is key 16s + 8(j>>2) + 4h + (j&3)), so the V^T fragment is two 8-byte LDS reads in
//     that same order.
//   * V arrives pre-transposed (Vt[b][h*64+d][key], keys contiguous): the producing GEMM is issued
//     with the operand roles swapped, so no transpose pass exists anywhere.
//   * softmax scale and log2(e) ride in the exp2 argument's FMA; masking runs only on a ragged last
//     tile; the O rescale is skipped (wave-uniformly) when no lane's running max moved.
//   * O leaves through a wave-private LDS transpose as whole 128-byte rows.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// RALD_ATTN_ABLATE (tools/probe/attn_ablate.hip only; 0 in the library): bit 0 no v_exp, bit 1 no running max, bit 2 V fragments read once
// per tile, bit 3 no PV MFMA, bit 4 K fragments read once per sub-tile, bit 5 no K/V staging, waits or barriers after the first tile - a way to see which unit bounds the kernel.
#ifndef RALD_ATTN_ABLATE
#define RALD_ATTN_ABLATE 0
#endif
__device__ __forceinline__ float fast_exp2(float x) {
    if constexpr (RALD_ATTN_ABLATE & 1) return x * 0.5f;
    else return __builtin_amdgcn_exp2f(x);
}

#ifndef RALD_ATTN_LAZY
#define RALD_ATTN_LAZY 8.f
#endif
#ifndef RALD_ATTN_PRIO
#define RALD_ATTN_PRIO 1   // raise the wave priority over the QK^T MFMAs (neutral at 512 keys, +3 % at 2048+)
#endif
#ifndef RALD_ATTN_STAGES
#define RALD_ATTN_STAGES 2
#endif
#ifdef RALD_ATTN_WPE
#define RALD_ATTN_ATTR __attribute__((amdgpu_waves_per_eu(RALD_ATTN_WPE, RALD_ATTN_WPE)))
#else
#define RALD_ATTN_ATTR
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// F16: K / V hold fp16 and the queries arrive in fp32 (a.Qf) - the folded set-encoder attention (ae_encode.hip), whose keys are
// Fourier features: 11 mantissa bits instead of 8 on operands that are bounded by construction.  Fragments stay `bf16x8` bit
// containers; only the conversions and the MFMA opcode differ.
template <bool F16>
__device__ __forceinline__ f32x16 attn_mfma(const bf16x8& x, const bf16x8& y, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}

template <bool PRESCALED, bool VROW, bool F16 = false>
__global__ __launch_bounds__(256) RALD_ATTN_ATTR void attention_d64_kernel(AttnArgs a) {
    constexpr int TILE_BYTES = 64 * 128;                       // 64 rows x 128 B (K: keys x d, Vt: d x keys)
    constexpr int NST = RALD_ATTN_STAGES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * 2 * TILE_BYTES];   // [buf][K | Vt]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5;
#if RALD_ATTN_ABLATE & 64
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in launch order, and all the
    // workgroups of one (batch, head) stream the same K/V.  Give every (batch, head) to ONE XCD - launch index L -> XCD L & 7,
    // slot L >> 3 -> (pair slot/nx of that XCD, block slot%nx) - so its K/V crosses the fabric once instead of once per query
    // block (4x at 512 latents: 268 MB -> 67 MB per launch at batch 64, which was the kernel's bound).
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    const int nx = ((a.nq + 127) >> 7) * ksplit;
    const int nbh = a.heads * a.batch;
    int bh, bx;
#ifdef RALD_ATTN_PLAIN_GRID   // probe builds only: the launch-order mapping this kernel had before (PMC before/after in profiles/pmc)
    if (false) {
#else
    if ((nbh & 7) == 0) {
#endif
        const int slot = blockIdx.x >> 3;
        bh = (slot / nx) * 8 + (blockIdx.x & 7);
        bx = slot % nx;
    } else {
        bh = blockIdx.x / nx;
        bx = blockIdx.x % nx;
    }
    const int h = bh % a.heads, b = bh / a.heads;
    const int qblk = bx / ksplit, ks = bx - qblk * ksplit;
    int q0 = (qblk * 4 + wave) * 32;
    const bool active = q0 < a.nq;                             // a ragged last workgroup still helps staging
    if (!active) q0 = a.nq - 32;
    const int64_t qoff = (int64_t)b * a.strideQ + (int64_t)(q0 + r) * a.ldq + h * 64 + 8 * hf;

    // DMA sources: piece p of this wave = tile rows 8*(wave + 4p) .. +7; lane -> (row l>>3, phys chunk l&7)
    // LDS swizzle: physical 16-byte chunk = logical chunk ^ ((row >> 1) & 7).  With 128-byte rows the
    // 256-byte bank row holds two tile rows, so (row parity, chunk ^ f(row)) must be a bijection over
    // the 16 rows a ds_read_b128 lane group touches: f = (row>>1)&7 is, f = row&7 is not (2-4-way).
    const int lr = lane >> 3;
    const bf16* gK[2];
    const bf16* gV[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 8 * (wave + 4 * p) + lr;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        gK[p] = a.K + (int64_t)b * a.strideK + (int64_t)row * a.ldk + h * a.hsk + lc * 8;
        if constexpr (VROW) {
            // V tile = [64 keys][64 d] like K, but read only through transposed 4x16 blocks (4 consecutive keys x 16 d):
            // swizzle chunk ^ 4 on rows with bit 1 set, so that the two row pairs of a block fall into different halves of
            // the 128-byte row (rows r and r+2 share a bank row)
            const int lcv = (lane & 7) ^ ((row & 2) << 1);
            gV[p] = a.V + (int64_t)b * a.strideV + (int64_t)row * a.ldv + h * a.hsk + lcv * 8;
        } else {
            gV[p] = a.Vt + (int64_t)b * a.strideVt + (int64_t)(h * a.hsk + row) * a.ldvt + lc * 8;
        }
    }
    auto stage = [&](int j0, int buf) {
        unsigned char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((glb_void*)(gK[p] + (int64_t)j0 * a.ldk), (lds_void*)(base + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gV[p] + (VROW ? (int64_t)j0 * a.ldv : (int64_t)j0)), (lds_void*)(base + TILE_BYTES + (wave + 4 * p) * 1024), 16, 0, 0);
        }
    };

    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if constexpr (F16) {
            const float4 x0 = *reinterpret_cast<const float4*>(a.Qf + qoff + 16 * s), x1 = *reinterpret_cast<const float4*>(a.Qf + qoff + 16 * s + 4);
            f16x8 f;
            f[0] = (_Float16)x0.x; f[1] = (_Float16)x0.y; f[2] = (_Float16)x0.z; f[3] = (_Float16)x0.w;
            f[4] = (_Float16)x1.x; f[5] = (_Float16)x1.y; f[6] = (_Float16)x1.z; f[7] = (_Float16)x1.w;
            qf[s] = __builtin_bit_cast(bf16x8, f);
        } else qf[s] = *reinterpret_cast<const bf16x8*>(a.Q + qoff + 16 * s);
    }

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    // Running max m and the exponent scale.  PRESCALED: q already carries scale*log2(e) (folded into
    // the projection GEMM's epilogue for free), scores are in exp2 units, and -m sits in a 16-register
    // splat that is the C-INPUT of the first QK^T MFMA: the accumulator comes out as (s - m) and feeds
    // v_exp_f32 directly - no per-score FMA.  The splat is rewritten only when a max moves (rare).
    float m = -1e30f, l = 0.f;
    const float c = PRESCALED ? 1.0f : a.scale * 1.4426950408889634f;
    f32x16 negm;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
    (void)negm;

    const int ntiles_all = (a.nk + 63) / 64;
    const int per = (ntiles_all + ksplit - 1) / ksplit;
    const int t0 = ks * per;
    const int ntiles = t0 + per < ntiles_all ? t0 + per : ntiles_all;         // this workgroup's key tiles: [t0, ntiles)
    if (t0 < ntiles) stage(t0 * 64, t0 % NST);
    if (NST == 3 && t0 + 1 < ntiles) stage((t0 + 1) * 64, (t0 + 1) % NST);
    // one key tile; FIRST (compile-time) peels the tile whose scores are still absolute
    auto tile = [&](const int t, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int j0 = t * 64;
        if (!(RALD_ATTN_ABLATE & 32) || FIRST) {
        if (NST == 3 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile t+1's four pieces may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                          // ... and everyone's; the buffer of tile t-1 is free
        asm volatile("" ::: "memory");
        }
        if (!(RALD_ATTN_ABLATE & 32) && t + NST - 1 < ntiles) stage(j0 + 64 * (NST - 1), (t + NST - 1) % NST);
        const unsigned char* sK = smem + (t % NST) * 2 * TILE_BYTES;
        const unsigned char* sV = sK + TILE_BYTES;

        // ---- S^T = K.Q^T for the two 32-key sub-tiles
#if RALD_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        f32x16 st[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (PRESCALED && !FIRST) st[u] = negm;                // accumulator starts at -m
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[u][i] = 0.f;
            }
            const int krow = 32 * u + r;
            bf16x8 kf;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (!(RALD_ATTN_ABLATE & 16) || s == 0) kf = *reinterpret_cast<const bf16x8*>(sK + krow * 128 + (((2 * s + hf) ^ ((krow >> 1) & 7)) << 4));
                st[u] = attn_mfma<F16>(kf, qf[s], st[u]);
            }
        }
#if RALD_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // st[u][i] = score(key j0 + 32u + (i&3) + 8*(i>>2) + 4*hf, query q0 + r)  [PRESCALED: minus m, exp2 units]
        if (j0 + 64 > a.nk) {                                  // ragged last tile only (wave-uniform)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (j0 + 32 * u + (i & 3) + 8 * (i >> 2) + 4 * hf >= a.nk) st[u][i] = -1e30f;
        }
        float mx = st[0][0];
        if constexpr (RALD_ATTN_ABLATE & 2) mx = FIRST ? 8.f : 0.f;
        else {
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
        if constexpr (PRESCALED) {
            // mx is relative to the reference max m (absolute on the first tile).  m only has to be NEAR the row max - the
            // softmax is invariant to it and fp32 / bf16 exponents have the room - so it moves only when some lane's scores
            // exceed it by more than RALD_ATTN_LAZY (exp2 units, p <= 2^LAZY): with an exact running max the branch below was
            // taken on practically every tile (P(no row of 32 sets a new max) ~ 0), costing 32 O multiplies + 32 subtracts each.
            if (FIRST || __any(mx > RALD_ATTN_LAZY)) {
                const float delta = FIRST ? mx : fmaxf(mx, 0.f);
                if constexpr (FIRST) m = mx;
                else {
                    const float alpha = fast_exp2(-delta);
                    m += delta;
                    l *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[i] = -m;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[u][i] -= delta;
            }
            f32x2 ps2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    st[u][i] = fast_exp2(st[u][i]);
                    st[u][i + 1] = fast_exp2(st[u][i + 1]);
                    ps2 += f32x2{st[u][i], st[u][i + 1]};
                }
            l += ps2[0] + ps2[1];                              // per-half partial; halves summed at the end
        } else {
            if (__any((mx - m) * c > RALD_ATTN_LAZY)) {        // somebody's scores left the reference max far behind: move it
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2((m - mn) * c);
                m = mn;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            const float mc = -m * c;
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    st[u][i] = fast_exp2(fmaf(st[u][i], c, mc));
                    ps += st[u][i];
                }
            l += ps;
        }

        // ---- O^T += V^T.P^T ; element j of the P fragment <-> key 32u + 16s + 8(j>>2) + 4hf + (j&3)
        bf16x4 lo_keep[2], hi_keep[2];
        (void)lo_keep; (void)hi_keep;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
                if constexpr (F16) {
                    f16x8 ph;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ph[j] = (_Float16)st[u][8 * s + j];
                    pf = __builtin_bit_cast(bf16x8, ph);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[u][8 * s + j];
                }
                const int ch = 4 * u + 2 * s;                  // 8-key chunk holding keys 32u+16s .. +7
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x4 lo, hi;
                    if constexpr ((RALD_ATTN_ABLATE & 4) != 0) {
                        if (u == 0 && s == 0) {
                            lo_keep[dt] = *reinterpret_cast<const bf16x4*>(sV + (32 * dt + r) * 128 + 8 * hf);
                            hi_keep[dt] = *reinterpret_cast<const bf16x4*>(sV + (32 * dt + r) * 128 + 8 * hf + 16);
                        }
                        lo = lo_keep[dt]; hi = hi_keep[dt];
                    } else if constexpr (VROW) {
                        // ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies row (key) q, columns 4p..4p+3 of a 4x16 block and
                        // lane i receives column i of the 4 rows.  Group g: d columns 32dt + 16(g&1) + i, keys 32u + 16s + 4(g>>1) + q
                        // (lo) and + 8 (hi) - exactly the element order of the P fragment above.
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                        const int gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
                        const int krow = 32 * u + 16 * s + 4 * (gq >> 1) + qq;
                        const int lch = 4 * dt + 2 * (gq & 1) + (pp >> 1);
                        const unsigned char* p_lo = sV + krow * 128 + ((lch ^ ((krow & 2) << 1)) << 4) + 8 * (pp & 1);
                        const unsigned char* p_hi = sV + (krow + 8) * 128 + ((lch ^ (((krow + 8) & 2) << 1)) << 4) + 8 * (pp & 1);
                        const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_lo);
                        const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_hi);
                        lo = __builtin_bit_cast(bf16x4, l4);
                        hi = __builtin_bit_cast(bf16x4, h4);
                    } else {
                        const int vrow = 32 * dt + r;
                        const unsigned char* vr = sV + vrow * 128 + 8 * hf;
                        lo = *reinterpret_cast<const bf16x4*>(vr + ((ch ^ ((vrow >> 1) & 7)) << 4));
                        hi = *reinterpret_cast<const bf16x4*>(vr + (((ch + 1) ^ ((vrow >> 1) & 7)) << 4));
                    }
                    const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    if constexpr ((RALD_ATTN_ABLATE & 8) != 0) {
                        if (dt == 0) o0[u * 2 + s] += (float)vf[0] * (float)pf[0] + (float)pf[7];
                        else o1[u * 2 + s] += (float)vf[0] * (float)pf[1] + (float)pf[6];
                    } else if (dt == 0) o0 = attn_mfma<F16>(vf, pf, o0);
                    else o1 = attn_mfma<F16>(vf, pf, o1);
                }
            }
    };
    if (t0 < ntiles) {
        tile(t0, std::true_type{});
        for (int t = t0 + 1; t < ntiles; ++t) tile(t, std::false_type{});
    }
    l += __shfl_xor(l, 32, 64);
#if RALD_ATTN_ABLATE & 64
    if (threadIdx.x == 0 && blockIdx.x == 1000 && a.part) { a.part[0] = (float)(clock64() - clk0); a.part[1] = (float)(wall_clock64() - wall0); }
#endif
    if (ksplit > 1) {
        // partial result of this key range: unnormalised O (fp32), running max in exp2 units and the sum - combined by
        // attention_combine_kernel.  Layout: part[((ks*batch + b)*heads + h)*nq + q][66] = {O[0..63], m, l}.
        if (active) {
            float* P = a.part + ((((int64_t)ks * a.batch + b) * a.heads + h) * a.nq + q0 + r) * 66;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float2*>(P + 8 * g + 4 * hf) = make_float2(o0[4 * g], o0[4 * g + 1]);
                *reinterpret_cast<float2*>(P + 8 * g + 4 * hf + 2) = make_float2(o0[4 * g + 2], o0[4 * g + 3]);
                *reinterpret_cast<float2*>(P + 32 + 8 * g + 4 * hf) = make_float2(o1[4 * g], o1[4 * g + 1]);
                *reinterpret_cast<float2*>(P + 32 + 8 * g + 4 * hf + 2) = make_float2(o1[4 * g + 2], o1[4 * g + 3]);
            }
            if (hf == 0) *reinterpret_cast<float2*>(P + 64) = make_float2(t0 < ntiles ? m * c : -1e30f, l);
        }
        return;
    }
    const float inv = 1.0f / l;

    // ---- O out: o{dt}[i] = O^T[d = 32dt + (i&3) + 8(i>>2) + 4hf][query q0+r].  Transpose through a
    // wave-private LDS patch [32 queries][128 B + 16] and store whole 128-byte rows.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // all waves are done with the K/V buffers
    asm volatile("" ::: "memory");
    unsigned char* patch = smem + wave * (32 * 144);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<bf16x4*>(patch + r * 144 + (8 * g + 4 * hf) * 2) =
            pack4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<bf16x4*>(patch + r * 144 + (32 + 8 * g + 4 * hf) * 2) =
            pack4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
    if (active) {
        bf16* O = a.O + (int64_t)b * a.strideO + (int64_t)q0 * a.ldo + h * 64;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
            const int row = r0 + (lane >> 3), pc = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(patch + row * 144 + pc * 16);
            *reinterpret_cast<uint4*>(O + (int64_t)row * a.ldo + pc * 8) = v;
        }
    }
}


// O[b][q][h*64 + d] = sum_s 2^(m_s - M) O_s[d] / sum_s 2^(m_s - M) l_s, M = max_s m_s; one wave per (b, h, q), lane = d.
// Lane s fetches split s's (m, l), so the weights of all <= 64 splits come from one round trip, and the partial rows are then
// independent loads (the first version walked the splits twice with dependent loads: 8-10 us for 16 splits).
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ part, int ksplit, int batch, int heads, int nq,
                                                                bf16* __restrict__ O, int64_t ldo, int64_t strideO) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (b*heads + h)*nq + q
    const int64_t rows = (int64_t)batch * heads * nq;
    if (row >= rows) return;
    float ms = -1e30f, ls = 0.f;
    if (lane < ksplit) {
        const float2 ml = *reinterpret_cast<const float2*>(part + ((int64_t)lane * rows + row) * 66 + 64);
        ms = ml.x; ls = ml.y;
    }
    const float M = wave_max(ms);
    const float w = lane < ksplit ? fast_exp2(ms - M) : 0.f;
    const float L = wave_sum(w * ls);
    float acc0 = 0.f, acc1 = 0.f;
    int s = 0;
    for (; s + 4 <= ksplit; s += 4) {
        const float p0 = part[((int64_t)s * rows + row) * 66 + lane], p1 = part[((int64_t)(s + 1) * rows + row) * 66 + lane];
        const float p2 = part[((int64_t)(s + 2) * rows + row) * 66 + lane], p3 = part[((int64_t)(s + 3) * rows + row) * 66 + lane];
        acc0 = fmaf(__shfl(w, s, 64), p0, acc0); acc1 = fmaf(__shfl(w, s + 1, 64), p1, acc1);
        acc0 = fmaf(__shfl(w, s + 2, 64), p2, acc0); acc1 = fmaf(__shfl(w, s + 3, 64), p3, acc1);
    }
    for (; s < ksplit; ++s) acc0 = fmaf(__shfl(w, s, 64), part[((int64_t)s * rows + row) * 66 + lane], acc0);
    const int q = (int)(row % nq);
    const int64_t bh = row / nq;
    const int h = (int)(bh % heads);
    const int64_t b = bh / heads;
    O[b * strideO + (int64_t)q * ldo + h * 64 + lane] = (bf16)((acc0 + acc1) / L);
}

// few (query block, head, batch) workgroups and many key tiles: split the keys so that ~512 workgroups run (two per CU: a lone workgroup
// walks its tiles as one dependent chain of ~1.2 us each), at least 4 tiles per workgroup
int attention_pick_ksplit(int nq, int nk, int heads, int batch) {
    const int64_t wgs = (int64_t)cdiv(nq, 128) * heads * batch;
    const int ntiles = cdiv(nk, 64);
    if (wgs >= 512 || ntiles < 16) return 1;
    int ks = (int)cdiv((int64_t)512, wgs);
    if (ks > ntiles / 4) ks = ntiles / 4;
    if (ks > 32) ks = 32;
    return ks < 2 ? 1 : ks;
}

int attention_d64(const AttnArgs& a, hipStream_t st) {
    RALD_CHECK(a.nq > 0 && a.nk > 0 && a.heads > 0 && a.batch > 0, "attention: empty problem");
    RALD_CHECK(a.nq % 32 == 0, "attention: nq must be a multiple of 32");
    RALD_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 8 == 0, "attention: leading dimensions must be multiples of 8 elements (16-byte rows)");
    RALD_CHECK(a.k_rows >= round_up(a.nk, 64), "attention: K must have rows allocated up to a multiple of 64 keys (the tail tile is staged whole)");
    RALD_CHECK(((uintptr_t)a.K % 16 == 0) && ((uintptr_t)a.O % 16 == 0), "attention: pointers must be 16-byte aligned");
    RALD_CHECK(a.hsk == 64 || a.hsk == 0, "attention: the heads read K/V 64 columns apart or all the same 64 columns");
    const bool vrow = a.V != nullptr;
    if (a.f16) {
        RALD_CHECK(a.Qf && (uintptr_t)a.Qf % 16 == 0 && a.ldq % 4 == 0 && a.q_prescaled && vrow, "attention: the fp16 form takes fp32 pre-scaled queries and a row-major V");
    } else {
        RALD_CHECK(a.Q && (uintptr_t)a.Q % 16 == 0, "attention: Q must be 16-byte aligned");
    }
    if (vrow) {
        RALD_CHECK(a.Vt == nullptr && (a.nk % 64 == 0 || a.v_padded) && a.ldv % 8 == 0 && (uintptr_t)a.V % 16 == 0,
                   "attention: row-major V needs 16-byte rows and nk % 64 == 0 (or rows zero-filled up to k_rows: v_padded)");
    } else {
        RALD_CHECK(a.Vt && a.ldvt % 8 == 0 && (uintptr_t)a.Vt % 16 == 0, "attention: Vt must be 16-byte aligned with 16-byte rows");
        RALD_CHECK(a.ldvt >= round_up(a.nk, 64), "attention: Vt rows must be padded (finite values) to a multiple of 64 keys");
    }
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    if (ksplit > 1) RALD_CHECK(a.part && ksplit <= 64 && (uintptr_t)a.part % 8 == 0, "attention: key split needs a scratch buffer (attention_split_scratch_bytes)");
    RALD_CHECK((int64_t)cdiv(a.nq, 128) * ksplit * a.heads * a.batch < (1ll << 31), "attention: too many workgroups");
    dim3 grid(cdiv(a.nq, 128) * ksplit * a.heads * a.batch);
    if (a.f16) hipLaunchKernelGGL((attention_d64_kernel<true, true, true>), grid, dim3(256), 0, st, a);
    else if (a.q_prescaled) {
        if (vrow) hipLaunchKernelGGL((attention_d64_kernel<true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attention_d64_kernel<true, false>), grid, dim3(256), 0, st, a);
    } else {
        if (vrow) hipLaunchKernelGGL((attention_d64_kernel<false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attention_d64_kernel<false, false>), grid, dim3(256), 0, st, a);
    }
    if (ksplit > 1) {
        const int64_t rows = (int64_t)a.batch * a.heads * a.nq;
        hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a.part, ksplit, a.batch, a.heads, a.nq, a.O,
                           a.ldo, a.strideO);
    }
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
