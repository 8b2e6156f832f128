// Residual GEMM with the NEXT LayerNorm fused into its epilogue (N = 512 = one full row per tile):
//
//     x[m][:]  += A[m][:] . W^T + bias                      (x = f(x) + x, fp32 residual stream)
//     h[m][:]   = LN(x[m][:]) * (add_one + g[s][:]) + b[s][:]    (bf16, the next GEMM's A operand)
//
// i.e. `to_out(...) + x` followed by AdaLayerNorm / PreNorm-LayerNorm of the following sub-block
// (models_radar_generation.py:166-168 + :127-131; models_ae.py:413-414 + :41-42).  Unfused, the
// LayerNorm is a separate HBM-bound pass that re-reads the 4-byte stream; fused, x is read once
// and written once per sub-block and three launches per transformer block disappear.
//
// Tile BM x 512 x 64, 8 waves as WM x WN, LDS-DMA staged double buffer (same engine as
// gemm_nt_glds_kernel; at BM = 128 the two stages take exactly the CU's 160 KiB).  Epilogue:
//   1. v = acc + bias + x_old in the accumulator layout (lane: 4 columns of 16 rows per m-tile);
//   2. per-row sum / sum-of-squares: 32 in-lane values, 2 cross-lane steps, then across the WN waves
//      through a 4 KiB LDS table and ONE workgroup barrier (single-pass variance in fp32: 512 terms);
//   3. x_new and h leave through the wave-private LDS transpose as whole rows (16 B per lane).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "mx8.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#ifdef RALD_LN_STAMPS   // tools/probe/ln_timeline.hip: shader clocks a workgroup spends waiting at the tile hand-overs vs in the whole k-loop
__device__ long long g_ln_stamps[1024][4];
#endif

// ---- epilogue shared by the main-loop forms: acc (+ x_old already inside unless XEPI) -> x_new, h ---------------------------------
template <int BM, int WM, int WN, bool XEPI>
__device__ __forceinline__ void resid_ln_epilogue(const GemmLnArgs& a, f32x4 (&acc)[BM / (16 * WM)][512 / (16 * WN)], unsigned char* smem, int lds_bytes, int m0) {
    constexpr int BN = 512;
    constexpr int WAVES = WM * WN;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    constexpr int ROWB_F = NT * 16 * 4, STRIDE_F = ROWB_F + 16;      // fp32 patch row (x_new)
    constexpr int ROWB_H = NT * 16 * 2, STRIDE_H = ROWB_H + 16;      // bf16 patch row (h)
    constexpr int PATCH = 16 * STRIDE_F;
    constexpr int RED_OFF = WAVES * PATCH;                            // float2 red[BM][WN]
    (void)lds_bytes; (void)ROWB_H;
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const bool nt_io = (a.nt_io & 1) != 0;
    const bool skip_x = RALD_ABLATED(a.nt_io, 2), skip_h = RALD_ABLATED(a.nt_io, 4);   // probe builds, RALD_NT_STORE bits 1 / 2: timing ablations
    // ---- 1. v = acc + bias + x_old (accumulator layout), row partial sums -------------------------
    const int mb = m0 + wm * (BM / WM);
    const int nb = wn * (BN / WN);
    float s1[MT], s2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = mb + i * 16 + fr;
        m = m < a.M ? m : a.M - 1;
        s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = nb + j * 16 + 4 * fq;
            const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
            // (bf16 operands: x_old is already in the accumulators, see the main loop)
            float4 xo = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XEPI) {
                const nt_f32x4 xv = nt_io ? __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.x + (int64_t)m * BN + n))
                                          : *reinterpret_cast<const nt_f32x4*>(a.x + (int64_t)m * BN + n);
                xo = make_float4(xv[0], xv[1], xv[2], xv[3]);
            }
            f32x4 v = acc[i][j];
            v[0] += b.x + xo.x; v[1] += b.y + xo.y; v[2] += b.z + xo.z; v[3] += b.w + xo.w;
            acc[i][j] = v;
            s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        s1[i] += __shfl_xor(s1[i], 16, 64); s2[i] += __shfl_xor(s2[i], 16, 64);
        s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
        asm volatile("" ::: "memory");            // keep only one m-tile's x_old loads in flight (register pressure)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // staging buffers are dead: reuse them (patches + red)
    asm volatile("" ::: "memory");
    float2* red = reinterpret_cast<float2*>(smem + RED_OFF);
    if (fq == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) red[(wm * (BM / WM) + i * 16 + fr) * WN + wn] = make_float2(s1[i], s2[i]);
    }
    __syncthreads();
    float mean[MT], rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) {
            const float2 p = red[(wm * (BM / WM) + i * 16 + fr) * WN + w];
            t1 += p.x; t2 += p.y;
        }
        mean[i] = t1 * (1.0f / BN);
        const float var = fmaxf(t2 * (1.0f / BN) - mean[i] * mean[i], 0.f);
        rstd[i] = rsqrtf(var + a.eps);
    }

    // ---- 2. x_new (fp32) and h (bf16) out as whole rows through the wave-private patch -------------
    unsigned char* patch = smem + wave * PATCH;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // x_new
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<float4*>(patch + fr * STRIDE_F + (16 * j + 4 * fq) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        {
            constexpr int LPR = ROWB_F / 16, RPI = 64 / LPR;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RPI) {
                const int r = r0 + lane / LPR, pc = lane % LPR;
                const int m = mb + i * 16 + r;
                const uint4 v = *reinterpret_cast<const uint4*>(patch + r * STRIDE_F + pc * 16);
                if (m < a.M && !skip_x) {
                    typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
                    if (nt_io) __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(a.x + (int64_t)m * BN + nb + pc * 4));
                    else *reinterpret_cast<uint4*>(a.x + (int64_t)m * BN + nb + pc * 4) = v;
                }
            }
        }
        // h = (v - mean) * rstd * (add_one + g) + b ; modulation row of this row's sample
        int m = mb + i * 16 + fr;
        m = m < a.M ? m : a.M - 1;
        const int64_t goff = (int64_t)(m / a.rows_per_group) * a.gstride;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = nb + j * 16 + 4 * fq;
            const float4 gg = *reinterpret_cast<const float4*>(a.g + goff + n);
            const float4 bb = *reinterpret_cast<const float4*>(a.b + goff + n);
            const f32x4 v = acc[i][j];
            *reinterpret_cast<bf16x4*>(patch + fr * STRIDE_H + (16 * j + 4 * fq) * 2) =
                pack4((v[0] - mean[i]) * rstd[i] * (a.add_one + gg.x) + bb.x, (v[1] - mean[i]) * rstd[i] * (a.add_one + gg.y) + bb.y,
                      (v[2] - mean[i]) * rstd[i] * (a.add_one + gg.z) + bb.z, (v[3] - mean[i]) * rstd[i] * (a.add_one + gg.w) + bb.w);
        }
        {
            constexpr int LPR = ROWB_H / 16, RPI = 64 / LPR;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RPI) {
                const int r = r0 + lane / LPR, pc = lane % LPR;
                const int mm = mb + i * 16 + r;
                const uint4 v = *reinterpret_cast<const uint4*>(patch + r * STRIDE_H + pc * 16);
                if (a.h8) {
                    // MXFP8 output: this lane's 16-byte piece is 8 consecutive columns, 4 consecutive lanes = one 32-column
                    // block (nb and the pieces are 32-column aligned); every lane of the wave takes part in the shuffles
                    const bf16x8 hv = *reinterpret_cast<const bf16x8*>(&v);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)hv[e];
                    const int64_t mc = mm < a.M ? mm : a.M - 1;
                    const int col = nb + pc * 8;
                    unsigned char q8[8] __attribute__((aligned(8)));
                    unsigned char sc;
                    mx8_block(f, q8, &sc, true);
                    if (mm < a.M) {
                        *reinterpret_cast<uint2*>(a.h8 + mc * BN + col) = *reinterpret_cast<const uint2*>(q8);
                        if ((pc & 3) == 0) a.hs[mc * (BN / 32) + col / 32] = sc;
                    }
                } else if (mm < a.M && !skip_h) {
                    typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
                    if (nt_io) __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(a.h + (int64_t)mm * BN + nb + pc * 8));
                    else *reinterpret_cast<uint4*>(a.h + (int64_t)mm * BN + nb + pc * 8) = v;
                }
            }
        }
    }
}

// BK = 64: LDS rows of 128 bytes, 8 chunks, chunk ^ (row & 7).  BK = 32 (bf16 only): rows of 64 bytes, 4 chunks, chunk ^ ((row >> 2) & 3) -
// rows r, r+4, r+8, r+12 share a 256-byte bank row, so a ds_read_b128 lane group (16 rows at one logical chunk) touches all 16 slots of it.
// With 64-row tiles the BK = 32 form needs 72 KiB of LDS: TWO workgroups per CU, so one streams its epilogue (196 KB of stores) while
// the other runs its k-loop, and a k-step that waits for HBM leaves the MFMAs to the other workgroup.
template <int BM, int WM, int WN, bool MX, int BK, bool XLOOP = true>
__device__ __forceinline__ void gemm_resid_ln_body(const GemmLnArgs& a) {
    constexpr int BN = 512, NSTAGE = 2;
    static_assert(BK == 64 || (BK == 32 && !MX), "k-step: 64, or 32 for bf16 operands");
    constexpr int WAVES = WM * WN;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    constexpr int ROWB = MX ? 128 : BK * 2;                      // bytes per LDS row
    constexpr int CPR = ROWB / 16;                               // 16-byte chunks per row
    constexpr int RPP = 1024 / ROWB;                             // rows per DMA piece (one wave instruction = 1 KiB)
    constexpr int PA = BM / RPP, PB = BN / RPP;                  // pieces per stage
    constexpr int CA = (PA + WAVES - 1) / WAVES;
    constexpr int CB = PB / WAVES;
    static_assert(CB >= 1 && PB % WAVES == 0 && MT >= 1 && NT >= 1 && (PA % WAVES == 0 || PA < WAVES), "tile/wave split");
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int ROWB_F = NT * 16 * 4, STRIDE_F = ROWB_F + 16;      // fp32 patch row (x_new)
    constexpr int ROWB_H = NT * 16 * 2, STRIDE_H = ROWB_H + 16;      // bf16 patch row (h)
    constexpr int PATCH = 16 * STRIDE_F;
    constexpr int RED_OFF = WAVES * PATCH;                            // float2 red[BM][WN]
    static_assert(RED_OFF + BM * WN * 8 <= NSTAGE * STAGE_BYTES, "epilogue scratch must fit in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mtile = blockIdx.x;
    if (a.strideW != 0) {
        // per-group weights: keep a group's tiles on one XCD (launch index g -> XCD g & 7), so its weights cross the fabric once
        const int tg = a.w_rows / BM, ngroups = (int)gridDim.x / (tg > 0 ? tg : 1);
        if (tg >= 1 && tg < 8 && ngroups * tg == (int)gridDim.x && (ngroups & 7) == 0) {
            const int slot = (int)blockIdx.x >> 3;
            mtile = ((slot / tg) * 8 + ((int)blockIdx.x & 7)) * tg + slot % tg;
        }
    }
    const int m0 = mtile * BM;
    const int lr = lane / CPR;                                    // row inside a DMA piece
    const int lc = CPR == 8 ? ((lane & 7) ^ lr) : ((lane & 3) ^ ((lr >> 2) & 3));     // source chunk that lands in physical chunk lane % CPR
    // operand rows per k-step: 128 bytes = 64 bf16 or 128 e4m3 (MX: e4m3 + e8m0 per 32, see gemm_fp8.hip); 64 bytes = 32 bf16
    constexpr int ESZ = MX ? 1 : 2;
    const unsigned char* A0 = MX ? a.A8 : reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* W0 = MX ? a.W8 : reinterpret_cast<const unsigned char*>(a.W + (int64_t)(m0 / a.w_rows) * a.strideW);
    const unsigned char* gA[CA];
    const unsigned char* gB[CB];
#pragma unroll
    for (int p = 0; p < CA; ++p) {
        int r = m0 + RPP * (wave + WAVES * p) + lr;
        r = r < a.M ? r : a.M - 1;
        gA[p] = A0 + ((int64_t)r * a.lda) * ESZ + lc * 16;
    }
#pragma unroll
    for (int p = 0; p < CB; ++p) gB[p] = W0 + ((int64_t)(RPP * (wave + WAVES * p) + lr) * a.ldw) * ESZ + lc * 16;
    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < CA; ++p)
            if (PA >= WAVES || wave < PA)                           // fewer A pieces than waves (64 rows x 64-byte rows): the first PA waves stage them
                __builtin_amdgcn_global_load_lds((glb_void*)(gA[p] + kt * ROWB), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < CB; ++p)
            __builtin_amdgcn_global_load_lds((glb_void*)(gB[p] + kt * ROWB), (lds_void*)(base + BM * ROWB + (wave + WAVES * p) * 1024), 16, 0, 0);
    };

    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    const bool nt_io_ = (a.nt_io & 1) != 0;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = MX ? a.K / 128 : a.K / BK;
    (void)lr;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (MX) {
        typedef int i32x8 __attribute__((ext_vector_type(8)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const int kb = a.K / 32;
        int offA[MT], offB[NT], sa_n[MT], sb_n[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int r = m0 + wm * (BM / WM) + i * 16 + fr;
            r = r < a.M ? r : a.M - 1;
            offA[i] = r * kb + fq;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) offB[j] = (wn * (BN / WN) + j * 16 + fr) * kb + fq;
        auto load_scales = [&](int kt) {
#pragma unroll
            for (int i = 0; i < MT; ++i) sa_n[i] = a.SA[offA[i] + kt * 4];
#pragma unroll
            for (int j = 0; j < NT; ++j) sb_n[j] = a.SW[offB[j] + kt * 4];
        };
        auto read_frag = [&](const unsigned char* tile_base, int row) -> i32x8 {
            const i32x4* sp = reinterpret_cast<const i32x4*>(tile_base) + row * 8;
            const i32x4 lo = sp[fq ^ (row & 7)], hi = sp[(fq + 4) ^ (row & 7)];
            return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        stage(0, 0);
        load_scales(0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int sa[MT], sb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) sa[i] = sa_n[i];
#pragma unroll
            for (int j = 0; j < NT; ++j) sb[j] = sb_n[j];
            if (kt + 1 < nk) { stage(kt + 1, (kt + 1) & 1); load_scales(kt + 1); }
            const unsigned char* tA = smem + (kt & 1) * STAGE_BYTES;
            const unsigned char* tB = tA + BM * 128;
            i32x8 fa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag(tA, wm * (BM / WM) + i * 16 + fr);
            i32x8 fb = read_frag(tB, wn * (BN / WN) + fr);          // one n-tile ahead of its MFMAs
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
    } else {
    // The residual x_old rides into the accumulators DURING the main loop: k-step q < 8 loads one eighth of this wave's x
    // tile (MT*NT/8 float4 per lane) right after the next stage's DMA is issued, and k-step q+1 adds it to the accumulators.
    // With one 160-KiB workgroup per CU every CU is in the same phase at the same time, so an epilogue that first READS
    // 67 MB of x leaves the MFMA pipe idle chip-wide while HBM streams, and HBM idle while the MFMAs run; spreading the read
    // over the k-loop overlaps the two (K = 512 at M = 32768: 47 -> 45 us; the rest of the epilogue is the 100 MB of stores, ~10 us,
    // and a main loop whose k-steps each wait one HBM latency for the A panel at prefetch distance 1).
    constexpr int XP = MT * NT / 8;                                   // float4 pieces per k-step
    static_assert(MT * NT % 8 == 0, "x pieces per k-step");
    nt_f32x4 xt[XP];
    const int mb_ = m0 + wm * (BM / WM), nb_ = wn * (BN / WN);
    // x_old addresses as uniform base + one 32-bit lane offset per m-tile + immediates (64-bit lane pointers per piece were hoisted out of
    // the k-loop by the compiler: 64 registers of loop-invariant addresses next to 128 accumulators)
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x) + (int64_t)mb_ * BN * 4;      // uniform
    unsigned xoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = i * 16 + fr;
        r = mb_ + r < a.M ? r : a.M - 1 - mb_;
        xoff[i] = (unsigned)r * (BN * 4) + (unsigned)(nb_ + 4 * fq) * 4;
    }
    auto x_load = [&](auto QC) {
        constexpr int q = decltype(QC)::value;
#pragma unroll
        for (int e = 0; e < XP; ++e) {
            const int idx = q * XP + e, i = idx / NT, j = idx % NT;
            const unsigned char* sp = xbase;
            asm volatile("" : "+s"(sp));                               // (keeps the address in the saddr + lane offset + immediate form)
            const nt_f32x4* px = reinterpret_cast<const nt_f32x4*>(sp + xoff[i] + j * 64);
            xt[e] = nt_io_ ? __builtin_nontemporal_load(px) : *px;
        }
    };
    auto x_add = [&](auto QC) {
        constexpr int q = decltype(QC)::value;
#pragma unroll
        for (int e = 0; e < XP; ++e) {
            const int idx = q * XP + e, i = idx / NT, j = idx % NT;
            acc[i][j][0] += xt[e][0]; acc[i][j][1] += xt[e][1]; acc[i][j][2] += xt[e][2]; acc[i][j][3] += xt[e][3];
        }
    };
    auto kstep = [&](int kt, auto QC) {
        constexpr int q = decltype(QC)::value;                        // 0..7: x piece of this k-step; 8: add the last piece; 9: none
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        if constexpr (q >= 1 && q <= 8) x_add(std::integral_constant<int, q - 1>{});
        if constexpr (q <= 7) x_load(QC);
        const bf16x8* sA = reinterpret_cast<const bf16x8*>(smem + (kt & 1) * STAGE_BYTES);
        const bf16x8* sB = sA + BM * CPR;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 fa[MT], fb[NT];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * (BM / WM) + i * 16 + fr;
                fa[i] = sA[r * CPR + (chunk ^ (CPR == 8 ? (r & 7) : ((r >> 2) & 3)))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * (BN / WN) + j * 16 + fr;
                fb[j] = sB[r * CPR + (chunk ^ (CPR == 8 ? (r & 7) : ((r >> 2) & 3)))];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (XLOOP && BK == 64 && MT == 4 && NT == 8 && CA + CB == 10) {
        // ---- pipelined main loop of the 128-row form (round 3) ----------------------------------------------------------------------
        // An iteration starts right AFTER a tile hand-over (barrier), so no LDS read is pending at the loop head and the compiler's
        // waits inside are exact counts.  Fragments: the A side (4 m-tiles) is double-buffered per 32-deep sub-step, the B side (8
        // n-tiles) is refreshed IN PLACE: the read of n-tile j for the next sub-step goes out right behind the 4 MFMAs that consumed
        // the current one (LDS returns in order; the register hazard is the hardware's).  The 10 DMA pieces of the next tile and the
        // 12 reads are spread between the MFMAs instead of standing in front of them (round 2: wait - barrier - 10 DMA issues - 12
        // reads - wait - 32 MFMAs - 12 reads - wait - 32 MFMAs, i.e. ~1300 idle matrix-pipe cycles per 2048-cycle k-step).
        constexpr int W_ALL = 0x0070;                                  // vmcnt(0) lgkmcnt(0)
        constexpr int W_ST1 = ((CA + CB) & 15) | (((CA + CB) >> 4) << 14) | 0x0f70;
        bf16x8 fa[MT], fb[NT];
        // DMA sources as scalar base + 32-bit lane offset (the saddr form of global_load_lds): 3 VGPRs instead of the 20 that ten
        // 64-bit lane pointers take - the register file has 128 accumulators, 48 fragment registers and 16 of x_old to hold
        const unsigned char* sbA = A0 + (int64_t)m0 * a.lda * ESZ;                    // uniform
        unsigned voA[CA];
#pragma unroll
        for (int p = 0; p < CA; ++p) {
            int r = RPP * (wave + WAVES * p) + lr;
            r = m0 + r < a.M ? r : a.M - 1 - m0;
            voA[p] = (unsigned)r * (unsigned)(a.lda * ESZ) + (unsigned)lc * 16u;
        }
        const unsigned voB = (unsigned)(RPP * wave + lr) * (unsigned)(a.ldw * ESZ) + (unsigned)lc * 16u;
        const unsigned stepB = (unsigned)(RPP * WAVES) * (unsigned)(a.ldw * ESZ);     // uniform: bytes between this wave's W pieces
        // MEASURED, NOT SHIPPED (-DRALD_KOFF builds only): +0.8 % per NFE at B = 64 and B = 128, but an element's fp32 accumulation order then
        // depends on its tile and on the engine that ran it, so the same sample comes out bit-different in batches of different size (the
        // bf16 roundings downstream flip) - every kernel here otherwise sums k in ascending 32-chunks whatever the tile shape.
        // k-steps are walked from a per-tile OFFSET, wrapping around (the sum over k does not care where it starts): launched together,
        // all 256 workgroups would otherwise ask their XCD's L2 for the SAME 64-KiB k-slice of W at the same moment, 32 requesters per
        // cache line, at every one of the k-steps - a single-round launch (B = 64: one tile per CU) ran its k-loop 37 % waiting at the
        // hand-overs, while the same kernel with two rounds of tiles (B = 128, naturally out of step) delivered 1.6 x the FLOP/s.
        // The offset is a function of the tile's row index alone ((index mod 256) / 8 + index mod 8, mod nk): the 32 tiles that share an XCD
        // (launch index mod 8 equal) start on different slices, the 4 tiles of a sample with per-sample weights too, and a sub-batch that
        // starts at a multiple of 64 samples (= 256 tiles) reproduces the whole batch bit for bit.
#ifndef RALD_KOFF            // off in the shipped build: see below (A/B builds: tools/build_variant.sh koff -DRALD_KOFF)
        const int koff = 0;
#else
        const int koff = RALD_ABLATED(a.nt_io, 8) ? 0 : (int)((unsigned)(((mtile & 255) >> 3) + (mtile & 7)) % (unsigned)nk);
#endif
        auto ksrc = [&](int kt) { const int k = kt + koff; return k >= nk ? k - nk : k; };
        auto stage2 = [&](int kt, int buf) {
            unsigned char* base = smem + buf * STAGE_BYTES;
            const unsigned char* ka = sbA + ksrc(kt) * ROWB;
            const unsigned char* kb = W0 + ksrc(kt) * ROWB;
#pragma unroll
            for (int p = 0; p < CA; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(ka + voA[p]), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
#pragma unroll
            for (int p = 0; p < CB; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(kb + (size_t)p * stepB + voB), (lds_void*)(base + BM * ROWB + (wave + WAVES * p) * 1024), 16, 0, 0);
        };
        auto rdA = [&](int buf, int kk, int i) -> bf16x8 {
            const bf16x8* sA = reinterpret_cast<const bf16x8*>(smem + buf * STAGE_BYTES);
            const int r = wm * (BM / WM) + i * 16 + fr;
            return sA[r * 8 + ((kk * 4 + fq) ^ (r & 7))];
        };
        auto rdB = [&](int buf, int kk, int j) -> bf16x8 {
            const bf16x8* sB = reinterpret_cast<const bf16x8*>(smem + buf * STAGE_BYTES) + BM * 8;
            const int r = wn * (BN / WN) + j * 16 + fr;
            return sB[r * 8 + ((kk * 4 + fq) ^ (r & 7))];
        };
        // One 32-deep sub-step = 32 MFMAs on the fragments in registers, with the 12 fragment reads of the NEXT sub-step (buffer nbuf,
        // half nkk) going out in place as soon as a fragment's last MFMA has been issued:
        //   head   (i,0) (i,1) for i = 0..3        then fb[0], fb[1] are re-read
        //   middle (i,j) for j = 2..5              fb[j] re-read behind its 4 MFMAs
        //   tail   (i,6) (i,7) for i = 0..3        fa[i] re-read behind its 2 MFMAs, fb[6], fb[7] at the end
        // so every A fragment is re-read >= 6 MFMAs (~100 clocks, an LDS latency) before the next sub-step's head needs it and no
        // fragment is double-buffered.  The DMA pieces of the next tile (first sub-step of an iteration) go between the head's MFMAs.
        // sched_barrier(0) after every group pins the order (hipcc otherwise pulls the reads to the front and waits for all of them).
        auto mm = [&](int i, int j) { acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0); };
        auto substep = [&](int nbuf, int nkk, auto DMA, int kt_next) {
            constexpr bool dma = decltype(DMA)::value;
            unsigned char* dbase = smem + (kt_next & 1) * STAGE_BYTES;
            const unsigned char* ka = sbA + ksrc(kt_next) * ROWB;
            const unsigned char* kb = W0 + ksrc(kt_next) * ROWB;
            // (the scalar bases pass through an empty asm: otherwise loop strength reduction turns every piece's address into a 64-bit
            //  lane pointer carried around the loop - the 20 registers this addressing form is here to save)
            auto dmaA = [&](int p) {
                const unsigned char* sp = ka;
                asm volatile("" : "+s"(sp));
                __builtin_amdgcn_global_load_lds((glb_void*)(sp + voA[p]), (lds_void*)(dbase + (wave + WAVES * p) * 1024), 16, 0, 0);
            };
            auto dmaB = [&](int p) {
                const unsigned char* sp = kb + (size_t)p * stepB;
                asm volatile("" : "+s"(sp));
                __builtin_amdgcn_global_load_lds((glb_void*)(sp + voB), (lds_void*)(dbase + BM * ROWB + (wave + WAVES * p) * 1024), 16, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < MT; ++i) {                              // head
                mm(i, 0);
                if constexpr (dma) { if (i < CA) dmaA(i); else dmaB(i - CA); }
                __builtin_amdgcn_sched_barrier(0);
                mm(i, 1);
                if constexpr (dma) dmaB(i + MT - CA);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nbuf >= 0) { fb[0] = rdB(nbuf, nkk, 0); fb[1] = rdB(nbuf, nkk, 1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 2; j < NT - 2; ++j) {                          // middle
                mm(0, j); mm(1, j);
                if constexpr (dma) { if (j - 2 + 2 * MT - CA < CB) dmaB(j - 2 + 2 * MT - CA); }
                __builtin_amdgcn_sched_barrier(0);
                mm(2, j); mm(3, j);
                if (nbuf >= 0) fb[j] = rdB(nbuf, nkk, j);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {                              // tail
                mm(i, NT - 2); mm(i, NT - 1);
                if (nbuf >= 0) fa[i] = rdA(nbuf, nkk, i);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nbuf >= 0) { fb[NT - 2] = rdB(nbuf, nkk, NT - 2); fb[NT - 1] = rdB(nbuf, nkk, NT - 1); }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(2 * MT - CA + (NT - 4) >= CB, "all DMA pieces of a stage find a slot in the first sub-step");
#ifdef RALD_LN_STAMPS
        long long st_wait = 0, st_t0 = clock64();
#endif
        auto hand_over = [&]() {
#ifdef RALD_LN_STAMPS
            const long long w0 = clock64();
#endif
            __builtin_amdgcn_s_waitcnt(W_ALL);                         // the next tile has landed; my reads of this one are done
            __builtin_amdgcn_s_barrier();
#ifdef RALD_LN_STAMPS
            st_wait += clock64() - w0;
#endif
        };
        stage2(0, 0);
        if (nk > 1) { stage2(1, 1); __builtin_amdgcn_s_waitcnt(W_ST1); } else __builtin_amdgcn_s_waitcnt(W_ALL);
        __builtin_amdgcn_s_barrier();
        // tile 0, sub-step 0: its fragments have nothing to hide behind
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = rdB(0, 0, j);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = rdA(0, 0, i);
        substep(0, 1, std::false_type{}, 0);                           // MFMAs (tile 0, kk 0); reads (tile 0, kk 1)
        hand_over();
        // iteration kt >= 1: [DMA tile kt+1] MFMAs (kt-1, kk 1) + reads (kt, kk 0) | MFMAs (kt, kk 0) + reads (kt, kk 1) | hand-over.
        // (x_old is added in the epilogue on this path: riding along in the loop - a uniform switch over the piece index - cost the
        //  register allocator PHI copies of all 128 accumulators and 140 spills)
        auto iter = [&](int kt, auto DMA) {
            substep(kt & 1, 0, DMA, kt + 1);                           // MFMAs (kt-1, kk 1) + DMA of tile kt+1 + reads (kt, kk 0)
            substep(kt & 1, 1, std::false_type{}, 0);                  // MFMAs (kt, kk 0) + reads (kt, kk 1)
            hand_over();
        };
        for (int kt = 1; kt + 1 < nk; ++kt) iter(kt, std::true_type{});
        if (nk > 1) iter(nk - 1, std::false_type{});
        substep(-1, 0, std::false_type{}, 0);                          // MFMAs (last tile, kk 1)
#ifdef RALD_LN_STAMPS
        if (threadIdx.x == 0) { g_ln_stamps[blockIdx.x & 1023][0] = st_wait; g_ln_stamps[blockIdx.x & 1023][1] = clock64() - st_t0; g_ln_stamps[blockIdx.x & 1023][2] = wall_clock64(); }
#endif
    } else {
    stage(0, 0);
    if constexpr (!XLOOP) {                                           // two workgroups per CU: the other one covers this one's epilogue reads
        for (int kt = 0; kt < nk; ++kt) kstep(kt, std::integral_constant<int, 9>{});
    } else if (nk >= 9) {                                             // K >= 576: pieces over the first eight k-steps
        kstep(0, std::integral_constant<int, 0>{}); kstep(1, std::integral_constant<int, 1>{});
        kstep(2, std::integral_constant<int, 2>{}); kstep(3, std::integral_constant<int, 3>{});
        kstep(4, std::integral_constant<int, 4>{}); kstep(5, std::integral_constant<int, 5>{});
        kstep(6, std::integral_constant<int, 6>{}); kstep(7, std::integral_constant<int, 7>{});
        kstep(8, std::integral_constant<int, 8>{});
        for (int kt = 9; kt < nk; ++kt) kstep(kt, std::integral_constant<int, 9>{});
    } else if (nk == 8) {                                             // K = 512: the last piece is added after the loop
        kstep(0, std::integral_constant<int, 0>{}); kstep(1, std::integral_constant<int, 1>{});
        kstep(2, std::integral_constant<int, 2>{}); kstep(3, std::integral_constant<int, 3>{});
        kstep(4, std::integral_constant<int, 4>{}); kstep(5, std::integral_constant<int, 5>{});
        kstep(6, std::integral_constant<int, 6>{}); kstep(7, std::integral_constant<int, 7>{});
        x_add(std::integral_constant<int, 7>{});
    } else {                                                          // short K: all of x after the loop
        for (int kt = 0; kt < nk; ++kt) kstep(kt, std::integral_constant<int, 9>{});
        x_load(std::integral_constant<int, 0>{}); x_add(std::integral_constant<int, 0>{});
        x_load(std::integral_constant<int, 1>{}); x_add(std::integral_constant<int, 1>{});
        x_load(std::integral_constant<int, 2>{}); x_add(std::integral_constant<int, 2>{});
        x_load(std::integral_constant<int, 3>{}); x_add(std::integral_constant<int, 3>{});
        x_load(std::integral_constant<int, 4>{}); x_add(std::integral_constant<int, 4>{});
        x_load(std::integral_constant<int, 5>{}); x_add(std::integral_constant<int, 5>{});
        x_load(std::integral_constant<int, 6>{}); x_add(std::integral_constant<int, 6>{});
        x_load(std::integral_constant<int, 7>{}); x_add(std::integral_constant<int, 7>{});
    }
    }

    }
    constexpr bool PIPE = XLOOP && !MX && BK == 64 && MT == 4 && NT == 8 && CA + CB == 10;     // the pipelined loop adds x_old in the epilogue
    resid_ln_epilogue<BM, WM, WN, (MX || !XLOOP || PIPE)>(a, acc, smem, NSTAGE * STAGE_BYTES, m0);
#ifdef RALD_LN_STAMPS
    if (threadIdx.x == 0) g_ln_stamps[blockIdx.x & 1023][3] = wall_clock64();
#endif
}

#ifdef RALD_PROBE
// ---- MEASURED DEAD END, probe builds only (tools/ab_ln_ring.py): 128-row form with two operand rings (bf16) ---------------------------
// Hypothesis: with one 80-KiB stage of A and W in flight (the kernel above) every k-step of a workgroup waits one HBM latency for the A
// panel (2.4 us per k-step measured for 0.9 us of MFMA work at the datasheet rate).  Result at B = 64: 51 vs 44.5 us (K = 512), 100 vs 88 us
// (K = 2048), NFE 12.10 vs 11.91 ms, same numbers out (x 1e-7, h 2e-5): with three A stages ahead the k-step takes just as long, so the
// A latency is not what the loop waits for - it runs at the pace of its LDS traffic (276 KB per k-step and CU) plus the MFMAs at the clock
// the chip sustains in such loops - and twice the barriers plus the up-front x_old read cost 6-12 us.
// The two operands have their own rings in the same 160 KiB:
//   A: 4 stages of 128 rows x 64 k (16 KiB each) - three stages (2.6 us of work) ahead of the MFMAs;
//   W: 3 stages of 512 rows x 32 k (32 KiB each) - two 32-deep steps ahead (L2 latency);
// one barrier per 32-deep step, counted vmcnt (the only vector-memory instructions inside the loop are the DMA pieces, issued in a fixed
// order, so "all but the pieces of the previous step" is a compile-time count).  The residual x_old is loaded STRAIGHT INTO the accumulators
// before the loop (its latency overlaps the first stages' DMA; no ordinary load sits beside the DMA stream inside the loop, which would make
// hipcc wait vmcnt(0) there).
template <int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_resid_ln_ring_kernel(GemmLnArgs a) {
    constexpr int BM = 128, BN = 512, WAVES = WM * WN;
    static_assert(WAVES == 8, "8 waves");
    constexpr int MT = BM / (16 * WM), NT = BN / (16 * WN);
    constexpr int NA = 4, A_STAGE = BM * 128, NWS = 3, W_STAGE = BN * 64;
    constexpr int W_OFF = NA * A_STAGE;
    static_assert(W_OFF + NWS * W_STAGE == 160 * 1024, "the two rings fill the CU's LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int fr = lane & 15, fq = lane >> 4;
    // DMA sources.  A: pieces of 8 rows x 128 B (2 per wave and stage); W: pieces of 16 rows x 64 B (4 per wave and stage)
    const unsigned char* gA[2];
    {
        const int lr = lane >> 3, lc = (lane & 7) ^ lr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int r = m0 + 8 * (wave + WAVES * p) + lr;
            r = r < a.M ? r : a.M - 1;
            gA[p] = reinterpret_cast<const unsigned char*>(a.A) + (int64_t)r * a.lda * 2 + lc * 16;
        }
    }
    const unsigned char* gW0;
    const int64_t w_piece = (int64_t)16 * WAVES * a.ldw * 2;               // bytes between this wave's pieces
    {
        const int lr = lane >> 2, lc = (lane & 3) ^ ((lr >> 2) & 3);
        gW0 = reinterpret_cast<const unsigned char*>(a.W) + (int64_t)(16 * wave + lr) * a.ldw * 2 + lc * 16;
    }
    auto stage_a = [&](int ka) {
        unsigned char* base = smem + (ka & (NA - 1)) * A_STAGE;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((glb_void*)(gA[p] + ka * 128), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
    };
    auto stage_w = [&](int t) {
        unsigned char* base = smem + W_OFF + (t % NWS) * W_STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((glb_void*)(gW0 + p * w_piece + t * 64), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
    };
    const int nk64 = a.K / 64, nk32 = a.K / 32;
    stage_a(0);
    if (nk64 > 1) stage_a(1);
    if (nk64 > 2) stage_a(2);
    stage_w(0);
    stage_w(1);                                                         // nk32 >= 2
    // accumulators start as x_old
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[MT][NT];
    {
        const int mb = m0 + wm * (BM / WM), nb = wn * (BN / WN);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = mb + i * 16 + fr;
            m = m < a.M ? m : a.M - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.x + (int64_t)m * BN + nb + j * 16 + 4 * fq));
                acc[i][j] = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
    auto compute = [&](int t) {
        const unsigned char* sA = smem + ((t >> 1) & (NA - 1)) * A_STAGE;
        const unsigned char* sW = smem + W_OFF + (t % NWS) * W_STAGE;
        bf16x8 fa[MT], fb[NT];
        const int chunk = (t & 1) * 4 + fq;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * (BM / WM) + i * 16 + fr;
            fa[i] = *reinterpret_cast<const bf16x8*>(sA + r * 128 + ((chunk ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = wn * (BN / WN) + j * 16 + fr;
            fb[j] = *reinterpret_cast<const bf16x8*>(sW + r * 64 + ((fq ^ ((r >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    auto issue = [&](int t) -> int {                                     // the loads issued during step t; returns their instruction count
        int n = 0;
        if (t + 2 < nk32) { stage_w(t + 2); n += 4; }
        if ((t & 1) == 0 && (t >> 1) + 3 < nk64) { stage_a((t >> 1) + 3); n += 2; }
        return n;
    };
    // step 0: everything issued so far (and x_old) has to be there
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(0);                                                         // (hipcc waits vmcnt(0) for x_old before the first MFMA: nothing may be in flight yet)
    int pend = issue(0);
    for (int t = 1; t < nk32; ++t) {
        // all but the pieces issued during step t-1 have landed: stage t of W (issued at step t-2) and its A stage (earlier still)
        if (pend == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (pend == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // ... for every wave; the slots overwritten below were read in step t-1
        asm volatile("" ::: "memory");
        pend = issue(t);
        compute(t);
    }
    asm volatile("" ::: "memory");
    resid_ln_epilogue<BM, WM, WN, false>(a, acc, smem, 160 * 1024, m0);
}

static int launch_ln_ring(const GemmLnArgs& a, hipStream_t st) {
    constexpr int smem = 160 * 1024;
    static bool attr_set = false;
    auto kern = gemm_resid_ln_ring_kernel<2, 4>;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(a.M, 128)), dim3(512), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}
#endif

template <int BM, int WM, int WN, bool MX>
__global__ __launch_bounds__(WM * WN * 64) void gemm_resid_ln_kernel(GemmLnArgs a) { gemm_resid_ln_body<BM, WM, WN, MX, 64>(a); }
#ifdef RALD_PROBE
// MEASURED DEAD END, probe builds only (tools/ab_ln_pair.py): 64-row tiles, 32-deep k-steps, two workgroups per CU (4 waves per SIMD, 128
// registers) so that one workgroup's epilogue streams while the other runs its k-loop.  Same results (x 1e-7, h 2e-5), but 67 vs 46 us at
// K = 512 and 140 vs 88 us at K = 2048 (B = 64), NFE 13.87 vs 12.41 ms: a 64-row tile re-reads W once per 64 rows - 1 GB of L2 -> LDS
// traffic per launch at K = 2048 (7.6 TB/s: the L2 is the bound) - and takes twice the barriers per FLOP.  The 128-row tile stays.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_resid_ln_pair_kernel(GemmLnArgs a) {
    gemm_resid_ln_body<64, 1, 8, false, 32, false>(a);
}

static int launch_ln_pair(const GemmLnArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (64 + 512) * 32 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)gemm_resid_ln_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_resid_ln_pair_kernel, dim3(cdiv(a.M, 64)), dim3(512), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}
#endif

template <int BM, int WM, int WN, bool MX>
static int launch_ln(const GemmLnArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BM + 512) * 64 * 2;
    static bool attr_set = false;
    auto kern = gemm_resid_ln_kernel<BM, WM, WN, MX>;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(a.M, BM)), dim3(WM * WN * 64), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

int gemm_resid_ln(const GemmLnArgs& a0, hipStream_t st) {
    static const int nt_env = RALD_PROBE_ENV("RALD_NT_STORE", 1);
    GemmLnArgs a = a0;
    a.nt_io = nt_env;
    const bool mx = a.A8 != nullptr;                         // MXFP8 operands: A8/SA and W8/SW instead of A and W
    RALD_CHECK(a.M > 0 && a.K > 0 && a.K % (mx ? 128 : 64) == 0, "gemm_resid_ln: bad shape");
    RALD_CHECK(a.lda % 16 == 0 && a.ldw % 16 == 0 && a.lda >= a.K && a.ldw >= a.K, "gemm_resid_ln: leading dimensions");
    RALD_CHECK((mx ? (a.SA && a.W8 && a.SW) : (a.A && a.W)) && a.bias && a.x && (a.h || (a.h8 && a.hs)) && a.g && a.b && a.rows_per_group > 0,
               "gemm_resid_ln: null argument");
    RALD_CHECK(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.A8 % 16 == 0) && ((uintptr_t)a.W8 % 16 == 0) &&
               ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.h % 16 == 0) && ((uintptr_t)a.g % 16 == 0) && ((uintptr_t)a.b % 16 == 0) &&
               a.gstride % 4 == 0, "gemm_resid_ln: 16-byte alignment");
    RALD_CHECK(!mx || (int64_t)a.M * (a.K / 32) < ((int64_t)1 << 31), "gemm_resid_ln: scale index overflow");
    RALD_CHECK(a.strideW == 0 || (!mx && a.w_rows % 128 == 0 && a.strideW % 8 == 0), "gemm_resid_ln: per-group weights need bf16 operands and groups of whole 128-row tiles");
    // 128-row tiles (all 160 KiB of LDS) when they cover the chip, 64-row tiles for smaller M
#ifdef RALD_PROBE
    if (!mx && RALD_PROBE_ENV("RALD_LN_PAIR", 0) && cdiv(a.M, 64) >= 384) return launch_ln_pair(a, st);
#endif
#ifdef RALD_PROBE
    if (!mx && cdiv(a.M, 128) >= 192 && a.K >= 64 && RALD_PROBE_ENV("RALD_LN_RING", 0)) return launch_ln_ring(a, st);
#endif
    if (cdiv(a.M, 128) >= 192) return mx ? launch_ln<128, 2, 4, true>(a, st) : launch_ln<128, 2, 4, false>(a, st);
    return mx ? launch_ln<64, 1, 8, true>(a, st) : launch_ln<64, 1, 8, false>(a, st);
}

}  // namespace rald
