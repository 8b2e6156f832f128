// NT bf16 GEMM on MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate, fused epilogues.
//
//   C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]      A: activations, B: weights ([out,in] as
//   torch.nn.Linear stores them) - both K-contiguous, which is exactly the MFMA fragment
//   order, so neither operand is ever transposed in memory.
//
// Two staging engines share one tile/epilogue design (BM x BN x 64, waves in a 2 x (WAVES/2) grid,
// 16x16 MFMA tiles, LDS rows of 128 B whose 16-byte chunks are XOR-swizzled by (row & 7) so every
// ds_read_b128 lane group is bank-conflict free):
//   * gemm_nt_kernel      register staging: global_load_dwordx4 -> VGPR -> ds_write_b128.  At 2
//                         workgroups/CU the ds_write path (~79 B/clk/CU) makes this LDS-bound.
//   * gemm_nt_glds_kernel LDS-DMA staging: global_load_lds_dwordx4 writes the tile straight into
//                         LDS (lane-linear destination; the swizzle is applied to each lane's
//                         SOURCE address and again on the read - CDNA guide rule 21), freeing the
//                         ds_write bandwidth and 32 staging VGPRs.
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
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace rald {

// ---- epilogue: lane owns row m = mb + 16i + fr, 4 consecutive columns n = nb + 16j + 4*fq + {0..3}
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MT][NT], const GemmArgs& a, int mb, int nb, int64_t coff, int fr, int fq) {
    if constexpr (EPI == EPI_GEGLU) {
        bf16* C = reinterpret_cast<bf16*>(a.C) + coff;
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
                const float al = n < a.alpha_ncols ? a.alpha : 1.0f;
                if constexpr (EPI == EPI_BF16) {
                    bf16* C = reinterpret_cast<bf16*>(a.C) + coff;
                    *reinterpret_cast<bf16x4*>(C + (int64_t)m * a.ldc + n) =
                        pack4(al * v[0] + b.x, al * v[1] + b.y, al * v[2] + b.z, al * v[3] + b.w);
                } else if constexpr (EPI == EPI_F32) {
                    float* C = reinterpret_cast<float*>(a.C) + coff;
                    *reinterpret_cast<float4*>(C + (int64_t)m * a.ldc + n) =
                        make_float4(al * v[0] + b.x, al * v[1] + b.y, al * v[2] + b.z, al * v[3] + b.w);
                } else {  // EPI_RESID
                    float* C = reinterpret_cast<float*>(a.C) + coff;
                    float4* p = reinterpret_cast<float4*>(C + (int64_t)m * a.ldc + n);
                    float4 r = *p;
                    *p = make_float4(r.x + v[0] + b.x, r.y + v[1] + b.y, r.z + v[2] + b.z, r.w + v[3] + b.w);
                }
            }
        }
    }
}


// =================================================================================================
// register-staged engine (4 waves, 2x2)
// =================================================================================================
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
    int64_t oa, ob, coff;
    gemm_batch_offsets(a, blockIdx.z, oa, ob, coff);
    const bf16* A = a.A + oa;
    const bf16* B = a.B + ob;

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
    gemm_epilogue<MT, NT, EPI>(acc, a, m0 + wm * (BM / 2), n0 + wn * (BN / 2), coff, fr, fq);
}

// =================================================================================================
// LDS-DMA engine: WM x WN waves; NSTAGE LDS buffers
// =================================================================================================
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#ifdef RALD_GEMM_CLOCK   // tools/probe/gemm_clock.hip: shader clocks vs the 100 MHz wall clock over one workgroup's life
__device__ long long g_gemm_clk[2];
#endif
#ifdef RALD_GEMM_STAMPS  // tools/probe/gemm_timeline.hip: per-workgroup wall-clock stamps (100 MHz) + hardware ids, one record per launch-order block
__device__ long long g_gemm_stamps[8192][8];
#define RALD_GSTAMP(i) do { if (threadIdx.x == 0) g_gemm_stamps[lin & 8191][i] = wall_clock64(); } while (0)
#else
#define RALD_GSTAMP(i) do { } while (0)
#endif
template <int BM, int BN, int WM, int WN, int NSTAGE, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_glds_kernel(GemmArgs a) {
#ifdef RALD_GEMM_CLOCK
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    constexpr int BK = 64;
    constexpr int WAVES = WM * WN;
    constexpr int MT = BM / (16 * WM);       // m-tiles per wave
    constexpr int NT = BN / (16 * WN);       // n-tiles per wave
    constexpr int CA = BM / 8 / WAVES;       // 1-KiB DMA pieces (8 rows x 128 B) per wave for A
    constexpr int CB = BN / 8 / WAVES;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSTAGE][A tile | B tile]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so the
    // blocks with equal id%8 share an L2.  Give each of those 8 groups a CONTIGUOUS span of the tile
    // walk (bijective for any grid size, guide 5 q/r form), and walk the tiles in column STRIPS of
    // GN n-tiles (n fastest inside a strip, then m, then the next strip): the 32 tiles an XCD runs at
    // once then share <= GN weight panels (2 MB at GN = 8, 256-row tiles, K = 512) that stay in its
    // 4 MB L2 while the A panels stream through.  Row-major order re-fetched the whole 4 MB FF1
    // weight matrix for every round of tiles (measured: 296 MB fetched vs 37.5 MB algorithmic).
    const int ntn = gridDim.x, ntm = gridDim.y, nt = ntn * ntm;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, q = nt >> 3, rr = nt & 7;
    const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lin >> 3);
    RALD_GSTAMP(0);
#ifdef RALD_GEMM_STAMPS
    if (threadIdx.x == 0) { g_gemm_stamps[lin & 8191][6] = __builtin_amdgcn_s_getreg(63492); g_gemm_stamps[lin & 8191][7] = __builtin_amdgcn_s_getreg(63508); }
#endif
#ifdef RALD_GN16               // A/B builds (tools/build_variant.sh): strips of 16 n-tiles (A panels fetched once at N = 4096)
    const int GN = 16;
#else
    const int GN = RALD_ABLATED(a.ablate, 128) ? 16 : 8;       // probe builds: bit 128 = strips of 16 n-tiles
#endif
    int tm, tn;
    if (ntn % GN == 0) {
        const int strip = tile / (ntm * GN), within = tile % (ntm * GN);
        tm = within / GN;
        tn = strip * GN + within % GN;
    } else {
        tm = tile / ntn;
        tn = tile % ntn;
    }
    int bz = blockIdx.z;
    if (nt < 8 && (gridDim.z & 7) == 0) {
        // batched problems of a few tiles each (the folded cross-attention: 2 x 2 tiles per sample, per-sample B operand): in launch order
        // a batch entry's tiles land on different XCDs and each re-reads the entry's operands from HBM.  Deal whole batch entries to the
        // XCDs instead: launch index g -> XCD g & 7, slot g >> 3 -> entry (slot / nt) * 8 + XCD, tile slot % nt.
        const int g = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int slot = g >> 3;
        bz = (slot / nt) * 8 + (g & 7);
        const int tl = slot % nt;
        tm = tl / ntn;
        tn = tl % ntn;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // k-steps are walked from a per-tile offset, wrapping around (2-stage engine): workgroups launched together otherwise ask the L2 for
    // the same k-slice of a shared operand panel at the same moment, at every k-step (gemm_ln.hip has the measurement).  Offset =
    // (m-tile + n-tile + batch entry) mod nk: the tiles that share an A panel (same m) or a B panel (same n) start on different slices,
    // and a tile's offset does not depend on how many other tiles the launch has (sub-batches reproduce the whole batch bit for bit).
    const int nk_ = a.K / BK;
#ifndef RALD_KOFF            // off in the shipped build: see below (A/B builds: tools/build_variant.sh koff -DRALD_KOFF)
    const int koff = 0;
#else
    const int koff = (NSTAGE == 2 && !RALD_ABLATED(a.ablate, 2048)) ? (int)((unsigned)(tm + tn + bz) % (unsigned)nk_) : 0;
#endif
    int64_t oa, ob, coff;
    gemm_batch_offsets(a, bz, oa, ob, coff);
    const bf16* A = a.A + oa;
    const bf16* B = a.B + ob;

    // DMA source addresses.  Piece p of this wave covers tile rows 8*(wave + WAVES*p) .. +7; lane l
    // lands in LDS at piece_base + 16*l = (row r = l>>3, physical chunk l&7), which must hold the
    // LOGICAL chunk (l&7) ^ (r&7): the XOR goes on the source address, the destination stays linear.
    const int lr = lane >> 3;
    const int lc = (lane & 7) ^ lr;
    const bf16* gA[CA];
    const bf16* gB[CB];
#pragma unroll
    for (int p = 0; p < CA; ++p) {
        int r = m0 + 8 * (wave + WAVES * p) + lr;
        r = r < a.M ? r : a.M - 1;
        gA[p] = A + (int64_t)r * a.lda + lc * 8;
    }
#pragma unroll
    for (int p = 0; p < CB; ++p) {
        int r = n0 + 8 * (wave + WAVES * p) + lr;
        r = r < a.N ? r : a.N - 1;
        gB[p] = B + (int64_t)r * a.ldb + lc * 8;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
        int ks = kt + koff;
        ks = ks >= nk_ ? ks - nk_ : ks;
        // probe builds (timing only, wrong results): bits 4096 / 8192 drop the B / A pieces of every stage after the first two - is a k-step
        // paced by the BYTES of its stage or by the latency of a stage, whatever its size?
        // (the probe library's main loop is ~2 x slower than the shipped one - its run-time switches sit between the MFMAs - so the same question
        // is asked of the shipped loop with compile-time variants: tools/build_variant.sh ska -DRALD_SKIP_A, skb -DRALD_SKIP_B, skab with both)
#if defined(RALD_SKIP_A)
        const bool skip_a = kt >= 2;
#else
        const bool skip_a = RALD_ABLATED(a.ablate, 8192) && kt >= 2;
#endif
#if defined(RALD_SKIP_B)
        const bool skip_b = kt >= 2;
#else
        const bool skip_b = RALD_ABLATED(a.ablate, 4096) && kt >= 2;
#endif
#pragma unroll
        for (int p = 0; p < CA; ++p)
            if (!skip_a) __builtin_amdgcn_global_load_lds((glb_void*)(gA[p] + ks * BK), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < CB; ++p)
            if (!skip_b) __builtin_amdgcn_global_load_lds((glb_void*)(gB[p] + ks * BK), (lds_void*)(base + BM * 128 + (wave + WAVES * p) * 1024), 16, 0, 0);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment reads of one 32-deep k sub-step (kk) of the tile in LDS buffer `buf`
    auto read_frags = [&](int buf, int kk, bf16x8 (&fa)[MT], bf16x8 (&fb)[NT]) {
        const bf16x8* sA = reinterpret_cast<const bf16x8*>(smem + buf * STAGE_BYTES);
        const bf16x8* sB = sA + BM * 8;
        const int chunk = kk * 4 + fq;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * (BM / WM) + i * 16 + fr;
            fa[i] = sA[r * 8 + (chunk ^ (r & 7))];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = wn * (BN / WN) + j * 16 + fr;
            fb[j] = sB[r * 8 + (chunk ^ (r & 7))];
        }
    };
    auto mfma_all = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fb)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    // Measured and removed (round 3, tools/ab_libs_gemm.py on compile-time variants of this loop): (a) dropping the A pieces, the B pieces or all
    // DMA after the prologue (-DRALD_SKIP_A / -DRALD_SKIP_B, wrong results, timing only) takes FF1 at B = 64 on random operands 167 -> 161 / 161
    // / 151 us and its K = 2048 form 442 -> 406 / 400 / 367 us: with NO global traffic a k-step still takes 1.25 us against 1.54 with it and
    // 1.09 of pure MFMA time at the ~1.9 GHz the chip holds in this loop - the loop is matrix-pipe-bound at the power-limited clock, the whole DMA
    // path costs 10-17 % (about what issuing 8-10 LDS-DMA instructions per wave and k-step costs), which is why no prefetch / ring / persistent
    // variant ever gained; (b) the same loop on v_mfma_f32_32x32x16_bf16 (half the MFMA instructions and operand-register reads per FLOP, same
    // LDS traffic): 5-8 % SLOWER (173 -> 184 us, 446 -> 480 us).
    if constexpr (NSTAGE == 2) {
        // Software-pipelined main loop, rotated so that an iteration starts right AFTER a tile hand-over (barrier): nothing is
        // pending on the LDS counter at the loop head, so the compiler's waits inside the iteration are exact counts (round 2: the
        // loop head carried pending fragment reads and hipcc answered with lgkmcnt(0) in front of every first MFMA burst, i.e. the
        // 12 reads just issued were waited for before any MFMA could go - a third of a k-step with the matrix pipe idle).
        // The fragment registers are double-buffered: F0 = 32-deep sub-step 0 of a tile, F1 = sub-step 1.  An iteration:
        //   DMA of tile kt+1 into the buffer that tile kt-1 has just left | reads F0(kt) under the MFMAs of F1(kt-1) |
        //   reads F1(kt) under the MFMAs of F0(kt) | wait (tile kt+1 landed, my reads of tile kt done) + barrier.
        // The DMA issues and the LDS reads are spread between the MFMAs (sched_group_barrier) instead of in front of them.
        bf16x8 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        constexpr int W_ALL = 0x0070;                                                  // vmcnt(0) lgkmcnt(0), expcnt untouched
        constexpr int W_ST1 = ((CA + CB) & 15) | (((CA + CB) >> 4) << 14) | 0x0f70;    // vmcnt(CA+CB): the older stage has landed
        stage(0, 0);
        if (nk > 1 && !RALD_ABLATED(a.ablate, 1)) { stage(1, 1); __builtin_amdgcn_s_waitcnt(W_ST1); }
        else __builtin_amdgcn_s_waitcnt(W_ALL);
        __builtin_amdgcn_s_barrier();
        RALD_GSTAMP(1);
        read_frags(0, 0, fa0, fb0);
        read_frags(0, 1, fa1, fb1);
        mfma_all(fa0, fb0);
        __builtin_amdgcn_s_waitcnt(W_ALL);
        __builtin_amdgcn_s_barrier();
        auto body = [&](int kt, auto with_dma) {
            const int cur = kt & 1;
            if constexpr (decltype(with_dma)::value) stage(kt + 1, cur ^ 1);
            read_frags(cur, 0, fa0, fb0);
            mfma_all(fa1, fb1);                                  // sub-step 1 of tile kt-1
            if constexpr (MT + NT == 12 && MT * NT == 32) {
                if constexpr (decltype(with_dma)::value) {
#pragma unroll
                    for (int g = 0; g < CA + CB; ++g) {          // 1 MFMA, then one DMA piece (8 x): the DMA goes out first, its latency is the long one
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 12; ++g) {               // 2 MFMAs, then one fragment read (12 x)
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 12; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(cur, 1, fa1, fb1);
            mfma_all(fa0, fb0);                                  // sub-step 0 of tile kt
            if constexpr (MT + NT == 12 && MT * NT == 32) {
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(W_ALL);                   // tile kt+1 landed; my reads of tile kt are done
            __builtin_amdgcn_s_barrier();
        };
        if (RALD_ABLATED(a.ablate, 1)) {
            for (int kt = 1; kt < nk; ++kt) body(kt, std::false_type{});
        } else {
            for (int kt = 1; kt + 1 < nk; ++kt) body(kt, std::true_type{});
            if (nk > 1) body(nk - 1, std::false_type{});
        }
        mfma_all(fa1, fb1);                                      // sub-step 1 of the last tile
    } else {
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nk) stage(s, s);
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt must have landed: allow the NSTAGE-2 younger tiles to stay in flight
            if (kt + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (CA + CB)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone's pieces of tile kt are in LDS; buffer (kt-1)%NSTAGE is free
            asm volatile("" ::: "memory");
            if (kt + NSTAGE - 1 < nk && !RALD_ABLATED(a.ablate, 1)) stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[MT], fb[NT];
                read_frags(kt % NSTAGE, kk, fa, fb);
                mfma_all(fa, fb);
            }
        }
    }
    if (RALD_ABLATED(a.ablate, 2)) {          // probe builds: keep the accumulators live, store (almost) nothing
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 1234.5678f) reinterpret_cast<float*>(a.C)[0] = s;
        return;
    }
    if constexpr (NSTAGE != 2) {              // (the 2-stage loop ends on a barrier behind its last fragment reads)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();         // every wave is done reading the staging buffers: reuse them as patches
        asm volatile("" ::: "memory");
    }
    RALD_GSTAMP(2);
    gemm_epilogue_lds<MT, NT, EPI>(acc, a, m0 + wm * (BM / WM), n0 + wn * (BN / WN), coff, lane, smem + wave * 8704);
    RALD_GSTAMP(3);
#ifdef RALD_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RALD_GSTAMP(4);
#endif
#ifdef RALD_GEMM_CLOCK
    if (tid == 0 && lin == nt / 2) { g_gemm_clk[0] = clock64() - clk0; g_gemm_clk[1] = wall_clock64() - wall0; }
#endif
}

#ifdef RALD_PROBE
// =================================================================================================
// Persistent form of the 256x256 engine for the GEGLU projection (FF1: the dominant kernel of an NFE).  PROBE builds only
// (RALD_GEMM_PERSIST=1): measured on MI355X against the plain launch of the same tiles, interleaved in one process
// (tools/ab_persist.py): FF1 alone 155.1 vs 151.6 us, whole NFE at B = 64 12.75 vs 12.60 ms, with the stagger 12.86 ms - the
// hardware dispatcher already starts the next workgroup of a CU while the previous one drains its stores, and the tile loop
// costs the double-buffered fragment registers of the plain kernel's main loop.  Kept as a measured dead end.
// Round 3 rebuilt it on the rotated main loop WITH the double-buffered fragments (scalar base + 32-bit lane offsets for the DMA sources,
// one code path for the last tile, bias through LDS-DMA; clean k-loop, 52 spilled registers at the tile boundaries): correct, and 7 % SLOWER
// per NFE (FF1 143 -> ~176 us at B = 64).  The reason is structural: vmcnt is per wave and in order, so the next tile's second hand-over
// (its stage 2 was issued behind the epilogue's stores) waits for this wave's own stores - and those are part of a chip-wide burst
// (16.8 MB per round of tiles, all CUs at once) that takes ~4 us to drain.  In the plain launch the next workgroup's waves start with
// empty counters and never wait for the previous workgroup's stores.  A persistent GEMM with an HBM-write-through epilogue pays its own
// store latency; the plain launch does not.
// =================================================================================================
// One workgroup per CU walks its tiles (virtual block id v = blockIdx.x + j * gridDim.x through the same XCD-aware strip order
// as above: v % 8 == blockIdx.x % 8, so a workgroup's tiles stay on its XCD's L2).  What the plain launch cannot do:
//   * the k-loop runs on ACROSS tile boundaries: the first two k-steps of the next tile are issued (LDS-DMA) during the last
//     two k-steps of the current one and land under its epilogue, so a tile starts with its operands in LDS instead of one
//     exposed HBM/L2 latency + a workgroup launch;
//   * the epilogue's transpose patches live in their own 18 KiB of LDS (GEGLU rows are 128 B: 16 x 144 B per wave), so the
//     two staging buffers stay untouched while the epilogue runs;
//   * its 8 output stores per wave stay in flight behind counted waits (vmcnt counts stores too): the next tile's first two
//     hand-overs wait for "all but the youngest 16 / 8" operations, i.e. for their DMA pieces only;
//   * STAGGER: the workgroups with an odd slot on their XCD start half a tile late, so that from then on half of the CUs
//     are in their HBM-heavy epilogue while the other half are in the MFMA loop (in a plain launch all 256 CUs run the same
//     phase at the same time; two independent streams gained 5-7 % from the same effect).
template <bool STAGGER>
__global__ __launch_bounds__(512) void gemm_geglu_persist_kernel(GemmArgs a, int ntn, int ntm) {
    constexpr int BM = 256, BN = 256, BK = 64, WM = 4, WN = 2, WAVES = 8;
    constexpr int MT = BM / (16 * WM), NT = BN / (16 * WN);
    constexpr int CA = BM / 8 / WAVES, CB = BN / 8 / WAVES;          // 4 + 4 DMA pieces per wave and stage
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;                  // 64 KiB
    constexpr int PATCH = 16 * (NT * 8 * 2 + 16);                    // 2304 B: one 16-row m-tile of GEGLU output per wave
    constexpr int NSTORE = MT * 2;                                   // output store instructions per wave and tile (8 rows each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][A tile | B tile] | 8 patches | [2 tiles][8 waves] 512 B of bias
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nt = ntn * ntm;
    const int nk = a.K / BK;
    const int lr = lane >> 3;
    const int lc = (lane & 7) ^ lr;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int GN = 8;
    auto tile_origin = [&](int v, int& m0, int& n0) {
        const int xcd = v & 7, q = nt >> 3, rr = nt & 7;
        const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (v >> 3);
        int tm, tn;
        if (ntn % GN == 0) {
            const int strip = tile / (ntm * GN), within = tile % (ntm * GN);
            tm = within / GN;
            tn = strip * GN + within % GN;
        } else {
            tm = tile / ntn;
            tn = tile % ntn;
        }
        m0 = tm * BM; n0 = tn * BN;
    };
    // one DMA stage: k-step kt of the tile at (m0, n0) into buffer buf.  Full tiles only (host contract: M, N multiples of 256), so
    // the per-lane part of every source address is a 32-bit element offset fixed for the whole launch (8 VGPRs) and the tile /
    // k-step part is scalar - the register file has no room for eight 64-bit pointers next to 128 accumulators and 96 fragment
    // registers (a first version spilled 99 VGPRs to scratch and ran at half the speed of the plain launch).
    unsigned offA[CA], offB[CB];
#pragma unroll
    for (int p = 0; p < CA; ++p) offA[p] = (unsigned)((8 * (wave + WAVES * p) + lr) * (int)a.lda + lc * 8);
#pragma unroll
    for (int p = 0; p < CB; ++p) offB[p] = (unsigned)((8 * (wave + WAVES * p) + lr) * (int)a.ldb + lc * 8);
    auto stage = [&](int m0, int n0, int kt, int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
        const bf16* sa = a.A + (int64_t)m0 * a.lda + kt * BK;       // scalar
        const bf16* sb = a.B + (int64_t)n0 * a.ldb + kt * BK;
#pragma unroll
        for (int p = 0; p < CA; ++p)
            __builtin_amdgcn_global_load_lds((glb_void*)(sa + offA[p]), (lds_void*)(base + (wave + WAVES * p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < CB; ++p)
            __builtin_amdgcn_global_load_lds((glb_void*)(sb + offB[p]), (lds_void*)(base + BM * 128 + (wave + WAVES * p) * 1024), 16, 0, 0);
    };
    // the wave's 128 bias values of the tile at n0 -> its LDS slot of parity `par` (two 256-byte DMA pieces, 4 bytes per lane)
    float* const s_bias = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES + WAVES * PATCH);
    auto stage_bias = [&](int n0, int par) {
        const float* src = a.bias + n0 + wn * (BN / WN) + lane;
        float* dst = s_bias + (par * WAVES + wave) * 128;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(src + 64), (lds_void*)(dst + 64), 4, 0, 0);
    };
    int v = blockIdx.x;
    if (v >= nt) return;
    if (STAGGER && ((blockIdx.x >> 3) & 1)) {
        // about half a tile of head start for the even slots: a tile is ~5 000 clocks per k-step, s_sleep 127 = 8 128 clocks
        for (int i = 0; i < (5 * nk + 8) / 16; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int m0, n0;
    tile_origin(v, m0, n0);
    stage_bias(n0, 0);
    stage(m0, n0, 0, 0);
    if (nk > 1) stage(m0, n0, 1, 1);
    bool first = true;
    int par = 0;
    for (;;) {
        const int vn = v + gridDim.x;
        int m1 = 0, n1 = 0;
        const bool more = vn < nt;
        if (more) tile_origin(vn, m1, n1);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k-step 0 of this tile (and its bias) must have landed.  Outstanding, oldest first: [bias][stage 0] [stage 1] [the previous
        // tile's NSTORE stores]
        if (first) { if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA + CB + NSTORE) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            // one 32-deep sub-step: the 4 A fragments at once, the B fragments one n-tile ahead of their 4 MFMAs (24 fragment
            // registers instead of the 96 of the plain kernel's double-buffered sets: the tile loop needs the difference)
            auto substep = [&](int kk) {
                const bf16x8* sA = reinterpret_cast<const bf16x8*>(smem + cur * STAGE_BYTES);
                const bf16x8* sB = sA + BM * 8;
                const int chunk = kk * 4 + fq;
                bf16x8 fa[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int r = wm * (BM / WM) + i * 16 + fr;
                    fa[i] = sA[r * 8 + (chunk ^ (r & 7))];
                }
                const int rb0 = wn * (BN / WN) + fr;
                bf16x8 fb = sB[rb0 * 8 + (chunk ^ (rb0 & 7))];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bf16x8 fbn = fb;
                    if (j + 1 < NT) {
                        const int r = rb0 + (j + 1) * 16;
                        fbn = sB[r * 8 + (chunk ^ (r & 7))];
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[i], acc[i][j], 0, 0, 0);
                    fb = fbn;
                }
            };
            substep(0);
            substep(1);
            if (kt + 1 < nk) {
                // k-step kt+1 landed; my reads of buffer `cur` are done.  After the first hand-over of a tile the stores of the
                // previous tile may still be in flight BEHIND the stage waited for (kt == 0: [stage 1][stores]); later the
                // only younger operations are DMA pieces issued after them, so everything is waited for.
                if (kt == 0 && !first) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 2 < nk) stage(m0, n0, kt + 2, cur);
                else if (more) { stage_bias(n1, par ^ 1); stage(m1, n1, kt + 2 - nk, cur); }     // the next tile's bias and k-step 0 (nk even: buffer 0)
            }
        }
        // every wave is done reading the last buffer (nk - 1) & 1 = 1: the next tile's k-step 1 goes there
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more && nk > 1) stage(m1, n1, 1, 1);
        gemm_epilogue_lds<MT, NT, EPI_GEGLU>(acc, a, m0 + wm * (BM / WM), n0 + wn * (BN / WN), 0, lane, smem + 2 * STAGE_BYTES + wave * PATCH,
                                             s_bias + (par * WAVES + wave) * 128);
        if (!more) break;
        v = vn; m0 = m1; n0 = n1;
        first = false;
        par ^= 1;
    }
}

#endif  // RALD_PROBE

// -------------------------------------------------------------------------------------------------
template <int BM, int BN>
static int launch_tile(const GemmArgs& a, int epi, hipStream_t st) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.batch * a.batch2);
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

template <int BM, int BN, int WM, int WN, int NSTAGE, int EPI>
static int launch_glds_epi(const GemmArgs& a, hipStream_t st) {
    constexpr int WAVES = WM * WN;
    constexpr int smem = NSTAGE * (BM + BN) * 64 * 2;
    static_assert(smem >= WAVES * 8704, "epilogue patches must fit in the staging buffers");
    static bool attr_set = false;
    auto kern = gemm_nt_glds_kernel<BM, BN, WM, WN, NSTAGE, EPI>;
    if (!attr_set && smem > 64 * 1024) {
        RALD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.batch * a.batch2);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}
template <int BM, int BN, int WM, int WN, int NSTAGE>
static int launch_glds(const GemmArgs& a, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_BF16:  return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_BF16>(a, st);
        case EPI_F32:   return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_F32>(a, st);
        case EPI_RESID: return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_RESID>(a, st);
        case EPI_GEGLU: return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_GEGLU>(a, st);
        case EPI_F16S:
            if constexpr ((BM == 64 && BN == 64) || (BM == 128 && BN == 128 && NSTAGE == 2)) return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_F16S>(a, st);
            else { set_error("gemm: the fp16-slab epilogue is built for the 64x64 ring and the 128x128 LDS-DMA engines"); return 1; }
        case EPI_SOFTMAX64:
            if constexpr ((BN / WN) % 64 == 0 && BM >= 128) return launch_glds_epi<BM, BN, WM, WN, NSTAGE, EPI_SOFTMAX64>(a, st);
            else { set_error("gemm: the softmax epilogue needs waves of 64-column groups"); return 1; }
        default: set_error("gemm: bad epilogue"); return 1;
    }
}

#ifdef RALD_PROBE
static int launch_geglu_persist(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (256 + 256) * 64 * 2 + 8 * 2304 + 2 * 8 * 512;
    static int n_cu = 0;
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)gemm_geglu_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        RALD_HIP(hipFuncSetAttribute((const void*)gemm_geglu_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        hipDeviceProp_t prop;
        RALD_HIP(hipGetDevice(&dev));
        RALD_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int ntn = a.N / 256, ntm = a.M / 256;
    int grid = n_cu - n_cu % 8;                               // a multiple of 8: a workgroup's tiles share its XCD (speed only)
    if (grid > ntn * ntm) grid = ntn * ntm;
    if (grid < 8) grid = ntn * ntm < 8 ? ntn * ntm : 8;
    const bool stagger = RALD_PROBE_ENV("RALD_GEMM_STAGGER", 1) != 0;
    if (stagger) hipLaunchKernelGGL(gemm_geglu_persist_kernel<true>, dim3(grid), dim3(512), smem, st, a, ntn, ntm);
    else hipLaunchKernelGGL(gemm_geglu_persist_kernel<false>, dim3(grid), dim3(512), smem, st, a, ntn, ntm);
    RALD_HIP(hipGetLastError());
    return 0;
}
#endif

int f16_saturation_gemm(unsigned* count, bool reset) {
    RALD_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_f16_sat_gemm), sizeof(unsigned)));
    if (reset) { const unsigned z = 0; RALD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_f16_sat_gemm), &z, sizeof(unsigned))); }
    return 0;
}

// Host-side shape contract is checked here, before any launch (an out-of-bounds MFMA tile
// can take the whole node down, so nothing is left to the kernel).
static int gemm_nt_impl(const GemmArgs& a, int epi, hipStream_t st);
int gemm_nt(const GemmArgs& a0, int epi, hipStream_t st) {
    // bf16 outputs are streamed (written once, read by the next kernel after the whole tensor has
    // passed through): non-temporal stores keep them from evicting the weight panels out of L2.
    static const int nt_store = RALD_PROBE_ENV("RALD_NT_STORE", 1);
    GemmArgs a = a0;
    if (nt_store) a.ablate |= 64;
    static const int st_flavour = RALD_PROBE_ENV("RALD_GEMM_STORE", 0);     // probe builds: 1 = sc0 sc1 (no nt), 2 = sc1
    if (st_flavour == 1) a.ablate |= 512;
    if (st_flavour == 2) a.ablate |= 1024;
    static const int diag = RALD_PROBE_ENV("RALD_GEMM_ABLATE", 0);   // probe builds (PMC runs): OR-ed into GemmArgs::ablate
    a.ablate |= diag;
    return gemm_nt_impl(a, epi, st);
}
static int gemm_nt_impl(const GemmArgs& a, int epi, hipStream_t st) {
    RALD_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0 && a.batch2 > 0, "gemm: empty problem");
    const int64_t nbatch = (int64_t)a.batch * a.batch2;
    RALD_CHECK(nbatch <= 65535, "gemm: batch * batch2 exceeds the grid z limit");
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
    if (epi == EPI_F16S) {                                        // split-K slabs (resid_splitk_ln): the 64x64 ring or 128x128 tiles, as EPI_F32 would get
        RALD_CHECK(!a.out8 && a.N % 8 == 0, "gemm: the fp16-slab epilogue needs N % 8 == 0");
        const int64_t w128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * nbatch, w64 = (int64_t)cdiv(a.M, 64) * cdiv(a.N, 64) * nbatch;
        RALD_CHECK(w128 >= 192 || w64 <= 256, "gemm: fp16 slabs are not built for the register-staged engine (ask gemm_f16s_ok first)");
        if (w128 >= 192) return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
        return launch_glds<64, 64, 2, 2, 8>(a, epi, st);
    }
    if (epi == EPI_SOFTMAX64) {
        // whole tiles only: every wave normalises complete 64-column groups of complete rows
        RALD_CHECK(a.M % 128 == 0 && a.N % 128 == 0 && !a.bias && !a.out8 && a.alpha_ncols >= a.N, "gemm: the softmax epilogue needs M, N % 128 == 0 and no bias");
        if (a.M % 256 == 0 && a.N % 256 == 0 && (int64_t)(a.M / 256) * (a.N / 256) * nbatch >= 256) return launch_glds<256, 256, 4, 2, 2>(a, epi, st);
        return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
    }
    if (a.out8) {
        RALD_CHECK(epi == EPI_GEGLU && a.outs && a.batch * a.batch2 == 1 && a.M % 256 == 0 && a.N % 256 == 0 && (a.N / 2) % 32 == 0 &&
                   (int64_t)(a.M / 256) * (a.N / 256) >= 256, "gemm: the MXFP8 output form needs the GEGLU epilogue on full 256x256 tiles");
        return launch_glds<256, 256, 4, 2, 2>(a, epi, st);
    }
    // engine selection.  Default (-1): LDS-DMA 256x256 tiles (8 waves) when they give every CU at
    // least one tile, LDS-DMA 128x128 (4 waves, 2 workgroups/CU) otherwise, register-staged 64x64 for
    // the small-M (batch-1) regime.  RALD_GEMM_IMPL forces a variant for A/B microbenchmarks:
    // 0 register-staged 128x128, 1 LDS-DMA 128x128 2 stages, 2 ... 3 stages, 3 LDS-DMA 256x128 2 stages,
    // 4 256x128 3 stages, 5 256x256 2 stages.
    int impl = -1;
    impl = RALD_PROBE_ENV("RALD_GEMM_IMPL", -1);
    const int64_t wg128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * nbatch;
    if (wg128 < RALD_PROBE_ENV("RALD_GEMM_SMALL_MAX", 192)) {
        // small-M (batch-1) regime: too few tiles to hide memory latency behind other workgroups, so put
        // (up to) the whole K extent in flight at once: 64x64 tiles, 8-stage LDS-DMA ring (128 KB).
        const int64_t wg64 = (int64_t)cdiv(a.M, 64) * cdiv(a.N, 64) * nbatch;
        static const int mid = RALD_PROBE_ENV("RALD_GEMM_MID", 0);   // A/B: 1 = LDS-DMA 128x128 from 96 tiles up, 2 = LDS-DMA 64x64 ring always
        if (mid == 1 && wg128 >= 96) return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
        if (mid == 2) return launch_glds<64, 64, 2, 2, 8>(a, epi, st);
        if (impl == 0 || wg64 > 256) return launch_tile<64, 64>(a, epi, st);   // more than one tile per CU: 5 small workgroups/CU hide latency
        return launch_glds<64, 64, 2, 2, 8>(a, epi, st);
    }
    if (impl < 0) {
        const int64_t wg256 = (int64_t)(a.M / 256) * (a.N / 256) * nbatch;
#ifdef RALD_PROBE
        // probe builds, RALD_GEMM_PERSIST=1: the persistent engine (one workgroup per CU) for GEGLU projections of >= 2 rounds of tiles
        if (epi == EPI_GEGLU && nbatch == 1 && a.M % 256 == 0 && a.N % 256 == 0 && wg256 >= 512 && a.K % 128 == 0 && !a.out8 &&
            RALD_PROBE_ENV("RALD_GEMM_PERSIST", 0) != 0)
            return launch_geglu_persist(a, st);
#endif
        if (a.M % 256 == 0 && a.N % 256 == 0 && wg256 >= 256) return launch_glds<256, 256, 4, 2, 2>(a, epi, st);
        // 192-320 tiles of 128 x 128 = about one 4-wave workgroup per CU, walking its k-steps as a latency chain (N = 512 projections at
        // M = 8192: 19-21 us for 4.3 GFLOP).  64 x 128 tiles with 3 stages give two workgroups per CU and a deeper prefetch: NFE at
        // B = 16 4.63 -> 4.36 ms, B = 8 2.94 -> 2.90 ms (RALD_GEMM_64x128=0 in probe builds: the old choice).
        if (epi != EPI_SOFTMAX64 && epi != EPI_GEGLU && wg128 <= RALD_PROBE_ENV("RALD_GEMM_64x128_MAX", 320) && a.M % 64 == 0 && RALD_PROBE_ENV("RALD_GEMM_64x128", 1)) return launch_glds<64, 128, 2, 2, 3>(a, epi, st);
        return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
    }
    switch (impl) {
        case 0: return launch_tile<128, 128>(a, epi, st);
        case 2: return launch_glds<128, 128, 2, 2, 3>(a, epi, st);
        case 3: if (a.M % 256 == 0 && wg128 >= 512) return launch_glds<256, 128, 4, 2, 2>(a, epi, st);
                return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
        case 4: if (a.M % 256 == 0 && wg128 >= 512) return launch_glds<256, 128, 4, 2, 3>(a, epi, st);
                return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
        case 5: if (a.M % 256 == 0 && a.N % 256 == 0 && wg128 >= 1024) return launch_glds<256, 256, 4, 2, 2>(a, epi, st);
                return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
        default: return launch_glds<128, 128, 2, 2, 2>(a, epi, st);
    }
}

}  // namespace rald
