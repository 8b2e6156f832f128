// Weight-gradient GEMM: C[n1][n2] += sum_m A[m][n1] . B[m][n2]  - both operands row-major with the CONTRACTED index on the rows.
//
// This is dW = dY^T . X of every Linear in the training step (autograd's mm(dY.t(), X) in the
// reference, engine_generation.py:93-104): dY [M, N1] and X [M, N2] are activations as the forward / backward pass leaves them,
// M = batch x 512 rows.  The NT engine (gemm.hip) needs both operands K-contiguous, i.e. two transposed bf16 copies per Linear
// (~900 transpose launches and 2 x the activation bytes per iteration) and then meets a GEMM with a handful of output tiles and
// K = M.  Here the tiles are staged as they lie in memory - [64 rows of m][128 columns] by LDS-DMA - and the MFMA fragments are
// read TRANSPOSED from LDS (ds_read_b64_tr_b16: a 16-lane group reads a 4 x 16 block and every lane receives one column of
// it), and the m range is split over blockIdx.z so that a 512 x 512 gradient still fills the chip; the splits meet in fp32
// atomics (the gradient buffers are accumulated into anyway; order-dependent in the last bits like the other atomics of the
// backward pass).
//
// LDS image of a tile: rows of 256 bytes (128 bf16), 16-byte chunk c of row m stored at c ^ f(m), f(m) = 2 (m & 3) | 8 ((m >> 3) & 1):
// the 4 rows of a transposed block (32 bytes each) land in 4 different 32-byte slots, and the row blocks of lane groups 0 / 1
// (rows 8 apart) in different halves of the 256-byte bank row.
#include "common.h"
#include "kernels.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct GemmTnArgs {
    const bf16* A; int64_t lda;      // [M][N1]
    const bf16* B; int64_t ldb;      // [M][N2]
    float* C; int64_t ldc;           // [N1][N2] fp32, accumulated into
    float* colsum;                   // optional [N1]: += sum_m A[m][n1] (the bias gradient of the same Linear)
    // slab form (no atomics, bit-reproducible): row range r writes its partial C to slab[r][N1][N2] and its column sums to slab_b[r][N1]
    // with plain stores; tn_reduce_kernel adds the ranges in order into C / colsum.  null = the atomic form.
    float* slab; float* slab_b;
    int M, N1, N2, rows_per_split;
    int t1, t2;                      // 128-wide tiles along n1, n2 (1-D launch: see the kernel)
    // CONV form (weight gradient of a 3x3x3 Conv3d, channels-last input x [B][ID][IH][IW][Cin]): row m = output voxel, column
    // n2 = tap * Cin + ci of the VIRTUAL patch matrix B[m][n2] = x[b][od*s - p + kd][oh*s - p + kh][ow*s - p + kw][ci] (zero outside);
    // C is the parameter's own [Cout][Cin][27] layout: output column n2 lands at ci * 27 + tap.
    int ID, IH, IW, Cin, OD, OH, OW, stride, pad;
    int lw, lh, ld;                  // log2 of OW, OH, OD when all three are powers of two (the decode is shifts then), else -1
};

__device__ __attribute__((aligned(16))) bf16 g_tn_zero[8];      // (zero-initialised) source of the patch matrix' padding

__device__ __forceinline__ int tn_swz(int m) { return ((m & 3) << 1) | (((m >> 3) & 1) << 3); }

// A-type fragment (16 columns n0..n0+15 as the MFMA's row index, k = 8 kq .. 8 kq + 7 of the 32-row k-step at tile row m0):
// lane (i = lane & 15, kq = lane >> 4) receives tile[m0 + 8 kq + e][n0 + i], e = 0..7
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* tile, int m0, int n0, int lane) {
    const int kq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = m0 + 8 * kq + qq, r1 = r0 + 4;
    const int c = (n0 >> 3) + (pp >> 1);                       // logical 16-byte chunk holding columns n0 + 4 pp .. + 3
    const unsigned char* p0 = tile + r0 * 256 + ((c ^ tn_swz(r0)) << 4) + 8 * (pp & 1);
    const unsigned char* p1 = tile + r1 * 256 + ((c ^ tn_swz(r1)) << 4) + 8 * (pp & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// NARROW (N1 <= 64): the n1 half of the 128-wide tile would be idle, so the two wave rows split each 64-row step between them instead
// (wave row wa takes the 32-row sub-step ks = wa); their partial sums meet in the same atomics as the row splits.
template <bool CONV, bool NARROW>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
    constexpr int TILE = 64 * 256;                              // one operand tile: 64 rows of m x 128 columns
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];      // [stage][A | B]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave >> 1, wb = wave & 1;                    // wave tile: 64 (n1) x 64 (n2)
    // 1-D launch, dealt to the XCDs by launch index (g & 7): every workgroup of one row range - they share its rows of A (per n1 tile) and of
    // B (per n2 tile) - runs on the same XCD, so the range crosses the fabric once (with (n2, n1, range) grid order the four n2 tiles of
    // a dY slice sat on four XCDs and each k-step waited for HBM: 3.5 us per step).  Range = 8 * (slot / tiles) + XCD, tile = slot % tiles.
    const int g_ = blockIdx.x, slot_ = g_ >> 3, ntile = a.t1 * a.t2;
    const int tile_ = slot_ % ntile, split_ = (slot_ / ntile) * 8 + (g_ & 7);
    const int bx = tile_ % a.t2, by = tile_ / a.t2;
    const int n1_0 = by * 128, n2_0 = bx * 128;
    const int m_begin = split_ * a.rows_per_split;
    int m_end = m_begin + a.rows_per_split;
    if (m_end > a.M) m_end = a.M;
    const int nsteps = (m_end - m_begin + 63) / 64;
    if (nsteps <= 0) return;
    // DMA: one instruction = 4 rows x 256 B; wave w stages rows 4 (w + 4 p) .. + 3 of each operand tile, p = 0..3
    const int lr = lane >> 4, pc = lane & 15;
    auto stage = [&](int s, int buf) {
        unsigned char* base = smem + buf * 2 * TILE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wave + 4 * p, r = 4 * piece + lr;
            const int lc = pc ^ tn_swz(r);
            int m = m_begin + 64 * s + r;
            m = m < m_end ? m : m_end - 1;                      // ragged tail: duplicated rows are masked out of the product below
            int ca = n1_0 + lc * 8, cb = n2_0 + lc * 8;
            ca = ca + 8 <= a.N1 ? ca : a.N1 - 8;                // columns past the matrix: re-read the last 8 (those outputs are not stored)
            cb = cb + 8 <= a.N2 ? cb : a.N2 - 8;
            __builtin_amdgcn_global_load_lds((glb_void*)(a.A + (int64_t)m * a.lda + ca), (lds_void*)(base + piece * 1024), 16, 0, 0);
            const bf16* srcb;
            if constexpr (CONV) {
                const int tap = cb / a.Cin, ci = cb - tap * a.Cin;              // (Cin % 8 == 0: a chunk never straddles two taps)
                const int kd = tap / 9, kh = (tap - 9 * kd) / 3, kw = tap - 9 * kd - 3 * kh;
                int ow, oh, od, b;
                if (a.lw >= 0) {
                    ow = m & (a.OW - 1);
                    int r2 = m >> a.lw;
                    oh = r2 & (a.OH - 1); r2 >>= a.lh;
                    od = r2 & (a.OD - 1);
                    b = r2 >> a.ld;
                } else {
                    ow = m % a.OW;
                    int r2 = m / a.OW;
                    oh = r2 % a.OH; r2 /= a.OH;
                    od = r2 % a.OD;
                    b = r2 / a.OD;
                }
                const int id = od * a.stride - a.pad + kd, ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
                const bool in = (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
                srcb = in ? a.B + ((((int64_t)b * a.ID + id) * a.IH + ih) * a.IW + iw) * a.Cin + ci : g_tn_zero;
            } else srcb = a.B + (int64_t)m * a.ldb + cb;
            __builtin_amdgcn_global_load_lds((glb_void*)srcb, (lds_void*)(base + TILE + piece * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) csum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_cs = a.colsum != nullptr && bx == 0 && wb == 0;
    const int n1_w = NARROW ? 0 : 64 * wa;                      // first n1 column of this wave inside the tile
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    stage(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < nsteps) stage(s + 1, (s + 1) & 1);
        const unsigned char* tA = smem + (s & 1) * 2 * TILE;
        const unsigned char* tB = tA + TILE;
        const int valid = m_end - (m_begin + 64 * s);           // rows of this step that exist (>= 64 except on the last step)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (NARROW && ks != wa) continue;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tn_frag(tA, 32 * ks, (NARROW ? 0 : 64 * wa) + 16 * i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = tn_frag(tB, 32 * ks, 64 * wb + 16 * j, lane);
            if (valid < 64) {                                   // ragged tail (wave-uniform): zero the k positions that are padding
                const int k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + e >= valid) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[i][e] = (bf16)0.f;
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (want_cs) csum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, csum[i], 0, 0, 0);
            }
        }
    }
    // acc[i][j][e] = C[n1_0 + 64 wa + 16 i + 4 (lane >> 4) + e][n2_0 + 64 wb + 16 j + (lane & 15)]
    const int rq = lane >> 4, cl = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n1 = n1_0 + n1_w + 16 * i + 4 * rq + e;
            if (n1 >= a.N1) continue;
            if (a.slab) {                                        // (NARROW: the two wave rows hold partial sums of the same outputs - atomic form only)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n2 = n2_0 + 64 * wb + 16 * j + cl;
                    if (n2 < a.N2) a.slab[((int64_t)split_ * a.N1 + n1) * a.N2 + n2] = acc[i][j][e];
                }
                if (want_cs && cl == 0) a.slab_b[(int64_t)split_ * a.N1 + n1] = csum[i][e];
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n2 = n2_0 + 64 * wb + 16 * j + cl;
                if (n2 >= a.N2) continue;
                if constexpr (CONV) {
                    const int tap = n2 / a.Cin, ci = n2 - tap * a.Cin;
                    unsafeAtomicAdd(a.C + (int64_t)n1 * a.ldc + ci * 27 + tap, acc[i][j][e]);
                } else unsafeAtomicAdd(a.C + (int64_t)n1 * a.ldc + n2, acc[i][j][e]);
            }
            if (want_cs && cl == 0) unsafeAtomicAdd(a.colsum + n1, csum[i][e]);
        }
}

// =====================================================================================================================
// Conv3d weight gradient, stride 1 / pad 1, line-staged: one workgroup = one (kd, kh) pair x one 64 x 64 (Cout x Cin) block of
// the parameter x a range of output lines.  Per step it stages 64 output voxels (64 / W whole lines) of dy and the matching
// INPUT lines (d + kd - 1, h + kh - 1) with a zero voxel on either side, and contracts them three times - the taps kw = 0, 1, 2
// are the same LDS rows shifted by one.  The generic form above reads every input voxel 27 times and dy once per 128 patch
// columns (11 GB from L2 for a 64 -> 64 convolution on 2.1 M voxels: 2 ms); this one reads both 9 times.
// =====================================================================================================================
struct WgradLineArgs {
    const bf16* dy; const bf16* x; float* dW; float* dbias;
    int D, H, W, lw, lh, ld, Cin, Cout, lines, steps_per_split, nsteps;   // OUTPUT dims; lines = B*D*H; a step = 64 / W lines; lh / ld = log2(H), log2(D) or -1
    int tiles;                                                            // 64 x 64 blocks of the parameter
    float* slab; float* slab_b;                                           // slab form (see GemmTnArgs): slab[range][Cout][27][Cin], slab_b[range][Cout]
    int ablate;                                                           // probe builds (RALD_WGRAD_ABLATE): 1 no MFMA, 2 no LDS reads, 4 no DMA after the first stage, 8 no atomics
};

// 8-row DMA pieces sit 1152 bytes apart (1024 + 128): rows 8 apart - the row blocks of neighbouring lane groups of one transposed read -
// then fall into different halves of the 256-byte bank row (with a plain 128-byte pitch all four groups hit the same banks).
constexpr int WL_PIECE = 1152;
// byte offset inside a 128-byte-pitch tile (64 columns) of the 8 bytes lane `lane` hands to ds_read_b64_tr_b16 for the 4-row block at
// row r0 and the 16 columns at n0: row r0 + qq, columns n0 + 4 pp .. + 3; chunk ^ 2 (row & 3)
__device__ __forceinline__ int wl_off(int r0, int n0, int lane, int rstep = 1) {
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    const int c = (n0 >> 3) + (pp >> 1);
    const int r = r0 + rstep * qq;
    return (r >> 3) * WL_PIECE + (r & 7) * 128 + ((c ^ ((r & 3) << 1)) << 4) + 8 * (pp & 1);
}
__device__ __forceinline__ bf16x8 wl_read(const unsigned char* tile, int off_lo, int off_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off_hi));
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// S = 1: stride 1, pad 1 (input dims = output dims; input rows of a line: W + 2, a zero voxel on either side).
// S = 2: Downsample (stride 2 over the input padded by one voxel at the far ends, models_radar_encoder.py:37-41): input dims = 2 x output
// dims, input voxel (2 d + kd, 2 h + kh, 2 w + kw), out of range only at index 2 W; 2 W + 1 input rows per line, the four voxels of a
// transposed read are two rows apart.
template <int S>
__global__ __launch_bounds__(256) void conv_wgrad_line_kernel(WgradLineArgs a) {
    constexpr int PB = S == 1 ? 12 : 20;                        // 8-row pieces of the input tile: 96 / 160 rows (>= 64 / W lines of W + 2 / 2 W + 1 rows)
    constexpr int PPW = (8 + PB) / 4;                           // DMA pieces per wave and stage: 5 / 7
    constexpr int A_BYTES = 8 * WL_PIECE, B_BYTES = PB * WL_PIECE, STAGE = A_BYTES + B_BYTES, NST = S == 1 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave >> 1, wb = wave & 1;                    // wave: 32 (Cout) x 32 (Cin) x 3 taps
    // The 9 (kd, kh) workgroups of one (block, line range) read the same dy lines and overlapping input lines: keep them on ONE XCD
    // (launch index g -> XCD g & 7), or each XCD fetches its own copy from HBM (4.8 GB per 64 -> 64 convolution on 2.1 M voxels, 1.25 ms).
    // Slot g >> 3 of an XCD walks (kd, kh) fastest, then the block, then its share of the line ranges (range = 8 * k + XCD).
    const int gl = blockIdx.x, xcd = gl & 7, slot = gl >> 3;
    const int grp = slot % 9, rest = slot / 9;
    const int tile_id = rest % a.tiles, split = (rest / a.tiles) * 8 + xcd;
    const int kd = grp / 3, kh = grp % 3;
    const int ncb = a.Cin >> 6;
    const int co0 = (tile_id / ncb) * 64, ci0 = (tile_id % ncb) * 64;
    const int W = a.W, Wp = S == 1 ? W + 2 : 2 * W + 1, lps = 64 >> a.lw;            // output width, input rows per line, lines per step
    const int ID = S * a.D, IH = S * a.H, IW = S * W;
    const int s_begin = split * a.steps_per_split;
    int s_end = s_begin + a.steps_per_split;
    if (s_end > a.nsteps) s_end = a.nsteps;
    const int nst = s_end - s_begin;
    if (nst <= 0) return;
    const int prow = lane >> 3, pch = lane & 7;                 // DMA piece: 8 rows x 128 B
    // 5 pieces per wave and stage: pieces 0..7 = dy rows, 8..19 = input rows (96; the unused ones read the zero line).  Everything that
    // does not change from step to step is worked out here: the loop was bound by its own address arithmetic (2.2 us per step).
    int64_t a_off[PPW];        // dy piece: element offset of this lane's 16 bytes relative to the step's first voxel; -1 = an input piece
    int b_l[PPW], b_col[PPW];  // input piece: line inside the step (or -1: zero line) and element offset inside the input line
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int piece = wave + 4 * p;
        a_off[p] = -1; b_l[p] = -1; b_col[p] = 0;
        if (piece < 8) {
            const int r = 8 * piece + prow;
            a_off[p] = (int64_t)r * a.Cout + co0 + (pch ^ ((r & 3) << 1)) * 8;
        } else {
            const int r = 8 * (piece - 8) + prow;
            const int l = r / Wp, iwp = r - l * Wp;
            const int iw = S == 1 ? iwp - 1 : iwp;
            if (l < lps && iw >= 0 && iw < IW) { b_l[p] = l; b_col[p] = iw * a.Cin + ci0 + (pch ^ ((r & 3) << 1)) * 8; }
        }
    }
    auto stage = [&](int s, int buf) {
        unsigned char* base = smem + buf * STAGE;
        const int line0 = (s_begin + s) * lps;
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const int piece = wave + 4 * p;
            if (piece < 8) {                                     // (wave-uniform)
                __builtin_amdgcn_global_load_lds((glb_void*)(a.dy + (int64_t)line0 * W * a.Cout + a_off[p]), (lds_void*)(base + piece * WL_PIECE), 16, 0, 0);
            } else {
                const bf16* src = g_tn_zero;
                if (b_l[p] >= 0) {
                    const int line = line0 + b_l[p];
                    int h, d, b;
                    if (a.lh >= 0) { h = line & (a.H - 1); d = (line >> a.lh) & (a.D - 1); b = line >> (a.lh + a.ld); }
                    else { h = line % a.H; const int t = line / a.H; d = t % a.D; b = t / a.D; }
                    const int id = S == 1 ? d + kd - 1 : 2 * d + kd, ih = S == 1 ? h + kh - 1 : 2 * h + kh;
                    if ((unsigned)id < (unsigned)ID && (unsigned)ih < (unsigned)IH)
                        src = a.x + (((int64_t)b * ID + id) * IH + ih) * IW * a.Cin + b_col[p];
                }
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(base + A_BYTES + (piece - 8) * WL_PIECE), 16, 0, 0);
            }
        }
    };
    // fragment read offsets (per lane, the same in every stage)
    const int kq = lane >> 4;
    int oa[2][2][2], ob[2][3][2][2];                            // [ks][i][lo/hi], [ks][tap][j][lo/hi]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int v0 = 32 * ks + 8 * kq;                         // this lane group's 8 voxels: two runs of 4 inside one line each
        const int rb0 = (v0 >> a.lw) * Wp + S * (v0 & (W - 1)), rb1 = ((v0 + 4) >> a.lw) * Wp + S * ((v0 + 4) & (W - 1));
#pragma unroll
        for (int i = 0; i < 2; ++i) { oa[ks][i][0] = wl_off(v0, 32 * wa + 16 * i, lane); oa[ks][i][1] = wl_off(v0 + 4, 32 * wa + 16 * i, lane); }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ob[ks][t][j][0] = A_BYTES + wl_off(rb0 + t, 32 * wb + 16 * j, lane, S);
                ob[ks][t][j][1] = A_BYTES + wl_off(rb1 + t, 32 * wb + 16 * j, lane, S);
            }
    }
    f32x4 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool want_cs = a.dbias != nullptr && grp == 0 && ci0 == 0 && wb == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    stage(0, 0);
    if (NST == 3 && nst > 1) stage(1, 1);
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        // three stages: the pieces of stage s+1 may still be in flight; two stages: nothing is (stage s+1 is issued below)
        if (NST == 3 && s + 1 < nst) { if (S == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NST - 1 < nst && !RALD_ABLATED(a.ablate, 4)) stage(s + NST - 1, buf >= 1 ? buf - 1 : NST - 1);          // (s + NST - 1) % NST
        const unsigned char* tS = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = RALD_ABLATED(a.ablate, 2) ? ones : wl_read(tS, oa[ks][i][0], oa[ks][i][1]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 fb = RALD_ABLATED(a.ablate, 2) ? ones : wl_read(tS, ob[ks][t][j][0], ob[ks][t][j][1]);
                    if (RALD_ABLATED(a.ablate, 1)) { acc[t][0][j][0] += (float)fb[0] * (float)fa[0][1]; acc[t][1][j][1] += (float)fb[2] * (float)fa[1][3]; continue; }
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[t][i][j], 0, 0, 0);
                }
            }
            if (want_cs) {
#pragma unroll
                for (int i = 0; i < 2; ++i) csum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, csum[i], 0, 0, 0);
            }
        }
        buf = buf + 1 < NST ? buf + 1 : 0;
    }
    // acc[t][i][j][e] = dW[co0 + 32 wa + 16 i + 4 (lane >> 4) + e][ci0 + 32 wb + 16 j + (lane & 15)][tap (kd, kh, t)]
    if (RALD_ABLATED(a.ablate, 8) && acc[0][0][0][0] != 12345.f) return;
    const int cl = lane & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + 32 * wa + 16 * i + 4 * kq + e;
            if (a.slab) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float* dst = a.slab + (((int64_t)split * a.Cout + co) * 27 + kd * 9 + kh * 3) * a.Cin + ci0 + 32 * wb + 16 * j + cl;
#pragma unroll
                    for (int t = 0; t < 3; ++t) dst[(int64_t)t * a.Cin] = acc[t][i][j][e];
                }
                if (want_cs && cl == 0) a.slab_b[(int64_t)split * a.Cout + co] = csum[i][e];
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + 32 * wb + 16 * j + cl;
                float* dst = a.dW + ((int64_t)co * a.Cin + ci) * 27 + kd * 9 + kh * 3;
#pragma unroll
                for (int t = 0; t < 3; ++t) unsafeAtomicAdd(dst + t, acc[t][i][j][e]);
            }
            if (want_cs && cl == 0) unsafeAtomicAdd(a.dbias + co, csum[i][e]);
        }
}

// C (+)= sum over the row ranges, in order, of slab[r][n1][n2]; CONV: C is the parameter's [Cout][Cin][27], n2 = tap * Cin + ci.
// One thread per output, reads coalesced along n2 (the slabs are a few MB and sit in L2 / MALL); colsum likewise from slab_b.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_b, int ranges, int N1, int N2, int Cin,
                                                        float* __restrict__ C, int64_t ldc, float* __restrict__ colsum) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)N1 * N2;
    if (idx < n) {
        float s0 = 0.f, s1 = 0.f;
        int r = 0;
        for (; r + 2 <= ranges; r += 2) { s0 += slab[(int64_t)r * n + idx]; s1 += slab[(int64_t)(r + 1) * n + idx]; }
        if (r < ranges) s0 += slab[(int64_t)r * n + idx];
        const int n1 = (int)(idx / N2), n2 = (int)(idx - (int64_t)n1 * N2);
        if (Cin > 0) { const int tap = n2 / Cin, ci = n2 - tap * Cin; C[(int64_t)n1 * ldc + ci * 27 + tap] += s0 + s1; }
        else C[(int64_t)n1 * ldc + n2] += s0 + s1;
    }
    if (colsum && idx < N1) {
        float s = 0.f;
        for (int r = 0; r < ranges; ++r) s += slab_b[(int64_t)r * N1 + idx];
        colsum[idx] += s;
    }
}
static int tn_reduce(const float* slab, const float* slab_b, int ranges, int N1, int N2, int Cin, float* C, int64_t ldc, float* colsum, hipStream_t st) {
    const int64_t n = (int64_t)N1 * N2;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, slab_b, ranges, N1, N2, Cin, C, ldc, colsum);
    RALD_HIP(hipGetLastError());
    return 0;
}

// row ranges of the generic kernel: a multiple of 8 (one per XCD), at least 256 rows (4 k-steps) per range.  Atomic form: ~1024 workgroups.
// Slab form: one range per XCD once there are 32 tiles (measured at 4 096 rows, tools/bench_tn_target.py: 512 x 2048 39.8 -> 29.6 us,
// 1536 x 512 34.5 -> 28.9 us against the atomic form's count; 4096 x 512 is 8 ranges either way) - every further range is another slab to
// write and to sum.
static void tn_ranges(int M, int N1, int N2, int& rows, int& used, int& launched, bool slab) {
    const int t1 = cdiv(N1, 128), t2 = cdiv(N2, 128);
    int splits = cdiv(RALD_PROBE_ENV("RALD_TN_TARGET", 1024), t1 * t2);
    if (slab && RALD_PROBE_ENV("RALD_TN_TARGET", 0) == 0) splits = t1 * t2 >= 32 ? 8 : (int)round_up(cdiv(256, t1 * t2), 8);
    const int max_splits = cdiv(M, 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    rows = (int)round_up(cdiv(M, splits), 64);
    used = cdiv(M, rows);
    launched = (int)round_up(used, 8);                          // (ranges past the last row return at once)
}

static int tn_launch(GemmTnArgs a, bool conv, hipStream_t st, float* workspace = nullptr) {
    const int t1 = cdiv(a.N1, 128), t2 = cdiv(a.N2, 128);
    int rows, used, splits;
    tn_ranges(a.M, a.N1, a.N2, rows, used, splits, workspace != nullptr && a.N1 > 64);
    RALD_CHECK((int64_t)t1 * t2 * splits < ((int64_t)1 << 31), "gemm_tn: grid too large");
    a.rows_per_split = rows; a.t1 = t1; a.t2 = t2;
    const dim3 grid((unsigned)(t1 * t2 * splits));
    const bool narrow = a.N1 <= 64;
    a.slab = nullptr; a.slab_b = nullptr;
    float* C = a.C; float* colsum = a.colsum;
    if (workspace && !narrow) {                                  // slab form: partials per row range, summed in order afterwards
        a.slab = workspace;
        a.slab_b = workspace + (int64_t)used * a.N1 * a.N2;
    }
    if (conv && narrow) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, dim3(256), 0, st, a);
    else if (conv) hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, dim3(256), 0, st, a);
    else if (narrow) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    if (a.slab) return tn_reduce(a.slab, a.slab_b, used, a.N1, a.N2, conv ? a.Cin : 0, C, a.ldc, colsum, st);
    return 0;
}

int64_t gemm_tn_workspace_floats(int M, int N1, int N2) {
    if (M < 1 || N1 <= 64 || N2 < 8) return 0;                  // the narrow form keeps its atomics
    int rows, used, launched;
    tn_ranges(M, N1, N2, rows, used, launched, true);
    return (int64_t)used * ((int64_t)N1 * N2 + N1);
}
int gemm_tn(const bf16* A, int64_t lda, const bf16* B, int64_t ldb, float* C, int64_t ldc, float* colsum, int M, int N1, int N2, hipStream_t st,
            float* workspace, int64_t workspace_floats) {
    RALD_CHECK(A && B && C && M >= 1 && N1 >= 8 && N2 >= 8, "gemm_tn: bad arguments");
    RALD_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= N1 && ldb >= N2 && ldc >= N2, "gemm_tn: leading dimensions (16-byte rows)");
    RALD_CHECK(N1 % 8 == 0 && N2 % 8 == 0, "gemm_tn: N1, N2 must be multiples of 8");
    RALD_CHECK((uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0, "gemm_tn: 16-byte alignment");
    GemmTnArgs a = {};
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.colsum = colsum; a.M = M; a.N1 = N1; a.N2 = N2;
    const int64_t need = gemm_tn_workspace_floats(M, N1, N2);
    if (workspace) RALD_CHECK(workspace_floats >= need && (uintptr_t)workspace % 16 == 0, "gemm_tn: workspace smaller than gemm_tn_workspace_floats");
    return tn_launch(a, false, st, need > 0 ? workspace : nullptr);
}

// dW [Cout][Cin][27] += sum over output voxels of dy[v][co] . x[v + offset(tap)][ci];  dbias [Cout] += column sums of dy.
// dy [B*OD*OH*OW][Cout] bf16, x [B][ID][IH][IW][Cin] bf16 (channels-last), OD = ID / stride etc. (3x3x3 kernel).
// workspace (conv3d_wgrad_workspace_floats floats, or null): the voxel ranges leave their partial gradients there with plain stores and a
// second launch adds them, in order, into dW / dbias - bit-reproducible, and 4-13 x faster below full resolution: the atomic form ends every
// workgroup in 12 288 atomics whose 64 lanes hit 64 different cache lines (the parameter layout puts the 27 taps innermost), which at
// 32 x 16 x 8 voxels was 93 % of the launch (928 -> 70 us without them, tools/bench_wgrad_levels.py).  null = the atomic form.
namespace {
struct WgradPlan { bool line; int splits, used, steps_per_split, nsteps, tiles, lw, lh, ld; };
int lg2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
WgradPlan wgrad_plan(int B, int ID, int IH, int IW, int Cin, int Cout, int stride, int pad) {
    WgradPlan p = {};
    const int OD = ID / stride, OH = IH / stride, OW = IW / stride;
    const bool s1 = stride == 1 && pad == 1, s2 = stride == 2 && pad == 0 && ID % 2 == 0 && IH % 2 == 0 && IW % 2 == 0;
    p.lw = lg2_exact(OW); p.lh = lg2_exact(OH); p.ld = lg2_exact(OD);
    p.line = (s1 || s2) && Cin % 64 == 0 && Cout % 64 == 0 && p.lw >= 2 && OW <= 64 && ((int64_t)B * OD * OH) % (64 / OW) == 0 &&
             (int64_t)B * OD * OH < ((int64_t)1 << 30);
    if (p.line) {
        p.nsteps = B * OD * OH / (64 / OW);
        p.tiles = (Cin / 64) * (Cout / 64);
        // Line ranges: a multiple of 8 (one per XCD and slot round), chosen so that the workgroups fill whole rounds of the chip's 512 slots
        // (2 per CU) - 1 080 workgroups ran as two rounds and a third one 11 % full - and as few as that allows (every range ends in a
        // 64 x 64 x 3 block of partial sums per workgroup)
        int splits = 8;
        double best = -1.0;
        const int forced = RALD_PROBE_ENV("RALD_WGRAD_SPLITS", 0);
        for (int r = 1; r <= 4; ++r) {
            int sp = 8 * ((512 * r) / (72 * p.tiles));
            if (sp < 8) sp = 8;
            if (sp > (int)round_up(cdiv(p.nsteps, 8), 8)) sp = (int)round_up(cdiv(p.nsteps, 8), 8);
            const int wgs = 9 * p.tiles * sp;
            const double eff = (double)wgs / (double)(cdiv(wgs, 512) * 512);
            if (eff > best + 0.05) { best = eff; splits = sp; }
        }
        if (forced > 0) splits = (int)round_up(forced, 8);
        p.steps_per_split = cdiv(p.nsteps, splits);
        p.used = cdiv(p.nsteps, p.steps_per_split);
        p.splits = (int)round_up(p.used, 8);                     // (ranges past the last step return at once)
    } else {
        int rows;
        tn_ranges(B * OD * OH * OW, Cout, 27 * Cin, rows, p.used, p.splits, Cout > 64);
    }
    return p;
}
}  // namespace

int64_t conv3d_wgrad_workspace_floats(int B, int ID, int IH, int IW, int Cin, int Cout, int stride, int pad) {
    if (B < 1 || ID < 1 || IH < 1 || IW < 1 || (stride != 1 && stride != 2) || Cin < 8 || Cout < 8) return 0;
    if ((int64_t)B * (ID / stride) * (IH / stride) * (IW / stride) >= ((int64_t)1 << 31)) return 0;
    if (Cout <= 64 && !wgrad_plan(B, ID, IH, IW, Cin, Cout, stride, pad).line) return 0;     // the narrow generic form keeps its atomics
    const WgradPlan p = wgrad_plan(B, ID, IH, IW, Cin, Cout, stride, pad);
    return (int64_t)p.used * ((int64_t)Cout * 27 * Cin + Cout);
}

int conv3d_wgrad_tn(const bf16* dy, const bf16* x, float* dW, float* dbias, int B, int ID, int IH, int IW, int Cin, int Cout, int stride, int pad,
                    hipStream_t st, float* workspace, int64_t workspace_floats) {
    RALD_CHECK(dy && x && dW && B >= 1 && ID >= 1 && IH >= 1 && IW >= 1 && (stride == 1 || stride == 2) && pad >= 0, "conv3d_wgrad_tn: bad arguments");
    RALD_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "conv3d_wgrad_tn: channel counts must be multiples of 8");
    RALD_CHECK((uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0, "conv3d_wgrad_tn: 16-byte alignment");
    const int OD = ID / stride, OH = IH / stride, OW = IW / stride;
    const int64_t M = (int64_t)B * OD * OH * OW;
    RALD_CHECK(M >= 1 && M < ((int64_t)1 << 31) && (int64_t)B * ID * IH * IW * Cin < ((int64_t)1 << 40), "conv3d_wgrad_tn: volume too large");
    const WgradPlan p = wgrad_plan(B, ID, IH, IW, Cin, Cout, stride, pad);
    const int64_t need = conv3d_wgrad_workspace_floats(B, ID, IH, IW, Cin, Cout, stride, pad);
    if (workspace) RALD_CHECK(workspace_floats >= need && (uintptr_t)workspace % 16 == 0, "conv3d_wgrad_tn: workspace smaller than conv3d_wgrad_workspace_floats");
    if (need == 0) workspace = nullptr;
    if (p.line) {
        WgradLineArgs w;
        w.dy = dy; w.x = x; w.dW = dW; w.dbias = dbias; w.D = OD; w.H = OH; w.W = OW; w.lw = p.lw; w.Cin = Cin; w.Cout = Cout;
        w.lh = p.lh; w.ld = p.ld;
        if (w.lh < 0 || w.ld < 0) w.lh = w.ld = -1;
        w.lines = B * OD * OH;
        w.nsteps = p.nsteps;
        w.steps_per_split = p.steps_per_split;
        w.tiles = p.tiles;
        w.slab = workspace; w.slab_b = workspace ? workspace + (int64_t)p.used * Cout * 27 * Cin : nullptr;
        w.ablate = RALD_PROBE_ENV("RALD_WGRAD_ABLATE", 0);
        RALD_CHECK((int64_t)9 * p.tiles * p.splits < ((int64_t)1 << 31), "conv3d_wgrad_tn: grid too large");
        constexpr int LDS1 = 3 * 20 * WL_PIECE, LDS2 = 2 * 28 * WL_PIECE;
        static bool attr_set = false;
        if (!attr_set) {
            RALD_HIP(hipFuncSetAttribute((const void*)conv_wgrad_line_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
            RALD_HIP(hipFuncSetAttribute((const void*)conv_wgrad_line_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
            attr_set = true;
        }
        if (stride == 1) hipLaunchKernelGGL(conv_wgrad_line_kernel<1>, dim3(9 * p.tiles * p.splits), dim3(256), LDS1, st, w);
        else hipLaunchKernelGGL(conv_wgrad_line_kernel<2>, dim3(9 * p.tiles * p.splits), dim3(256), LDS2, st, w);
        RALD_HIP(hipGetLastError());
        if (workspace) return tn_reduce(w.slab, w.slab_b, p.used, Cout, 27 * Cin, Cin, dW, (int64_t)Cin * 27, dbias, st);
        return 0;
    }
    GemmTnArgs a = {};
    a.A = dy; a.lda = Cout; a.B = x; a.ldb = 0; a.C = dW; a.ldc = (int64_t)Cin * 27; a.colsum = dbias; a.M = (int)M; a.N1 = Cout; a.N2 = 27 * Cin;
    a.ID = ID; a.IH = IH; a.IW = IW; a.Cin = Cin; a.OD = OD; a.OH = OH; a.OW = OW; a.stride = stride; a.pad = pad;
    a.lw = p.lw; a.lh = p.lh; a.ld = p.ld;
    if (a.lw < 0 || a.lh < 0 || a.ld < 0) a.lw = a.lh = a.ld = -1;
    return tn_launch(a, true, st, workspace);
}

}  // namespace rald
