// Radar-spectrum encoder (model/models_radar_encoder.py Encoder :137-241) and the tokeniser half of
// EDMPrecond.process_radar_cond (models_radar_generation.py:363-407).
//
// Data layout (MI355X-first): activations are channels-LAST (NDHWC = [b][r][a][e][c]) so that one
// voxel's channels are contiguous - exactly an MFMA A-fragment row - and the encoder's output
// [B,8,4,2,16] is already in the `permute(0,2,3,4,1)` order the tokeniser wants (:387).  The
// residual trunk stays fp32; every conv input is the bf16 tensor swish(GroupNorm(x)) written by
// one HBM-bound pass (gn_apply), so the conv kernel is a pure implicit GEMM:
//     M = output voxels, N = Cout, K = 27*Cin  (weights packed [Cout][tap][Cin], K-contiguous).
// Conv3d zero padding applies to the normalised+activated tensor, i.e. out-of-range taps
// contribute exact zeros (register zero-fill, no padded copy).  Downsample = F.pad(0,1) + conv
// k3 s2 p0 (:37-41) is the same kernel with stride 2 and left pad 0.
#include "dit.h"

#include <cmath>
#include <cstdio>
#include <map>

namespace rald {

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// conv_in: Cin = 1 -> ch (64); the cube's channel 0 is read in place ([B,R,A,E,cube_ch], :378).
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ cube, int cube_ch, int Cin, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B, int D,
                                                      int H, int Wd, int Cout) {
    extern __shared__ float sw[];            // [Cin][27][Cout] (tap-major so 8 consecutive co are contiguous)
    for (int i = threadIdx.x; i < Cin * 27 * Cout; i += 256) {
        const int co = i % Cout, t = (i / Cout) % 27, ci = i / (27 * Cout);
        sw[i] = W[(co * Cin + ci) * 27 + t];
    }
    __syncthreads();
    const int groups = Cout / 8;                          // threads per voxel
    const int vpb = 256 / groups;                         // voxels per block
    const int64_t v = (int64_t)blockIdx.x * vpb + threadIdx.x / groups;
    const int cg = threadIdx.x % groups;
    const int64_t nvox = (int64_t)B * D * H * Wd;
    if (v >= nvox) return;
    int w = (int)(v % Wd);
    int64_t r = v / Wd;
    int h = (int)(r % H); r /= H;
    int d = (int)(r % D);
    int b = (int)(r / D);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = bias[cg * 8 + i];
    for (int kd = 0; kd < 3; ++kd)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int id = d + kd - 1, ih = h + kh - 1, iw = w + kw - 1;
                if ((unsigned)id >= (unsigned)D || (unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)Wd) continue;
                const float* xp = cube + ((((int64_t)b * D + id) * H + ih) * Wd + iw) * cube_ch;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float x = xp[ci];
                    const float* wt = sw + (ci * 27 + (kd * 3 + kh) * 3 + kw) * Cout + cg * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += x * wt[i];
                }
            }
    float4* o = reinterpret_cast<float4*>(out + v * Cout + cg * 8);
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// GroupNorm statistics (32 groups), x [B][S][C] fp32 -> stats[b][g] = {sum, sumsq} in double.  Deterministic (no atomics: the
// radar condition - and with it every sample drawn from it - is bit-reproducible run to run): each block reduces its
// voxels in a fixed order and writes one partial per group, gn_finish_kernel adds the partials in block order.
// part[b][blk][g] = {sum, sumsq}, blk < gridDim.x.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ part, int S, int C,
                                                       int vox_per_block) {
    __shared__ float4 sp[256];
    const int b = blockIdx.y;
    const int quads = C / 4;                               // float4 pieces per voxel
    const int q = threadIdx.x % quads;
    const int vstep = 256 / quads;
    const int v0 = blockIdx.x * vox_per_block;
    const int v1 = min(S, v0 + vox_per_block);
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;         // channel pairs (4q,4q+1) and (4q+2,4q+3)
    int v = v0 + threadIdx.x / quads;
    for (; v + 3 * vstep < v1; v += 4 * vstep) {           // four loads in flight per lane (the loop is latency-bound otherwise)
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = reinterpret_cast<const float4*>(x + ((int64_t)b * S + v + u * vstep) * C)[q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += t[u].x + t[u].y; q0 += t[u].x * t[u].x + t[u].y * t[u].y;
            s1 += t[u].z + t[u].w; q1 += t[u].z * t[u].z + t[u].w * t[u].w;
        }
    }
    for (; v < v1; v += vstep) {
        const float4 t = reinterpret_cast<const float4*>(x + ((int64_t)b * S + v) * C)[q];
        s0 += t.x + t.y; q0 += t.x * t.x + t.y * t.y;
        s1 += t.z + t.w; q1 += t.z * t.z + t.w * t.w;
    }
    sp[threadIdx.x] = make_float4(s0, q0, s1, q1);
    __syncthreads();
    if (threadIdx.x < 32) {
        // C is 64 * 2^k (host-checked): channels per group cpg = 2^lcpg >= 2.  Group g owns quads [g*cpg/4, (g+1)*cpg/4) of every
        // voxel row of the block (cpg = 2: one half of quad g/2); visit exactly those, rows then quads, in a fixed order
        const int g = threadIdx.x, lcpg = 31 - __clz(C / 32), rows = 256 / quads;
        const int tq0 = lcpg >= 2 ? g << (lcpg - 2) : g >> 1, tq1 = lcpg >= 2 ? (g + 1) << (lcpg - 2) : tq0 + 1;
        double su = 0.0, sq = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int tq = tq0; tq < tq1; ++tq) {
                const float4 p = sp[r * quads + tq];
                if (lcpg >= 2 || !(g & 1)) { su += (double)p.x; sq += (double)p.y; }
                if (lcpg >= 2 || (g & 1)) { su += (double)p.z; sq += (double)p.w; }
            }
        double* o = part + (((int64_t)b * gridDim.x + blockIdx.x) * 32 + g) * 2;
        o[0] = su; o[1] = sq;
    }
}
// stats[b][g] = sum over the sample's partial slots part[b*nblk + k][g] (statistics blocks or conv tiles) in a fixed order:
// sixteen contiguous ranges in parallel (threads 64q .. 64q+63), then the ranges in order
__global__ __launch_bounds__(1024) void gn_finish_kernel(const double* __restrict__ part, double* __restrict__ stats, int nblk) {
    __shared__ double sq[16][64];
    const int b = blockIdx.x, gw = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k0 = (int)((int64_t)nblk * q / 16), k1 = (int)((int64_t)nblk * (q + 1) / 16);
    double acc = 0.0;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc += part[((int64_t)b * nblk + k) * 64 + gw];
    sq[q][gw] = acc;
    __syncthreads();
    if (q == 0) {
        double t = sq[0][gw];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += sq[i][gw];
        stats[(int64_t)b * 64 + gw] = t;
    }
}
constexpr int GN_VPB = 512;                                // voxels per statistics block
static inline int gn_blocks(int S) { return (S + GN_VPB - 1) / GN_VPB; }

// y_bf16 = act(GroupNorm(x)) with per-channel affine; act = swish (x*sigmoid(x), :5-7) or identity.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16* __restrict__ y, int S, int C, float eps, int do_swish) {
    __shared__ float smean[32], srstd[32];
    const int b = blockIdx.y;
    const int quads = C / 4;
    const int cpg = C / 32;
    if (threadIdx.x < 32) {
        const double n = (double)S * cpg;
        const double su = stats[((int64_t)b * 32 + threadIdx.x) * 2], sq = stats[((int64_t)b * 32 + threadIdx.x) * 2 + 1];
        const double mean = su / n;
        const double var = sq / n - mean * mean;               // biased variance, as torch GroupNorm
        smean[threadIdx.x] = (float)mean;
        srstd[threadIdx.x] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)S * quads; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % quads);
        const int64_t idx = (int64_t)b * S * quads + i;
        const float4 t = reinterpret_cast<const float4*>(x)[idx];
        const float4 gm = reinterpret_cast<const float4*>(gamma)[q];
        const float4 bt = reinterpret_cast<const float4*>(beta)[q];
        const float v[4] = {t.x, t.y, t.z, t.w};
        const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
        const float bb[4] = {bt.x, bt.y, bt.z, bt.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (4 * q + j) / cpg;
            float yv = (v[j] - smean[g]) * srstd[g] * gg[j] + bb[j];
            if (do_swish) yv = yv / (1.0f + __expf(-yv));
            o[j] = yv;
        }
        reinterpret_cast<bf16x4*>(y)[idx] = pack4(o[0], o[1], o[2], o[3]);
    }
}

// Implicit-GEMM Conv3d k3: same tile machinery as gemm_nt_kernel (128 voxels x 64 couts x 64 K,
// 4 waves 2x2, XOR-swizzled double-buffered LDS); the A rows are gathered per tap.
struct ConvArgs {
    const bf16* in;      // [B][ID][IH][IW][Cin]
    const bf16* w;       // [Cout][27][Cin]
    const float* bias;   // [Cout]
    const float* resid;  // [M][Cout] or nullptr
    float* out;          // [M][Cout]
    int B, ID, IH, IW, Cin, OD, OH, OW, Cout, stride, pad;
    // optional: GroupNorm(32) partial statistics of the OUTPUT, one slot per 128-voxel tile (requires OD*OH*OW % 128 == 0 so that
    // no tile straddles two samples): gn_part[tile][32 groups][2] doubles = {sum, sumsq}; saves the statistics pass over the fp32
    // activation (537 MB at full resolution, B = 8) that the next GroupNorm would otherwise make
    double* gn_part = nullptr;
    // optional split-K (few tiles, long K: the 64- and 512-voxel levels): gridDim.z workgroups share a tile, each takes a contiguous
    // range of k-steps and writes its raw accumulators to split_part[z][M][Cout]; conv_split_reduce_kernel adds them in order
    float* split_part = nullptr;
    // optional: the result as bf16 INSTEAD of fp32 (training: a data gradient whose only reader is the GroupNorm backward - half the bytes
    // written here and read twice there); no residual, no split-K
    bf16* out16 = nullptr;
};

// Epilogue shared by the two convolution kernels: bias (+ fp32 residual), fp32 store, optional GroupNorm partials of the output.
// acc[i][j][e] <-> voxel m0 + wm*64 + i*16 + fr, channel n0 + wn*32 + j*16 + 4*fq + e.  Must be reached by the whole workgroup
// after the K-loop's last barrier (it reuses the staging LDS when a.gn_part is set).
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[4][2], const ConvArgs& a, int64_t M, int64_t m0, int n0, int wm, int wn, int fr,
                                              int fq, int tid, unsigned char* smem, int64_t tile128 = -1, const float4* pre_bias = nullptr,
                                              const float4* pre_resid = nullptr) {   // pre_*: this lane's bias [NT] / residual [MT][NT] values, loaded earlier
    constexpr int BM = 128, BN = 64, MT = 4, NT = 2;
    float cs[NT][4], cq[NT][4];                 // per-lane channel sums over this lane's MT voxels (GroupNorm partials)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[j][e] = 0.f; cq[j][e] = 0.f; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int64_t m = m0 + wm * (BM / 2) + i * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + 4 * fq;
            if (n >= a.Cout) continue;
            const float4 b = pre_bias ? pre_bias[j] : *reinterpret_cast<const float4*>(a.bias + n);
            f32x4 v = acc[i][j];
            float4 o = make_float4(v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w);
            if (a.resid) {
                const float4 r = pre_resid ? pre_resid[i * NT + j] : *reinterpret_cast<const float4*>(a.resid + m * a.Cout + n);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            if (a.out16) *reinterpret_cast<bf16x4*>(a.out16 + m * a.Cout + n) = pack4(o.x, o.y, o.z, o.w);
            else *reinterpret_cast<float4*>(a.out + m * a.Cout + n) = o;
            cs[j][0] += o.x; cs[j][1] += o.y; cs[j][2] += o.z; cs[j][3] += o.w;
            cq[j][0] += o.x * o.x; cq[j][1] += o.y * o.y; cq[j][2] += o.z * o.z; cq[j][3] += o.w * o.w;
        }
    }
    if (a.gn_part) {
        // fixed-shape reduction (bit-reproducible): 16 voxel lanes by butterfly, the two voxel halves (wm) and the channels of a
        // group in index order.  Host contract: full tiles (M % 128 == 0, Cout % 64 == 0), the K-loop's last barrier has passed.
        float2* sc = reinterpret_cast<float2*>(smem);                       // [wm][64 channels of this n-tile]
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the 16 voxel lanes of one fq share a DPP row: four row-local adds (quad swaps, half-mirror, mirror) leave the row's total in every
                // lane at VALU rate (the ds_bpermute butterfly this replaces was 2.3 us of a 24 us tile: 128 LDS-crossbar ops and their waits)
                float s1 = cs[j][e], s2 = cq[j][e];
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xf, 0xf, true));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0xB1, 0xf, 0xf, true));
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x4E, 0xf, 0xf, true));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x4E, 0xf, 0xf, true));
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x141, 0xf, 0xf, true));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x141, 0xf, 0xf, true));
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x140, 0xf, 0xf, true));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x140, 0xf, 0xf, true));
                if (fr == 0) sc[wm * 64 + wn * 32 + j * 16 + 4 * fq + e] = make_float2(s1, s2);
            }
        __syncthreads();
        const int cpg = a.Cout / 32, gpt = 64 / cpg;                         // channels per group, groups in this 64-channel tile
        if (tid < gpt) {
            double su = 0.0, sq = 0.0;
            for (int w = 0; w < 2; ++w)
                for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { su += (double)sc[w * 64 + c].x; sq += (double)sc[w * 64 + c].y; }
            double* o = a.gn_part + ((tile128 >= 0 ? tile128 : (int64_t)blockIdx.y) * 32 + n0 / cpg + tid) * 2;
            o[0] = su; o[1] = sq;
        }
    }
}

__global__ __launch_bounds__(256) void conv3d_igemm_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 64, BK = 64;
    constexpr int MT = BM / 32, NT = BN / 32, PA = BM / 32, PB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
    bf16x8* sA = reinterpret_cast<bf16x8*>(smem);
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem + 2 * BM * BK * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)a.B * a.OD * a.OH * a.OW;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3, schunk = tid & 7;

    // per staged row: base voxel coordinates of the receptive field
    int rb[PA], rd[PA], rh[PA], rw[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int64_t m = m0 + srow + 32 * p;
        if (m >= M) m = M - 1;
        const int ow = (int)(m % a.OW);
        int64_t r = m / a.OW;
        const int oh = (int)(r % a.OH); r /= a.OH;
        const int od = (int)(r % a.OD);
        rb[p] = (int)(r / a.OD);
        rd[p] = od * a.stride - a.pad; rh[p] = oh * a.stride - a.pad; rw[p] = ow * a.stride - a.pad;
    }
    const bf16* gB[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        int r = n0 + srow + 32 * p;
        r = r < a.Cout ? r : a.Cout - 1;
        gB[p] = a.w + (int64_t)r * 27 * a.Cin + schunk * 8;
    }
    const int cpk = a.Cin / BK;                 // K-steps per tap
    const int nk = 27 * cpk;
    bf16x8 rA[PA], rB[PB];
    auto load_tile = [&](int kt) {
        const int tap = kt / cpk, c0 = (kt - tap * cpk) * BK;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int id = rd[p] + kd, ih = rh[p] + kh, iw = rw[p] + kw;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
            if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
                v = *reinterpret_cast<const bf16x8*>(a.in + ((((int64_t)rb[p] * a.ID + id) * a.IH + ih) * a.IW + iw) * a.Cin + c0 + schunk * 8);
            rA[p] = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) rB[p] = *reinterpret_cast<const bf16x8*>(gB[p] + (int64_t)kt * BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int r = srow + 32 * p;
            sA[(buf * BM + r) * 8 + (schunk ^ (r & 7))] = rA[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int r = srow + 32 * p;
            sB[(buf * BN + r) * 8 + (schunk ^ (r & 7))] = rB[p];
        }
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int k_begin = (int)((int64_t)nk * blockIdx.z / gridDim.z), k_end = (int)((int64_t)nk * (blockIdx.z + 1) / gridDim.z);
    load_tile(k_begin);
    store_tile(k_begin & 1);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = k_begin; kt < k_end; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < k_end) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MT], fb[NT];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * (BM / 2) + i * 16 + fr;
                fa[i] = sA[(buf * BM + r) * 8 + (chunk ^ (r & 7))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                fb[j] = sB[(buf * BN + r) * 8 + (chunk ^ (r & 7))];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < k_end) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (a.split_part) {                          // raw partial sums of this k-range (bias / residual are added by the reduce pass)
        float* part = a.split_part + (int64_t)blockIdx.z * M * a.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int64_t m = m0 + wm * (BM / 2) + i * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + 4 * fq;
                if (n >= a.Cout) continue;
                *reinterpret_cast<float4*>(part + m * a.Cout + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    conv_epilogue(acc, a, M, m0, n0, wm, wn, fr, fq, tid, smem);
}
// out = bias + sum_z part[z] (+ resid), z in order; one float4 per thread
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float* __restrict__ part, int splits, int64_t MC, int Cout,
                                                                const float* __restrict__ bias, const float* __restrict__ resid,
                                                                float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MC) return;
    const float4 b = *reinterpret_cast<const float4*>(bias + (int)(i % Cout));
    float4 acc = *reinterpret_cast<const float4*>(part + i);
    for (int z = 1; z < splits; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(part + (int64_t)z * MC + i);
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    if (resid) {
        const float4 r = *reinterpret_cast<const float4*>(resid + i);
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    *reinterpret_cast<float4*>(out + i) = acc;
}

// Stride-1, pad-1 variant that stages whole input LINES.  A tile is 128 consecutive output voxels = L = 128/OW complete w-lines
// (OW = 2^lw in {8, 16, 32}); for a fixed (kd, kh) the three kw taps read the same L input lines shifted by one voxel, so the
// lines are staged ONCE with a one-voxel apron (L x (OW+2) rows of 128 B) and the taps index into them: a third of the gathers of
// conv3d_igemm_kernel, whose full-resolution launches run at the L2 -> CU rate of those gathers.  Same MFMA fragment layout
// and epilogue.  Host contract: stride 1, pad 1, M % 128 == 0, Cin % 64 == 0.
__global__ __launch_bounds__(256) void conv3d_line_kernel(ConvArgs a, int lw) {
    constexpr int BM = 128, BN = 64, BK = 64, MT = 4, NT = 2, PB = 2, RE_MAX = 160, PAX = RE_MAX / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RE_MAX * 128 + 2 * BN * 128];
    bf16x8* sA = reinterpret_cast<bf16x8*>(smem);                         // [2][RE_MAX rows][8 chunks]
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem + 2 * RE_MAX * 128);      // [2][64 couts][8 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)a.B * a.OD * a.OH * a.OW;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3, schunk = tid & 7;
    const int OW = 1 << lw, EW = OW + 2, RE = (BM >> lw) * EW;
    // staged row e = srow + 32p -> (line l, apron column we): the line's (b, od, oh) and the input column iw = we - 1
    int eb[PAX], ed[PAX], eh[PAX], ewi[PAX];
    const int64_t ln0 = m0 >> lw;
#pragma unroll
    for (int p = 0; p < PAX; ++p) {
        const int e = srow + 32 * p;
        const int l = e / EW;
        ewi[p] = e < RE ? e - l * EW - 1 : -2;                               // -2: no such row (never valid)
        int64_t ln = ln0 + l;
        const int64_t nlines = (int64_t)a.B * a.OD * a.OH;
        if (ln >= nlines) ln = nlines - 1;
        eh[p] = (int)(ln % a.OH);
        const int64_t r = ln / a.OH;
        ed[p] = (int)(r % a.OD);
        eb[p] = (int)(r / a.OD);
    }
    const bf16* gB[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        int r = n0 + srow + 32 * p;
        r = r < a.Cout ? r : a.Cout - 1;
        gB[p] = a.w + (int64_t)r * 27 * a.Cin + schunk * 8;
    }
    const int cpk = a.Cin / BK;
    const int nsteps = 27 * cpk;                                             // step s = (pair * cpk + c) * 3 + kw, pair = kd*3 + kh
    // Global loads run TWO k-steps ahead of their first use (a k-step is only 16 MFMAs per wave, far less than one memory
    // latency): weights in two register sets that alternate by step parity, input lines loaded at kw = 0 of the previous
    // (kd, kh) pair and written to LDS at its kw = 2.
    bf16x8 rA[PAX], rB[2][PB];
    auto load_A = [&](int pc) {                                              // pc = pair * cpk + c
        const int pair = pc / cpk, c0 = (pc - pair * cpk) * BK;
        const int kd = pair / 3, kh = pair - 3 * kd;
#pragma unroll
        for (int p = 0; p < PAX; ++p) {
            const int id = ed[p] + kd - 1, ih = eh[p] + kh - 1, iw = ewi[p];
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
            if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
                v = *reinterpret_cast<const bf16x8*>(a.in + ((((int64_t)eb[p] * a.ID + id) * a.IH + ih) * a.IW + iw) * a.Cin + c0 + schunk * 8);
            rA[p] = v;
        }
    };
    auto store_A = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PAX; ++p) {
            const int e = srow + 32 * p;
            if (e < RE) sA[(buf * RE_MAX + e) * 8 + (schunk ^ (e & 7))] = rA[p];
        }
    };
    auto load_B = [&](int s, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        const int pc = s / 3, kw = s - 3 * pc;
        const int pair = pc / cpk, c = pc - pair * cpk;
        const int kt = (pair * 3 + kw) * cpk + c;                            // weight K index: tap-major, then the Cin chunk
#pragma unroll
        for (int p = 0; p < PB; ++p) rB[par][p] = *reinterpret_cast<const bf16x8*>(gB[p] + (int64_t)kt * BK);
    };
    auto store_B = [&](int buf, auto PAR) {
        constexpr int par = decltype(PAR)::value;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int r = srow + 32 * p;
            sB[(buf * BN + r) * 8 + (schunk ^ (r & 7))] = rB[par][p];
        }
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    int ebase[MT];                                                           // staged row of this lane's voxel at kw = 0
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * (BM / 2) + i * 16 + fr;
        ebase[i] = (r >> lw) * EW + (r & (OW - 1));
    }
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    const int npc = 9 * cpk;
    load_A(0); load_B(0, P0{});
    store_A(0); store_B(0, P0{});
    if (nsteps > 1) load_B(1, P1{});
    if (npc > 1) load_A(1);
    __syncthreads();
    // one k-step; KW = s % 3, PAR = s & 1 (compile-time so that the register sets are not indexed dynamically)
    auto step = [&](int pc, auto KWC, auto PARC) {
        constexpr int kw = decltype(KWC)::value, par = decltype(PARC)::value;
        const int s = 3 * pc + kw;
        const int abuf = pc & 1;
        if (s + 2 < nsteps) load_B(s + 2, PARC);                             // set `par` held B(s), already in LDS
        if (kw == 0 && pc >= 1 && pc + 1 < npc) load_A(pc + 1);               // (pc = 0: issued in the prologue)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MT], fb[NT];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int e = ebase[i] + kw;
                fa[i] = sA[(abuf * RE_MAX + e) * 8 + (chunk ^ (e & 7))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                fb[j] = sB[(par * BN + r) * 8 + (chunk ^ (r & 7))];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_B(par ^ 1, std::integral_constant<int, par ^ 1>{});   // B(s+1), loaded one step ago
        if (kw == 2 && pc + 1 < npc) store_A(abuf ^ 1);                       // lines of the next pair, loaded at its kw = 0
        __syncthreads();
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    for (int pc = 0; pc < npc; pc += 2) {                                    // s = 3 pc + kw: parity (pc + kw) & 1
        step(pc, K0{}, P0{}); step(pc, K1{}, P1{}); step(pc, K2{}, P0{});
        if (pc + 1 < npc) { step(pc + 1, K0{}, P1{}); step(pc + 1, K1{}, P0{}); step(pc + 1, K2{}, P1{}); }
    }
    conv_epilogue(acc, a, M, m0, n0, wm, wn, fr, fq, tid, smem);
}

// Plane-staged form for the 64-channel levels: a workgroup of 8 waves takes 256 output voxels = L = 256 / OW whole w-lines of ONE (b, d) plane
// and stages everything those voxels ever read - the three d-planes x (L + 2) h-lines x (OW + 2) columns, <= 1 024 rows of 128 B = 128 KiB -
// ONCE; the 27 taps then only index into that image, and the only traffic of the main loop is the weights of one (kd, kh) pair at a time
// (3 taps, 24 KiB, from L2).  conv3d_line_kernel re-stages its lines for each of the nine (kd, kh) pairs (360 KB per 256 voxels instead of
// 130 KB) and, worse, WAITS for them: with one or two k-steps of prefetch in registers a step lasts as long as a trip to memory - 3.6 us per
// pair for 0.7 us of MFMA work, 18 GB/s per CU, whatever the inner loop looks like (a 3-taps-per-barrier form of that kernel with
// software-pipelined fragment reads measured exactly the same 515 us per full-resolution convolution of 4 samples).  Here a tile pays the trip
// once.  Measured (radar-condition encode at B = 8, eight full-resolution launches): 7.72 -> 7.06 ms, 515 -> ~455 us per launch; with the
// staging loads redirected to one address (compile-time variant) 7.05 ms, without the epilogue 6.06 ms - what is left of a 28 us tile is ~7 us of
// prologue (16 rows of address arithmetic, loads and LDS stores per thread, one exposed trip to memory), ~12 us of main loop (6.6 us of MFMA work:
// two barriers and one exposed fragment-read latency per pair, one workgroup per CU) and 8.5 us of epilogue (residual read, GroupNorm partial
// sums, fp32 stores) that nothing overlaps.  Host contract: stride 1, pad 1, Cin = 64, Cout % 64 == 0, OW in {16, 32}, OH % L == 0 (a tile never leaves its plane), M % 256 == 0.
// LDS (dynamic): 128 KiB of lines + 24 KiB of weights; the epilogue's scratch reuses the weights' 24 KiB.
__global__ __launch_bounds__(512) void conv3d_plane_kernel(ConvArgs a, int lw) {
    constexpr int BN = 64, MT = 4, NT = 2, RE_MAX = 1024, PAX = RE_MAX / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    bf16x8* sA = reinterpret_cast<bf16x8*>(smem3);                        // [RE rows][8 chunks], row e = ((kd * (L + 2)) + l) * (OW + 2) + we
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem3 + RE_MAX * 128);         // [3 taps][64 couts][8 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)a.B * a.OD * a.OH * a.OW;
    const int64_t m0 = (int64_t)blockIdx.y * 256;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3, schunk = tid & 7;
    const int OW = 1 << lw, EW = OW + 2, L = 256 >> lw, LH = L + 2, PL = LH * EW, RE = 3 * PL;
    const int64_t ln0 = m0 >> lw;
    const int oh0 = (int)(ln0 % a.OH);
    const int64_t pl = ln0 / a.OH;
    const int od = (int)(pl % a.OD), b = (int)(pl / a.OD);
    int rw = n0 + srow;
    rw = rw < a.Cout ? rw : a.Cout - 1;
    const bf16* gW = a.w + (int64_t)rw * 27 * 64 + schunk * 8;               // (Cin = 64: tap t of output channel rw at element t * 64)
    // the weights run THREE pairs ahead of their use in a register ring (12 VGPRs per pair): with one pair of prefetch every pair waited for its
    // weights - all 256 CUs ask the L2 for the same 24 KiB at the same moment, ~3 us a trip - and a tile took 30 us for 6.6 us of MFMA work
    bf16x8 rB[3][3];
    auto loadW = [&](int pair, auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rB[sl][kw] = *reinterpret_cast<const bf16x8*>(gW + (pair * 3 + kw) * 64);
    };
    auto storeW = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) sB[(kw * BN + srow) * 8 + (schunk ^ (srow & 7))] = rB[sl][kw];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    {   // the tile's whole input image, once: unconditional loads from clamped addresses, zeros selected at the LDS store
        bf16x8 rA[PAX];
        unsigned rvalid = 0;
#pragma unroll
        for (int p = 0; p < PAX; ++p) {
            const int e = srow + 64 * p;
            const int kd = e / PL, rem = e - kd * PL;
            const int l = rem / EW, we = rem - l * EW;
            const int id = od + kd - 1, ih = oh0 - 1 + l, iw = we - 1;
            const bool ok = e < RE && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
            const int64_t off = ok ? ((((int64_t)b * a.ID + id) * a.IH + ih) * a.IW + iw) * 64 + schunk * 8 : (int64_t)schunk * 8;
            rA[p] = *reinterpret_cast<const bf16x8*>(a.in + off);
            rvalid |= ok ? (1u << p) : 0u;
        }
        loadW(0, S0{});
        loadW(1, S1{});
        loadW(2, S2{});
        bf16x8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
#pragma unroll
        for (int p = 0; p < PAX; ++p) {
            const int e = srow + 64 * p;
            if (e < RE) sA[e * 8 + (schunk ^ (e & 7))] = (rvalid >> p) & 1 ? rA[p] : zero;
        }
        storeW(S0{});
        loadW(3, S0{});
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    int ebase[MT];                                                           // staged row of this lane's voxel for tap (0, 0, 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        ebase[i] = (r >> lw) * EW + (r & (OW - 1));
    }
    // fragments of one tap (both 32-deep halves), double-buffered: tap kw + 1 is read under the MFMAs of tap kw
    bf16x8 fa[2][2][MT], fb[2][2][NT];
    auto rd = [&](int poff, int kw, auto BUF) {
        constexpr int bb = decltype(BUF)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int e = ebase[i] + poff + kw;
                fa[bb][kk][i] = sA[e * 8 + (chunk ^ (e & 7))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                fb[bb][kk][j] = sB[(kw * BN + r) * 8 + (chunk ^ (r & 7))];
            }
        }
    };
    auto mm = [&](auto BUF) {
        constexpr int bb = decltype(BUF)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bb][kk][j], fa[bb][kk][i], acc[i][j], 0, 0, 0);
    };
    using F0 = std::integral_constant<int, 0>;
    using F1 = std::integral_constant<int, 1>;
    __syncthreads();
    auto step = [&](auto PAIR) {
        constexpr int pair = decltype(PAIR)::value;
        constexpr int kd = pair / 3, kh = pair - 3 * kd;
        const int poff = (kd * LH + kh) * EW;
        rd(poff, 0, F0{});
        __builtin_amdgcn_sched_barrier(0);
        rd(poff, 1, F1{});
        mm(F0{});
#pragma unroll
        for (int g = 0; g < 12; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(poff, 2, F0{});
        mm(F1{});
#pragma unroll
        for (int g = 0; g < 12; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(F0{});
        __syncthreads();                                                      // every wave has read this pair's weights
        if constexpr (pair + 1 < 9) {
            using SL = std::integral_constant<int, (pair + 1) % 3>;
            storeW(SL{});                                                     // pair + 1, loaded three pairs ago
            if constexpr (pair + 4 < 9) loadW(pair + 4, SL{});
            __syncthreads();
        }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    // what the epilogue adds - bias and fp32 residual of this lane's outputs - is fetched under the last three pairs (one workgroup per CU:
    // nothing else would hide that trip to memory; M % 256 == 0 and Cout % 64 == 0, so every index exists)
    float4 pb[NT], pr[MT * NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) pb[j] = *reinterpret_cast<const float4*>(a.bias + n0 + wn * (BN / 2) + j * 16 + 4 * fq);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            pr[i * NT + j] = a.resid ? *reinterpret_cast<const float4*>(a.resid + (m0 + wm * 64 + i * 16 + fr) * a.Cout + n0 + wn * (BN / 2) + j * 16 + 4 * fq)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    // the epilogue is the 128-voxel one, once per half of the tile (waves 0-3 / 4-7); its scratch lies in the weights' area
    conv_epilogue(acc, a, M, m0 + (wm >> 1) * 128, n0, wm & 1, wn, fr, fq, tid & 255, smem3 + RE_MAX * 128 + (wm >> 1) * 2048,
                  2 * (int64_t)blockIdx.y + (wm >> 1), pb, pr);
}

// The plane-staged form made persistent along d: a workgroup walks `ds` consecutive d-planes of one (b, 256-voxel h-block) column with a rolling window of
// three input planes in LDS (plane id lives in slot id mod 3).  Per output tile it stages ONE new plane instead of three - loaded into registers at the start
// of the tile, stored over the plane that leaves the window as soon as the kd = 0 pairs have read it - the weights' register ring runs on across tiles (pair (p + 4) mod 9 is
// fetched under pair p, whatever tile that belongs to), and the next tile's first fragment reads follow this tile's output stores directly: the prologue
// of conv3d_plane_kernel (address arithmetic, one exposed trip to memory, 16 LDS stores per thread) is paid once per `ds` tiles.
// grid.y = columns x segments: column = (b, oh0 / L), segment = ds planes.  Host contract: conv3d_plane_kernel's, and OD % ds == 0.
__global__ __launch_bounds__(512) void conv3d_pplane_kernel(ConvArgs a, int lw, int ds) {
    constexpr int BN = 64, MT = 4, NT = 2, RE_MAX = 1024, PPX = 6;                 // a plane image is <= 340 rows: 6 staging rows per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    bf16x8* sA = reinterpret_cast<bf16x8*>(smem3);                        // [3 slots][PL rows][8 chunks]; swizzle by the absolute row index
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem3 + RE_MAX * 128);         // [3 taps][64 couts][8 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)a.B * a.OD * a.OH * a.OW;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3, schunk = tid & 7;
    const int OW = 1 << lw, EW = OW + 2, L = 256 >> lw, LH = L + 2, PL = LH * EW;
    const int nseg = a.OD / ds, nhb = a.OH / L;
    const int seg = blockIdx.y % nseg, col = blockIdx.y / nseg;
    const int hb = col % nhb, b = col / nhb;
    const int oh0 = hb * L, d_lo = seg * ds;
    // this thread's staging rows of a plane image: row e = l * EW + we <-> input (ih = oh0 - 1 + l, iw = we - 1); the same for every plane
    int rowoff[PPX];                                                         // element offset inside a plane, or -1
#pragma unroll
    for (int p = 0; p < PPX; ++p) {
        const int e = srow + 64 * p;
        const int l = e / EW, we = e - l * EW;
        const int ih = oh0 - 1 + l, iw = we - 1;
        rowoff[p] = (e < PL && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW) ? (ih * a.IW + iw) * 64 + schunk * 8 : -1;
    }
    const int64_t plane_elems = (int64_t)a.IH * a.IW * 64;
    const bf16* inb = a.in + (int64_t)b * a.ID * plane_elems;
    bf16x8 zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    auto loadP = [&](int id, bf16x8 (&r)[PPX]) {                              // plane id (may lie outside the volume: zeros at the store)
        const bool inside = (unsigned)id < (unsigned)a.ID;
        const bf16* pb_ = inb + (int64_t)(inside ? id : 0) * plane_elems;
#pragma unroll
        for (int p = 0; p < PPX; ++p) r[p] = *reinterpret_cast<const bf16x8*>(pb_ + (rowoff[p] >= 0 ? rowoff[p] : schunk * 8));
    };
    auto storeP = [&](int id, const bf16x8 (&r)[PPX]) {
        const bool inside = (unsigned)id < (unsigned)a.ID;
        const int slot = (id + 3) % 3;
#pragma unroll
        for (int p = 0; p < PPX; ++p) {
            const int e = srow + 64 * p;
            if (e < PL) { const int ea = slot * PL + e; sA[ea * 8 + (schunk ^ (ea & 7))] = (inside && rowoff[p] >= 0) ? r[p] : zero; }
        }
    };
    int rw = n0 + srow;
    rw = rw < a.Cout ? rw : a.Cout - 1;
    const bf16* gW = a.w + (int64_t)rw * 27 * 64 + schunk * 8;
    bf16x8 rB[3][3];
    auto loadW = [&](int pair, auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rB[sl][kw] = *reinterpret_cast<const bf16x8*>(gW + (pair * 3 + kw) * 64);
    };
    auto storeW = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) sB[(kw * BN + srow) * 8 + (schunk ^ (srow & 7))] = rB[sl][kw];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    bf16x8 rP[PPX];
    {   // the first window: planes d_lo - 1, d_lo, d_lo + 1
        bf16x8 r0[PPX], r1[PPX];
        loadP(d_lo - 1, r0); loadP(d_lo, r1); loadP(d_lo + 1, rP);
        loadW(0, S0{}); loadW(1, S1{}); loadW(2, S2{});
        storeP(d_lo - 1, r0); storeP(d_lo, r1); storeP(d_lo + 1, rP);
        storeW(S0{});
        loadW(3, S0{});
    }
    const int fr = lane & 15, fq = lane >> 4;
    int ebase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        ebase[i] = (r >> lw) * EW + (r & (OW - 1));
    }
    float4 pb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) pb[j] = *reinterpret_cast<const float4*>(a.bias + n0 + wn * (BN / 2) + j * 16 + 4 * fq);
    f32x4 acc[MT][NT];
    bf16x8 fa[2][2][MT], fb[2][2][NT];
    auto rd = [&](int poff, int kw, auto BUF) {
        constexpr int bb = decltype(BUF)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int e = ebase[i] + poff + kw;
                fa[bb][kk][i] = sA[e * 8 + (chunk ^ (e & 7))];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                fb[bb][kk][j] = sB[(kw * BN + r) * 8 + (chunk ^ (r & 7))];
            }
        }
    };
    auto mm = [&](auto BUF) {
        constexpr int bb = decltype(BUF)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bb][kk][j], fa[bb][kk][i], acc[i][j], 0, 0, 0);
    };
    using F0 = std::integral_constant<int, 0>;
    using F1 = std::integral_constant<int, 1>;
    __syncthreads();
    for (int d = d_lo; d < d_lo + ds; ++d) {
        const bool more = d + 1 < d_lo + ds;
        if (more) loadP(d + 2, rP);                                          // the plane the NEXT tile adds to the window: a whole tile ahead
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int64_t m0 = (((int64_t)b * a.OD + d) * a.OH + oh0) * OW;
        const int sl0 = (d + 2) % 3;                                          // slot of plane d - 1; d -> (sl0 + 1) % 3, d + 1 -> (sl0 + 2) % 3
        float4 pr[MT * NT];
        auto step = [&](auto PAIR) {
            constexpr int pair = decltype(PAIR)::value;
            constexpr int kd = pair / 3, kh = pair - 3 * kd;
            int slot = sl0 + kd;
            slot = slot >= 3 ? slot - 3 : slot;
            const int poff = slot * PL + kh * EW;
            rd(poff, 0, F0{});
            __builtin_amdgcn_sched_barrier(0);
            rd(poff, 1, F1{});
            mm(F0{});
#pragma unroll
            for (int g = 0; g < 12; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd(poff, 2, F0{});
            mm(F1{});
#pragma unroll
            for (int g = 0; g < 12; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(F0{});
            __syncthreads();                                                  // every wave has read this pair's weights
            if constexpr (pair < 8) {
                using SL = std::integral_constant<int, (pair + 1) % 3>;
                storeW(SL{});                                                 // pair + 1, loaded three pairs ago
                loadW((pair + 4) % 9, SL{});                                  // (wraps into the next tile: the weights do not depend on the tile)
                if constexpr (pair == 2) { if (more) storeP(d + 2, rP); }     // over plane d - 1, whose last readers were pairs 0-2 (frees rP's registers)
                __syncthreads();
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                pr[i * NT + j] = a.resid ? *reinterpret_cast<const float4*>(a.resid + (m0 + wm * 64 + i * 16 + fr) * a.Cout + n0 + wn * (BN / 2) + j * 16 + 4 * fq)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        // (after pair 8's barrier nobody reads the weights' area: the epilogue's scratch lies there)
        conv_epilogue(acc, a, M, m0 + (wm >> 1) * 128, n0, wm & 1, wn, fr, fq, tid & 255, smem3 + RE_MAX * 128 + (wm >> 1) * 2048, m0 / 128 + (wm >> 1), pb, pr);
        if (more) {
            __syncthreads();                                                  // the epilogue's scratch has been read
            storeW(S0{});                                                     // pair 0 of the next tile (fetched under pair 5)
            loadW(3, S0{});
            __syncthreads();
        }
    }
}

// engine choice for one convolution (RALD_CONV_LINE=0 keeps the per-tap gather kernel everywhere: A/B switch)
static void launch_conv(const ConvArgs& a, hipStream_t st) {
    static const bool line = RALD_PROBE_ENV("RALD_CONV_LINE", 1) != 0;
    const int64_t M = (int64_t)a.B * a.OD * a.OH * a.OW;
    const dim3 grid(cdiv(a.Cout, 64), (unsigned)((M + 127) / 128));
    const bool pow2 = a.OW == 8 || a.OW == 16 || a.OW == 32;
    if (line && a.stride == 1 && a.pad == 1 && pow2 && M % 128 == 0 && a.Cin % 64 == 0 && a.OD == a.ID && a.OH == a.IH && a.OW == a.IW) {
        const int lw = a.OW == 8 ? 3 : a.OW == 16 ? 4 : 5;
        static const bool plane = RALD_PROBE_ENV("RALD_CONV_PLANE", 1) != 0;
        static const int pds = RALD_PROBE_ENV("RALD_CONV_PPLANE_DS", 8);      // planes per persistent workgroup (0 = the one-tile kernel)
        const int Lp = 256 >> lw;
        if (plane && pds > 0 && a.Cin == 64 && a.Cout % 64 == 0 && lw >= 4 && a.OH % Lp == 0 && a.OD % pds == 0 && 3 * (Lp + 2) * (a.OW + 2) <= 1024 &&
            (int64_t)a.B * (a.OH / Lp) * (a.OD / pds) >= 256) {
            constexpr int LDSP = 1024 * 128 + 3 * 64 * 128;
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3d_pplane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSP); attr_set = true; }
            hipLaunchKernelGGL(conv3d_pplane_kernel, dim3(cdiv(a.Cout, 64), (unsigned)(a.B * (a.OH / Lp) * (a.OD / pds))), dim3(512), LDSP, st, a, lw, pds);
        } else
        if (plane && a.Cin == 64 && a.Cout % 64 == 0 && lw >= 4 && a.OH % Lp == 0 && M % 256 == 0 && M / 256 >= 256 && 3 * (Lp + 2) * (a.OW + 2) <= 1024) {
            constexpr int LDSP = 1024 * 128 + 3 * 64 * 128;                  // the tile's input image staged once + one (kd, kh) pair of weights
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3d_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSP); attr_set = true; }
            hipLaunchKernelGGL(conv3d_plane_kernel, dim3(cdiv(a.Cout, 64), (unsigned)(M / 256)), dim3(512), LDSP, st, a, lw);
        } else
        hipLaunchKernelGGL(conv3d_line_kernel, grid, dim3(256), 0, st, a, lw);
    } else {
        hipLaunchKernelGGL(conv3d_igemm_kernel, grid, dim3(256), 0, st, a);
    }
}

// tokens[b][t][c] = z[b][t][:].Wp[c][:] + bp[c] + r_emb[r][c] + a_emb[a][c] + e_emb[e][c], t = (r*A + a)*E + e
__global__ void radar_token_kernel(const float* __restrict__ z, const float* __restrict__ Wp, const float* __restrict__ bp,
                                   const float* __restrict__ re, const float* __restrict__ ae, const float* __restrict__ ee,
                                   float* __restrict__ tok, int R, int A, int E, int zc, int C) {
    const int t = blockIdx.x;                 // token within the batch
    const int b = blockIdx.y;
    const int e = t % E, aa = (t / E) % A, r = t / (E * A);
    const float* zz = z + ((int64_t)b * R * A * E + t) * zc;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = bp[c];
        for (int k = 0; k < zc; ++k) acc += zz[k] * Wp[c * zc + k];
        tok[((int64_t)b * R * A * E + t) * C + c] = acc + re[r * C + c] + ae[aa * C + c] + ee[e * C + c];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// ---- decoder helpers --------------------------------------------------------------------------------------------------------
// Upsample (:18-27): nearest-neighbour x2 of the fp32 trunk [B][D][H][W][C] (channels last) -> bf16 [B][2D][2H][2W][C], the input of
// the 3x3x3 convolution that follows.  One thread per 4 channels of an OUTPUT voxel.
__global__ __launch_bounds__(256) void upsample2_cast_kernel(const float* __restrict__ x, bf16* __restrict__ y, int B, int D, int H, int W, int C) {
    const int64_t quads = (int64_t)B * 8 * D * H * W * (C / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % (C / 4));
        int64_t v = i / (C / 4);
        const int ow = (int)(v % (2 * W)); v /= 2 * W;
        const int oh = (int)(v % (2 * H)); v /= 2 * H;
        const int od = (int)(v % (2 * D));
        const int b = (int)(v / (2 * D));
        const float4 s = *reinterpret_cast<const float4*>(x + ((((int64_t)b * D + od / 2) * H + oh / 2) * W + ow / 2) * C + 4 * c4);
        *reinterpret_cast<bf16x4*>(y + i * 4) = pack4(s.x, s.y, s.z, s.w);
    }
}
// z [rows][zc] fp32 -> bf16 [rows][64], zero beyond zc (the decoder's conv_in reads 64-channel rows)
__global__ __launch_bounds__(256) void pad_cast64_kernel(const float* __restrict__ z, bf16* __restrict__ y, int64_t rows, int zc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * 64) return;
    const int c = (int)(i & 63);
    y[i] = (bf16)(c < zc ? z[(i >> 6) * zc + c] : 0.f);
}

struct RadarEncoder::Impl {
    int ch = 64, zc = 16, R = 128, A = 64, E = 32, token_ch = 512, cin = 1;
    DeviceArena* arena = nullptr;
    enum Kind { CONV3, CONV1, VEC, CONVIN };
    struct Tensor { Kind kind; int cout, cin; void* ptr = nullptr; bool loaded = false; int cin_pad = 0, cout_pad = 0; };   // pads: decoder conv_in / conv_out
    std::map<std::string, Tensor> tensors;
    // tokeniser
    float *w_tok = nullptr, *b_tok = nullptr, *emb_r = nullptr, *emb_a = nullptr, *emb_e = nullptr;
    bool tok_loaded[5] = {false, false, false, false, false};
    // workspace (for `sub` samples at a time)
    int sub = 0;
    float *f0 = nullptr, *f1 = nullptr, *f2 = nullptr, *z = nullptr, *tok = nullptr, *sbuf = nullptr;
    bf16 *n16 = nullptr, *q16 = nullptr, *k16 = nullptr, *vt16 = nullptr, *p16 = nullptr, *o16 = nullptr;
    double* stats = nullptr;
    int gn_part_blocks = 0;
    double* cpart = nullptr;            // per-tile GroupNorm partials written by the conv epilogue
    size_t cpart_bytes = 0;
    float* spart = nullptr;             // split-K partial accumulators of the small-level convolutions
    size_t spart_bytes = 0;
    const float* fused_src = nullptr;   // activation whose partials sit in cpart (nullptr: none)
    int fused_B = 0, fused_S = 0, fused_C = 0;
    int tok_batch = 0;

    void add(const std::string& name, Kind k, int cout, int cin) { Tensor t{k, cout, cin}; t.cin_pad = cin; t.cout_pad = cout; tensors[name] = t; }
    // 3x3x3 convolution whose channel counts are padded up for the implicit-GEMM kernel (Cin to 64, Cout to 4): zero weights / bias in the pad
    void add_conv3_padded(const std::string& n, int cout, int cin, int cout_pad, int cin_pad) {
        add_conv3(n, cout, cin);
        tensors[n + ".weight"].cin_pad = cin_pad; tensors[n + ".weight"].cout_pad = cout_pad; tensors[n + ".bias"].cout_pad = cout_pad;
    }
    bool is_decoder = false;
    int out_ch = 2;
    void add_conv3(const std::string& n, int cout, int cin) { add(n + ".weight", CONV3, cout, cin); add(n + ".bias", VEC, cout, 0); }
    void add_conv1(const std::string& n, int cout, int cin) { add(n + ".weight", CONV1, cout, cin); add(n + ".bias", VEC, cout, 0); }
    void add_norm(const std::string& n, int c) { add(n + ".weight", VEC, c, 0); add(n + ".bias", VEC, c, 0); }
    void add_res(const std::string& n, int cin, int cout) {
        add_norm(n + ".norm1", cin); add_conv3(n + ".conv1", cout, cin); add_norm(n + ".norm2", cout); add_conv3(n + ".conv2", cout, cout);
        if (cin != cout) add_conv1(n + ".nin_shortcut", cout, cin);
    }
    void add_attn(const std::string& n, int c) {
        add_norm(n + ".norm", c);
        for (const char* p : {".q", ".k", ".v", ".proj_out"}) add_conv1(n + p, c, c);
    }
    template <class T> T* P(const std::string& n) { return (T*)tensors.at(n).ptr; }

    int ensure_ws(int nsub);
    int run_conv(const bf16* in, const std::string& name, const float* resid, float* out, int B, int ID, int IH, int IW, int cin,
                 int cout, int stride, int pad, hipStream_t st);
    int gn(const float* x, const std::string& name, bf16* y, int B, int S, int C, bool swish, hipStream_t st);
    int resblock(float*& x, float*& t1, float*& t2, const std::string& name, int B, int Dd, int Hh, int Ww, int cin, int cout, hipStream_t st);
    int attnblock(float* x, const std::string& name, int B, int S, int C, hipStream_t st);
    int forward(const float* cube, int cube_ch, int B, float* zout, hipStream_t st);
    int decode(const float* z, int B, float* out, hipStream_t st);     // Decoder.forward (:333-359), nsub samples
};

static const int kChMult[5] = {1, 1, 2, 2, 4};

static int radar_alloc_tensors(RadarEncoder::Impl& m, DeviceArena* arena) {
    for (auto& kv : m.tensors) {
        auto& t = kv.second;
        size_t bytes = 0;
        switch (t.kind) {
            case RadarEncoder::Impl::CONV3: bytes = (size_t)t.cout_pad * 27 * t.cin_pad * 2; break;
            case RadarEncoder::Impl::CONV1: bytes = (size_t)t.cout * t.cin * 2; break;
            case RadarEncoder::Impl::VEC: bytes = (size_t)t.cout_pad * 4; break;
            case RadarEncoder::Impl::CONVIN: bytes = (size_t)t.cout * t.cin * 27 * 4; break;
        }
        t.ptr = arena->alloc(bytes < 64 ? 64 : bytes, true);       // >= 16 floats so float4 bias reads of a 16-ch tail stay in bounds; pads stay zero
        RALD_CHECK(t.ptr, "radar encoder: weight allocation failed");
    }
    return 0;
}

int RadarEncoder::create(int ch, int z_ch, int R, int A, int E, int token_ch, DeviceArena* arena, int in_channels) {
    RALD_CHECK(in_channels == 1 || in_channels == 2, "radar encoder: in_channels must be 1 or 2");
    RALD_CHECK(ch % 64 == 0 && ch >= 64, "radar encoder: hidden channels must be a multiple of 64");
    RALD_CHECK(R % 16 == 0 && A % 16 == 0 && E % 16 == 0, "radar encoder: cube dims must be multiples of 16");
    RALD_CHECK(z_ch % 4 == 0 && z_ch <= 64, "radar encoder: z channels must be a multiple of 4, <= 64");
    impl = new Impl();
    Impl& m = *impl;
    m.ch = ch; m.zc = z_ch; m.R = R; m.A = A; m.E = E; m.token_ch = token_ch; m.arena = arena; m.cin = in_channels;
    m.add("conv_in.weight", Impl::CONVIN, ch, in_channels);
    m.add("conv_in.bias", Impl::VEC, ch, 0);
    int block_in = ch;
    for (int l = 0; l < 5; ++l) {
        const int block_out = ch * kChMult[l];
        block_in = ch * (l == 0 ? 1 : kChMult[l - 1]);
        for (int b = 0; b < 2; ++b) {
            m.add_res("down." + std::to_string(l) + ".block." + std::to_string(b), block_in, block_out);
            block_in = block_out;
        }
        if (l == 4)
            for (int b = 0; b < 2; ++b) m.add_attn("down.4.attn." + std::to_string(b), block_in);
        if (l != 4) m.add_conv3("down." + std::to_string(l) + ".downsample.conv", block_in, block_in);
    }
    m.add_res("mid.block_1", block_in, block_in);
    m.add_attn("mid.attn_1", block_in);
    m.add_res("mid.block_2", block_in, block_in);
    m.add_norm("norm_out", block_in);
    m.add_conv3("conv_out", z_ch, block_in);
    RALD_TRY(radar_alloc_tensors(m, arena));
    const int nt_r = R / 16, nt_a = A / 16, nt_e = E / 16;
    m.w_tok = (float*)arena->alloc((size_t)token_ch * z_ch * 4, true);
    m.b_tok = (float*)arena->alloc((size_t)token_ch * 4, true);
    m.emb_r = (float*)arena->alloc((size_t)nt_r * token_ch * 4, true);
    m.emb_a = (float*)arena->alloc((size_t)nt_a * token_ch * 4, true);
    m.emb_e = (float*)arena->alloc((size_t)nt_e * token_ch * 4, true);
    RALD_CHECK(m.w_tok && m.b_tok && m.emb_r && m.emb_a && m.emb_e, "radar encoder: allocation failed");
    return 0;
}

// Decoder of the RadarAutoencoder (models_radar_encoder.py:243-359): conv_in z_ch -> 4ch, mid (ResnetBlock, AttnBlock, ResnetBlock), five
// levels from the coarsest up - three ResnetBlocks each, nearest-neighbour x2 + conv3x3 between levels -, GroupNorm + swish + conv_out.
// Same kernels as the encoder; conv_in's z_ch inputs are zero-padded to 64 channels and conv_out's out_ch outputs to 4.
int RadarEncoder::create_decoder(int ch, int z_ch, int out_ch, int R, int A, int E, DeviceArena* arena) {
    RALD_CHECK(ch % 64 == 0 && ch >= 64, "radar decoder: hidden channels must be a multiple of 64");
    RALD_CHECK(R % 16 == 0 && A % 16 == 0 && E % 16 == 0, "radar decoder: cube dims must be multiples of 16");
    RALD_CHECK(z_ch >= 1 && z_ch <= 64 && out_ch >= 1 && out_ch <= 4, "radar decoder: z channels <= 64, output channels <= 4");
    impl = new Impl();
    Impl& m = *impl;
    m.ch = ch; m.zc = z_ch; m.R = R; m.A = A; m.E = E; m.arena = arena; m.is_decoder = true; m.out_ch = out_ch;
    int block_in = ch * kChMult[4];
    m.add_conv3_padded("conv_in", block_in, z_ch, block_in, 64);
    m.add_res("mid.block_1", block_in, block_in);
    m.add_attn("mid.attn_1", block_in);
    m.add_res("mid.block_2", block_in, block_in);
    for (int l = 4; l >= 0; --l) {
        const int block_out = ch * kChMult[l];
        for (int b = 0; b < 3; ++b) {
            m.add_res("up." + std::to_string(l) + ".block." + std::to_string(b), block_in, block_out);
            block_in = block_out;
        }
        if (l != 0) m.add_conv3("up." + std::to_string(l) + ".upsample.conv", block_in, block_in);
    }
    m.add_norm("norm_out", block_in);
    m.add_conv3_padded("conv_out", out_ch, block_in, 4, block_in);
    RALD_TRY(radar_alloc_tensors(m, arena));
    return 0;
}

bool RadarEncoder::all_loaded(std::string* missing) const {
    if (!impl) return false;
    for (const auto& kv : impl->tensors)
        if (!kv.second.loaded) { if (missing) *missing = kv.first; return false; }
    return true;
}

void RadarEncoder::expected_keys(const std::string& prefix, std::set<std::string>& out) const {
    if (!impl) return;
    for (const auto& kv : impl->tensors) out.insert(prefix + kv.first);
}

int RadarEncoder::load_weight(const std::string& name, const float* data, int64_t nelem, Stager& st) {
    RALD_CHECK(impl, "radar encoder: not created");
    auto it = impl->tensors.find(name);
    RALD_CHECK(it != impl->tensors.end(), "radar encoder: unknown key '" + name + "'");
    auto& t = it->second;
    switch (t.kind) {
        case Impl::VEC:
            RALD_CHECK(nelem == t.cout, "radar encoder: size mismatch for '" + name + "'");
            RALD_TRY(st.to_f32(data, (float*)t.ptr, 1, t.cout, t.cout, nullptr));      // (a padded tail stays zero)
            break;
        case Impl::CONVIN:
            RALD_CHECK(nelem == (int64_t)t.cout * t.cin * 27, "radar encoder: size mismatch for '" + name + "'");
            RALD_TRY(st.to_f32(data, (float*)t.ptr, t.cout, t.cin * 27, t.cin * 27, nullptr));
            break;
        case Impl::CONV1:
            RALD_CHECK(nelem == (int64_t)t.cout * t.cin, "radar encoder: size mismatch for '" + name + "'");
            RALD_TRY(st.to_bf16(data, (bf16*)t.ptr, t.cout, t.cin, t.cin, nullptr));
            break;
        case Impl::CONV3: {
            RALD_CHECK(nelem == (int64_t)t.cout * t.cin * 27, "radar encoder: size mismatch for '" + name + "'");
            // torch [Cout][Cin][3][3][3] -> [Cout][tap][Cin] (host permute, then bf16 upload)
            std::vector<float> src((size_t)nelem), dst((size_t)nelem);
            RALD_HIP(hipMemcpy(src.data(), data, (size_t)nelem * 4, hipMemcpyDefault));
            for (int co = 0; co < t.cout; ++co)
                for (int ci = 0; ci < t.cin; ++ci)
                    for (int tp = 0; tp < 27; ++tp)
                        dst[((size_t)co * 27 + tp) * t.cin + ci] = src[((size_t)co * t.cin + ci) * 27 + tp];
            RALD_TRY(st.to_bf16(dst.data(), (bf16*)t.ptr, t.cout * 27, t.cin, t.cin_pad, nullptr));   // rows of cin_pad, zero beyond cin
            break;
        }
    }
    t.loaded = true;
    return 0;
}

int RadarEncoder::load_token_weight(const std::string& name, const float* data, int64_t nelem, Stager& st) {
    RALD_CHECK(impl, "radar encoder: not created");
    Impl& m = *impl;
    const int C = m.token_ch;
    if (name == "radar_token_project.weight") { RALD_CHECK(nelem == (int64_t)C * m.zc, "size mismatch: " + name); m.tok_loaded[0] = true; return st.to_f32(data, m.w_tok, C, m.zc, m.zc, nullptr); }
    if (name == "radar_token_project.bias") { RALD_CHECK(nelem == C, "size mismatch: " + name); m.tok_loaded[1] = true; return st.to_f32(data, m.b_tok, 1, C, C, nullptr); }
    if (name == "radar_r_emb.weight") { RALD_CHECK(nelem == (int64_t)(m.R / 16) * C, "size mismatch: " + name); m.tok_loaded[2] = true; return st.to_f32(data, m.emb_r, m.R / 16, C, C, nullptr); }
    if (name == "radar_a_emb.weight") { RALD_CHECK(nelem == (int64_t)(m.A / 16) * C, "size mismatch: " + name); m.tok_loaded[3] = true; return st.to_f32(data, m.emb_a, m.A / 16, C, C, nullptr); }
    if (name == "radar_e_emb.weight") { RALD_CHECK(nelem == (int64_t)(m.E / 16) * C, "size mismatch: " + name); m.tok_loaded[4] = true; return st.to_f32(data, m.emb_e, m.E / 16, C, C, nullptr); }
    RALD_CHECK(false, "radar encoder: unknown key '" + name + "'");
}

int RadarEncoder::Impl::ensure_ws(int nsub) {
    if (nsub <= sub) return 0;
    RALD_HIP(hipDeviceSynchronize());
    for (void* p : {(void*)f0, (void*)f1, (void*)f2, (void*)n16, (void*)stats, (void*)cpart, (void*)spart, (void*)q16, (void*)k16, (void*)vt16, (void*)p16, (void*)o16, (void*)sbuf})
        if (p) arena->release(p);
    const size_t vox = (size_t)R * A * E;
    const size_t act = (size_t)nsub * vox * ch;                 // level-0 activation (the largest)
    const int ntok = (R / 16) * (A / 16) * (E / 16), cl = ch * 4;
    f0 = (float*)arena->alloc(act * 4, true);
    f1 = (float*)arena->alloc(act * 4, true);
    f2 = (float*)arena->alloc(act * 4, true);
    n16 = (bf16*)arena->alloc(act * 2, true);
    gn_part_blocks = gn_blocks((int)vox);
    stats = (double*)arena->alloc((size_t)nsub * 64 * (1 + gn_part_blocks) * 8, true);   // final {sum, sumsq} + per-block partials
    spart_bytes = (size_t)8 << 20;                              // splits * tiles <= 256 tiles of 128 x 64 fp32
    spart = (float*)arena->alloc(spart_bytes, true);
    cpart_bytes = (size_t)nsub * (vox / 128) * 64 * 8;
    cpart = (double*)arena->alloc(cpart_bytes, true);
    q16 = (bf16*)arena->alloc((size_t)nsub * ntok * cl * 2, true);
    k16 = (bf16*)arena->alloc((size_t)nsub * ntok * cl * 2, true);
    vt16 = (bf16*)arena->alloc((size_t)nsub * cl * ntok * 2, true);
    p16 = (bf16*)arena->alloc((size_t)nsub * ntok * ntok * 2, true);
    o16 = (bf16*)arena->alloc((size_t)nsub * ntok * cl * 2, true);
    sbuf = (float*)arena->alloc((size_t)nsub * ntok * ntok * 4, true);
    RALD_CHECK(f0 && f1 && f2 && n16 && stats && cpart && spart && q16 && k16 && vt16 && p16 && o16 && sbuf, "radar encoder: workspace allocation failed");
    sub = nsub;
    return 0;
}

int RadarEncoder::Impl::gn(const float* x, const std::string& name, bf16* y, int B, int S, int C, bool swish, hipStream_t st) {
    RALD_CHECK(gn_blocks(S) <= gn_part_blocks, "radar encoder: GroupNorm partial buffer too small");
    static const bool fuse = RALD_PROBE_ENV("RALD_GN_FUSE", 1) != 0;   // A/B switch
    if (fuse && x == fused_src && B == fused_B && S == fused_S && C == fused_C) {
        // the convolution that produced x left per-tile partials: no statistics pass over x
        hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(1024), 0, st, cpart, stats, S / 128);
    } else {
        double* part = stats + (size_t)B * 64;
        hipLaunchKernelGGL(gn_stats_kernel, dim3(gn_blocks(S), B), dim3(256), 0, st, x, part, S, C, GN_VPB);
        hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(1024), 0, st, part, stats, gn_blocks(S));
    }
    const int64_t quads = (int64_t)S * C / 4;
    const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, st, x, stats, P<float>(name + ".weight"), P<float>(name + ".bias"), y,
                       S, C, 1e-6f, swish ? 1 : 0);
    RALD_HIP(hipGetLastError());
    return 0;
}

int RadarEncoder::Impl::run_conv(const bf16* in, const std::string& name, const float* resid, float* out, int B, int ID, int IH,
                                 int IW, int cin, int cout, int stride, int pad, hipStream_t st) {
    ConvArgs a;
    a.in = in; a.w = P<bf16>(name + ".weight"); a.bias = P<float>(name + ".bias"); a.resid = resid; a.out = out;
    a.B = B; a.ID = ID; a.IH = IH; a.IW = IW; a.Cin = cin; a.Cout = cout; a.stride = stride; a.pad = pad;
    a.OD = ID / stride; a.OH = IH / stride; a.OW = IW / stride;
    RALD_CHECK(cin % 64 == 0 && cout % 4 == 0, "conv3d: Cin must be a multiple of 64 and Cout of 4");
    const int64_t M = (int64_t)B * a.OD * a.OH * a.OW;
    const int So = a.OD * a.OH * a.OW;
    // few tiles x long K (the 512- and 64-voxel levels: 8-64 workgroups looping over 54-108 k-steps): split K over gridDim.z
    static const bool split_ok = RALD_PROBE_ENV("RALD_CONV_SPLITK", 1) != 0;   // A/B switch
    const int64_t tiles = (int64_t)cdiv(cout, 64) * ((M + 127) / 128);
    const int nk = 27 * (cin / 64);
    const bool line_engine = stride == 1 && pad == 1 && (a.OW == 8 || a.OW == 16 || a.OW == 32) && M % 128 == 0;
    if (split_ok && !line_engine && tiles <= 64 && nk >= 12 && cout % 4 == 0) {
        int splits = (int)(256 / tiles);
        if (splits > 16) splits = 16;
        if (splits > nk / 3) splits = nk / 3;
        if (splits >= 2 && (size_t)splits * M * cout * 4 <= spart_bytes) {
            if (out == fused_src) fused_src = nullptr;         // no GroupNorm partials from this path
            a.split_part = spart;
            hipLaunchKernelGGL(conv3d_igemm_kernel, dim3(cdiv(cout, 64), (unsigned)((M + 127) / 128), splits), dim3(256), 0, st, a);
            const int64_t MC = M * cout;
            hipLaunchKernelGGL(conv_split_reduce_kernel, dim3((unsigned)((MC / 4 + 255) / 256)), dim3(256), 0, st, spart, splits, MC, cout,
                               a.bias, resid, out);
            RALD_HIP(hipGetLastError());
            return 0;
        }
    }
    if (out == f0 || out == f1 || out == f2) {
        if (So % 128 == 0 && cout % 64 == 0 && (cout == 64 || cout == 128 || cout == 256) && (size_t)B * (So / 128) * 64 * 8 <= cpart_bytes) {
            a.gn_part = cpart;
            fused_src = out; fused_B = B; fused_S = So; fused_C = cout;
        } else if (out == fused_src) fused_src = nullptr;      // this buffer is being overwritten without new partials
    }
    (void)M;
    launch_conv(a, st);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ResnetBlock.forward (:82-100): x -> x + conv2(swish(GN(conv1(swish(GN(x)))))), 1x1 shortcut if channels change.
int RadarEncoder::Impl::resblock(float*& x, float*& t1, float*& t2, const std::string& name, int B, int Dd, int Hh, int Ww, int cin,
                                 int cout, hipStream_t st) {
    const int S = Dd * Hh * Ww;
    RALD_TRY(gn(x, name + ".norm1", n16, B, S, cin, true, st));
    RALD_TRY(run_conv(n16, name + ".conv1", nullptr, t1, B, Dd, Hh, Ww, cin, cout, 1, 1, st));
    const float* res = x;
    if (cin != cout) {
        RALD_TRY(cast_f32_bf16(x, n16, (int64_t)B * S * cin, st));
        GemmArgs g = gemm_args(n16, cin, P<bf16>(name + ".nin_shortcut.weight"), cin, t2, cout, P<float>(name + ".nin_shortcut.bias"), B * S, cout, cin);
        RALD_TRY(gemm_nt(g, EPI_F32, st));
        if (t2 == fused_src) fused_src = nullptr;
        res = t2;
    }
    RALD_TRY(gn(t1, name + ".norm2", n16, B, S, cout, true, st));
    float* dst = (cin != cout) ? t2 : x;                       // in-place residual add (same index read then written)
    RALD_TRY(run_conv(n16, name + ".conv2", res, dst, B, Dd, Hh, Ww, cout, cout, 1, 1, st));
    if (cin != cout) std::swap(x, t2);
    return 0;
}

// AttnBlock.forward (:112-135): single head over the S tokens, scale C^-1/2, residual.
int RadarEncoder::Impl::attnblock(float* x, const std::string& name, int B, int S, int C, hipStream_t st) {
    RALD_CHECK(S % 64 == 0, "radar attention: token count must be a multiple of 64");
    RALD_TRY(gn(x, name + ".norm", n16, B, S, C, false, st));
    GemmArgs q = gemm_args(n16, C, P<bf16>(name + ".q.weight"), C, q16, C, P<float>(name + ".q.bias"), B * S, C, C);
    RALD_TRY(gemm_nt(q, EPI_BF16, st));
    GemmArgs k = gemm_args(n16, C, P<bf16>(name + ".k.weight"), C, k16, C, P<float>(name + ".k.bias"), B * S, C, C);
    RALD_TRY(gemm_nt(k, EPI_BF16, st));
    GemmArgs v = gemm_args(P<bf16>(name + ".v.weight"), C, n16, C, vt16, S, nullptr, C, S, C);   // V^T (bias folded below)
    v.batch = B; v.strideB = (int64_t)S * C; v.strideC = (int64_t)C * S;
    RALD_TRY(gemm_nt(v, EPI_BF16, st));
    GemmArgs s = gemm_args(q16, C, k16, C, sbuf, S, nullptr, S, S, C);
    s.batch = B; s.strideA = (int64_t)S * C; s.strideB = (int64_t)S * C; s.strideC = (int64_t)S * S; s.alpha = 1.0f / sqrtf((float)C);
    RALD_TRY(gemm_nt(s, EPI_F32, st));
    RALD_TRY(softmax_rows(sbuf, S, p16, S, B * S, S, st));
    // rows of P sum to 1, so P.(v + 1.b_v^T) = P.v + b_v: the v bias becomes the epilogue bias
    GemmArgs pv = gemm_args(p16, S, vt16, S, o16, C, P<float>(name + ".v.bias"), S, C, S);
    pv.batch = B; pv.strideA = (int64_t)S * S; pv.strideB = (int64_t)C * S; pv.strideC = (int64_t)S * C;
    RALD_TRY(gemm_nt(pv, EPI_BF16, st));
    GemmArgs o = gemm_args(o16, C, P<bf16>(name + ".proj_out.weight"), C, x, C, P<float>(name + ".proj_out.bias"), B * S, C, C);
    RALD_TRY(gemm_nt(o, EPI_RESID, st));
    if (x == fused_src) fused_src = nullptr;                  // x changed in place: its conv partials are stale
    return 0;
}

int RadarEncoder::Impl::forward(const float* cube, int cube_ch, int B, float* zout, hipStream_t st) {
    for (const auto& kv : tensors) RALD_CHECK(kv.second.loaded, "radar encoder: missing key '" + kv.first + "'");
    RALD_TRY(ensure_ws(B));
    float *x = f0, *t1 = f1, *t2 = f2;
    fused_src = nullptr;
    int Dd = R, Hh = A, Ww = E;
    {
        const int groups = ch / 8, vpb = 256 / groups;
        const int64_t nvox = (int64_t)B * Dd * Hh * Ww;
        RALD_CHECK(cube_ch >= cin, "radar encoder: cube has fewer channels than conv_in expects");
        hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((nvox + vpb - 1) / vpb)), dim3(256), cin * 27 * ch * 4, st, cube, cube_ch, cin,
                           P<float>("conv_in.weight"), P<float>("conv_in.bias"), x, B, Dd, Hh, Ww, ch);
        RALD_HIP(hipGetLastError());
    }
    int cin = ch;
    for (int l = 0; l < 5; ++l) {
        const int cout = ch * kChMult[l];
        for (int b = 0; b < 2; ++b) {
            const std::string nm = "down." + std::to_string(l) + ".block." + std::to_string(b);
            RALD_TRY(resblock(x, t1, t2, nm, B, Dd, Hh, Ww, cin, cout, st));
            cin = cout;
            if (l == 4) RALD_TRY(attnblock(x, "down.4.attn." + std::to_string(b), B, Dd * Hh * Ww, cin, st));
        }
        if (l != 4) {
            // Downsample (:37-41): input is the raw trunk (no norm/activation), cast to bf16
            RALD_TRY(cast_f32_bf16(x, n16, (int64_t)B * Dd * Hh * Ww * cin, st));
            RALD_TRY(run_conv(n16, "down." + std::to_string(l) + ".downsample.conv", nullptr, t1, B, Dd, Hh, Ww, cin, cin, 2, 0, st));
            std::swap(x, t1);
            Dd /= 2; Hh /= 2; Ww /= 2;
        }
    }
    RALD_TRY(resblock(x, t1, t2, "mid.block_1", B, Dd, Hh, Ww, cin, cin, st));
    RALD_TRY(attnblock(x, "mid.attn_1", B, Dd * Hh * Ww, cin, st));
    RALD_TRY(resblock(x, t1, t2, "mid.block_2", B, Dd, Hh, Ww, cin, cin, st));
    RALD_TRY(gn(x, "norm_out", n16, B, Dd * Hh * Ww, cin, true, st));
    RALD_TRY(run_conv(n16, "conv_out", nullptr, zout, B, Dd, Hh, Ww, cin, zc, 1, 1, st));
    return 0;
}

int RadarEncoder::Impl::decode(const float* zin, int B, float* out, hipStream_t st) {
    for (const auto& kv : tensors) RALD_CHECK(kv.second.loaded, "radar decoder: missing key '" + kv.first + "'");
    RALD_TRY(ensure_ws(B));
    float *x = f0, *t1 = f1, *t2 = f2;
    fused_src = nullptr;
    int Dd = R / 16, Hh = A / 16, Ww = E / 16;
    int cin = ch * kChMult[4];
    {
        const int64_t rows = (int64_t)B * Dd * Hh * Ww;
        hipLaunchKernelGGL(pad_cast64_kernel, dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, st, zin, n16, rows, zc);
        RALD_HIP(hipGetLastError());
        RALD_TRY(run_conv(n16, "conv_in", nullptr, x, B, Dd, Hh, Ww, 64, cin, 1, 1, st));
    }
    RALD_TRY(resblock(x, t1, t2, "mid.block_1", B, Dd, Hh, Ww, cin, cin, st));
    RALD_TRY(attnblock(x, "mid.attn_1", B, Dd * Hh * Ww, cin, st));
    RALD_TRY(resblock(x, t1, t2, "mid.block_2", B, Dd, Hh, Ww, cin, cin, st));
    for (int l = 4; l >= 0; --l) {
        const int cout = ch * kChMult[l];
        for (int b = 0; b < 3; ++b) {
            RALD_TRY(resblock(x, t1, t2, "up." + std::to_string(l) + ".block." + std::to_string(b), B, Dd, Hh, Ww, cin, cout, st));
            cin = cout;
        }
        if (l != 0) {
            const int64_t quads = (int64_t)B * 8 * Dd * Hh * Ww * (cin / 4);
            const unsigned blocks = (unsigned)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
            hipLaunchKernelGGL(upsample2_cast_kernel, dim3(blocks), dim3(256), 0, st, x, n16, B, Dd, Hh, Ww, cin);
            RALD_HIP(hipGetLastError());
            Dd *= 2; Hh *= 2; Ww *= 2;
            RALD_TRY(run_conv(n16, "up." + std::to_string(l) + ".upsample.conv", nullptr, t1, B, Dd, Hh, Ww, cin, cin, 1, 1, st));
            std::swap(x, t1);
        }
    }
    RALD_TRY(gn(x, "norm_out", n16, B, Dd * Hh * Ww, cin, true, st));
    RALD_TRY(run_conv(n16, "conv_out", nullptr, out, B, Dd, Hh, Ww, cin, 4, 1, 1, st));       // [B][R][A][E][4]: channels 0..out_ch-1 are the reconstruction
    return 0;
}

// z [B][R/16][A/16][E/16][z_ch] fp32 (the layout RadarAutoencoder._encode returns) -> pred4 [B][R][A][E][4] fp32 (out_ch real channels + zero pad)
int RadarEncoder::decode(const float* z, int B, float* pred4, hipStream_t st) {
    RALD_CHECK(impl && impl->is_decoder && z && pred4 && B >= 1, "radar decoder: bad arguments");
    Impl& m = *impl;
    const int SUB = 4;
    const size_t zs = (size_t)(m.R / 16) * (m.A / 16) * (m.E / 16) * m.zc, os = (size_t)m.R * m.A * m.E * 4;
    for (int b0 = 0; b0 < B; b0 += SUB) {
        const int nb = B - b0 < SUB ? B - b0 : SUB;
        RALD_TRY(m.decode(z + (size_t)b0 * zs, nb, pred4 + (size_t)b0 * os, st));
    }
    return 0;
}

int RadarEncoder::encode(const float* cube, int cube_ch, int B, float** zptr, hipStream_t st) {
    RALD_CHECK(impl && cube && B >= 1, "radar encoder: bad arguments");
    Impl& m = *impl;
    const int ntok = (m.R / 16) * (m.A / 16) * (m.E / 16);
    if (B > m.tok_batch) {
        RALD_HIP(hipDeviceSynchronize());
        if (m.z) m.arena->release(m.z);
        if (m.tok) m.arena->release(m.tok);
        m.z = (float*)m.arena->alloc((size_t)B * ntok * m.zc * 4, true);
        m.tok = (float*)m.arena->alloc((size_t)B * ntok * m.token_ch * 4, true);
        RALD_CHECK(m.z && m.tok, "radar encoder: allocation failed");
        m.tok_batch = B;
    }
    const int SUB = 4;                                         // samples per pass: bounds the 67 MB/sample fp32 trunk buffers
    const size_t cube_stride = (size_t)m.R * m.A * m.E * cube_ch;
    for (int b0 = 0; b0 < B; b0 += SUB) {
        const int nb = B - b0 < SUB ? B - b0 : SUB;
        RALD_TRY(m.forward(cube + (size_t)b0 * cube_stride, cube_ch, nb, m.z + (size_t)b0 * ntok * m.zc, st));
    }
    *zptr = m.z;
    return 0;
}

int RadarEncoder::tokens(const float* cube, int B, float** tokens_out, hipStream_t st) {
    RALD_CHECK(impl, "radar encoder: not created");
    Impl& m = *impl;
    for (int i = 0; i < 5; ++i) RALD_CHECK(m.tok_loaded[i], "radar encoder: tokeniser weights missing");
    float* z = nullptr;
    RALD_TRY(encode(cube, 2, B, &z, st));                      // intensity channel of the [.,2] cube (:378)
    const int nr = m.R / 16, na = m.A / 16, ne = m.E / 16;
    hipLaunchKernelGGL(radar_token_kernel, dim3(nr * na * ne, B), dim3(256), 0, st, z, m.w_tok, m.b_tok, m.emb_r, m.emb_a, m.emb_e, m.tok, nr,
                       na, ne, m.zc, m.token_ch);
    RALD_HIP(hipGetLastError());
    *tokens_out = m.tok;
    return 0;
}

RadarEncoder::~RadarEncoder() { delete impl; }

// ---- op-level launchers of the kernels above (used by the training path, radar_train.hip / train_encoder.py) ------
int conv3d_igemm(const bf16* in, const bf16* w_packed, const float* bias, const float* resid, float* out, int B, int ID, int IH, int IW,
                 int Cin, int Cout, int stride, int pad, hipStream_t st, bf16* out_bf16) {
    RALD_CHECK(in && w_packed && bias && (out || out_bf16), "conv3d: null pointer");
    RALD_CHECK(!out_bf16 || (!out && !resid && (uintptr_t)out_bf16 % 8 == 0), "conv3d: the bf16 result replaces the fp32 one and takes no residual");
    RALD_CHECK(B > 0 && ID > 0 && IH > 0 && IW > 0 && (stride == 1 || stride == 2), "conv3d: bad geometry");
    RALD_CHECK(Cin % 64 == 0 && Cout % 4 == 0, "conv3d: Cin must be a multiple of 64 and Cout of 4");
    ConvArgs a;
    a.in = in; a.w = w_packed; a.bias = bias; a.resid = resid; a.out = out; a.out16 = out_bf16;
    a.B = B; a.ID = ID; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.pad = pad;
    a.OD = ID / stride; a.OH = IH / stride; a.OW = IW / stride;
    launch_conv(a, st);
    RALD_HIP(hipGetLastError());
    return 0;
}

int groupnorm_fwd(const float* x, const float* gamma, const float* beta, bf16* y, double* stats, int B, int S, int C, int swish,
                  hipStream_t st) {
    RALD_CHECK(x && gamma && beta && y && stats, "groupnorm: null pointer");
    RALD_CHECK(B > 0 && S > 0 && C % 64 == 0 && 256 % (C / 4) == 0, "groupnorm: channel count must be 64, 128 or 256");
    double* part = stats + (size_t)B * 64;                 // caller contract: B*64*(1 + ceil(S/512)) doubles
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gn_blocks(S), B), dim3(256), 0, st, x, part, S, C, GN_VPB);
    hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(1024), 0, st, part, stats, gn_blocks(S));
    const int64_t quads = (int64_t)S * C / 4;
    const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, st, x, stats, gamma, beta, y, S, C, 1e-6f, swish ? 1 : 0);
    RALD_HIP(hipGetLastError());
    return 0;
}

// the normalisation alone from statistics already in hand (the training backward re-creates the activations it did not keep)
int groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, bf16* y, int B, int S, int C, int swish, hipStream_t st) {
    RALD_CHECK(x && gamma && beta && y && stats, "groupnorm_apply: null pointer");
    RALD_CHECK(B > 0 && S > 0 && C % 64 == 0 && 256 % (C / 4) == 0, "groupnorm_apply: channel count must be 64, 128 or 256");
    const int64_t quads = (int64_t)S * C / 4;
    const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, st, x, stats, gamma, beta, y, S, C, 1e-6f, swish ? 1 : 0);
    RALD_HIP(hipGetLastError());
    return 0;
}

int conv_in_fwd(const float* cube, int cube_ch, int Cin, const float* W, const float* bias, float* out, int B, int D, int H, int Wd, int Cout,
                hipStream_t st) {
    RALD_CHECK(cube && W && bias && out && Cout % 8 == 0 && 256 % (Cout / 8) == 0 && cube_ch >= Cin, "conv_in: bad arguments");
    const int groups = Cout / 8, vpb = 256 / groups;
    const int64_t nvox = (int64_t)B * D * H * Wd;
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((nvox + vpb - 1) / vpb)), dim3(256), Cin * 27 * Cout * 4, st, cube, cube_ch, Cin, W, bias, out,
                       B, D, H, Wd, Cout);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
