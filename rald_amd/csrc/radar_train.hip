// Backward-pass building blocks of the radar-spectrum encoder (model/models_radar_encoder.py:29-241; the shipped
// configuration trains it jointly with the denoiser, `unfreeze_radar_enc: true`).  Layout as in radar.hip:
// activations channels-last [b][d][h][w][c], fp32 trunk, bf16 conv inputs.
//   * Conv3d data gradient = the forward implicit-GEMM kernel on dY with flipped, transposed weights
//     (conv_pack_weights with dgrad = 1); the stride-2 Downsample's gradient first spreads dY onto the even
//     positions of a zero grid (zero_insert2) and then runs the same stride-1 kernel with pad 2.
//   * Conv3d weight gradient = one MFMA GEMM per voxel chunk: dW[co][ci*27+t] += dY^T[co][m] . patches^T[ci*27+t][m],
//     with the transposed im2col matrix written by im2col_t (rows in the weight tensor's own order, so the result
//     accumulates straight into the fp32 gradient of the [Cout, Cin, 3, 3, 3] parameter).
//   * GroupNorm(32 groups, eps 1e-6) + swish backward in two streaming passes (group sums, then dx).
#include "common.h"
#include "kernels.h"

namespace rald {

// W [Cout][Cin][27] fp32 (the parameter's layout) -> packed bf16 for conv3d_igemm.
//   forward: out[co][t][ci]  (ci < Cin_pad; zero beyond Cin)
//   dgrad  : out[ci][t][co] = W[co][ci][26 - t]  (co < Cout_pad; zero beyond Cout) - the kernel then maps dY -> dX
__global__ void conv_pack_weights_kernel(const float* __restrict__ W, bf16* __restrict__ out, int Cout, int Cin, int pad_to, int dgrad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int inner = (int)(i % pad_to);
    const int t = (int)((i / pad_to) % 27);
    const int outer = (int)(i / ((int64_t)pad_to * 27));
    float v = 0.f;
    if (!dgrad) { if (inner < Cin) v = W[((int64_t)outer * Cin + inner) * 27 + t]; }
    else { if (inner < Cout) v = W[((int64_t)inner * Cin + outer) * 27 + (26 - t)]; }
    out[i] = (bf16)v;
}
int conv_pack_weights(const float* W, bf16* out, int Cout, int Cin, int pad_to, int dgrad, hipStream_t st) {
    RALD_CHECK(W && out && Cout > 0 && Cin > 0 && pad_to >= (dgrad ? Cout : Cin), "conv_pack_weights: bad arguments");
    const int64_t total = (int64_t)(dgrad ? Cin : Cout) * 27 * pad_to;
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, out, Cout, Cin, pad_to, dgrad, total);
    RALD_HIP(hipGetLastError());
    return 0;
}

// x [M][C] f32 -> bf16 [M][Cpad], zero-filled channels (conv_out has 16 output channels, the kernel wants 64 inputs)
__global__ void pad_channels_kernel(const float* __restrict__ x, bf16* __restrict__ out, int C, int Cpad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    out[i] = (bf16)(c < C ? x[(i / Cpad) * C + c] : 0.f);
}
int pad_channels(const float* x, bf16* out, int64_t M, int C, int Cpad, hipStream_t st) {
    RALD_CHECK(x && out && M > 0 && C > 0 && Cpad >= C, "pad_channels: bad arguments");
    const int64_t total = M * Cpad;
    hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, out, C, Cpad, total);
    RALD_HIP(hipGetLastError());
    return 0;
}

// dY [B][OD][OH][OW][C] f32 -> bf16 [B][2OD][2OH][2OW][C] with dY on the even positions, zeros elsewhere
__global__ void zero_insert2_kernel(const float* __restrict__ dy, bf16* __restrict__ out, int OD, int OH, int OW, int C, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int w = (int)(r % (2 * OW)); r /= 2 * OW;
    const int h = (int)(r % (2 * OH)); r /= 2 * OH;
    const int d = (int)(r % (2 * OD));
    const int64_t b = r / (2 * OD);
    float v = 0.f;
    if (!((w | h | d) & 1)) v = dy[((((b * OD + d / 2) * OH + h / 2) * OW + w / 2)) * C + c];
    out[i] = (bf16)v;
}
int zero_insert2(const float* dy, bf16* out, int B, int OD, int OH, int OW, int C, hipStream_t st) {
    RALD_CHECK(dy && out && B > 0 && OD > 0 && OH > 0 && OW > 0 && C > 0, "zero_insert2: bad arguments");
    const int64_t total = (int64_t)B * 8 * OD * OH * OW * C;
    hipLaunchKernelGGL(zero_insert2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, out, OD, OH, OW, C, total);
    RALD_HIP(hipGetLastError());
    return 0;
}

// Transposed im2col of a chunk of output voxels: out[(ci*27 + t)][j] = x[b][od*s - p + kd][..][ci] for output voxel
// m0 + j (zero outside the volume).  One 64-voxel x 64-channel tile per (blockIdx.x, tap, channel block), through LDS.
__global__ __launch_bounds__(256) void im2col_t_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int ID, int IH, int IW, int C, int OD, int OH,
                                                       int OW, int stride, int pad, int64_t m0, int nchunk, int64_t Mtot) {
    __shared__ bf16 tile[64][72];
    const int j0 = blockIdx.x * 64, tap = blockIdx.y, c0 = blockIdx.z * 64;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    for (int p = 0; p < 2; ++p) {
        const int jr = (threadIdx.x >> 3) + 32 * p, cc = (threadIdx.x & 7) * 8;
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
        const int64_t m = m0 + j0 + jr;
        if (j0 + jr < nchunk && m < Mtot) {
            const int ow = (int)(m % OW);
            int64_t r = m / OW;
            const int oh = (int)(r % OH); r /= OH;
            const int od = (int)(r % OD);
            const int64_t b = r / OD;
            const int id = od * stride - pad + kd, ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
            if ((unsigned)id < (unsigned)ID && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW)
                v = *reinterpret_cast<const bf16x8*>(x + ((((b * ID + id) * IH + ih) * IW + iw)) * C + c0 + cc);
        }
        *reinterpret_cast<bf16x8*>(&tile[jr][cc]) = v;
    }
    __syncthreads();
    for (int p = 0; p < 2; ++p) {
        const int c = (threadIdx.x >> 3) + 32 * p, jr = (threadIdx.x & 7) * 8;
        if (j0 + jr < nchunk) {
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = tile[jr + i][c];
            *reinterpret_cast<bf16x8*>(out + ((int64_t)(c0 + c) * 27 + tap) * nchunk + j0 + jr) = v;
        }
    }
}
int im2col_t(const bf16* x, bf16* out, int B, int ID, int IH, int IW, int C, int stride, int pad, int64_t m0, int nchunk, hipStream_t st) {
    RALD_CHECK(x && out && C % 64 == 0 && nchunk > 0 && nchunk % 8 == 0, "im2col_t: C must be a multiple of 64 and the chunk of 8");
    const int OD = ID / stride, OH = IH / stride, OW = IW / stride;
    const int64_t Mtot = (int64_t)B * OD * OH * OW;
    RALD_CHECK(m0 >= 0 && m0 < Mtot, "im2col_t: chunk start out of range");
    hipLaunchKernelGGL(im2col_t_kernel, dim3(cdiv(nchunk, 64), 27, C / 64), dim3(256), 0, st, x, out, ID, IH, IW, C, OD, OH, OW, stride, pad, m0, nchunk,
                       Mtot);
    RALD_HIP(hipGetLastError());
    return 0;
}

// conv_in (Cin = 1) weight gradient as a row-contracting GEMM (gemm_tn.hip): the 27-neighbourhood of every voxel of the single input channel
// as one bf16 row of 32 (27 taps + 5 zeros) - 134 MB for 2.1 M voxels - then dW [Cout][27] = dy^T . patches.  conv_in_wgrad_kernel below
// (one serial pass per workgroup over its voxels, LDS-staged) takes 1.74 ms for 3.6 GFLOP; this pair ~0.15 ms.
__global__ __launch_bounds__(256) void patches27_kernel(const float* __restrict__ cube, int cube_ch, bf16* __restrict__ out, int D, int H, int Wd, int64_t nvox) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t v = gid >> 2;                                   // 4 threads per voxel, 8 slots each
    const int q = (int)(gid & 3);
    if (v >= nvox) return;
    const int w = (int)(v % Wd);
    int64_t r = v / Wd;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D);
    const int64_t b = r / D;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = 8 * q + i;
        float x = 0.f;
        if (t < 27) {
            const int id = d + t / 9 - 1, ih = h + (t / 3) % 3 - 1, iw = w + t % 3 - 1;
            if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)Wd)
                x = cube[((((b * D + id) * H + ih) * Wd + iw)) * cube_ch];
        }
        o[i] = (bf16)x;
    }
    reinterpret_cast<bf16x8*>(out)[gid] = o;
}
int patches27(const float* cube, int cube_ch, bf16* out, int B, int D, int H, int Wd, hipStream_t st) {
    RALD_CHECK(cube && out && B >= 1 && cube_ch >= 1, "patches27: bad arguments");
    const int64_t nvox = (int64_t)B * D * H * Wd;
    hipLaunchKernelGGL(patches27_kernel, dim3((unsigned)cdiv(nvox * 4, (int64_t)256)), dim3(256), 0, st, cube, cube_ch, out, D, H, Wd, nvox);
    RALD_HIP(hipGetLastError());
    return 0;
}

// conv_in (Cin = 1): dW[co][t] += sum_v dY[v][co] * cube[v + off(t)][0].  A workgroup walks `chunks` tiles of 256
// voxels (neighbourhoods staged in LDS), thread (co, tap group) keeps 7 partial sums, one atomicAdd per output per
// workgroup.  Cout <= 64.
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const float* __restrict__ cube, int cube_ch, const float* __restrict__ dy, int B, int D, int H,
                                                            int Wd, int Cout, int chunks, float* __restrict__ dW) {
    __shared__ float sx[256][28];
    const int64_t nvox = (int64_t)B * D * H * Wd;
    const int co = threadIdx.x & 63, g = threadIdx.x >> 6;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < chunks; ++ch) {
        const int64_t v0 = ((int64_t)blockIdx.x * chunks + ch) * 256;
        if (v0 >= nvox) break;
        __syncthreads();
        {
            const int64_t v = v0 + threadIdx.x;
            int w = 0, h = 0, d = 0;
            int64_t b = 0;
            if (v < nvox) { w = (int)(v % Wd); int64_t r = v / Wd; h = (int)(r % H); r /= H; d = (int)(r % D); b = r / D; }
            for (int t = 0; t < 27; ++t) {
                const int id = d + t / 9 - 1, ih = h + (t / 3) % 3 - 1, iw = w + t % 3 - 1;
                float x = 0.f;
                if (v < nvox && (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)Wd)
                    x = cube[((((b * D + id) * H + ih) * Wd + iw)) * cube_ch];
                sx[threadIdx.x][t] = x;
            }
        }
        __syncthreads();
        if (co < Cout) {
            const int vend = (int)(nvox - v0 < 256 ? nvox - v0 : 256);
            for (int j = 0; j < vend; ++j) {
                const float g_ = dy[(v0 + j) * Cout + co];
#pragma unroll
                for (int q = 0; q < 7; ++q) { const int t = g + 4 * q; if (t < 27) acc[q] += g_ * sx[j][t]; }
            }
        }
    }
    if (co < Cout) {
#pragma unroll
        for (int q = 0; q < 7; ++q) { const int t = g + 4 * q; if (t < 27) atomicAdd(dW + co * 27 + t, acc[q]); }
    }
}
int conv_in_wgrad(const float* cube, int cube_ch, const float* dy, int B, int D, int H, int Wd, int Cout, float* dW, hipStream_t st) {
    RALD_CHECK(cube && dy && dW && Cout <= 64, "conv_in_wgrad: Cout must be <= 64");
    const int64_t nvox = (int64_t)B * D * H * Wd;
    const int64_t tiles = (nvox + 255) / 256;
    const int chunks = (int)((tiles + 2047) / 2048);          // <= 2048 workgroups
    hipLaunchKernelGGL(conv_in_wgrad_kernel, dim3((unsigned)((tiles + chunks - 1) / chunks)), dim3(256), 0, st, cube, cube_ch, dy, B, D, H, Wd, Cout,
                       chunks, dW);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- GroupNorm (+ swish) backward -------------------------------------------------------------------------------
// forward: y = xhat * gamma + beta (xhat over the group's S*cpg elements), a = swish(y) or y.  Given da:
//   pass 1: per workgroup (1 024 voxels of one sample) the partial sums  sum dy*xhat, sum dy  per channel and  sum dy*gamma, sum dy*gamma*xhat
//           per group, reduced in a FIXED order through LDS and stored as that workgroup's row of the scratch table (no atomics);
//   finish: per sample, the rows added in order -> gsum[b][g] (double) and the sample's channel sums;
//   pass 2: dx (+)= rstd * (dy*gamma - gsum0/n - xhat * gsum1/n); its first workgroup adds the samples' channel sums, in order, into dgamma / dbeta.
// Bit-reproducible run to run (round 2's form met in fp32 / fp64 atomics: 2 048 workgroups on the same 128 + 64 addresses at full resolution).
struct GnBwdArgs {
    const float* x; const double* stats; const float* gamma; const float* beta; const float* da;
    float* dx; float* dgamma; float* dbeta; double* gsum;
    int S, C, swish, accumulate; float eps;
    float* part; float* csum; int nblk;                     // scratch: part[b][blk][2C + 64] (pass 1), csum[b][2C] (finish); nblk = workgroups per sample
    int da16;                                               // da holds bf16 (a data-gradient convolution's bf16 result) instead of fp32
    bf16* dxb;                                              // optional: the (accumulated) dx rounded to bf16 - what the convolution gradients read; dx may then be null
};
__device__ __forceinline__ float gn_dy(float yv, float da, int swish) {
    if (!swish) return da;
    const float sg = 1.0f / (1.0f + __expf(-yv));
    return da * sg * (1.0f + yv * (1.0f - sg));
}
__global__ __launch_bounds__(256) void gn_bwd_pass_kernel(GnBwdArgs a, int pass, int vox_per_block) {
    __shared__ float smean[32], srstd[32], sm1[32], sm2[32];
    __shared__ float red[4][256][4];                        // pass 1: every thread's four partial sums of its four channels
    const int b = blockIdx.y, C = a.C, cpg = C / 32, quads = C / 4;
    if (threadIdx.x < 32) {
        const double n = (double)a.S * cpg;
        const double su = a.stats[((int64_t)b * 32 + threadIdx.x) * 2], sq = a.stats[((int64_t)b * 32 + threadIdx.x) * 2 + 1];
        const double mean = su / n, var = sq / n - mean * mean;
        smean[threadIdx.x] = (float)mean;
        srstd[threadIdx.x] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.eps));
        if (pass == 2) {
            sm1[threadIdx.x] = (float)(a.gsum[((int64_t)b * 32 + threadIdx.x) * 2] / n);
            sm2[threadIdx.x] = (float)(a.gsum[((int64_t)b * 32 + threadIdx.x) * 2 + 1] / n);
        }
    }
    if (pass == 2 && blockIdx.x == 0 && b == 0) {           // the samples' channel sums, in order, into the parameter gradients (once per launch)
        for (int t = threadIdx.x; t < 2 * C; t += 256) {
            float sacc = 0.f;
            for (int bb_ = 0; bb_ < (int)gridDim.y; ++bb_) sacc += a.csum[(int64_t)bb_ * 2 * C + t];
            if (t < C) a.dgamma[t] += sacc; else a.dbeta[t - C] += sacc;
        }
    }
    __syncthreads();
    const int q = threadIdx.x % quads, vstep = 256 / quads;
    const int v0 = blockIdx.x * vox_per_block, v1 = min(a.S, v0 + vox_per_block);
    const float4 gm = reinterpret_cast<const float4*>(a.gamma)[q], bt = reinterpret_cast<const float4*>(a.beta)[q];
    const float gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
    float pg[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0}, g2[4] = {0, 0, 0, 0};
    for (int v = v0 + threadIdx.x / quads; v < v1; v += vstep) {
        const int64_t idx = ((int64_t)b * a.S + v) * quads + q;
        const float4 t = reinterpret_cast<const float4*>(a.x)[idx];
        float4 d;
        if (a.da16) { const bf16x4 h = reinterpret_cast<const bf16x4*>(a.da)[idx]; d = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
        else d = reinterpret_cast<const float4*>(a.da)[idx];
        const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (4 * q + j) / cpg;
            const float xh = (xv[j] - smean[g]) * srstd[g];
            const float dy = gn_dy(xh * gg[j] + bb[j], dv[j], a.swish);
            if (pass == 1) { pg[j] += dy * xh; pb[j] += dy; g1[j] += dy * gg[j]; g2[j] += dy * gg[j] * xh; }
            else o[j] = srstd[g] * (dy * gg[j] - sm1[g] - xh * sm2[g]);
        }
        if (pass == 2) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.dx) {
                float4* dst = reinterpret_cast<float4*>(a.dx) + idx;
                if (a.accumulate) r = *dst;
                r.x += o[0]; r.y += o[1]; r.z += o[2]; r.w += o[3];
                *dst = r;
            } else { r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3]; }
            if (a.dxb) reinterpret_cast<bf16x4*>(a.dxb)[idx] = pack4(r.x, r.y, r.z, r.w);
        }
    }
    if (pass == 2) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = pg[j]; red[1][threadIdx.x][j] = pb[j]; red[2][threadIdx.x][j] = g1[j]; red[3][threadIdx.x][j] = g2[j]; }
    __syncthreads();
    // fixed order: the vstep row lanes of a channel, then (group sums) the channels of the group
    float* prow = a.part + ((int64_t)b * a.nblk + blockIdx.x) * (2 * C + 64);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int L = 0; L < vstep; ++L) { s0 += red[0][L * quads + (c >> 2)][c & 3]; s1 += red[1][L * quads + (c >> 2)][c & 3]; }
        prow[c] = s0; prow[C + c] = s1;
    }
    if (threadIdx.x >= 192 && threadIdx.x < 224) {          // (a wave of its own next to the channel sums above)
        const int g = threadIdx.x - 192;
        float s0 = 0.f, s1 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c)
            for (int L = 0; L < vstep; ++L) { s0 += red[2][L * quads + (c >> 2)][c & 3]; s1 += red[3][L * quads + (c >> 2)][c & 3]; }
        prow[2 * C + g] = s0; prow[2 * C + 32 + g] = s1;
    }
}
// per sample: the workgroups' rows added in order
__global__ __launch_bounds__(256) void gn_bwd_finish_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ csum, double* __restrict__ gsum) {
    const int b = blockIdx.x, W = 2 * C + 64;
    for (int t = threadIdx.x; t < W; t += 256) {
        const float* p = part + (int64_t)b * nblk * W + t;
        if (t < 2 * C) {
            float sacc = 0.f;
            for (int k = 0; k < nblk; ++k) sacc += p[(int64_t)k * W];
            csum[(int64_t)b * 2 * C + t] = sacc;
        } else {
            double sacc = 0.0;
            for (int k = 0; k < nblk; ++k) sacc += (double)p[(int64_t)k * W];
            const int g = (t - 2 * C) & 31, which = (t - 2 * C) >> 5;
            gsum[((int64_t)b * 32 + g) * 2 + which] = sacc;
        }
    }
}
int64_t groupnorm_bwd_scratch_bytes(int B, int S, int C) {
    if (B < 1 || S < 1 || C < 64) return 0;
    const int64_t nblk = (S + 1023) / 1024;
    return (int64_t)B * 64 * 8 + ((int64_t)B * nblk * (2 * C + 64) + (int64_t)B * 2 * C) * 4;
}
int groupnorm_bwd(const float* x, const double* stats, const float* gamma, const float* beta, const float* da, float* dx, float* dgamma, float* dbeta,
                  double* gsum_scratch, int B, int S, int C, int swish, int accumulate, hipStream_t st, bf16* dx_bf16, int da_is_bf16) {
    RALD_CHECK(x && stats && gamma && beta && da && (dx || dx_bf16) && dgamma && dbeta && gsum_scratch, "groupnorm_bwd: null pointer");
    RALD_CHECK(dx || !accumulate, "groupnorm_bwd: accumulating needs the fp32 dx");
    RALD_CHECK((uintptr_t)dx_bf16 % 8 == 0, "groupnorm_bwd: the bf16 copy must be 8-byte aligned");
    RALD_CHECK(B > 0 && S > 0 && C % 64 == 0 && C <= 256 && 256 % (C / 4) == 0, "groupnorm_bwd: channel count must be 64, 128 or 256");
    GnBwdArgs a;
    a.x = x; a.stats = stats; a.gamma = gamma; a.beta = beta; a.da = da; a.dx = dx; a.dgamma = dgamma; a.dbeta = dbeta; a.gsum = gsum_scratch;
    a.S = S; a.C = C; a.swish = swish; a.accumulate = accumulate; a.eps = 1e-6f; a.dxb = dx_bf16; a.da16 = da_is_bf16;
    const int vpb = 1024;
    a.nblk = cdiv(S, vpb);
    a.part = reinterpret_cast<float*>(gsum_scratch + (size_t)B * 64);                   // caller contract: groupnorm_bwd_scratch_bytes(B, S, C)
    a.csum = a.part + (size_t)B * a.nblk * (2 * C + 64);
    hipLaunchKernelGGL(gn_bwd_pass_kernel, dim3(a.nblk, B), dim3(256), 0, st, a, 1, vpb);
    hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(B), dim3(256), 0, st, a.part, a.nblk, C, a.csum, gsum_scratch);
    hipLaunchKernelGGL(gn_bwd_pass_kernel, dim3(a.nblk, B), dim3(256), 0, st, a, 2, vpb);
    RALD_HIP(hipGetLastError());
    return 0;
}

// delta[m] = sum_c a[m][c] * b[m][c]  (bf16 rows of any width; the single-head attention of AttnBlock :102-135)
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, int64_t M, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += (float)a[row * C + c] * (float)b[row * C + c];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}
int rowdot(const bf16* a, const bf16* b, int64_t M, int C, float* out, hipStream_t st) {
    RALD_CHECK(a && b && out && M > 0 && C > 0, "rowdot: bad arguments");
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, a, b, M, C, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
