// Streaming query decoder of the set-latent autoencoder (KLAutoEncoder.decode, model/models_ae.py:417-424):
//
//     queries [Q,3] -> PointEmbed -> PreNorm(LN q, LN ctx) 1-head d=dim cross-attention over the M latents
//                   -> to_out -> to_outputs -> one logit per query
//
// ONE kernel, 12 bytes in and 4 bytes out per query; nothing else touches HBM.  What makes that possible is that every
// step between the 51 Fourier features of a query and its attention scores is linear except the query LayerNorm, whose
// statistics are themselves a quadratic form of the features:
//
//     qe      = feat.Wpe^T + b_pe                          (PointEmbed :128-138)
//     qe - mu = feat.Wc^T + b_c                            Wc, b_c = Wpe, b_pe centred over the d outputs (mean is linear)
//     var     = |L.[feat;1]|^2                             L^T.L = [Wc|b_c]^T.[Wc|b_c] / d     (52 x 52, weights only)
//     S[q,l]  = rstd_q * (feat.H_l + h0_l) + hb_l          H = LN_ctx(x).T2, h0 = LN_ctx(x).t20, hb = LN_ctx(x).t2b
//     logit   = softmax_l(S[q,:]) . u + c0                 u = LN_ctx(x).w_fold   (value path folded, see ae.hip)
//
// with T2 = Wk^T.(scale.Wq.diag(g).Wc) etc. computed once per weight load on the host in double (Ae::finalize).  Per sample
// the decoder context is therefore H [M x 51] + three vectors instead of K,V [M x d]: 64 KiB in LDS for M = 512, resident
// for the whole launch, and the per-query score GEMM has K = 64 instead of K = d (+ the d x 64 embedding GEMM + LayerNorm
// that it replaces).  Exact in real arithmetic; numerically the 51-term sums run on v_mfma_f32_32x32x16_f16 (fp16 operands:
// 11-bit mantissas; features are in [-1,1], H carries one power-of-two scale per sample) with fp32 accumulation.
//
// Operand layout (k = slot 0..63 of the K = 64 contraction; MFMA 32x32x16 B-operand: lane (q = lane&31, h = lane>>5) holds
// k = 16s + 8h + j of k-step s): the h = 0 lanes hold sin(p_e) (e = 8s + j < 24), x, y, z, 1, 1, std, std_lo, std; the h = 1
// lanes hold cos(p_e) and zeros - every lane computes its own 32 slots from its query's 3 coordinates in registers, no LDS.
// The two 1-slots multiply h0 split in fp16 hi + lo, the three std-slots (std_q = sqrt(var_q + eps) in hi / lo) multiply
// hb (hi, hi, lo), so that rstd_q * acc = S exactly as above with ~22-bit bias terms.
//
// S^T = H~.f~^T puts the latent index on the accumulator registers and the query on the lane: the softmax over the M
// latents is lane-local (online, lazily rescaled), u is read from LDS as a broadcast.  Bound: v_exp_f32 + the 3 VALU per
// score next to 4 MFMA per 32x32 score tile - not HBM (16 B/query) and not the 2.15 MFLOP/query the reference executes.
#include <cmath>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace rald {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// slot of reference feature f (0..23 sin, 24..47 cos, 48..50 xyz), see the header
static inline int slot_of_feature(int f) {
    if (f < 24) return 16 * (f >> 3) + (f & 7);
    if (f < 48) { const int e = f - 24; return 16 * (e >> 3) + 8 + (e & 7); }
    return 48 + (f - 48);
}
constexpr int SLOT_ONE = 51;      // (s=3, h=0, j=3): 1.0 -> h0_hi ; j=4 (slot 52): 1.0 -> h0_lo
constexpr int SLOT_STD = 53;      // j=5,6,7 (slots 53,54,55): std_hi, std_lo, std_hi -> hb_hi, hb_hi, hb_lo
constexpr int SLOT_U = 63;        // column of the projection that carries u (its feature slot is always zero)

// byte offset of element (row, k) in a [rows][64] fp16 image with 128-byte rows, 16-byte chunks XOR-swizzled by the row
__host__ __device__ static inline int img_off(int row, int k) { return row * 128 + (((k >> 3) ^ (row & 7)) << 4) + (k & 7) * 2; }

// ---------------------------------------------------------------------------------------------------------------
// host: weight-only tables (double precision)
// ---------------------------------------------------------------------------------------------------------------
// t2aug [d][64] fp32 in slot order (cols of features: H; SLOT_ONE: h0; SLOT_STD: hb; SLOT_U: u) and the var factor L
// as a [64][64] fp16 image.  Wq [d][d] (to_q), Wk [d][d] (first half of to_kv), ng / nb [d] (query LayerNorm),
// Wpe [d][51], bpe [d], wfold [d] (value path, ae.hip).
int ae_decode_tables(int d, const float* Wq, const float* Wk, const float* ng, const float* nb, const float* Wpe, const float* bpe,
                     const float* wfold, std::vector<float>& t2aug, std::vector<unsigned short>& l_img) {
    const int F = 51, FA = 52;
    std::vector<double> Wc((size_t)d * FA);                 // [c][f], column 51 = centred bias
    for (int f = 0; f < FA; ++f) {
        double mu = 0.0;
        for (int c = 0; c < d; ++c) mu += f < F ? (double)Wpe[(size_t)c * F + f] : (double)bpe[c];
        mu /= d;
        for (int c = 0; c < d; ++c) Wc[(size_t)c * FA + f] = (f < F ? (double)Wpe[(size_t)c * F + f] : (double)bpe[c]) - mu;
    }
    // ---- var factor: modified Gram-Schmidt on [Wc|bc] / sqrt(d): R [52][52] upper triangular, R^T.R = Gram matrix
    std::vector<double> Qm((size_t)d * FA), R((size_t)FA * FA, 0.0);
    const double inv_sd = 1.0 / std::sqrt((double)d);
    for (int j = 0; j < FA; ++j) {
        std::vector<double> v(d);
        double n0 = 0.0;
        for (int c = 0; c < d; ++c) { v[c] = Wc[(size_t)c * FA + j] * inv_sd; n0 += v[c] * v[c]; }
        for (int i = 0; i < j; ++i) {
            double r = 0.0;
            for (int c = 0; c < d; ++c) r += Qm[(size_t)c * FA + i] * v[c];
            R[(size_t)i * FA + j] = r;
            for (int c = 0; c < d; ++c) v[c] -= r * Qm[(size_t)c * FA + i];
        }
        double nn = 0.0;
        for (int c = 0; c < d; ++c) nn += v[c] * v[c];
        const double rjj = (nn > 1e-24 * (n0 > 0 ? n0 : 1.0)) ? std::sqrt(nn) : 0.0;     // dependent column: contributes nothing new
        R[(size_t)j * FA + j] = rjj;
        for (int c = 0; c < d; ++c) Qm[(size_t)c * FA + j] = rjj > 0 ? v[c] / rjj : 0.0;
    }
    l_img.assign(64 * 64, 0);
    for (int i = 0; i < FA; ++i)
        for (int f = i; f < FA; ++f) {
            const int k = f < F ? slot_of_feature(f) : SLOT_ONE;
            const f16 hv = (f16)(float)R[(size_t)i * FA + f];
            unsigned short bits;
            __builtin_memcpy(&bits, &hv, 2);
            l_img[img_off(i, k) / 2] = bits;
        }
    // ---- T [j][53]: scale*log2e * Wq . diag(g) . [Wc | bc] and scale*log2e * Wq . nb
    const int NC = 53;
    const double s = 1.4426950408889634 / std::sqrt((double)d);
    std::vector<double> T((size_t)d * NC, 0.0);
    for (int j = 0; j < d; ++j) {
        double* tj = &T[(size_t)j * NC];
        for (int c = 0; c < d; ++c) {
            const double wq = (double)Wq[(size_t)j * d + c];
            const double wg = wq * (double)ng[c] * s;
            const double* wc = &Wc[(size_t)c * FA];
            for (int f = 0; f < FA; ++f) tj[f] += wg * wc[f];
            tj[52] += wq * (double)nb[c] * s;
        }
    }
    // ---- T2 [c'][53] = Wk^T . T, scattered into slot order
    t2aug.assign((size_t)d * 64, 0.f);
    std::vector<double> acc(NC);
    for (int cp = 0; cp < d; ++cp) {
        for (int f = 0; f < NC; ++f) acc[f] = 0.0;
        for (int j = 0; j < d; ++j) {
            const double wk = (double)Wk[(size_t)j * d + cp];
            const double* tj = &T[(size_t)j * NC];
            for (int f = 0; f < NC; ++f) acc[f] += wk * tj[f];
        }
        float* o = &t2aug[(size_t)cp * 64];
        for (int f = 0; f < F; ++f) o[slot_of_feature(f)] = (float)acc[f];
        o[SLOT_ONE] = (float)acc[51];
        o[SLOT_STD] = (float)acc[52];
        o[SLOT_U] = wfold[cp];
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// per-sample context: Y = LN_ctx(x) . t2aug  (fp32), then the fp16 LDS image + u + scale
// ---------------------------------------------------------------------------------------------------------------
template <int VPL>     // d = 64 * VPL
__global__ __launch_bounds__(256) void ae_ctx_project_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ t2,
                                                             float* __restrict__ Y, unsigned* __restrict__ absmax, int rows, int M) {
    constexpr int D = 64 * VPL;
    __shared__ float xs[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    {
        const int r = row < rows ? row : rows - 1;
        float v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { v[i] = x[(int64_t)r * D + lane + 64 * i]; s += v[i]; }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float dv = v[i] - mean; q += dv * dv; }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            xs[w][c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
    }
    __syncthreads();
    __shared__ float red[4][4][64];
    const float y = project4_rows<D>(xs, t2, red, lane, w);
    if (row < rows) Y[(int64_t)row * 64 + lane] = y;
    // largest |coefficient| of the sample (the fp16 image's scale): max is order-independent, so the atomic keeps the result
    // reproducible; non-negative floats compare like their bit patterns
    float mx = (row < rows && (lane <= SLOT_ONE || lane == SLOT_STD)) ? fabsf(y) : 0.f;
    mx = wave_max(mx);
    if (lane == 0 && row < rows) atomicMax(absmax + row / M, __float_as_uint(mx));
}

// ctx = [ H~ image: M x 128 B | u: M floats | inv_scale, 3 pad floats ] per sample; 4 latent rows per workgroup
__global__ __launch_bounds__(256) void ae_ctx_pack_kernel(const float* __restrict__ Y, const unsigned* __restrict__ absmax,
                                                          unsigned char* __restrict__ ctx, int M, int64_t ctx_stride) {
    const int b = blockIdx.y;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63;
    const float* y = Y + (int64_t)b * M * 64;
    unsigned char* out = ctx + (int64_t)b * ctx_stride;
    const float mx = __uint_as_float(absmax[b]);
    // power-of-two scale that puts the largest entry into [2^13, 2^14): fp16 keeps 11 bits for everything within 2^-27 of it
    float scale = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) scale = exp2f((float)(13 - ilogbf(mx)));
    float v = 0.f;
    if (k < SLOT_ONE) v = y[l * 64 + k] * scale;
    else if (k == SLOT_ONE || k == SLOT_ONE + 1) {
        const float t = y[l * 64 + SLOT_ONE] * scale;
        const f16 hi = (f16)t;
        v = k == SLOT_ONE ? (float)hi : t - (float)hi;
    } else if (k >= SLOT_STD && k <= SLOT_STD + 2) {
        const float t = y[l * 64 + SLOT_STD] * scale;
        const f16 hi = (f16)t;
        v = k == SLOT_STD + 2 ? t - (float)hi : (float)hi;
    }
    *reinterpret_cast<f16*>(out + img_off(l, k)) = (f16)v;
    float* u = reinterpret_cast<float*>(out + (int64_t)M * 128);
    if (k == SLOT_U) u[l] = y[l * 64 + SLOT_U];
    if (l == 0 && k == 0) u[M] = 1.0f / scale;
}

int ae_ctx_build(const float* x, const float* gamma, const float* beta, const float* t2aug, float* Yscratch, void* ctx, int B, int M, int d,
                 hipStream_t st) {
    RALD_CHECK(d == 256 || d == 512, "ae_ctx_build: dim must be 256 or 512");
    RALD_CHECK(M % 4 == 0, "ae_ctx_build: num_latents must be a multiple of 4");
    const int rows = B * M;
    unsigned* absmax = reinterpret_cast<unsigned*>(Yscratch + (int64_t)rows * 64);      // B words behind the projection (scratch is sized for them)
    RALD_HIP(hipMemsetAsync(absmax, 0, (size_t)B * 4, st));
    if (d == 256) hipLaunchKernelGGL((ae_ctx_project_kernel<4>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, gamma, beta, t2aug, Yscratch, absmax, rows, M);
    else hipLaunchKernelGGL((ae_ctx_project_kernel<8>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, gamma, beta, t2aug, Yscratch, absmax, rows, M);
    RALD_HIP(hipGetLastError());
    hipLaunchKernelGGL(ae_ctx_pack_kernel, dim3(M / 4, B), dim3(256), 0, st, Yscratch, absmax, (unsigned char*)ctx, M, ae_ctx_stride(M));
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the streaming kernel
// ---------------------------------------------------------------------------------------------------------------
struct DecodeArgs {
    const unsigned char* ctx; int64_t ctx_stride;     // per sample: image | u | inv_scale
    const unsigned short* l_img;                      // [64][64] fp16 image of the var factor
    const float* queries;                             // [B][Q][3]
    float* out;                                       // [B][Q]
    const float* basis;                               // [3][24]
    int64_t Q;
    int M;
    float c0, eps;
};

template <bool DIAG, int NW>
__global__ __launch_bounds__(NW * 64) void ae_decode_stream_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = a.M;
    unsigned char* s_h = smem;                                   // M * 128
    unsigned char* s_l = smem + (size_t)M * 128;                 // 8192
    float* s_u = reinterpret_cast<float*>(s_l + 8192);           // M + 4
    float* s_basis = s_u + M + 4;                                // 72 (+ pad)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.ctx + (int64_t)b * a.ctx_stride);
        uint4* dst = reinterpret_cast<uint4*>(s_h);
        const int n16 = M * 8;                                   // image
        for (int i = tid; i < n16; i += NW * 64) dst[i] = src[i];
        const uint4* su = src + n16;                             // u | inv_scale
        uint4* du = reinterpret_cast<uint4*>(s_u);
        for (int i = tid; i < M / 4 + 1; i += NW * 64) du[i] = su[i];
        const uint4* sl = reinterpret_cast<const uint4*>(a.l_img);
        uint4* dl = reinterpret_cast<uint4*>(s_l);
        for (int i = tid; i < 512; i += NW * 64) dl[i] = sl[i];
        if (tid < 72) s_basis[tid] = a.basis[tid] * 0.15915494309189535f;      // radians -> revolutions (v_sin_f32 takes revolutions)
    }
    __syncthreads();
    const float inv_scale = s_u[M];
    const int r = lane & 31, h = lane >> 5;
    const float* qin = a.queries + (int64_t)b * a.Q * 3;
    float* qout = a.out + (int64_t)b * a.Q;
    const int64_t nchunks = (a.Q + 63) / 64;
    const float quarter = h ? 0.25f : 0.0f;                       // cos(t) = sin(t + 1/4 revolution)

    for (int64_t chunk = (int64_t)blockIdx.x * NW + wave; chunk < nchunks; chunk += (int64_t)gridDim.x * NW) {
        f16x8 bq[2][4];
        float rq[2];                                              // rstd_q / scale
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            int64_t q = chunk * 64 + qb * 32 + r;
            q = q < a.Q ? q : a.Q - 1;
            const float x = qin[q * 3 + 0], y = qin[q * 3 + 1], z = qin[q * 3 + 2];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                f16x8 f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 8 * s + j;
                    float p;
                    if (DIAG) p = (s == 0 ? x : (s == 1 ? y : z)) * s_basis[24 * s + e];
                    else p = fmaf(z, s_basis[48 + e], fmaf(y, s_basis[24 + e], x * s_basis[e]));
                    p += quarter;
                    p = __builtin_amdgcn_fractf(p);
                    f[j] = (f16)__builtin_amdgcn_sinf(p);
                }
                bq[qb][s] = f;
            }
            f16x8 f;
            f[0] = (f16)(h ? 0.f : x); f[1] = (f16)(h ? 0.f : y); f[2] = (f16)(h ? 0.f : z);
            f[3] = (f16)(h ? 0.f : 1.f); f[4] = f[3];
            f[5] = (f16)0.f; f[6] = (f16)0.f; f[7] = (f16)0.f;
            bq[qb][3] = f;
        }
        // ---- LayerNorm statistics of the query embedding: var_q = |L.f~|^2 (two 32-row tiles x 4 k-steps)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const int row = 32 * t + r;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f16x8 la = *reinterpret_cast<const f16x8*>(s_l + row * 128 + (((2 * s + h) ^ (row & 7)) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(la, bq[qb][s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) ss = fmaf(acc[i], acc[i], ss);
            }
            ss += __shfl_xor(ss, 32, 64);
            const float var = ss + a.eps;
            const float rstd = rsqrtf(var);
            const float sd = var * rstd;                           // sqrt(var + eps)
            const f16 sd_hi = (f16)sd;
            const f16 sd_lo = (f16)(sd - (float)sd_hi);
            f16x8 f = bq[qb][3];
            f[5] = h ? (f16)0.f : sd_hi; f[6] = h ? (f16)0.f : sd_lo; f[7] = h ? (f16)0.f : sd_hi;
            bq[qb][3] = f;
            rq[qb] = rstd * inv_scale;
        }
        // ---- scores + online softmax over the latents
        float m[2] = {-INFINITY, -INFINITY}, den[2] = {0.f, 0.f}, num[2] = {0.f, 0.f};
        const int ntile = M >> 5;
        for (int t = 0; t < ntile; ++t) {
            const int row = 32 * t + r;
            f16x8 fa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const f16x8*>(s_h + row * 128 + (((2 * s + h) ^ (row & 7)) << 4));
            float4 uv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) uv[g] = *reinterpret_cast<const float4*>(s_u + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bq[qb][s], acc, 0, 0, 0);
                float tm = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
                for (int i = 4; i < 16; i += 4) tm = fmaxf(tm, fmaxf(fmaxf(acc[i], acc[i + 1]), fmaxf(acc[i + 2], acc[i + 3])));
                const float vm = tm * rq[qb];
                if (__any(vm > m[qb] + 8.0f)) {                    // lazy running maximum: p stays <= 2^8
                    const float mn = fmaxf(m[qb], vm);
                    const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
                    den[qb] *= alpha; num[qb] *= alpha;
                    m[qb] = mn;
                }
                const float nm = -m[qb];
                float d0 = 0.f, d1 = 0.f, n0 = 0.f, n1 = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(acc[4 * g + 0], rq[qb], nm));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(acc[4 * g + 1], rq[qb], nm));
                    const float p2 = __builtin_amdgcn_exp2f(fmaf(acc[4 * g + 2], rq[qb], nm));
                    const float p3 = __builtin_amdgcn_exp2f(fmaf(acc[4 * g + 3], rq[qb], nm));
                    d0 += p0; d1 += p1; d0 += p2; d1 += p3;
                    n0 = fmaf(p0, uv[g].x, n0); n1 = fmaf(p1, uv[g].y, n1); n0 = fmaf(p2, uv[g].z, n0); n1 = fmaf(p3, uv[g].w, n1);
                }
                den[qb] += d0 + d1;
                num[qb] += n0 + n1;
            }
        }
        // ---- merge the two lane halves of each query and store
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mo = __shfl_xor(m[qb], 32, 64), dn = __shfl_xor(den[qb], 32, 64), nn = __shfl_xor(num[qb], 32, 64);
            const float mm = fmaxf(m[qb], mo);
            const float wa = __builtin_amdgcn_exp2f(m[qb] - mm), wb = __builtin_amdgcn_exp2f(mo - mm);
            const float dsum = den[qb] * wa + dn * wb, nsum = num[qb] * wa + nn * wb;
            const int64_t q = chunk * 64 + qb * 32 + r;
            if (h == 0 && q < a.Q) qout[q] = nsum / dsum + a.c0;
        }
    }
}

template <bool DIAG, int NW>
static int launch_decode(const DecodeArgs& a, int B, size_t smem, hipStream_t st) {
    auto kern = ae_decode_stream_kernel<DIAG, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 128 + 8192 + (1024 + 4 + 76) * 4));
        attr_set = true;
    }
    const int64_t nchunks = (a.Q + 63) / 64;
    int64_t per_sample = (nchunks + NW - 1) / NW;                  // workgroups that have at least one chunk per wave
    const int64_t cap = B >= 256 ? 1 : 256 / B;                    // about one workgroup per CU over the whole batch
    if (per_sample > cap) per_sample = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)per_sample, (unsigned)B), dim3(NW * 64), smem, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

// nw: waves per workgroup (8, 12 or 16; 0 = the measured default) - one workgroup per CU holds the sample's image in LDS
int ae_decode_stream(const void* ctx, const unsigned short* l_img, const float* queries, float* out, const float* basis, int basis_diag,
                     int B, int64_t Q, int M, float c0, hipStream_t st, int nw) {
    RALD_CHECK(M % 32 == 0 && M >= 32 && M <= 1024, "ae_decode_stream: num_latents must be a multiple of 32 in [32,1024]");
    RALD_CHECK(B >= 1 && Q >= 1 && B <= 65535, "ae_decode_stream: bad batch / query count");
    DecodeArgs a;
    a.ctx = (const unsigned char*)ctx; a.ctx_stride = ae_ctx_stride(M); a.l_img = l_img; a.queries = queries; a.out = out; a.basis = basis;
    a.Q = Q; a.M = M; a.c0 = c0; a.eps = 1e-5f;
    const size_t smem = (size_t)M * 128 + 8192 + (size_t)(M + 4 + 76) * 4;
    if (nw == 0) nw = 12;
    if (nw == 8) return basis_diag ? launch_decode<true, 8>(a, B, smem, st) : launch_decode<false, 8>(a, B, smem, st);
    if (nw == 12) return basis_diag ? launch_decode<true, 12>(a, B, smem, st) : launch_decode<false, 12>(a, B, smem, st);
    if (nw == 16) return basis_diag ? launch_decode<true, 16>(a, B, smem, st) : launch_decode<false, 16>(a, B, smem, st);
    RALD_CHECK(false, "ae_decode_stream: nw must be 0, 8, 12 or 16");
}

}  // namespace rald
