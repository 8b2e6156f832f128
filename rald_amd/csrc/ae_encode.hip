// Folded encoder of the set-latent autoencoder (KLAutoEncoder.encode, model/models_ae.py:351-399): the two attentions of the
// latent queries over the P input points, without ever forming the points' d-wide embeddings, keys or values.
//
//     pc [P,3] -> PointEmbed (:355)  emb_p = Wa.f_p,  f_p = [sin(p.basis), cos(p.basis), p, 1]  (52 numbers, Wa = [Wpe | b_pe])
//     mix query (:380-386)           8-head attention of the (weight-only) query LN(d_latents).Wq over k, v = to_kv(emb)
//     cross_attend (:395)            1-head d-wide attention of LN(x).Wq over k, v = to_kv(LN_ctx(emb))
//
// Everything between f_p and a score or a value is linear, except LN_ctx, whose statistics are a quadratic form of f_p
// (as in the query decoder, ae_decode.hip): emb_p - mean = Wc.f_p, var_p = |R.f_p|^2, R^T.R = Wc^T.Wc / d.  So
//
//   mix    score_h[m,p] = (q1_h[m].Wk_h.Wa).f_p       (the part multiplying the constant 1 is the same for every key: dropped)
//          out[m]       = sum_h Wo_h.Wv_h.Wa.(sum_p P_h[m,p] f_p)
//          x            = query_proj(s_latents + out) = X0 + O1.T4^T          O1[m][64h + j] = sum_p P_h[m,p] f_p[j]
//   cross  score[m,p]   = (LN(x)[m].T1).g_p,  g_p = rstd_p.f_p                (the LN_ctx bias term is again constant over p)
//          x           += (sum_p P[m,p] g_p).T3^T + c3
//
// i.e. both attentions have head dimension 64 (52 used) with KEYS = VALUES = one fp16 feature row per point - F for the mix
// layer (all 8 heads read the same rows), G = rstd.F for cross_attend - and the weight-only tables Q1 (the mix queries), T4,
// X0, T1, T3, c3 are built once per weight load on the host in double.  Per cloud of 10 000 points this removes the
// PointEmbed GEMM, four 10 000 x 512 x 512 K/V projections, the context LayerNorm and the d = 512 score / PV products
// (21 of 47 GFLOP, ~250 MB of activations); what is left runs on attention_d64_kernel's fp16 form (attention.hip).
//
// Exact in real arithmetic (tests/test_encode_fold.py checks the tables against the oracle in float64); numerically the
// features are bounded by construction (sin, cos, normalised coordinates), fp16 keeps 11 bits of them.
#include <cmath>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace rald {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int NF = 51, FA = 52;

// C[m][n] = sum_k A[m][k] B[k][n]  (row-major doubles; the inner loop runs over contiguous n)
void matmul(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; ++i) {
        double* c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int t = 0; t < k; ++t) {
            const double a = A[(size_t)i * k + t];
            const double* b = B + (size_t)t * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}
}  // namespace

// R [52][52] (upper triangular, R^T.R = Wc^T.Wc / d) and Wc [d][52] (columns centred over the d outputs) of PointEmbed's Linear
int ae_embed_factor(int d, const float* Wpe, const float* bpe, std::vector<double>& Wc, std::vector<double>& R) {
    Wc.assign((size_t)d * FA, 0.0);
    for (int f = 0; f < FA; ++f) {
        double mu = 0.0;
        for (int c = 0; c < d; ++c) mu += f < NF ? (double)Wpe[(size_t)c * NF + f] : (double)bpe[c];
        mu /= d;
        for (int c = 0; c < d; ++c) Wc[(size_t)c * FA + f] = (f < NF ? (double)Wpe[(size_t)c * NF + f] : (double)bpe[c]) - mu;
    }
    // modified Gram-Schmidt on Wc / sqrt(d)
    std::vector<double> Qm((size_t)d * FA);
    R.assign((size_t)FA * FA, 0.0);
    const double inv_sd = 1.0 / std::sqrt((double)d);
    std::vector<double> v(d);
    for (int j = 0; j < FA; ++j) {
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
    return 0;
}

// Weight-only tables of the folded encoder.  All inputs are the reference's tensors in their own layouts (fp32, host):
//   Wpe [d][51], bpe [d]                                             point_embed.mlp
//   mix (query_type 'mix' only, else pass mixq = false and lat = latents.weight):
//     d_lat [M][d], mng / mnb [d] (mix_attn_layer.norm), mWq [I][d], mWkv [2I][d], mWo [d][I], mbo [d], lat = s_latents [M][d],
//     Wqp [d][d], bqp [d] (query_proj)
//   cross_attend_blocks.0: cg / cb [d] (norm_context), cWq [d][d], cWkv [2d][d], cWo [d][d], cbo [d]
// Outputs:
//   Rf  [52][52] fp32                   variance factor
//   Q1  [M][I]  fp32, [m][64h + j]      mix queries against the feature rows, times dim_head^-1/2 . log2(e); columns j >= 51 zero
//   T4  [d][I]  fp32, row n, k = 64h+j  x = X0 + O1.T4^T
//   X0  [M][d]  fp32                    everything of x that does not depend on the points ('learnable': the latents)
//   T1  [d][64] fp32, [c][j]            cross-attention queries Q' = LN(x).T1, times d^-1/2 . log2(e)
//   T3  [d][64] fp32, row n             x += O'.T3^T + c3
//   c3  [d]
int ae_encode_tables(int d, int I, int M, int heads, bool mixq, const float* Wpe, const float* bpe, const float* d_lat, const float* mng,
                     const float* mnb, const float* mWq, const float* mWkv, const float* mWo, const float* mbo, const float* lat,
                     const float* Wqp, const float* bqp, const float* cg, const float* cb, const float* cWq, const float* cWkv,
                     const float* cWo, const float* cbo, std::vector<float>& Rf, std::vector<float>& Q1, std::vector<float>& T4,
                     std::vector<float>& X0, std::vector<float>& T1, std::vector<float>& T3, std::vector<float>& c3) {
    RALD_CHECK(d > 0 && M > 0 && heads > 0 && I == heads * 64, "ae_encode_tables: bad sizes");
    constexpr double LOG2E = 1.4426950408889634;
    std::vector<double> Wc, R;
    RALD_TRY(ae_embed_factor(d, Wpe, bpe, Wc, R));
    Rf.assign((size_t)FA * FA, 0.f);
    for (size_t i = 0; i < Rf.size(); ++i) Rf[i] = (float)R[i];
    std::vector<double> Wa((size_t)d * FA);                                  // [c][f] = [Wpe | bpe]
    for (int c = 0; c < d; ++c) {
        for (int f = 0; f < NF; ++f) Wa[(size_t)c * FA + f] = Wpe[(size_t)c * NF + f];
        Wa[(size_t)c * FA + NF] = bpe[c];
    }
    auto dbl = [](const float* p, size_t n) { return std::vector<double>(p, p + n); };
    X0.assign((size_t)M * d, 0.f);
    if (mixq) {
        // q1 = LN(d_latents).Wq^T                                             (:383-384, PreNorm of the mix layer)
        std::vector<double> xn((size_t)M * d), WqT((size_t)d * I), q1((size_t)M * I);
        for (int m = 0; m < M; ++m) {
            double mu = 0.0, var = 0.0;
            for (int c = 0; c < d; ++c) mu += d_lat[(size_t)m * d + c];
            mu /= d;
            for (int c = 0; c < d; ++c) { const double t = d_lat[(size_t)m * d + c] - mu; var += t * t; }
            const double rstd = 1.0 / std::sqrt(var / d + 1e-5);
            for (int c = 0; c < d; ++c) xn[(size_t)m * d + c] = (d_lat[(size_t)m * d + c] - mu) * rstd * mng[c] + mnb[c];
        }
        for (int n = 0; n < I; ++n)
            for (int c = 0; c < d; ++c) WqT[(size_t)c * I + n] = mWq[(size_t)n * d + c];
        matmul(xn.data(), WqT.data(), q1.data(), M, d, I);
        // per head: A_h = Wk_h.Wa [64][52], V_h = Wv_h.Wa [64][52]
        const double s1 = LOG2E / std::sqrt(64.0);
        Q1.assign((size_t)M * I, 0.f);
        std::vector<double> Wk = dbl(mWkv, (size_t)I * d), Wv = dbl(mWkv + (size_t)I * d, (size_t)I * d);
        std::vector<double> Ah((size_t)I * FA), Vh((size_t)I * FA);
        matmul(Wk.data(), Wa.data(), Ah.data(), I, d, FA);                   // rows 64h..64h+63 = A_h
        matmul(Wv.data(), Wa.data(), Vh.data(), I, d, FA);
        std::vector<double> qa(FA);
        for (int m = 0; m < M; ++m)
            for (int h = 0; h < heads; ++h) {
                for (int f = 0; f < FA; ++f) qa[f] = 0.0;
                for (int j = 0; j < 64; ++j) {
                    const double q = q1[(size_t)m * I + 64 * h + j];
                    const double* ar = &Ah[(size_t)(64 * h + j) * FA];
                    for (int f = 0; f < FA; ++f) qa[f] += q * ar[f];
                }
                for (int f = 0; f < NF; ++f) Q1[(size_t)m * I + 64 * h + f] = (float)(qa[f] * s1);
            }
        // T [d][I]: column block h = Wo[:, 64h:64h+64].V_h; the constant column (f = 51) multiplies sum_p P = 1: into X0
        std::vector<double> T((size_t)d * I, 0.0), tconst(d, 0.0);
        for (int n = 0; n < d; ++n)
            for (int h = 0; h < heads; ++h) {
                double* tr = &T[(size_t)n * I + 64 * h];
                for (int j = 0; j < 64; ++j) {
                    const double w = mWo[(size_t)n * I + 64 * h + j];
                    const double* vr = &Vh[(size_t)(64 * h + j) * FA];
                    for (int f = 0; f < NF; ++f) tr[f] += w * vr[f];
                    tconst[n] += w * vr[NF];
                }
            }
        // x = (s_lat + T.O1 + tconst + bo).Wqp^T + bqp
        std::vector<double> WqpD = dbl(Wqp, (size_t)d * d), T4d((size_t)d * I);
        matmul(WqpD.data(), T.data(), T4d.data(), d, d, I);
        T4.assign((size_t)d * I, 0.f);
        for (size_t i = 0; i < T4.size(); ++i) T4[i] = (float)T4d[i];
        std::vector<double> sl((size_t)M * d), WqpT((size_t)d * d), x0((size_t)M * d);
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < d; ++c) sl[(size_t)m * d + c] = (double)lat[(size_t)m * d + c] + tconst[c] + mbo[c];
        for (int n = 0; n < d; ++n)
            for (int c = 0; c < d; ++c) WqpT[(size_t)c * d + n] = Wqp[(size_t)n * d + c];
        matmul(sl.data(), WqpT.data(), x0.data(), M, d, d);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < d; ++n) X0[(size_t)m * d + n] = (float)(x0[(size_t)m * d + n] + bqp[n]);
    } else {
        Q1.clear(); T4.clear();
        for (size_t i = 0; i < X0.size(); ++i) X0[i] = lat[i];
    }
    // cross_attend: U = Wk.diag(cg).Wc [d][52];  T1 = Wq^T.U . d^-1/2 log2e;  T3 = Wo.(Wv.diag(cg).Wc);  c3 = Wo.(Wv.cb) + bo
    {
        std::vector<double> gWc((size_t)d * FA), U((size_t)d * FA), Vv((size_t)d * FA), t((size_t)d * FA);
        for (int c = 0; c < d; ++c)
            for (int f = 0; f < FA; ++f) gWc[(size_t)c * FA + f] = cg[c] * Wc[(size_t)c * FA + f];
        std::vector<double> Wk = dbl(cWkv, (size_t)d * d), Wv = dbl(cWkv + (size_t)d * d, (size_t)d * d), Wo = dbl(cWo, (size_t)d * d);
        matmul(Wk.data(), gWc.data(), U.data(), d, d, FA);
        matmul(Wv.data(), gWc.data(), Vv.data(), d, d, FA);
        std::vector<double> WqT((size_t)d * d);
        for (int n = 0; n < d; ++n)
            for (int c = 0; c < d; ++c) WqT[(size_t)c * d + n] = cWq[(size_t)n * d + c];
        matmul(WqT.data(), U.data(), t.data(), d, d, FA);
        const double s2 = LOG2E / std::sqrt((double)d);
        T1.assign((size_t)d * 64, 0.f);
        for (int c = 0; c < d; ++c)
            for (int f = 0; f < FA; ++f) T1[(size_t)c * 64 + f] = (float)(t[(size_t)c * FA + f] * s2);
        matmul(Wo.data(), Vv.data(), t.data(), d, d, FA);
        T3.assign((size_t)d * 64, 0.f);
        for (int n = 0; n < d; ++n)
            for (int f = 0; f < FA; ++f) T3[(size_t)n * 64 + f] = (float)t[(size_t)n * FA + f];
        std::vector<double> vb(d, 0.0);
        for (int j = 0; j < d; ++j)
            for (int c = 0; c < d; ++c) vb[j] += Wv[(size_t)j * d + c] * cb[c];
        c3.assign(d, 0.f);
        for (int n = 0; n < d; ++n) {
            double s = cbo[n];
            for (int j = 0; j < d; ++j) s += Wo[(size_t)n * d + j] * vb[j];
            c3[n] = (float)s;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// per point: the 51 Fourier features (+ the constant) and 1 / std of the point's embedding, as two fp16 key/value rows
// ---------------------------------------------------------------------------------------------------------------
// F [b][Pp][64]: f_0..f_50, 0...          (mix layer: the constant's score is the same for every key, its value is in X0)
// G [b][Pp][64]: rstd.f_0..f_50, rstd, 0...
// Rows P..Pp-1 (the attention stages whole 64-key tiles) are zero.  One thread per point, one wave per workgroup (157 workgroups for
// 10 000 points); R sits in LDS and is read as broadcast float4s - read wave-uniformly from global memory instead (scalar loads) the 1 378
// dependent s_load / v_fma pairs took 28 us.
__global__ __launch_bounds__(64) void ae_enc_features_kernel(const float* __restrict__ pc, const float* __restrict__ basis,
                                                             const float* __restrict__ Rf, f16* __restrict__ F, f16* __restrict__ G,
                                                             int P, int Pp, int64_t total) {
    __shared__ __attribute__((aligned(16))) float sR[FA * FA];
    for (int i = threadIdx.x; i < FA * FA / 4; i += 64) reinterpret_cast<float4*>(sR)[i] = reinterpret_cast<const float4*>(Rf)[i];
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= total) return;
    const int p = (int)(idx % Pp);
    const int64_t b = idx / Pp;
    f16x8* fo = reinterpret_cast<f16x8*>(F + idx * 64);
    f16x8* go = reinterpret_cast<f16x8*>(G + idx * 64);
    if (p >= P) {
        f16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (f16)0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { fo[c] = z; go[c] = z; }
        return;
    }
    const float* pt = pc + (b * P + p) * 3;
    const float x = pt[0], y = pt[1], z = pt[2];
    float f[64];
    constexpr float INV_2PI = 0.15915494309189535f;
#pragma unroll
    for (int e = 0; e < 24; ++e) {
        const float rev = (x * basis[e] + y * basis[24 + e] + z * basis[48 + e]) * INV_2PI;     // revolutions
        const float fr = rev - floorf(rev);
        f[e] = __builtin_amdgcn_sinf(fr);
        const float fc = fr + 0.25f;
        f[24 + e] = __builtin_amdgcn_sinf(fc - floorf(fc));
    }
    f[48] = x; f[49] = y; f[50] = z; f[51] = 1.f;
#pragma unroll
    for (int j = 52; j < 64; ++j) f[j] = 0.f;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < FA; ++i) {                       // row i of the upper-triangular factor: columns 4*(i/4) .. 51 (zeros left of the diagonal)
        constexpr int NCH = FA / 4;
        float4 r[NCH];                                   // the whole row in flight before its FMAs (the compiler kept 2 reads in flight otherwise)
#pragma unroll
        for (int c = i / 4; c < NCH; ++c) r[c] = *reinterpret_cast<const float4*>(sR + i * FA + 4 * c);
        asm volatile("" ::: "memory");
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = i / 4; c < NCH; ++c) {
            s0 = fmaf(r[c].x, f[4 * c], s0); s1 = fmaf(r[c].y, f[4 * c + 1], s1);
            s0 = fmaf(r[c].z, f[4 * c + 2], s0); s1 = fmaf(r[c].w, f[4 * c + 3], s1);
        }
        const float s = s0 + s1;
        var = fmaf(s, s, var);
    }
    const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        f16x8 a, g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * c + j;
            a[j] = (f16)(k == 51 ? 0.f : f[k]);
            g[j] = (f16)(f[k] * rstd);
        }
        fo[c] = a; go[c] = g;
    }
}

int ae_enc_features(const float* pc, const float* basis, const float* Rf, void* F, void* G, int B, int P, int Pp, hipStream_t st) {
    RALD_CHECK(pc && basis && Rf && F && G && B >= 1 && P >= 1 && Pp >= P && Pp % 64 == 0, "ae_enc_features: bad arguments");
    RALD_CHECK((uintptr_t)F % 16 == 0 && (uintptr_t)G % 16 == 0, "ae_enc_features: 16-byte alignment");
    const int64_t total = (int64_t)B * Pp;
    hipLaunchKernelGGL(ae_enc_features_kernel, dim3((unsigned)cdiv(total, (int64_t)64)), dim3(64), 0, st, pc, basis, Rf, (f16*)F, (f16*)G, P, Pp, total);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// x[row] = (xin ? xin[row] : 0) + X0[row % M]  (kept, fp32);  Q'[row] = LN(x[row]; g, b) . T1   [rows][64] fp32
// ---------------------------------------------------------------------------------------------------------------
template <int VPL>     // d = 64 * VPL; one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void ae_enc_qproj_kernel(const float* __restrict__ xin, const float* __restrict__ X0, float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ T1, float* __restrict__ Qo, int rows, int M) {
    constexpr int D = 64 * VPL;
    __shared__ float xs[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    {
        const int r = row < rows ? row : rows - 1;
        const float* x0 = X0 + (int64_t)(r % M) * D;
        float v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i] = x0[lane + 64 * i] + (xin ? xin[(int64_t)r * D + lane + 64 * i] : 0.f);
            s += v[i];
        }
        if (row < rows) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) x[(int64_t)r * D + lane + 64 * i] = v[i];
        }
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
    const float q = project4_rows<D>(xs, T1, red, lane, w);
    if (row < rows) Qo[(int64_t)row * 64 + lane] = q;
}

int ae_enc_qproj(const float* xin, const float* X0, float* x, const float* gamma, const float* beta, const float* T1, float* Qo, int rows, int M,
                 int d, hipStream_t st) {
    RALD_CHECK(X0 && x && gamma && beta && T1 && Qo && rows >= 1 && M >= 1, "ae_enc_qproj: bad arguments");
    RALD_CHECK(d == 256 || d == 512, "ae_enc_qproj: d must be 256 or 512");
    dim3 grid(cdiv(rows, 4)), block(256);
    if (d == 256) hipLaunchKernelGGL((ae_enc_qproj_kernel<4>), grid, block, 0, st, xin, X0, x, gamma, beta, T1, Qo, rows, M);
    else hipLaunchKernelGGL((ae_enc_qproj_kernel<8>), grid, block, 0, st, xin, X0, x, gamma, beta, T1, Qo, rows, M);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
