// Kernels specific to the set-latent autoencoder (model/models_ae.py): Fourier point features,
// row softmax for the single-head d=dim cross-attention, the folded decode epilogue, the
// diagonal-Gaussian posterior.  HBM-bound, coalesced; the contractions run in gemm.hip.
#include "common.h"
#include "kernels.h"

namespace rald {

// ---- PointEmbed features (models_ae.py:128-133): feat = [sin(p.basis) (24), cos(p.basis) (24), p (3)],
// zero-padded to 64 bf16 so the 51->dim Linear runs as one K=64 MFMA step.
__global__ __launch_bounds__(256) void point_features_kernel(const float* __restrict__ pts, const float* __restrict__ basis,
                                                             bf16* __restrict__ feat, int64_t n) {
    __shared__ float sb[3 * 24];
    if (threadIdx.x < 72) sb[threadIdx.x] = basis[threadIdx.x];
    __syncthreads();
    // 8 threads per point: thread j writes 8 consecutive bf16 (16 B) -> a wave writes 1 KiB contiguous
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = gid >> 3;
    const int j = (int)(gid & 7);
    if (p >= n) return;
    const float x = pts[p * 3 + 0], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = j * 8 + i;
        float v = 0.f;
        if (c < 48) {
            const int e = c < 24 ? c : c - 24;
            const float pr = x * sb[e] + y * sb[24 + e] + z * sb[48 + e];
            v = c < 24 ? sinf(pr) : cosf(pr);
        } else if (c < 51) {
            v = c == 48 ? x : (c == 49 ? y : z);
        }
        o[i] = (bf16)v;
    }
    reinterpret_cast<bf16x8*>(feat)[gid] = o;
}
int point_features(const float* pts, const float* basis, bf16* feat, int64_t n, hipStream_t st) {
    RALD_CHECK(n > 0, "point_features: empty");
    const int64_t threads = n * 8;
    hipLaunchKernelGGL(point_features_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pts, basis, feat, n);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- row softmax fp32 -> bf16 with the row zero-padded to ld_out (so it can be the A operand of
// the P.V GEMM whose K must be a multiple of 64).  One workgroup per row; rows live in L2.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t ld_s, bf16* __restrict__ P,
                                                           int64_t ld_p, int n) {
    __shared__ float red[4];
    const float* s = S + (int64_t)blockIdx.x * ld_s;
    bf16* p = P + (int64_t)blockIdx.x * ld_p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -1e30f;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, s[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sum += __expf(s[i] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < (int)ld_p; i += 256) p[i] = (bf16)(i < n ? __expf(s[i] - mx) * inv : 0.f);
}
int softmax_rows(const float* S, int64_t ld_s, bf16* P, int64_t ld_p, int rows, int n, hipStream_t st) {
    RALD_CHECK(rows > 0 && n > 0 && ld_s >= n && ld_p >= n, "softmax_rows: bad shape");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, S, ld_s, P, ld_p, n);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- folded decoder epilogue: logit[q] = sum_k softmax(S[q,:])[k] * u[b][k] + c0
// (decoder_cross_attn value path + to_out + to_outputs collapse to one vector u per sample because
// the reference applies no nonlinearity between them: models_ae.py:417-424, :103-105).
// One wave per query row, nkeys (= num_latents, <= 1024) scores in registers.
template <int VPL>
__global__ __launch_bounds__(256) void softmax_dot_kernel(const float* __restrict__ S, const float* __restrict__ u,
                                                          float* __restrict__ out, int64_t rows, int rows_per_batch, float c0) {
    constexpr int NC = VPL / 4;
    constexpr int NK = VPL * 64;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* s = reinterpret_cast<const float4*>(S + row * NK);
    const float4* uu = reinterpret_cast<const float4*>(u + (row / rows_per_batch) * NK);
    float4 v[NC];
    float mx = -1e30f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = s[lane + 64 * c];
        mx = fmaxf(mx, fmaxf(fmaxf(v[c].x, v[c].y), fmaxf(v[c].z, v[c].w)));
    }
    mx = wave_max(mx);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float4 w = uu[lane + 64 * c];
        const float e0 = __expf(v[c].x - mx), e1 = __expf(v[c].y - mx), e2 = __expf(v[c].z - mx), e3 = __expf(v[c].w - mx);
        den += e0 + e1 + e2 + e3;
        num += e0 * w.x + e1 * w.y + e2 * w.z + e3 * w.w;
    }
    den = wave_sum(den);
    num = wave_sum(num);
    if (lane == 0) out[row] = num / den + c0;
}
int softmax_dot(const float* S, const float* u, float* out, int64_t rows, int nkeys, int rows_per_batch, float c0, hipStream_t st) {
    RALD_CHECK(rows > 0 && rows_per_batch > 0, "softmax_dot: empty");
    RALD_CHECK(nkeys == 256 || nkeys == 512 || nkeys == 1024, "softmax_dot: num_latents must be 256, 512 or 1024");
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (nkeys == 256) hipLaunchKernelGGL((softmax_dot_kernel<4>), grid, block, 0, st, S, u, out, rows, rows_per_batch, c0);
    else if (nkeys == 512) hipLaunchKernelGGL((softmax_dot_kernel<8>), grid, block, 0, st, S, u, out, rows, rows_per_batch, c0);
    else hipLaunchKernelGGL((softmax_dot_kernel<16>), grid, block, 0, st, S, u, out, rows, rows_per_batch, c0);
    RALD_HIP(hipGetLastError());
    return 0;
}

// generic-width variant (any nkeys): one wave per row, strided loop (used when num_latents is not 256/512/1024)
__global__ __launch_bounds__(256) void softmax_dot_generic_kernel(const float* __restrict__ S, const float* __restrict__ u,
                                                                  float* __restrict__ out, int64_t rows, int nk,
                                                                  int rows_per_batch, float c0) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * nk;
    const float* uu = u + (row / rows_per_batch) * nk;
    float mx = -1e30f;
    for (int i = lane; i < nk; i += 64) mx = fmaxf(mx, s[i]);
    mx = wave_max(mx);
    float den = 0.f, num = 0.f;
    for (int i = lane; i < nk; i += 64) {
        const float e = __expf(s[i] - mx);
        den += e;
        num += e * uu[i];
    }
    den = wave_sum(den);
    num = wave_sum(num);
    if (lane == 0) out[row] = num / den + c0;
}
int softmax_dot_generic(const float* S, const float* u, float* out, int64_t rows, int nkeys, int rows_per_batch, float c0, hipStream_t st) {
    RALD_CHECK(rows > 0 && nkeys > 0 && rows_per_batch > 0, "softmax_dot: empty");
    hipLaunchKernelGGL(softmax_dot_generic_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, u, out, rows, nkeys, rows_per_batch, c0);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- u[row] = LN_affine(x[row]) . w   (the folded value vector; one wave per latent row)
template <int VPL>
__global__ __launch_bounds__(256) void ln_dot_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ w,
                                                     float* __restrict__ out, int M) {
    constexpr int NC = VPL / 4;
    constexpr int D = VPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = xr[lane + 64 * c];
        s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[lane + 64 * c];
        const float4 b = reinterpret_cast<const float4*>(beta)[lane + 64 * c];
        const float4 ww = reinterpret_cast<const float4*>(w)[lane + 64 * c];
        acc += ((v[c].x - mean) * rstd * g.x + b.x) * ww.x + ((v[c].y - mean) * rstd * g.y + b.y) * ww.y +
               ((v[c].z - mean) * rstd * g.z + b.z) * ww.z + ((v[c].w - mean) * rstd * g.w + b.w) * ww.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}
int ln_dot(const float* x, const float* gamma, const float* beta, const float* w, float* out, int M, int D, hipStream_t st) {
    RALD_CHECK(D == 256 || D == 512, "ln_dot: D must be 256 or 512");
    dim3 grid(cdiv(M, 4)), block(256);
    if (D == 256) hipLaunchKernelGGL((ln_dot_kernel<4>), grid, block, 0, st, x, gamma, beta, w, out, M);
    else hipLaunchKernelGGL((ln_dot_kernel<8>), grid, block, 0, st, x, gamma, beta, w, out, M);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- out_bf16[b][m][c] = bf16(a[m][c] + d[b][m][c])   (static + dynamic query, models_ae.py:385)
__global__ void add_bcast_cast_kernel(const float* __restrict__ a, const float* __restrict__ d, bf16* __restrict__ out,
                                      int64_t per_batch4, int64_t total4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const float4 av = reinterpret_cast<const float4*>(a)[i % per_batch4];
    const float4 dv = reinterpret_cast<const float4*>(d)[i];
    reinterpret_cast<bf16x4*>(out)[i] = pack4(av.x + dv.x, av.y + dv.y, av.z + dv.z, av.w + dv.w);
}
int add_bcast_cast(const float* a, const float* d, bf16* out, int64_t per_batch, int batch, hipStream_t st) {
    RALD_CHECK(per_batch % 4 == 0 && batch > 0, "add_bcast_cast: bad shape");
    const int64_t total4 = per_batch / 4 * batch;
    hipLaunchKernelGGL(add_bcast_cast_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, a, d, out, per_batch / 4, total4);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- DiagonalGaussianDistribution (models_ae.py:141-163): ml = [mean | logvar] per row (2L wide)
//   z = mean + exp(0.5*clamp(logvar,-30,20)) * eps ;  kl[b] = 0.5 * mean(mean^2 + var - 1 - logvar)
__global__ __launch_bounds__(1024) void posterior_kernel(const float* __restrict__ ml, const float* __restrict__ eps,
                                                         float* __restrict__ mean_o, float* __restrict__ logvar_o,
                                                         float* __restrict__ z, float* __restrict__ kl, int rows, int L) {
    // one workgroup of 16 waves per sample (kl[b] is a mean over the sample: a fixed-order reduction, no atomics), 4 elements per
    // thread and iteration (L % 4 == 0 is checked by the caller); the first version's 256 threads x 64 scalar iterations took 36 us
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int n4 = (L % 4 == 0) ? rows * L / 4 : 0, L4 = L / 4;
    float acc = 0.f;
    if (L % 4 != 0) {                                                  // latent_dim 1 / 2 (factories kl_d512_m512_l1, _l2): scalar form
        for (int i = threadIdx.x; i < rows * L; i += 1024) {
            const int r = i / L, c = i % L;
            const int64_t row = (int64_t)b * rows + r, o = row * L + c;
            const float mu = ml[row * 2 * L + c], lv_raw = ml[row * 2 * L + L + c];
            const float lv = fminf(fmaxf(lv_raw, -30.f), 20.f);
            z[o] = mu + expf(0.5f * lv) * eps[o];
            if (mean_o) mean_o[o] = mu;
            if (logvar_o) logvar_o[o] = lv_raw;
            acc += mu * mu + expf(lv) - 1.0f - lv;
        }
    }
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const int r = i / L4, c = (i % L4) * 4;
        const int64_t row = (int64_t)b * rows + r;
        const float4 mu = *reinterpret_cast<const float4*>(ml + row * 2 * L + c);
        const float4 lr = *reinterpret_cast<const float4*>(ml + row * 2 * L + L + c);
        const int64_t o = row * L + c;
        const float4 e = *reinterpret_cast<const float4*>(eps + o);
        const float lv0 = fminf(fmaxf(lr.x, -30.f), 20.f), lv1 = fminf(fmaxf(lr.y, -30.f), 20.f), lv2 = fminf(fmaxf(lr.z, -30.f), 20.f),
                    lv3 = fminf(fmaxf(lr.w, -30.f), 20.f);
        *reinterpret_cast<float4*>(z + o) = make_float4(mu.x + expf(0.5f * lv0) * e.x, mu.y + expf(0.5f * lv1) * e.y, mu.z + expf(0.5f * lv2) * e.z,
                                                        mu.w + expf(0.5f * lv3) * e.w);
        if (mean_o) *reinterpret_cast<float4*>(mean_o + o) = mu;
        if (logvar_o) *reinterpret_cast<float4*>(logvar_o + o) = lr;
        acc += (mu.x * mu.x + expf(lv0) - 1.0f - lv0) + (mu.y * mu.y + expf(lv1) - 1.0f - lv1) + (mu.z * mu.z + expf(lv2) - 1.0f - lv2) +
               (mu.w * mu.w + expf(lv3) - 1.0f - lv3);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        kl[b] = 0.5f * t / (float)(rows * L);
    }
}
int posterior(const float* ml, const float* eps, float* mean_o, float* logvar_o, float* z, float* kl, int B, int rows, int L,
              hipStream_t st) {
    hipLaunchKernelGGL(posterior_kernel, dim3(B), dim3(1024), 0, st, ml, eps, mean_o, logvar_o, z, kl, rows, L);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- x_f32[m][n] = sum_k in[m][k]*W[n][k] + bias[n], tiny K (the 'proj' Linear latent_dim -> dim, :410)
__global__ __launch_bounds__(256) void small_k_linear_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out, int M,
                                                             int K, int N) {
    __shared__ float sx[8][64];
    const int m0 = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * K; i += 256) {
        const int r = i / K, c = i % K;
        sx[r][c] = (m0 + r < M) ? in[(int64_t)(m0 + r) * K + c] : 0.f;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float acc[8];
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = bv;
        const float* w = W + (int64_t)n * K;
        for (int c = 0; c < K; ++c) {
            const float wv = w[c];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += sx[r][c] * wv;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (m0 + r < M) out[(int64_t)(m0 + r) * N + n] = acc[r];
    }
}
int small_k_linear(const float* in, const float* W, const float* bias, float* out, int M, int K, int N, hipStream_t st) {
    RALD_CHECK(K >= 1 && K <= 64, "small_k_linear: K must be in [1,64]");
    hipLaunchKernelGGL(small_k_linear_kernel, dim3(cdiv(M, 8)), dim3(256), 0, st, in, W, bias, out, M, K, N);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
