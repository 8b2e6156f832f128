// Backward-pass building blocks of the denoiser's transformer block ("backward kernels
// for K2-K5"): what autograd derives for models_radar_generation.py:35-169 (CrossAttention, GEGLU
// FeedForward, AdaLayerNorm, BasicTransformerBlock), as explicit HIP kernels around the NT GEMM engine.
//
// GEMM shapes.  The engine contracts over the contiguous dimension of both operands (C = A.B^T), so
//   dX = dY . W      ->  gemm_nt(A = dY [M,N],   B = W^T [K,N])          (weight transposed once per step)
//   dW = dY^T . X    ->  gemm_nt(A = dY^T [N,M], B = X^T [K,M])          (activation transposes: transpose_rows)
// Everything here is HBM-bound streaming work (transposes, LayerNorm / GEGLU / softmax backward, column
// sums); the FLOPs stay in the MFMA GEMMs.
#include "common.h"
#include "kernels.h"

namespace rald {

// ---- batched 2-D transpose with cast to bf16: in [rows][cols] (f32 or bf16) -> out [cols][rows] bf16 -------
// 64x64 tiles through LDS; 16-byte global accesses on both sides when the shapes allow (every use in the training
// step does: rows, cols, leading dimensions multiples of 8), element-wise otherwise.
template <typename T>
__global__ __launch_bounds__(256) void transpose_rows_kernel(TransposeArgs a) {
    __shared__ bf16 tile[64][72];                             // [row][col], 144-byte rows: 16-byte aligned, conflict-light
    const int b1 = blockIdx.z / a.batch2, b2 = blockIdx.z - b1 * a.batch2;
    const T* in = reinterpret_cast<const T*>(a.in) + (int64_t)b1 * a.stride_in + (int64_t)b2 * a.stride_in2;
    bf16* out = a.out + (int64_t)b1 * a.stride_out + (int64_t)b2 * a.stride_out2;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bool vec = (a.rows % 8 == 0) && (a.cols % 8 == 0) && (a.ld_in % 8 == 0) && (a.ld_out % 8 == 0) && (a.stride_in % 8 == 0) &&
                     (a.stride_in2 % 8 == 0) && (a.stride_out % 8 == 0) && (a.stride_out2 % 8 == 0) &&
                     ((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.out % 16 == 0);
    if (vec) {
        // load: thread -> (row = tid / 8 (+32), 8 consecutive columns)
        for (int p = 0; p < 2; ++p) {
            const int r = (threadIdx.x >> 3) + 32 * p, c = (threadIdx.x & 7) * 8;
            bf16x8 v;
            if (r0 + r < a.rows && c0 + c < a.cols) {
                const T* src = in + (int64_t)(r0 + r) * a.ld_in + c0 + c;
                if constexpr (sizeof(T) == 4) {
                    const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
                    v = bf16x8{(bf16)x.x, (bf16)x.y, (bf16)x.z, (bf16)x.w, (bf16)y.x, (bf16)y.y, (bf16)y.z, (bf16)y.w};
                } else {
                    v = *reinterpret_cast<const bf16x8*>(src);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(&tile[r][c]) = v;
        }
        __syncthreads();
        // store: thread -> (output row = input column c = tid / 8 (+32), 8 consecutive input rows)
        for (int p = 0; p < 2; ++p) {
            const int c = (threadIdx.x >> 3) + 32 * p, r = (threadIdx.x & 7) * 8;
            if (c0 + c < a.cols && r0 + r < a.rows) {
                bf16x8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = tile[r + i][c];
                *reinterpret_cast<bf16x8*>(out + (int64_t)(c0 + c) * a.ld_out + r0 + r) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < a.rows && c < a.cols) ? (bf16)(float)in[(int64_t)r * a.ld_in + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < a.cols && r < a.rows) out[(int64_t)c * a.ld_out + r] = tile[tx][i];
    }
}

int transpose_rows(const TransposeArgs& a, int in_is_bf16, hipStream_t st) {
    RALD_CHECK(a.rows > 0 && a.cols > 0 && a.batch > 0 && a.batch2 > 0 && a.in && a.out, "transpose_rows: bad arguments");
    RALD_CHECK((int64_t)a.batch * a.batch2 <= 65535, "transpose_rows: too many batches");
    dim3 grid(cdiv(a.cols, 64), cdiv(a.rows, 64), a.batch * a.batch2);
    if (in_is_bf16) hipLaunchKernelGGL(transpose_rows_kernel<bf16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(transpose_rows_kernel<float>, grid, dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- AdaLayerNorm / LayerNorm backward, D = 512 -------------------------------------------------------------
// forward: h = xhat * (add_one + s[g]) + t[g], xhat = (x - mean) * rstd.  Given dh:
//   dx += rstd * (gh - mean(gh) - xhat * mean(gh * xhat)),  gh = dh * (add_one + s[g])
//   ds[g] += sum_rows dh * xhat,   dt[g] += sum_rows dh
// One wave per row (the row stays in registers), 16 rows per workgroup (all in one group when
// rows_per_group % 16 == 0), column sums reduced through LDS, one atomicAdd per column per workgroup.
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dh, const float* __restrict__ s,
                                                         int64_t gstride, int rows_per_group, float add_one, float eps, int64_t rows,
                                                         float* __restrict__ dx, bf16* __restrict__ dxb, float* __restrict__ ds, float* __restrict__ dt) {
    constexpr int D = 512, RPW = 4;                          // 4 waves x 4 rows
    __shared__ float red[2][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float as[8], at[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) as[i] = at[i] = 0.f;
    const int64_t row0 = (int64_t)blockIdx.x * (4 * RPW);
    const int64_t grp = row0 / rows_per_group;               // host guarantees rows_per_group % 16 == 0
    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t row = row0 + wave * RPW + rr;
        if (row >= rows) break;
        const float4 a = *reinterpret_cast<const float4*>(x + row * D + lane * 8), b = *reinterpret_cast<const float4*>(x + row * D + lane * 8 + 4);
        const float4 c = *reinterpret_cast<const float4*>(dh + row * D + lane * 8), d = *reinterpret_cast<const float4*>(dh + row * D + lane * 8 + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const float g[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
        const float mean = wave_sum(sum) * (1.0f / D);
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] -= mean; var += v[i] * v[i]; }
        const float rstd = rsqrtf(wave_sum(var) * (1.0f / D) + eps);
        const float* sp = s + grp * gstride + lane * 8;
        float gh[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] *= rstd;                                     // xhat
            gh[i] = g[i] * (add_one + sp[i]);
            m1 += gh[i];
            m2 += gh[i] * v[i];
            as[i] += g[i] * v[i];
            at[i] += g[i];
        }
        m1 = wave_sum(m1) * (1.0f / D);
        m2 = wave_sum(m2) * (1.0f / D);
        float* o = dx + row * D + lane * 8;
        float4 o0 = *reinterpret_cast<float4*>(o), o1 = *reinterpret_cast<float4*>(o + 4);
        o0.x += rstd * (gh[0] - m1 - v[0] * m2); o0.y += rstd * (gh[1] - m1 - v[1] * m2);
        o0.z += rstd * (gh[2] - m1 - v[2] * m2); o0.w += rstd * (gh[3] - m1 - v[3] * m2);
        o1.x += rstd * (gh[4] - m1 - v[4] * m2); o1.y += rstd * (gh[5] - m1 - v[5] * m2);
        o1.z += rstd * (gh[6] - m1 - v[6] * m2); o1.w += rstd * (gh[7] - m1 - v[7] * m2);
        *reinterpret_cast<float4*>(o) = o0;
        *reinterpret_cast<float4*>(o + 4) = o1;
        if (dxb) {                                            // the bf16 copy the next weight-gradient / input-gradient GEMMs read
            bf16x8 ob;
            ob[0] = (bf16)o0.x; ob[1] = (bf16)o0.y; ob[2] = (bf16)o0.z; ob[3] = (bf16)o0.w;
            ob[4] = (bf16)o1.x; ob[5] = (bf16)o1.y; ob[6] = (bf16)o1.z; ob[7] = (bf16)o1.w;
            *reinterpret_cast<bf16x8*>(dxb + row * D + lane * 8) = ob;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][wave][lane * 8 + i] = as[i]; red[1][wave][lane * 8 + i] = at[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c % D;
        const float v = red[which][0][col] + red[which][1][col] + red[which][2][col] + red[which][3][col];
        atomicAdd((which ? dt : ds) + grp * gstride + col, v);
    }
}

int ln_mod_bwd(const float* x, const float* dh, const float* s, int64_t gstride, int rows_per_group, float add_one, float eps, int64_t rows,
               int D, float* dx, float* ds, float* dt, hipStream_t st, bf16* dx_bf16) {
    RALD_CHECK(D == 512, "ln_mod_bwd: D must be 512");
    RALD_CHECK((uintptr_t)dx_bf16 % 16 == 0, "ln_mod_bwd: the bf16 copy must be 16-byte aligned");
    RALD_CHECK(rows > 0 && rows_per_group > 0 && (rows_per_group % 16 == 0 || rows_per_group >= rows), "ln_mod_bwd: rows_per_group must be a multiple of 16");
    hipLaunchKernelGGL(ln_mod_bwd_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, x, dh, s, gstride, rows_per_group, add_one, eps, rows,
                       dx, dx_bf16, ds, dt);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- GEGLU (models_radar_generation.py:88-95), natural layout: u [M][2I] = [a | g], hid = a * gelu_erf(g) ---------
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16* __restrict__ u, bf16* __restrict__ hid, int64_t M, int I) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (idx >= M * I) return;
    const int64_t m = idx / I;
    const int c = (int)(idx % I);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(u + m * 2 * I + c), g = *reinterpret_cast<const bf16x8*>(u + m * 2 * I + I + c);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)((float)a[i] * gelu_erf((float)g[i]));
    *reinterpret_cast<bf16x8*>(hid + m * I + c) = o;
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16* __restrict__ u, const bf16* __restrict__ dhid, bf16* __restrict__ du, int64_t M, int I) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (idx >= M * I) return;
    const int64_t m = idx / I;
    const int c = (int)(idx % I);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(u + m * 2 * I + c), g = *reinterpret_cast<const bf16x8*>(u + m * 2 * I + I + c);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dhid + m * I + c);
    bf16x8 da, dg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float gv = (float)g[i], dv = (float)d[i];
        const float cdf = 0.5f * (1.0f + erff(gv * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * gv * gv);
        da[i] = (bf16)(dv * gv * cdf);                        // d/da (a * gelu(g))
        dg[i] = (bf16)(dv * (float)a[i] * (cdf + gv * pdf));  // gelu'(g) = Phi(g) + g * phi(g)
    }
    *reinterpret_cast<bf16x8*>(du + m * 2 * I + c) = da;
    *reinterpret_cast<bf16x8*>(du + m * 2 * I + I + c) = dg;
}

int geglu_fwd(const bf16* u, bf16* hid, int64_t M, int I, hipStream_t st) {
    RALD_CHECK(M > 0 && I > 0 && I % 8 == 0, "geglu_fwd: inner width must be a multiple of 8");
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)((M * I / 8 + 255) / 256)), dim3(256), 0, st, u, hid, M, I);
    RALD_HIP(hipGetLastError());
    return 0;
}
int geglu_bwd(const bf16* u, const bf16* dhid, bf16* du, int64_t M, int I, hipStream_t st) {
    RALD_CHECK(M > 0 && I > 0 && I % 8 == 0, "geglu_bwd: inner width must be a multiple of 8");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((M * I / 8 + 255) / 256)), dim3(256), 0, st, u, dhid, du, M, I);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- bias gradients: out[n] += sum_m X[m][n] (X f32 or bf16) ----------------------------------------------------
// A workgroup covers `cw` columns (cw = min(N, 256) when that divides 256, else 256) and `rows_per_block` rows: the
// 256 / cw row lanes stride over the rows, partials meet in LDS, one atomicAdd per column per workgroup (tall, narrow
// inputs - a 64-channel bias over 2 M voxels - would otherwise serialise on 64 addresses).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ld, int64_t M, int N, int cw, int rows_per_block,
                                                     float* __restrict__ out) {
    __shared__ float red[256];
    const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, nty = 256 / cw;
    const int c = blockIdx.x * cw + tx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float s = 0.f;
    if (c < N)
        for (int64_t r = r0 + ty; r < r1; r += nty) s += (float)X[r * ld + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && c < N) {
        for (int k = 1; k < nty; ++k) s += red[k * cw + tx];
        atomicAdd(out + c, s);
    }
}
int colsum(const void* X, int is_bf16, int64_t ld, int64_t M, int N, float* out, hipStream_t st) {
    RALD_CHECK(M > 0 && N > 0 && X && out, "colsum: bad arguments");
    const int cw = (N < 256 && 256 % N == 0) ? N : 256;
    const int col_blocks = cdiv(N, cw);
    int64_t rpb = (M * col_blocks + 1023) / 1024;           // ~1024 workgroups in all
    const int min_rows = 8 * (256 / cw);
    if (rpb < min_rows) rpb = min_rows;
    dim3 grid(col_blocks, (unsigned)((M + rpb - 1) / rpb));
    if (is_bf16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)X, ld, M, N, cw, (int)rpb, out);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, ld, M, N, cw, (int)rpb, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

// ---- attention backward, element-wise parts (the products are batched NT GEMMs) -----------------------------
// lse[r] = log sum_c exp(scale * S[r][c])  (one wave per row)
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ S, int64_t rows, int cols, float scale, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = S + row * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(p[c] * scale - mx);
    sum = wave_sum(sum);
    if (lane == 0) lse[row] = mx + __logf(sum);
}
// delta[b][h][q] = sum_d dO[m][h*64 + d] * O[m][h*64 + d], m = b*nq + q  (the [batch*heads][nq] layout of lse)
__global__ __launch_bounds__(256) void rowdot_heads_kernel(const bf16* __restrict__ dO, const bf16* __restrict__ O, int64_t M, int heads, int nq,
                                                           float* __restrict__ delta) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (m, h, octet)
    const int64_t mh = idx >> 3;
    if (mh >= M * heads) return;
    const int oct = (int)(idx & 7);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(dO + mh * 64 + oct * 8), b = *reinterpret_cast<const bf16x8*>(O + mh * 64 + oct * 8);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (oct == 0) {
        const int64_t m = mh / heads;
        const int h = (int)(mh - m * heads);
        const int64_t bb = m / nq;
        delta[(bb * heads + h) * nq + (m - bb * nq)] = s;
    }
}
// P = exp(scale*S - lse[i]), dS = P * (dP - delta[i]) * scale, i = row (by_col = 0) or column (by_col = 1: the
// transposed orientation S^T = K.Q^T, whose softmax normaliser belongs to the COLUMN's query).
// S, dP: [batch][R][C] f32; lse/delta: [batch][R or C] with element stride `vstride`.
__global__ __launch_bounds__(256) void attn_bwd_elem_kernel(const float* __restrict__ S, const float* __restrict__ dP, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, int R, int Cc, int64_t vbatch_stride, int vstride,
                                                            float scale, int by_col, bf16* __restrict__ P, bf16* __restrict__ dS, int64_t total) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= total) return;
    const int64_t per = (int64_t)R * Cc;
    const int64_t b = idx / per;
    const int64_t rem = idx - b * per;
    const int r = (int)(rem / Cc), c = (int)(rem % Cc);
    const float4 s = *reinterpret_cast<const float4*>(S + idx), d = *reinterpret_cast<const float4*>(dP + idx);
    const float sv[4] = {s.x, s.y, s.z, s.w}, dv[4] = {d.x, d.y, d.z, d.w};
    bf16x4 po, so;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t vi = b * vbatch_stride + (int64_t)(by_col ? c + i : r) * vstride;
        const float p = __expf(sv[i] * scale - lse[vi]);
        po[i] = (bf16)p;
        so[i] = (bf16)(p * (dv[i] - delta[vi]) * scale);
    }
    if (P) *reinterpret_cast<bf16x4*>(P + idx) = po;
    *reinterpret_cast<bf16x4*>(dS + idx) = so;
}

int row_lse(const float* S, int64_t rows, int cols, float scale, float* lse, hipStream_t st) {
    RALD_CHECK(rows > 0 && cols > 0, "row_lse: empty");
    hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, rows, cols, scale, lse);
    RALD_HIP(hipGetLastError());
    return 0;
}
int rowdot_heads(const bf16* dO, const bf16* O, int64_t M, int heads, int nq, float* delta, hipStream_t st) {
    RALD_CHECK(M > 0 && heads > 0 && nq > 0 && M % nq == 0, "rowdot_heads: M must be a multiple of nq");
    hipLaunchKernelGGL(rowdot_heads_kernel, dim3((unsigned)((M * heads * 8 + 255) / 256)), dim3(256), 0, st, dO, O, M, heads, nq, delta);
    RALD_HIP(hipGetLastError());
    return 0;
}
int attn_bwd_elem(const float* S, const float* dP, const float* lse, const float* delta, int64_t batch, int R, int Cc, int64_t vbatch_stride,
                  int vstride, float scale, int by_col, bf16* P, bf16* dS, hipStream_t st) {
    RALD_CHECK(batch > 0 && R > 0 && Cc > 0 && Cc % 4 == 0, "attn_bwd_elem: the column count must be a multiple of 4");
    const int64_t total = batch * R * Cc;
    hipLaunchKernelGGL(attn_bwd_elem_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, S, dP, lse, delta, R, Cc, vbatch_stride,
                       vstride, scale, by_col, P, dS, total);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald

// =================================================================================================
// Small fp32 pieces of the training step: the timestep-embedding MLP, the 72 AdaLN linears, proj_in /
// proj_out and their gradients have one dimension of 32..512 or a contraction over the batch - far too thin
// for the MFMA tile engine.  One generic FMA kernel covers them:
//   C[m][n] += alpha * sum_k A(m,k) * B(n,k),  A(m,k) = A[m*lda + k] or (trans_a) A[k*lda + m],
//                                              B(n,k) = B[n*ldb + k] or (trans_b) B[k*ldb + n]
// 64x64 tiles, 16-deep k chunks through LDS, 4x4 outputs per thread; long contractions are split over
// grid.z and combined with fp32 atomics.
// =================================================================================================
namespace rald {

__global__ __launch_bounds__(256) void sgemm_acc_kernel(const float* __restrict__ A, int64_t lda, int trans_a, const float* __restrict__ B, int64_t ldb,
                                                        int trans_b, float* __restrict__ Cm, int64_t ldc, int M, int N, int K, int k_chunk, float alpha,
                                                        int use_atomics) {
    __shared__ float sA[16][65], sB[16][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_lo = blockIdx.z * k_chunk, k_hi = k_lo + k_chunk < K ? k_lo + k_chunk : K;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            int kk, mm;
            if (trans_a) { mm = e & 63; kk = e >> 6; } else { kk = e & 15; mm = e >> 4; }     // contiguous index fastest
            const int m = m0 + mm, k = k0 + kk;
            sA[kk][mm] = (m < M && k < k_hi) ? (trans_a ? A[(int64_t)k * lda + m] : A[(int64_t)m * lda + k]) : 0.f;
            int nn;
            if (trans_b) { nn = e & 63; kk = e >> 6; } else { kk = e & 15; nn = e >> 4; }
            const int n = n0 + nn, kb = k0 + kk;
            sB[kk][nn] = (n < N && kb < k_hi) ? (trans_b ? B[(int64_t)kb * ldb + n] : B[(int64_t)n * ldb + kb]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[kk][ty * 4 + i]; b[i] = sB[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float* c = Cm + (int64_t)m * ldc + n;
            if (use_atomics) atomicAdd(c, alpha * acc[i][j]);
            else *c += alpha * acc[i][j];
        }
    }
}

int sgemm_acc(const float* A, int64_t lda, int trans_a, const float* B, int64_t ldb, int trans_b, float* Cm, int64_t ldc, int M, int N, int K,
              float alpha, hipStream_t st) {
    RALD_CHECK(M > 0 && N > 0 && K > 0 && A && B && Cm, "sgemm_acc: bad arguments");
    const int tiles = cdiv(M, 64) * cdiv(N, 64);
    int splits = 1;
    if (tiles < 256 && K > 256) {                      // too few tiles to fill the chip: split the contraction
        splits = cdiv(512, tiles);
        const int max_splits = cdiv(K, 128);
        if (splits > max_splits) splits = max_splits;
        if (splits > 1024) splits = 1024;
    }
    const int k_chunk = cdiv(cdiv(K, splits), 16) * 16;
    splits = cdiv(K, k_chunk);
    dim3 grid(cdiv(N, 64), cdiv(M, 64), splits);
    hipLaunchKernelGGL(sgemm_acc_kernel, grid, dim3(256), 0, st, A, lda, trans_a, B, ldb, trans_b, Cm, ldc, M, N, K, k_chunk, alpha, splits > 1 ? 1 : 0);
    RALD_HIP(hipGetLastError());
    return 0;
}

// silu(x) = x * sigmoid(x)  (models_radar_generation.py:217-219)
__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.0f + __expf(-v)); }
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = x[i], s = 1.0f / (1.0f + __expf(-v));
        dx[i] = dy[i] * s * (1.0f + v * (1.0f - s));
    }
}
int silu_fwd(const float* x, float* y, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    RALD_HIP(hipGetLastError());
    return 0;
}
int silu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, dy, dx, n);
    RALD_HIP(hipGetLastError());
    return 0;
}

// EDMLoss (models_radar_generation.py:283-295) + EDMPrecond.forward's output mix (:422-430), per element:
//   D = c_skip*xn + c_out*F;  loss += w*(D - y)^2 / numel;  dF = 2*w*c_out*(D - y) / numel
// coef[b] = {c_skip, c_out, w}.  Also writes D (optional).
__global__ __launch_bounds__(256) void edm_loss_kernel(const float* __restrict__ F, const float* __restrict__ xn, const float* __restrict__ y,
                                                       const float* __restrict__ coef, int64_t per_sample, int64_t total, float* __restrict__ dF,
                                                       float* __restrict__ D_out, double* __restrict__ loss) {
    __shared__ double sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double part = 0.0;
    if (i < total) {
        const float* c = coef + (i / per_sample) * 3;
        const float d = c[0] * xn[i] + c[1] * F[i];
        const float r = d - y[i];
        part = (double)(c[2] * r * r);
        dF[i] = 2.0f * c[2] * c[1] * r / (float)total;
        if (D_out) D_out[i] = d;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (sh[0] + sh[1] + sh[2] + sh[3]) / (double)total);
}
int edm_loss_grad(const float* F, const float* xn, const float* y, const float* coef, int64_t per_sample, int64_t total, float* dF, float* D_out,
                  double* loss, hipStream_t st) {
    RALD_CHECK(total > 0 && per_sample > 0 && total % per_sample == 0, "edm_loss_grad: bad sizes");
    RALD_HIP(hipMemsetAsync(loss, 0, sizeof(double), st));
    hipLaunchKernelGGL(edm_loss_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, F, xn, y, coef, per_sample, total, dF, D_out, loss);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
