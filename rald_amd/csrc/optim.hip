// Optimizer step of the training loop on flat parameter storage (SURVEY.md 8f rank 1;
// engine_generation.py:96-110, utils/misc.py:249-269, main_generation.py:161):
//   clip_grad_norm_(parameters, max_norm)  ->  torch.optim.AdamW step  ->  update_ema(rate 0.999)
// The reference walks ~560 parameter tensors three times (norm, AdamW foreach, EMA foreach).  Here the
// model's parameters, gradients, both Adam moments and the EMA copy are five flat fp32 arrays with the
// same layout, so the whole step is two streaming passes:
//   1. sum of squares of the gradient (fp64 partials, one atomicAdd per workgroup);
//   2. one fused pass: g *= clip coefficient (read from DEVICE memory: no host sync), decoupled weight
//      decay, both moments, bias-corrected update, EMA - 20 B read + 16 B written per parameter.
// HBM-bound: 36 B/parameter.  Arithmetic follows torch's single-tensor AdamW op by op.
#include "common.h"
#include "kernels.h"

namespace rald {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += (double)v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// clip_grad_norm_: total_norm = sqrt(sum sq); coef = min(max_norm / (total_norm + 1e-6), 1);
// the extra pre-scale (1/world for a SUM all-reduce, 1/loss-scale) is folded in.  out[0] = total_norm
// (of the pre-scaled gradient), out[1] = the factor the fused pass multiplies gradients with.
__global__ void clip_coef_kernel(const double* __restrict__ sumsq, float pre_scale, float max_norm, float* __restrict__ out) {
    const float total = (float)sqrt(*sumsq) * pre_scale;
    float coef = 1.0f;
    if (max_norm > 0.f) {
        coef = max_norm / (total + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
    }
    out[0] = total;
    out[1] = coef * pre_scale;
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; float* ema;
    const float* gscale;                 // device scalar multiplied into the gradient (may be null)
    int64_t n;
    float decay, omb1, beta2, omb2, eps, step_size, bc2_sqrt, ema_rate, om_ema;   // python-double scalars rounded to fp32 once, like torch
    int write_grad;                      // also store the scaled gradient back (clip_grad_norm_ scales in place)
};

__device__ __forceinline__ void adam_one(const AdamArgs& a, float gs, float& p, float& g, float& m, float& v, float& e) {
#pragma clang fp contract(off)                                // torch rounds after every op
    g = g * gs;
    p = p * a.decay;                                         // param.mul_(1 - lr * weight_decay)
    m = m + a.omb1 * (g - m);                                // exp_avg.lerp_(grad, 1 - beta1), weight < 0.5 form
    v = v * a.beta2 + a.omb2 * g * g;                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;       // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p + (-a.step_size) * (m / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (a.ema) e = e * a.ema_rate + p * a.om_ema;            // targ.mul_(rate).add_(src, alpha=1-rate)
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
    const float gs = a.gscale ? a.gscale[0] : 1.0f;
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4*>(a.p)[i], g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
        float4 e = a.ema ? reinterpret_cast<float4*>(a.ema)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        adam_one(a, gs, p.x, g.x, m.x, v.x, e.x);
        adam_one(a, gs, p.y, g.y, m.y, v.y, e.y);
        adam_one(a, gs, p.z, g.z, m.z, v.z, e.z);
        adam_one(a, gs, p.w, g.w, m.w, v.w, e.w);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.ema) reinterpret_cast<float4*>(a.ema)[i] = e;
        if (a.write_grad) reinterpret_cast<float4*>(const_cast<float*>(a.g))[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float p = a.p[i], g = a.g[i], m = a.m[i], v = a.v[i], e = a.ema ? a.ema[i] : 0.f;
        adam_one(a, gs, p, g, m, v, e);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (a.ema) a.ema[i] = e;
        if (a.write_grad) const_cast<float*>(a.g)[i] = g;
    }
}

// update_ema on its own (the reference calls it every iteration, also on accumulation steps)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float rate, float om_rate) {
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ema[i] = ema[i] * rate + p[i] * om_rate;
}

static int stream_grid(int64_t n4) {
    const int64_t want = (n4 + 255) / 256;
    return (int)(want < 1 ? 1 : (want > 256 * 8 ? 256 * 8 : want));   // 8 workgroups per CU, grid-stride
}

int optim_grad_sumsq(const float* g, int64_t n, double* out, hipStream_t st) {
    RALD_CHECK(n >= 0 && ((uintptr_t)g % 16 == 0), "optim_grad_sumsq: gradient buffer must be 16-byte aligned");
    RALD_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
    if (n == 0) return 0;
    hipLaunchKernelGGL(sumsq_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, st, g, n, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

int optim_clip_coef(const double* sumsq, float pre_scale, float max_norm, float* out2, hipStream_t st) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, st, sumsq, pre_scale, max_norm, out2);
    RALD_HIP(hipGetLastError());
    return 0;
}

int optim_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gscale, double lr, double beta1,
                    double beta2, double eps, double wd, int64_t step, double ema_rate, int write_grad, hipStream_t st) {
    RALD_CHECK(n >= 0 && step >= 1, "optim_adamw_ema: n >= 0 and step >= 1 expected");
    RALD_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) % 16 == 0, "optim_adamw_ema: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.gscale = gscale; a.n = n;
    a.decay = (float)(1.0 - lr * wd); a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.ema_rate = (float)ema_rate; a.om_ema = (float)(1.0 - ema_rate); a.write_grad = write_grad;
    // python-float (double) scalars in torch: bias_correction = 1 - beta ** step, step_size = lr / bc1
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

int optim_ema(float* ema, const float* p, int64_t n, double rate, hipStream_t st) {
    RALD_CHECK(n >= 0, "optim_ema: n >= 0 expected");
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n)), dim3(256), 0, st, ema, p, n, (float)rate, (float)(1.0 - rate));
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
