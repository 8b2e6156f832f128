// Query generation + refine on the device: the host numpy code that sits between
// `model.sample` and `vae.decode` in engine_generation.evaluate (:250-300):
//   * generate_query_points (utils/utils.py:147-175): uniform queries in the normalised box;
//   * the use_cart_query chain (engine_generation.py:251-256): inverse_norm_points(cart range) ->
//     cartesian2polar (dataset_preprocessor/lidar.py:49-55) -> norm_points (utils/utils.py:77-104) ->
//     remove_points_outside_fov (:106-112), all float64 in the reference, cast to float32 at the end;
//   * aug_query_helper (datasets/utils/query_helper.py:3-42) + norm_points: the refine-query jitter.
// The random draws are INPUTS (device arrays): the host mirror fills them either from numpy's global
// RNG in the reference's draw order (bit-identical queries) or from a device generator.
// All HBM-bound streaming work; arithmetic follows numpy's dtype promotion so results are bit-exact
// up to the last-ulp behaviour of atan2/asin.
#include "common.h"
#include "kernels.h"

namespace rald {

struct QRange {
    double off[3], scale[3], smax;     // (max+min)/2, (max-min)/2 per axis, max scale
    float foff[3], fscale[3];          // the same as float32 (numpy casts python scalars to the array dtype)
};

static QRange make_qrange(const double* r) {
    QRange q;
    q.smax = 0.0;
    for (int a = 0; a < 3; ++a) {
        q.off[a] = (r[3 + a] + r[a]) / 2;
        q.scale[a] = (r[3 + a] - r[a]) / 2;
        q.foff[a] = (float)q.off[a];
        q.fscale[a] = (float)q.scale[a];
        if (q.scale[a] > q.smax) q.smax = q.scale[a];
    }
    return q;
}

struct QBox { double lo[3], span[3]; };

// x_min/x_max of generate_query_points (utils/utils.py:157-169); iso overrides aniso like the reference
static QBox make_qbox(const QRange& q, int aniso, int iso) {
    QBox b;
    for (int a = 0; a < 3; ++a) {
        double lo = -1.0, hi = 1.0;
        (void)aniso;
        if (iso) { lo = -(q.scale[a] / q.smax); hi = q.scale[a] / q.smax; }
        b.lo[a] = lo;
        b.span[a] = hi - lo;
    }
    return b;
}

// np.random.uniform(lo, hi, n) = lo + (hi - lo) * u, float64, then .astype('float32')
__global__ void query_uniform_kernel(const double* __restrict__ u, int64_t n, QBox b, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) out[i * 3 + a] = (float)(b.lo[a] + b.span[a] * u[(int64_t)a * n + i]);
}

// the float64 chain of the use_cart_query branch for one query; returns the FoV verdict
__device__ __forceinline__ bool cart_query(const double* __restrict__ u, int64_t n, int64_t i, const QBox& b, const QRange& cart,
                                           const QRange& pol, int aniso, int iso, double* p) {
#pragma clang fp contract(off)
    double g[3], c[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = b.lo[a] + b.span[a] * u[(int64_t)a * n + i];
    if (aniso) for (int a = 0; a < 3; ++a) c[a] = g[a] * cart.scale[a] + cart.off[a];
    if (iso) for (int a = 0; a < 3; ++a) c[a] = g[a] * cart.smax + cart.off[a];
    const double r = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double r2d = 180.0 / 3.141592653589793238462643383279502884;
    const double pl[3] = {r, -(atan2(c[1], c[0]) * r2d), asin(c[2] / r) * r2d};
    p[0] = p[1] = p[2] = 0.0;
    if (aniso) for (int a = 0; a < 3; ++a) p[a] = (pl[a] - pol.off[a]) / pol.scale[a];
    if (iso) for (int a = 0; a < 3; ++a) p[a] = (pl[a] - pol.off[a]) / pol.smax;
    return p[0] > -1.0 && p[0] < 1.0 && p[1] > -1.0 && p[1] < 1.0 && p[2] > -1.0 && p[2] < 1.0;
}

constexpr int QB = 1024;

__global__ __launch_bounds__(256) void cart_count_kernel(const double* __restrict__ u, int64_t n, QBox b, QRange cart, QRange pol, int aniso,
                                                         int iso, int* __restrict__ counts) {
    __shared__ int sh[4];
    const int64_t base = (int64_t)blockIdx.x * QB;
    int c = 0;
    double p[3];
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        c += (i < n && cart_query(u, n, i, b, cart, pol, aniso, iso, p)) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void cart_scatter_kernel(const double* __restrict__ u, int64_t n, QBox b, QRange cart, QRange pol, int aniso,
                                                           int iso, const int* __restrict__ offsets, float* __restrict__ out) {
    __shared__ int wave_base[4];
    const int64_t base = (int64_t)blockIdx.x * QB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = offsets[blockIdx.x];
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        double p[3];
        const bool keep = i < n && cart_query(u, n, i, b, cart, pol, aniso, iso, p);
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_base[wave] = __popcll(m);
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wave_base[w];
        const int sub_total = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
        if (keep) {
            const int64_t o = (int64_t)running + wb + before;
            out[o * 3] = (float)p[0]; out[o * 3 + 1] = (float)p[1]; out[o * 3 + 2] = (float)p[2];
        }
        running += sub_total;
        __syncthreads();
    }
}

// norm_points on a float32 array (utils/utils.py:77-104): the anisotropic branch stays float32, the
// isotropic one subtracts a float64 offset array and is rounded on assignment into the float32 result
__device__ __forceinline__ void norm_point_f32(const QRange& q, int aniso, int iso, const float* p, float* out) {
#pragma clang fp contract(off)
    float r[3] = {0.f, 0.f, 0.f};
    if (aniso) for (int a = 0; a < 3; ++a) r[a] = (p[a] - q.foff[a]) / q.fscale[a];
    if (iso) for (int a = 0; a < 3; ++a) r[a] = (float)(((double)p[a] - q.off[a]) / q.smax);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

__global__ void norm_points_kernel(const float* __restrict__ in, int64_t n, QRange q, int aniso, int iso, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {in[i * 3], in[i * 3 + 1], in[i * 3 + 2]};
    norm_point_f32(q, aniso, iso, p, out + i * 3);
}

struct RefineArgs {
    const float* pred; int64_t n_pred, aug_num;
    const int64_t* sel; const int64_t* scales; const double* u;
    double lo[3], hi[3], voxel[3];
    QRange q; int aniso, iso, normalise;
    float* out;
};

// aug_query_helper (query_helper.py:3-42) row j, then norm_points if asked
__global__ void refine_kernel(RefineArgs a) {
#pragma clang fp contract(off)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.aug_num) return;
    float p[3];
    if (j < a.n_pred) {                                           // the helper points themselves come first
        for (int c = 0; c < 3; ++c) p[c] = a.pred[j * 3 + c];
    } else {
        const int64_t g = j - a.n_pred, s = a.sel[g];
        const double sc = (double)a.scales[g];
        for (int c = 0; c < 3; ++c) {
            const double bias = (a.u[g * 3 + c] * 2 - 1) * (a.voxel[c] * sc);
            double v = (double)a.pred[s * 3 + c] + bias;
            v = fmin(fmax(v, a.lo[c]), a.hi[c]);                  // np.clip
            p[c] = (float)v;                                      // assignment into the float32 result
        }
    }
    if (a.normalise) norm_point_f32(a.q, a.aniso, a.iso, p, a.out + j * 3);
    else for (int c = 0; c < 3; ++c) a.out[j * 3 + c] = p[c];
}

// ---- launchers ---------------------------------------------------------------------------------
int query_uniform(const double* u, int64_t n, const double* pc_range, int aniso, int iso, float* out, hipStream_t st) {
    RALD_CHECK(n >= 0 && (aniso || iso), "query_uniform: n >= 0 and one of norm_anisotropy / norm_isotropy expected");
    if (n == 0) return 0;
    const QBox b = make_qbox(make_qrange(pc_range), aniso, iso);
    hipLaunchKernelGGL(query_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, u, n, b, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

int query_uniform_cart(const double* u, int64_t n, const double* range_cart, const double* range_polar, int aniso, int iso, float* out,
                       int64_t* out_count, int* scratch, hipStream_t st) {
    RALD_CHECK(n >= 0 && n < (int64_t)1 << 30 && (aniso || iso), "query_uniform_cart: bad size or normalisation flags");
    if (n == 0) { RALD_HIP(hipMemsetAsync(out_count, 0, sizeof(int64_t), st)); return 0; }
    const QRange cart = make_qrange(range_cart), pol = make_qrange(range_polar);
    const QBox b = make_qbox(cart, aniso, iso);
    const int nblocks = (int)((n + QB - 1) / QB);
    hipLaunchKernelGGL(cart_count_kernel, dim3(nblocks), dim3(256), 0, st, u, n, b, cart, pol, aniso, iso, scratch);
    post_scan_counts(scratch, nblocks, out_count, st);
    hipLaunchKernelGGL(cart_scatter_kernel, dim3(nblocks), dim3(256), 0, st, u, n, b, cart, pol, aniso, iso, scratch, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

int query_norm_points(const float* in, int64_t n, const double* pc_range, int aniso, int iso, float* out, hipStream_t st) {
    RALD_CHECK(n >= 0, "query_norm_points: n >= 0 expected");
    if (n == 0) return 0;
    hipLaunchKernelGGL(norm_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, make_qrange(pc_range), aniso, iso, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

int query_refine(const float* pred, int64_t n_pred, int64_t aug_num, const int64_t* sel, const int64_t* scales, const double* u,
                 const double* pc_range, const double* voxel, int aniso, int iso, int normalise, float* out, hipStream_t st) {
    RALD_CHECK(n_pred >= 0 && aug_num >= 0, "query_refine: sizes must be non-negative");
    RALD_CHECK(n_pred >= aug_num || n_pred > 0, "query_refine: cannot augment an empty set of helper points");
    RALD_CHECK(n_pred >= aug_num || (sel && scales && u), "query_refine: random draws required when n_pred < aug_num");
    if (aug_num == 0) return 0;
    RefineArgs a;
    a.pred = pred; a.n_pred = n_pred < aug_num ? n_pred : aug_num; a.aug_num = aug_num;
    a.sel = sel; a.scales = scales; a.u = u;
    for (int c = 0; c < 3; ++c) { a.lo[c] = pc_range[c]; a.hi[c] = pc_range[3 + c]; a.voxel[c] = voxel[c]; }
    a.q = make_qrange(pc_range); a.aniso = aniso; a.iso = iso; a.normalise = normalise; a.out = out;
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((aug_num + 255) / 256)), dim3(256), 0, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
