// Decode post-processing on the device: what engine_generation.evaluate does on
// the host after vae.decode (engine_generation.py:229-243, :283-322; utils/utils.py:50-75, :116-142;
// dataset_preprocessor/lidar.py:57-63):
//   * occupancy threshold + ORDER-PRESERVING compaction of the positive queries (np.where(logits > 0)),
//     fused with inverse normalisation and polar (r, az deg, el deg) -> cartesian;
//   * Chamfer distance = 0.5*mean_gt(min_pred ||.||) + 0.5*mean_pred(min_gt ||.||): the reference
//     queries a cKDTree point by point in a Python loop; here an exact brute-force nearest neighbour
//     in fp64 (10^9 pairs is ~0.1 ms of fp64 FMA on MI355X), LDS-tiled;
//   * accuracy / IoU of the thresholded logits against labels.
// HBM-bound integer/byte work except the Chamfer kernel (fp64 VALU-bound).
#include "common.h"
#include "kernels.h"

namespace rald {

constexpr int CB = 1024;   // elements per compaction block

// pass 1: positives per block of 1024 queries
__global__ __launch_bounds__(256) void count_pos_kernel(const float* __restrict__ logits, int64_t Q, float thr, int* __restrict__ counts) {
    __shared__ int sh[4];
    const int64_t base = (int64_t)blockIdx.x * CB;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        c += (i < Q && logits[i] > thr) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// pass 2: exclusive scan of the block counts (one workgroup; nblocks <= 1M), total -> *total
__global__ __launch_bounds__(1024) void scan_counts_kernel(int* __restrict__ counts, int nblocks, int64_t* __restrict__ total) {
    __shared__ int sh[1024];
    int carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? counts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) counts[i] = carry + sh[threadIdx.x] - v;
        const int blk_total = sh[1023];
        __syncthreads();
        carry += blk_total;
    }
    if (threadIdx.x == 0) *total = carry;
}

struct PostXform {
    float sx, sy, sz, ox, oy, oz;    // anisotropic scale / offset (fp32, as numpy computes them on float32 arrays)
    float smax;                      // isotropic scale
    double dox, doy, doz;            // isotropic offsets are a float64 array in the reference
    int aniso, iso, view_cone;
};

__device__ __forceinline__ void xform_point(const PostXform& t, float px, float py, float pz, float* out) {
#pragma clang fp contract(off)                               // numpy rounds after every op: no FMA contraction here
    float x = 0.f, y = 0.f, z = 0.f;                         // inverse_norm_points (utils/utils.py:50-75)
    if (t.aniso) {
        x = px * t.sx + t.ox;                                // two roundings, like numpy
        y = py * t.sy + t.oy;
        z = pz * t.sz + t.oz;
    }
    if (t.iso) {
        x = (float)((double)(px * t.smax) + t.dox);
        y = (float)((double)(py * t.smax) + t.doy);
        z = (float)((double)(pz * t.smax) + t.doz);
    }
    if (t.view_cone) {                                       // polar2cartesian (lidar.py:57-63), fp32 like numpy on float32
        const float d2r = 0.017453292519943295f;
        const float az = -(y * d2r), el = z * d2r;
        const float ce = cosf(el);
        const float r = x;
        x = r * ce * cosf(az);
        y = r * ce * sinf(az);
        z = r * sinf(el);
    }
    out[0] = x; out[1] = y; out[2] = z;
}

// pass 3: scatter the positives in index order, transformed
__global__ __launch_bounds__(256) void scatter_pos_kernel(const float* __restrict__ logits, const float* __restrict__ queries, int64_t Q,
                                                          float thr, const int* __restrict__ offsets, PostXform t,
                                                          float* __restrict__ out_pts, int64_t* __restrict__ out_idx) {
    __shared__ int wave_base[4];
    const int64_t base = (int64_t)blockIdx.x * CB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = offsets[blockIdx.x];
    for (int k = 0; k < 4; ++k) {                            // 4 sub-blocks of 256 consecutive queries, in order
        const int64_t i = base + k * 256 + threadIdx.x;
        const bool pos = i < Q && logits[i] > thr;
        const unsigned long long m = __ballot(pos);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_base[wave] = __popcll(m);
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wave_base[w];
        const int sub_total = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
        if (pos) {
            const int64_t o = (int64_t)running + wb + before;
            xform_point(t, queries[i * 3], queries[i * 3 + 1], queries[i * 3 + 2], out_pts + o * 3);
            if (out_idx) out_idx[o] = i;
        }
        running += sub_total;
        __syncthreads();
    }
}

__global__ void xform_points_kernel(const float* __restrict__ in, int64_t n, PostXform t, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xform_point(t, in[i * 3], in[i * 3 + 1], in[i * 3 + 2], out + i * 3);
}

// ---- Chamfer: sum_i min_j ||a_i - b_j||  (fp64), b staged through LDS in tiles of 1024 points
__global__ __launch_bounds__(256) void nn_dist_sum_kernel(const float* __restrict__ a, int64_t na, const float* __restrict__ b, int64_t nb,
                                                          double* __restrict__ sum) {
    __shared__ double sb[3][1024];
    __shared__ double red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double ax = 0, ay = 0, az = 0;
    if (i < na) { ax = a[i * 3]; ay = a[i * 3 + 1]; az = a[i * 3 + 2]; }
    double best = 1e300;
    for (int64_t j0 = 0; j0 < nb; j0 += 1024) {
        const int cnt = (int)((nb - j0) < 1024 ? (nb - j0) : 1024);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            sb[0][j] = b[(j0 + j) * 3]; sb[1][j] = b[(j0 + j) * 3 + 1]; sb[2][j] = b[(j0 + j) * 3 + 2];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const double dx = ax - sb[0][j], dy = ay - sb[1][j], dz = az - sb[2][j];
            const double d = dx * dx + dy * dy + dz * dz;
            best = d < best ? d : best;
        }
    }
    double v = (i < na) ? sqrt(best) : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
}

// ---- accuracy / IoU per sample (engine_generation.py:229-243): pred = logits >= 0
__global__ __launch_bounds__(256) void iou_kernel(const float* __restrict__ logits, const float* __restrict__ labels, int64_t Q,
                                                  float* __restrict__ acc, float* __restrict__ iou) {
    __shared__ float red[3][4];
    const int b = blockIdx.x;
    float eq = 0.f, inter = 0.f, uni = 0.f;
    for (int64_t i = threadIdx.x; i < Q; i += 256) {
        const float p = logits[(int64_t)b * Q + i] >= 0.f ? 1.f : 0.f;
        const float l = labels[(int64_t)b * Q + i];
        eq += (p == l) ? 1.f : 0.f;
        inter += p * l;
        uni += (p + l) > 0.f ? 1.f : 0.f;
    }
    eq = wave_sum(eq); inter = wave_sum(inter); uni = wave_sum(uni);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = eq; red[1][threadIdx.x >> 6] = inter; red[2][threadIdx.x >> 6] = uni; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float e = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float in = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float un = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        acc[b] = e / (float)Q;
        iou[b] = in * 1.0f / un + 1e-5f;
    }
}

static PostXform make_xform(const double* pc_range, int aniso, int iso, int view_cone) {
    PostXform t;
    // offsets / scales are Python floats (double) in the reference; the anisotropic branch multiplies a
    // float32 array by them (result float32), the isotropic branch adds a float64 offset array.
    const double xo = ((double)pc_range[3] + pc_range[0]) / 2, yo = ((double)pc_range[4] + pc_range[1]) / 2, zo = ((double)pc_range[5] + pc_range[2]) / 2;
    const double xs = ((double)pc_range[3] - pc_range[0]) / 2, ys = ((double)pc_range[4] - pc_range[1]) / 2, zs = ((double)pc_range[5] - pc_range[2]) / 2;
    t.sx = (float)xs; t.sy = (float)ys; t.sz = (float)zs; t.ox = (float)xo; t.oy = (float)yo; t.oz = (float)zo;
    double mx = xs > ys ? xs : ys; mx = mx > zs ? mx : zs;
    t.smax = (float)mx; t.dox = xo; t.doy = yo; t.doz = zo;
    t.aniso = aniso; t.iso = iso; t.view_cone = view_cone;
    return t;
}

void post_scan_counts(int* counts, int nblocks, int64_t* total, hipStream_t st) {
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, counts, nblocks, total);
}

int post_scratch_ints(int64_t Q) { return (int)((Q + CB - 1) / CB) + 8; }

int post_occupied_points(const float* logits, const float* queries, int64_t Q, const double* pc_range_host, int aniso, int iso,
                         int view_cone, float thr, float* out_pts, int64_t* out_idx, int64_t* out_count, int* scratch, hipStream_t st) {
    RALD_CHECK(logits && queries && out_pts && out_count && scratch && pc_range_host && Q >= 1, "post_occupied_points: bad argument");
    RALD_CHECK(Q <= (int64_t)1 << 30, "post_occupied_points: too many queries");
    const int nblocks = (int)((Q + CB - 1) / CB);
    const PostXform t = make_xform(pc_range_host, aniso, iso, view_cone);
    hipLaunchKernelGGL(count_pos_kernel, dim3(nblocks), dim3(256), 0, st, logits, Q, thr, scratch);
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, scratch, nblocks, out_count);
    hipLaunchKernelGGL(scatter_pos_kernel, dim3(nblocks), dim3(256), 0, st, logits, queries, Q, thr, scratch, t, out_pts, out_idx);
    RALD_HIP(hipGetLastError());
    return 0;
}

int post_transform_points(const float* in, int64_t n, const double* pc_range_host, int aniso, int iso, int view_cone, float* out, hipStream_t st) {
    RALD_CHECK(in && out && pc_range_host && n >= 1, "post_transform_points: bad argument");
    const PostXform t = make_xform(pc_range_host, aniso, iso, view_cone);
    hipLaunchKernelGGL(xform_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, t, out);
    RALD_HIP(hipGetLastError());
    return 0;
}

// sums[0] = sum over a of min-dist to b, sums[1] = sum over b of min-dist to a (device doubles, zeroed here)
int post_chamfer_sums(const float* a, int64_t na, const float* b, int64_t nb, double* sums, hipStream_t st) {
    RALD_CHECK(a && b && sums && na >= 1 && nb >= 1, "post_chamfer_sums: empty point set");
    RALD_HIP(hipMemsetAsync(sums, 0, 16, st));
    hipLaunchKernelGGL(nn_dist_sum_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, a, na, b, nb, sums);
    hipLaunchKernelGGL(nn_dist_sum_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, b, nb, a, na, sums + 1);
    RALD_HIP(hipGetLastError());
    return 0;
}

int post_iou(const float* logits, const float* labels, int B, int64_t Q, float* acc, float* iou, hipStream_t st) {
    RALD_CHECK(logits && labels && acc && iou && B >= 1 && Q >= 1, "post_iou: bad argument");
    hipLaunchKernelGGL(iou_kernel, dim3(B), dim3(256), 0, st, logits, labels, Q, acc, iou);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald

namespace rald {

// ---- radar cube preprocessing (datasets/aligned_coloradar/Coloradar_dataset.py:432-475, process_radar_data):
// raw [B][R][A][E][Craw] (intensity dB, doppler, ..., validity mask LAST) -> out [B][R][tA][tE][2]
//   ch0 = clip(intensity, 0, max_i) / max_i ; ch1 = doppler * mask (/ max_dopp), then bilinear upsampling
//   over (A, E) with align_corners=True (F.interpolate, :465-470).  One thread per output element pair.
__global__ void radar_cube_prepare_kernel(const float* __restrict__ raw, float* __restrict__ out, int64_t n_out, int A, int E, int Craw,
                                          int tA, int tE, int norm_i, float max_i, int norm_d, float max_d) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int te = (int)(i % tE);
    const int ta = (int)((i / tE) % tA);
    const int64_t br = i / ((int64_t)tE * tA);                       // b*R + r
    // source coordinates, align_corners=True: src = dst * (in-1)/(out-1)   (torch upsample_bilinear2d)
    const float ra = tA > 1 ? (float)(A - 1) / (float)(tA - 1) : 0.f;
    const float re = tE > 1 ? (float)(E - 1) / (float)(tE - 1) : 0.f;
    const float a1r = ra * ta, e1r = re * te;
    const int a1 = (int)a1r, e1 = (int)e1r;
    const int a1p = a1 < A - 1 ? 1 : 0, e1p = e1 < E - 1 ? 1 : 0;
    const float a1l = a1r - a1, a0l = 1.f - a1l, e1l = e1r - e1, e0l = 1.f - e1l;
    float v[2][2][2];                                                // [da][de][channel]
#pragma unroll
    for (int da = 0; da < 2; ++da)
#pragma unroll
        for (int de = 0; de < 2; ++de) {
            const float* p = raw + ((br * A + (a1 + da * a1p)) * E + (e1 + de * e1p)) * Craw;
            float in = 0.f;
            if (norm_i) in = fminf(fmaxf(p[0], 0.f), max_i) / max_i;
            float dp = p[1] * p[Craw - 1];
            if (norm_d) dp = dp / max_d;
            v[da][de][0] = in; v[da][de][1] = dp;
        }
#pragma unroll
    for (int c = 0; c < 2; ++c)
        out[i * 2 + c] = a0l * (e0l * v[0][0][c] + e1l * v[0][1][c]) + a1l * (e0l * v[1][0][c] + e1l * v[1][1][c]);
}

int radar_cube_prepare(const float* raw, int B, int R, int A, int E, int Craw, int tA, int tE, int norm_i, float max_i, int norm_d,
                       float max_d, float* out, hipStream_t st) {
    RALD_CHECK(raw && out && B >= 1 && R >= 1 && A >= 1 && E >= 1 && Craw >= 3 && tA >= 1 && tE >= 1, "radar_cube_prepare: bad argument");
    RALD_CHECK(max_i > 0.f && max_d > 0.f, "radar_cube_prepare: normalisation constants must be positive");
    const int64_t n_out = (int64_t)B * R * tA * tE;
    hipLaunchKernelGGL(radar_cube_prepare_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, raw, out, n_out, A, E, Craw, tA,
                       tE, norm_i, max_i, norm_d, max_d);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
