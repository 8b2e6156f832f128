// Backward of head-dim-64 attention O = softmax(q k^T * scale) v (CrossAttention, models_radar_generation.py:66-75) for the training
// step: two launches instead of the 14 of the unfused form (four [nq x nk] fp32 score-shaped GEMMs, two
// element-wise passes, three transposes, three thin GEMMs), and nothing score-shaped reaches memory.
//
//   attn_bwd_q_kernel   one wave = 32 queries (on the LANES, as in attention.hip), a workgroup of four streams the keys twice:
//                       pass 1  S^T = K.Q^T -> running max / sum -> lse (exp2 units); delta = rowsum(dO * O)
//                       pass 2  P^T = exp2(S^T c - lse), dP^T = V.dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T.dS^T
//                       (S^T / dS^T accumulators are directly the B operands of the next MFMA: keys on the accumulator rows)
//   attn_bwd_kv_kernel  one wave = 32 keys on the lanes, streams the queries once with lse / delta from the first kernel:
//                       S = Q.K^T, P = exp2(S c - lse[q]), dP = dO.V^T, dS = P (dP - delta[q]),
//                       dV^T += dO^T.P, dK^T += Q^T.dS           (queries on the accumulator rows)
//
// Operand plumbing is attention.hip's: 64-row x 128-byte tiles written by LDS-DMA with the XOR swizzle applied to the source
// address; row fragments (A operand = tile rows) from a "natural" image, column gathers (A operand = the tile transposed) through
// ds_read_b64_tr_b16 from a second image of the same tile with the transposed-read swizzle.  P and dS are rounded to bf16 for the
// MFMAs, as the unfused form stored them.  Gradient parity: tests/test_gpu_train_ops.py against autograd in fp32.
#include "common.h"
#include "kernels.h"

namespace rald {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

namespace {
constexpr int TB = 64 * 128;                                   // one staged tile

__device__ __forceinline__ f32x16 mfma32(const bf16x8& x, const bf16x8& y, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}
// rows `row`, elements 16s + 8hf .. +7 of a natural image
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* img, int row, int s, int hf) {
    return *reinterpret_cast<const bf16x8*>(img + row * 128 + (((2 * s + hf) ^ ((row >> 1) & 7)) << 4));
}
// A operand = the tile transposed: lane (r, hf) gets column 32dt + r of rows row0 + 4hf + (0..3) and row0 + 8 + 4hf + (0..3)
// (row0 = 32u + 16s) - the order of a bf16 fragment cut from accumulator elements 8s .. 8s+7
__device__ __forceinline__ bf16x8 frag_col(const unsigned char* img, int row0, int dt, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int row = row0 + 4 * (gq >> 1) + qq;
    const int lch = 4 * dt + 2 * (gq & 1) + (pp >> 1);
    const unsigned char* p_lo = img + row * 128 + ((lch ^ ((row & 2) << 1)) << 4) + 8 * (pp & 1);
    const unsigned char* p_hi = img + (row + 8) * 128 + ((lch ^ (((row + 8) & 2) << 1)) << 4) + 8 * (pp & 1);
    const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_lo);
    const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_hi);
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, l4), __builtin_bit_cast(bf16x4, h4), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 cut8(const f32x16& v, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)v[8 * s + j];
    return f;
}
}  // namespace

// ------------------------------------------------------------------------------------------------- query side
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 3 * TB];     // [buf][K natural | K transposed-read | V natural]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5, lr = lane >> 3;
    const int nx = a.nq >> 7;
    const int bh = blockIdx.x / nx, qblk = blockIdx.x - bh * nx;
    const int h = bh % a.heads, b = bh / a.heads;
    const int q0 = (qblk * 4 + wave) * 32;
    const float c = a.scale * 1.4426950408889634f;

    const bf16 *gKn[2], *gKt[2], *gVn[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 8 * (wave + 4 * p) + lr;
        const int lcn = (lane & 7) ^ ((row >> 1) & 7), lct = (lane & 7) ^ ((row & 2) << 1);
        gKn[p] = a.K + (int64_t)b * a.sk + (int64_t)row * a.ldk + h * 64 + lcn * 8;
        gKt[p] = a.K + (int64_t)b * a.sk + (int64_t)row * a.ldk + h * 64 + lct * 8;
        gVn[p] = a.V + (int64_t)b * a.sv + (int64_t)row * a.ldv + h * 64 + lcn * 8;
    }
    auto stage = [&](int j0, int buf) {
        unsigned char* base = smem + buf * 3 * TB;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((glb_void*)(gKn[p] + (int64_t)j0 * a.ldk), (lds_void*)(base + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gKt[p] + (int64_t)j0 * a.ldk), (lds_void*)(base + TB + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gVn[p] + (int64_t)j0 * a.ldv), (lds_void*)(base + 2 * TB + (wave + 4 * p) * 1024), 16, 0, 0);
        }
    };
    const int ntiles = a.nk >> 6;
    stage(0, 0);

    bf16x8 qf[4], dof[4];
    float delta = 0.f;
    {
        const int64_t row = q0 + r;
        const bf16* qp = a.Q + (int64_t)b * a.sq + row * a.ldq + h * 64 + 8 * hf;
        const bf16* dp = a.dO + (int64_t)b * a.sdo + row * a.lddo + h * 64 + 8 * hf;
        const bf16* op = a.O + (int64_t)b * a.so + row * a.ldo + h * 64 + 8 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + 16 * s);
#pragma unroll
            for (int j = 0; j < 8; ++j) delta = fmaf((float)dof[s][j], (float)of[j], delta);
        }
        delta += __shfl_xor(delta, 32, 64);
    }

    float m = -1e30f, l = 0.f, lse = 0.f;
    f32x16 dq0, dq1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq0[i] = 0.f; dq1[i] = 0.f; }

    for (int it = 0; it < 2 * ntiles; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 1 < 2 * ntiles) stage(((it + 1) % ntiles) * 64, (it + 1) & 1);
        const unsigned char* sKn = smem + (it & 1) * 3 * TB;
        const unsigned char* sKt = sKn + TB;
        const unsigned char* sVn = sKn + 2 * TB;
        f32x16 st[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[u][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st[u] = mfma32(frag_row(sKn, 32 * u + r, s, hf), qf[s], st[u]);
        }
        if (it < ntiles) {                                      // pass 1: log-sum-exp of this lane's query (exp2 units)
            float mx = st[0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx * c);
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) ps += __builtin_amdgcn_exp2f(fmaf(st[u][i], c, -mn));
            l = l * __builtin_amdgcn_exp2f(m - mn) + ps;
            m = mn;
            if (it == ntiles - 1) {
                l += __shfl_xor(l, 32, 64);
                lse = m + __builtin_amdgcn_logf(l);             // v_log_f32 = log2
                if (hf == 0) {
                    a.lse[(int64_t)bh * a.nq + q0 + r] = lse;
                    a.delta[(int64_t)bh * a.nq + q0 + r] = delta;
                }
            }
            continue;
        }
        // pass 2
        f32x16 dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[u][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) dp[u] = mfma32(frag_row(sVn, 32 * u + r, s, hf), dof[s], dp[u]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[u][i], c, -lse));
                st[u][i] = p * (dp[u][i] - delta);              // dS^T (without the softmax scale: applied once at the end)
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsf = cut8(st[u], s);
                dq0 = mfma32(frag_col(sKt, 32 * u + 16 * s, 0, lane), dsf, dq0);
                dq1 = mfma32(frag_col(sKt, 32 * u + 16 * s, 1, lane), dsf, dq1);
            }
    }
    // dq{dt}[i] = dQ^T[d = 32dt + (i&3) + 8(i>>2) + 4hf][query q0 + r]
    bf16* out = a.dQ + (int64_t)b * a.sdq + (int64_t)(q0 + r) * a.lddq + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<bf16x4*>(out + 8 * g + 4 * hf) =
            pack4(dq0[4 * g] * a.scale, dq0[4 * g + 1] * a.scale, dq0[4 * g + 2] * a.scale, dq0[4 * g + 3] * a.scale);
        *reinterpret_cast<bf16x4*>(out + 32 + 8 * g + 4 * hf) =
            pack4(dq1[4 * g] * a.scale, dq1[4 * g + 1] * a.scale, dq1[4 * g + 2] * a.scale, dq1[4 * g + 3] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------------- key side
constexpr int KV_STAGE = 4 * TB + 512;                          // Q natural | Q transposed-read | dO natural | dO transposed-read | lse[64] | delta[64]
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_kv[];
    unsigned char* smem = smem_kv;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hf = lane >> 5, lr = lane >> 3;
    const int nx = (a.nk + 127) >> 7;
    const int bh = blockIdx.x / nx, kblk = blockIdx.x - bh * nx;
    const int h = bh % a.heads, b = bh / a.heads;
    int key0 = (kblk * 4 + wave) * 32;
    const bool active = key0 < a.nk;
    if (!active) key0 = a.nk - 32;
    const float c = a.scale * 1.4426950408889634f;

    const bf16 *gQn[2], *gQt[2], *gDn[2], *gDt[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 8 * (wave + 4 * p) + lr;
        const int lcn = (lane & 7) ^ ((row >> 1) & 7), lct = (lane & 7) ^ ((row & 2) << 1);
        gQn[p] = a.Q + (int64_t)b * a.sq + (int64_t)row * a.ldq + h * 64 + lcn * 8;
        gQt[p] = a.Q + (int64_t)b * a.sq + (int64_t)row * a.ldq + h * 64 + lct * 8;
        gDn[p] = a.dO + (int64_t)b * a.sdo + (int64_t)row * a.lddo + h * 64 + lcn * 8;
        gDt[p] = a.dO + (int64_t)b * a.sdo + (int64_t)row * a.lddo + h * 64 + lct * 8;
    }
    const float* grow = (wave == 0 ? a.lse : a.delta) + (int64_t)bh * a.nq + lane;     // waves 0 / 1 fetch the tile's lse / delta
    auto stage = [&](int i0, int buf) {
        unsigned char* base = smem + buf * KV_STAGE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((glb_void*)(gQn[p] + (int64_t)i0 * a.ldq), (lds_void*)(base + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gQt[p] + (int64_t)i0 * a.ldq), (lds_void*)(base + TB + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gDn[p] + (int64_t)i0 * a.lddo), (lds_void*)(base + 2 * TB + (wave + 4 * p) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gDt[p] + (int64_t)i0 * a.lddo), (lds_void*)(base + 3 * TB + (wave + 4 * p) * 1024), 16, 0, 0);
        }
        if (wave < 2) __builtin_amdgcn_global_load_lds((glb_void*)(grow + i0), (lds_void*)(base + 4 * TB + wave * 256), 4, 0, 0);
    };
    const int ntiles = a.nq >> 6;
    stage(0, 0);

    bf16x8 kf[4], vf[4];
    {
        const bf16* kp = a.K + (int64_t)b * a.sk + (int64_t)(key0 + r) * a.ldk + h * 64 + 8 * hf;
        const bf16* vp = a.V + (int64_t)b * a.sv + (int64_t)(key0 + r) * a.ldv + h * 64 + 8 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
            vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s);
        }
    }
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) stage((t + 1) * 64, (t + 1) & 1);
        const unsigned char* sQn = smem + (t & 1) * KV_STAGE;
        const unsigned char* sQt = sQn + TB;
        const unsigned char* sDn = sQn + 2 * TB;
        const unsigned char* sDt = sQn + 3 * TB;
        const float* sl = reinterpret_cast<const float*>(sQn + 4 * TB);
        const float* sd = sl + 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x16 s_, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s_[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s_ = mfma32(frag_row(sQn, 32 * u + r, s, hf), kf[s], s_);
                dp = mfma32(frag_row(sDn, 32 * u + r, s, hf), vf[s], dp);
            }
            // element i <-> query 32u + 8(i>>2) + 4hf + (i&3) of the tile, key key0 + r
            f32x16 pp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(sl + 32 * u + 8 * g + 4 * hf);
                const float4 d4 = *reinterpret_cast<const float4*>(sd + 32 * u + 8 * g + 4 * hf);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s_[4 * g + j], c, -lv[j]));
                    pp[4 * g + j] = p;
                    s_[4 * g + j] = p * (dp[4 * g + j] - dl[j]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = cut8(pp, s), dsf = cut8(s_, s);
                dv0 = mfma32(frag_col(sDt, 32 * u + 16 * s, 0, lane), pf, dv0);
                dv1 = mfma32(frag_col(sDt, 32 * u + 16 * s, 1, lane), pf, dv1);
                dk0 = mfma32(frag_col(sQt, 32 * u + 16 * s, 0, lane), dsf, dk0);
                dk1 = mfma32(frag_col(sQt, 32 * u + 16 * s, 1, lane), dsf, dk1);
            }
        }
    }
    if (!active) return;
    bf16* ok = a.dK + (int64_t)b * a.sdk + (int64_t)(key0 + r) * a.lddk + h * 64;
    bf16* ov = a.dV + (int64_t)b * a.sdv + (int64_t)(key0 + r) * a.lddv + h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<bf16x4*>(ok + 8 * g + 4 * hf) =
            pack4(dk0[4 * g] * a.scale, dk0[4 * g + 1] * a.scale, dk0[4 * g + 2] * a.scale, dk0[4 * g + 3] * a.scale);
        *reinterpret_cast<bf16x4*>(ok + 32 + 8 * g + 4 * hf) =
            pack4(dk1[4 * g] * a.scale, dk1[4 * g + 1] * a.scale, dk1[4 * g + 2] * a.scale, dk1[4 * g + 3] * a.scale);
        *reinterpret_cast<bf16x4*>(ov + 8 * g + 4 * hf) = pack4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
        *reinterpret_cast<bf16x4*>(ov + 32 + 8 * g + 4 * hf) = pack4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
    }
}

int attention_bwd_d64(const AttnBwdArgs& a, hipStream_t st) {
    RALD_CHECK(a.Q && a.K && a.V && a.O && a.dO && a.dQ && a.dK && a.dV && a.lse && a.delta, "attention_bwd: null argument");
    RALD_CHECK(a.nq > 0 && a.nk > 0 && a.heads > 0 && a.batch > 0, "attention_bwd: empty problem");
    RALD_CHECK(a.nq % 128 == 0 && a.nk % 64 == 0, "attention_bwd: nq must be a multiple of 128 and nk of 64");
    RALD_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddo % 8 == 0 && a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0,
               "attention_bwd: leading dimensions must be multiples of 8 elements (inputs) / 4 elements (outputs)");
    RALD_CHECK(a.ldq >= a.heads * 64 && a.ldk >= a.heads * 64 && a.ldv >= a.heads * 64 && a.ldo >= a.heads * 64 && a.lddo >= a.heads * 64 &&
               a.lddq >= a.heads * 64 && a.lddk >= a.heads * 64 && a.lddv >= a.heads * 64, "attention_bwd: leading dimension smaller than heads * 64");
    RALD_CHECK(((uintptr_t)a.Q % 16 == 0) && ((uintptr_t)a.K % 16 == 0) && ((uintptr_t)a.V % 16 == 0) && ((uintptr_t)a.O % 16 == 0) && ((uintptr_t)a.dO % 16 == 0) &&
               ((uintptr_t)a.dQ % 8 == 0) && ((uintptr_t)a.dK % 8 == 0) && ((uintptr_t)a.dV % 8 == 0) && ((uintptr_t)a.lse % 16 == 0) && ((uintptr_t)a.delta % 16 == 0),
               "attention_bwd: pointer alignment (16 bytes for inputs and scratch, 8 for outputs)");
    RALD_CHECK(a.sq % 8 == 0 && a.sk % 8 == 0 && a.sv % 8 == 0 && a.so % 8 == 0 && a.sdo % 8 == 0 && a.sdq % 4 == 0 && a.sdk % 4 == 0 && a.sdv % 4 == 0,
               "attention_bwd: batch strides must keep the row alignment");
    RALD_CHECK((int64_t)a.batch * a.heads * (a.nq / 128) < (1ll << 31) && (int64_t)a.batch * a.heads * cdiv(a.nk, 128) < (1ll << 31), "attention_bwd: too many workgroups");
    static bool attr_set = false;
    if (!attr_set) {
        RALD_HIP(hipFuncSetAttribute((const void*)attn_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KV_STAGE));
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(a.batch * a.heads * (a.nq / 128)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(a.batch * a.heads * cdiv(a.nk, 128)), dim3(256), 2 * KV_STAGE, st, a);
    RALD_HIP(hipGetLastError());
    return 0;
}

}  // namespace rald
