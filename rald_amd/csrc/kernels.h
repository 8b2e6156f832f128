// Internal launch interface of the kernel files (not part of the C-ABI; include/rald_hip.h is).
#pragma once
#include <cstdlib>
#include <vector>

#include "common.h"

namespace rald {

// ---------------------------------------------------------------- gemm.hip
enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID = 2, EPI_GEGLU = 3,
       EPI_SOFTMAX64 = 4,
       EPI_F16S = 5 };        // C_fp16 = 2^-6 (alpha*acc + bias), saturating: split-K slabs for reduce_resid_ln(part_f16) (norm.hip)   // C_bf16 = softmax over every aligned group of 64 output columns of alpha*acc, in exp2 units (folded cross-attention, dit.hip)
struct GemmArgs {
    const bf16* A; int64_t lda; int64_t strideA;   // [batch][M][K] activations (K contiguous)
    const bf16* B; int64_t ldb; int64_t strideB;   // [batch][N][K] weights     (K contiguous)
    void* C;       int64_t ldc; int64_t strideC;   // bf16 or f32, see epilogue
    const float* bias;                             // [N] or nullptr
    int M, N, K, batch;
    float alpha;
    int alpha_ncols;                               // alpha applies to output columns n < alpha_ncols only (others use 1)
    int ablate;                                    // bit 64: non-temporal output stores.  Probe builds only (RALD_ABLATED): 1 = no DMA in the loop, 2 = no epilogue, 16 = no stores
    // optional inner batch (attention heads): grid z = batch * batch2, blockIdx.z = b1 * batch2 + b2,
    // operand offset = b1 * stride + b2 * stride2
    int batch2 = 1;
    int64_t strideA2 = 0, strideB2 = 0, strideC2 = 0;
    // optional MXFP8 form of a bf16 / GEGLU output INSTEAD of C: e4m3 [M][ldc] + e8m0 scales [M][ncols/32] (batch 1 only)
    unsigned char *out8 = nullptr, *outs = nullptr;
};
#ifdef __HIPCC__
__device__ __forceinline__ void gemm_batch_offsets(const GemmArgs& a, int bz, int64_t& oa, int64_t& ob, int64_t& oc) {
    const int b1 = bz / a.batch2, b2 = bz - b1 * a.batch2;
    oa = (int64_t)b1 * a.strideA + (int64_t)b2 * a.strideA2;
    ob = (int64_t)b1 * a.strideB + (int64_t)b2 * a.strideB2;
    oc = (int64_t)b1 * a.strideC + (int64_t)b2 * a.strideC2;
}
#endif
int gemm_nt(const GemmArgs& a, int epi, hipStream_t st);
// diagnostic: lanes that clamped an fp16 slab value since the last reset (EPI_F16S here, the per-head slabs in attn_small.hip); blocking
int f16_saturation_gemm(unsigned* count, bool reset);
int f16_saturation_attn(unsigned* count, bool reset);
// EPI_F16S exists on the LDS-DMA engines only: true when gemm_nt would run this shape on one of them (else use EPI_F32)
inline bool gemm_f16s_ok(int M, int N, int batch) {
    const int64_t w128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch, w64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch;
    return w128 >= 192 || w64 <= 256;
}
inline GemmArgs gemm_args(const bf16* A, int64_t lda, const bf16* B, int64_t ldb, void* C, int64_t ldc,
                          const float* bias, int M, int N, int K) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.strideA = 0; g.B = B; g.ldb = ldb; g.strideB = 0;
    g.C = C; g.ldc = ldc; g.strideC = 0; g.bias = bias; g.M = M; g.N = N; g.K = K; g.batch = 1; g.alpha = 1.f; g.alpha_ncols = 1 << 30; g.ablate = 0;
    return g;
}

// ---------------------------------------------------------------- gemm_ln.hip
// x[M][512] += A[M][K].W[512][K]^T + bias ;  h = LN(x) * (add_one + g[s]) + b[s]  (s = row / rows_per_group * gstride)
struct GemmLnArgs {
    const bf16* A; int64_t lda;
    const bf16* W; int64_t ldw;
    const float* bias;
    float* x;                 // [M][512] fp32, updated in place
    bf16* h;                  // [M][512] bf16 out (skipped when h8 is set)
    unsigned char* h8 = nullptr;   // optional MXFP8 form of h instead: e4m3 [M][512] ...
    unsigned char* hs = nullptr;   // ... + e8m0 scales [M][16]
    int nt_io = 1;                 // non-temporal residual / output traffic (RALD_NT_STORE=0 turns it off for A/B runs)
    // MXFP8 operands instead of A / W when A8 is set: e4m3 [M][K], [512][K] + e8m0 scales [..][K/32]; lda/ldw in elements
    const unsigned char *A8 = nullptr, *SA = nullptr, *W8 = nullptr, *SW = nullptr;
    const float* g; const float* b; int64_t gstride; int rows_per_group; float add_one, eps;
    int M, K;
    // optional per-group weights (bf16 operands): rows [i*w_rows, (i+1)*w_rows) multiply W + i*strideW (w_rows a multiple of 128)
    int64_t strideW = 0; int w_rows = 1 << 30;
};
int gemm_resid_ln(const GemmLnArgs& a, hipStream_t st);
// true when the fused kernel beats GEMM + separate LayerNorm (measured on MI355X, tools/bench_resid_ln.py and
// whole-NFE sweeps: wins from M = 16384 on for K = 512 and K = 2048, +1 % per NFE at B = 32; at M = 8192 it loses)
inline bool gemm_resid_ln_pays(int M, int K = 512) { (void)K; return M >= 16384; }

// radar_train.hip: out [B*D*H*W][32] bf16 = the 27-neighbourhood (zero outside the volume) of channel 0 of cube [B][D][H][W][cube_ch], 5 zero pads
int patches27(const float* cube, int cube_ch, bf16* out, int B, int D, int H, int Wd, hipStream_t st);
// ---------------------------------------------------------------- gemm_tn.hip
// C[n1][n2] += sum_m A[m][n1] . B[m][n2] (fp32, atomics); colsum (optional) [n1] += sum_m A[m][n1].  Weight / bias gradient of a Linear.
int gemm_tn(const bf16* A, int64_t lda, const bf16* B, int64_t ldb, float* C, int64_t ldc, float* colsum, int M, int N1, int N2, hipStream_t st,
            float* workspace = nullptr, int64_t workspace_floats = 0);
int64_t gemm_tn_workspace_floats(int M, int N1, int N2);
// the same with B = the virtual patch matrix of a 3x3x3 Conv3d over channels-last x: dW [Cout][Cin][27] += ..., dbias += column sums of dy
int conv3d_wgrad_tn(const bf16* dy, const bf16* x, float* dW, float* dbias, int B, int ID, int IH, int IW, int Cin, int Cout, int stride, int pad,
                    hipStream_t st, float* workspace = nullptr, int64_t workspace_floats = 0);
int64_t conv3d_wgrad_workspace_floats(int B, int ID, int IH, int IW, int Cin, int Cout, int stride, int pad);

// ---------------------------------------------------------------- norm.hip
// out_bf16[m][c] = LN(x[m])[c] * (add_one + g[s][c]) + b[s][c],  s = (m / rows_per_group) * gstride
int layernorm_mod(const float* x, bf16* out, int M, int D, const float* g, const float* b,
                  int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st);
// x_f32[m][n] = coef[s].c_in * sum_k xin[m][k] * W[n][k]      (proj_in fused with EDM c_in)
int proj_in(const float* xin, const float* W, float* x, int M, int C, int D, const float* coef,
            int coef_stride, int rows_per_group, hipStream_t st);
// D_x[m][c] = c_skip * xin[m][c] + c_out * sum_k LN_affine(x[m])[k] * Wout[c][k]
int final_norm_proj(const float* x, const float* gamma, const float* beta, const float* Wout,
                    const float* xin, float* out, int M, int D, int C, const float* coef,
                    int coef_stride, int rows_per_group, hipStream_t st, const bf16* Whl = nullptr);
// out [hi | lo][n] bf16 with w = hi + lo (to ~2^-17): the pre-split proj_out weight that final_norm_proj's large-M form takes as Whl
int split_hi_lo(const float* w, bf16* out, int n, hipStream_t st);

// ---------------------------------------------------------------- attention.hip
struct AttnArgs {
    const bf16* Q;  int64_t ldq,  strideQ;     // Q [b][i][h*64+d]
    const bf16* K;  int64_t ldk,  strideK;     // K [b][j][h*64+d]
    const bf16* Vt; int64_t ldvt, strideVt;    // Vt[b][h*64+d][j]  (keys contiguous, zero padded to 32)
    // alternative V operand, row-major like K (V[b][j][h*64+d]): set V and leave Vt null; needs nk % 64 == 0.  Read through
    // ds_read_b64_tr_b16, so a fused q|k|v projection can feed the kernel without a transposed copy of V.
    const bf16* V = nullptr; int64_t ldv = 0, strideV = 0;
    // key split (few queries x many keys, e.g. 512 latents x 10 000 points at batch 1): ksplit > 1 workgroups share a query block,
    // each takes a contiguous range of key tiles and writes unnormalised partials to `part` (attention_split_scratch_bytes),
    // a second kernel combines them.  0 / 1 = off; attention_pick_ksplit() suggests a value.
    int ksplit = 1; float* part = nullptr;
    bf16* O;        int64_t ldo,  strideO;     // O [b][i][h*64+d]
    int nq, nk, heads, batch;
    int k_rows;                                // rows allocated per batch in K (>= round_up(nk,64): the tail tile reads them)
    float scale;                               // softmax scale; ignored when q_prescaled
    int q_prescaled;                           // Q already multiplied by scale*log2(e) (done for free in the producing GEMM's epilogue)
    // fp16 form (the folded set-encoder attention, ae_encode.hip): K and V point at fp16 data, the queries are fp32 in Qf (same
    // ldq / strideQ, in elements), converted on load.  Needs q_prescaled and the row-major V.
    int f16 = 0; const float* Qf = nullptr;
    int hsk = 64;                              // column offset between the heads' K / V slices; 0 = every head reads the same 64 columns
    int v_padded = 0;                          // row-major V: rows nk .. k_rows-1 are finite (zero), so nk need not be a multiple of 64
};
int attention_d64(const AttnArgs& a, hipStream_t st);
int attention_pick_ksplit(int nq, int nk, int heads, int batch);
inline int64_t attention_split_scratch_bytes(int ksplit, int nq, int heads, int batch) { return (int64_t)ksplit * batch * heads * nq * 66 * 4; }

// ---------------------------------------------------------------- attn_bwd.hip (training: gradients of the attention above)
// dQ, dK, dV of O = softmax(Q K^T scale) V per head; every operand bf16 [b][row][h*64 + d] with its row and batch strides (column
// slices of fused buffers are fine).  lse / delta: fp32 scratch of batch*heads*nq elements each (written, then read).
struct AttnBwdArgs {
    const bf16 *Q, *K, *V, *O, *dO;
    int64_t ldq, sq, ldk, sk, ldv, sv, ldo, so, lddo, sdo;
    bf16 *dQ, *dK, *dV;
    int64_t lddq, sdq, lddk, sdk, lddv, sdv;
    float *lse, *delta;
    int nq, nk, heads, batch;
    float scale;
};
int attention_bwd_d64(const AttnBwdArgs& a, hipStream_t st);

// ---------------------------------------------------------------- attn_small.hip (<= 2048 rows: fused per-head sub-blocks)
// part[h][row][512] = (softmax(q_h k_h^T) v_h) . Wo[:, 64h:64h+64]^T  for the fused q|k|v buffer of a self-attention
int attn_self_proj(const bf16* qkv, int64_t ld, const bf16* Wo, float* part, int NL, int heads, int batch, hipStream_t st, bool part_f16 = false);
// part[h][row][512] = (softmax(to_q(hin)_h Kc_h^T) Vc_h) . Wo[:, 64h:64h+64]^T against 64 cached condition tokens
int xattn_q2_proj(const bf16* hin, const bf16* Wq, const bf16* Kc, int64_t ldk, int64_t strideK, const bf16* Vt, int64_t ldvt, int64_t strideVt,
                  const bf16* Wo, float* part, int M, int NL, int heads, int n_keys, float qscale, hipStream_t st, bool part_f16 = false);
// x[M][512] += bias + sum_s part[s]; optionally h = LN(x) * (add_one + g) + b  (norm.hip; deterministic order)
// (part_f16: the slabs are fp16 scaled by 2^-6, as attn_self_proj / xattn_q2_proj write them with part_f16; S = 8 only)
int reduce_resid_ln(const float* part, int S, int64_t part_stride, const float* bias, float* x, bf16* h, int M, const float* g, const float* b,
                    int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st, bool part_f16 = false);
// row count up to which the fused small-batch sub-blocks are used (measured on MI355X, tools/sweep_nfe.py: per NFE at B = 1 / 2 / 4
// 1.41 / 1.70 / 2.59 ms fused against 1.64 / 1.89 / 2.27 ms unfused - from 2048 rows on the plain kernels fill the chip)
inline bool small_m_fused(int M, int NL, int heads, int D, int n_keys) { return M <= 1024 && NL == 512 && heads == 8 && D == 512 && n_keys == 64; }

// ---------------------------------------------------------------- small.hip
enum { ACT_NONE = 0, ACT_SILU = 1 };
// out[s][n] = act(sum_k in[s][k] * W[n][k] + bias[n]); fp32 throughout, S small (t-embed path)
int skinny_linear(const float* in, const float* W, const float* bias, float* out, int S, int N, int K,
                  int act, hipStream_t st);
// pe[s][0:128] = cos(c_noise[s]*f_i), pe[s][128:256] = sin(..)   (PositionalEmbedding)
int positional_embedding(const float* c_noise, float* pe, int S, int channels, hipStream_t st);
int cast_f32_bf16(const float* in, bf16* out, int64_t n, hipStream_t st);
// diagnostic: fills the LDS of every CU with NaN patterns (see small.hip)
int poison_lds(hipStream_t st);
// dst_bf16[map(r)][c] = src_f32[r][c] ; map==nullptr -> identity; ld_dst >= cols (pad zero-filled by caller)
int pack_rows_bf16(const float* src, bf16* dst, int rows, int cols, int64_t ld_dst, const int* rowmap, hipStream_t st);
int scale_f32(const float* in, float* out, float s, int64_t n, hipStream_t st);
// Heun/Euler updates of edm_sampler (models_radar_generation.py:265-273)
int heun_euler(const float* x_hat, const float* denoised, float t_hat, float t_next, float* d_cur,
               float* x_next, int64_t n, hipStream_t st);
int heun_correct(const float* x_hat, const float* x_euler, const float* denoised, const float* d_cur,
                 float t_hat, float t_next, float* x_next, int64_t n, hipStream_t st);

// ---------------------------------------------------------------- ae_kernels.hip
int point_features(const float* pts, const float* basis, bf16* feat, int64_t n, hipStream_t st);
int softmax_rows(const float* S, int64_t ld_s, bf16* P, int64_t ld_p, int rows, int n, hipStream_t st);
int softmax_dot(const float* S, const float* u, float* out, int64_t rows, int nkeys, int rows_per_batch, float c0, hipStream_t st);
int softmax_dot_generic(const float* S, const float* u, float* out, int64_t rows, int nkeys, int rows_per_batch, float c0, hipStream_t st);
int ln_dot(const float* x, const float* gamma, const float* beta, const float* w, float* out, int M, int D, hipStream_t st);
int add_bcast_cast(const float* a, const float* d, bf16* out, int64_t per_batch, int batch, hipStream_t st);
int posterior(const float* ml, const float* eps, float* mean_o, float* logvar_o, float* z, float* kl, int B, int rows, int L, hipStream_t st);
int small_k_linear(const float* in, const float* W, const float* bias, float* out, int M, int K, int N, hipStream_t st);

#ifdef __HIPCC__
// y[r][lane] = sum_c xs[r][c] * T[c*64 + lane] for the 4 rows a 4-wave workgroup holds (normalised) in LDS: wave w sums the K-quarter
// [w*D/4, (w+1)*D/4) for ALL 4 rows - each table element is fetched once per workgroup instead of once per row (one wave per row read the
// whole 128-KiB table per row: 64 MB of L2 traffic for 512 rows) - and the partials meet in LDS.  Returns row w's 64 outputs on wave w.
template <int D>
__device__ __forceinline__ float project4_rows(const float (*xs)[D], const float* __restrict__ T, float (*red)[4][64], int lane, int w) {
    constexpr int KQ = D / 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = w * KQ; c < (w + 1) * KQ; c += 32) {
        float t[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) t[j] = T[(c + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(xs[r][c + j], t[j], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][r][lane] = acc[r];
    __syncthreads();
    return (red[0][w][lane] + red[1][w][lane]) + (red[2][w][lane] + red[3][w][lane]);
}
#endif

// ---------------------------------------------------------------- ae_encode.hip (folded encoder attentions)
int ae_embed_factor(int d, const float* Wpe, const float* bpe, std::vector<double>& Wc, std::vector<double>& R);
int ae_encode_tables(int d, int I, int M, int heads, bool mixq, const float* Wpe, const float* bpe, const float* d_lat, const float* mng,
                     const float* mnb, const float* mWq, const float* mWkv, const float* mWo, const float* mbo, const float* lat,
                     const float* Wqp, const float* bqp, const float* cg, const float* cb, const float* cWq, const float* cWkv,
                     const float* cWo, const float* cbo, std::vector<float>& Rf, std::vector<float>& Q1, std::vector<float>& T4,
                     std::vector<float>& X0, std::vector<float>& T1, std::vector<float>& T3, std::vector<float>& c3);
int ae_enc_features(const float* pc, const float* basis, const float* Rf, void* F, void* G, int B, int P, int Pp, hipStream_t st);
int ae_enc_qproj(const float* xin, const float* X0, float* x, const float* gamma, const float* beta, const float* T1, float* Qo, int rows, int M,
                 int d, hipStream_t st);
// ---------------------------------------------------------------- ae_decode.hip (streaming query decoder)
// per-sample decoder context: [ fp16 image M x 128 B | u: M floats | inv_scale + 3 pad floats ]
inline int64_t ae_ctx_stride(int M) { return (int64_t)M * 128 + (int64_t)M * 4 + 16; }
int ae_decode_tables(int d, const float* Wq, const float* Wk, const float* ng, const float* nb, const float* Wpe, const float* bpe,
                     const float* wfold, std::vector<float>& t2aug, std::vector<unsigned short>& l_img);   // host, double precision
int ae_ctx_build(const float* x, const float* gamma, const float* beta, const float* t2aug, float* Yscratch, void* ctx, int B, int M, int d,
                 hipStream_t st);
int ae_decode_stream(const void* ctx, const unsigned short* l_img, const float* queries, float* out, const float* basis, int basis_diag,
                     int B, int64_t Q, int M, float c0, hipStream_t st, int nw = 0);

// ---------------------------------------------------------------- post.hip
int post_scratch_ints(int64_t Q);
void post_scan_counts(int* counts, int nblocks, int64_t* total, hipStream_t st);   // exclusive scan of per-block counts
int post_occupied_points(const float* logits, const float* queries, int64_t Q, const double* pc_range_host, int aniso, int iso,
                         int view_cone, float thr, float* out_pts, int64_t* out_idx, int64_t* out_count, int* scratch, hipStream_t st);
int post_transform_points(const float* in, int64_t n, const double* pc_range_host, int aniso, int iso, int view_cone, float* out, hipStream_t st);
int post_chamfer_sums(const float* a, int64_t na, const float* b, int64_t nb, double* sums, hipStream_t st);
int post_iou(const float* logits, const float* labels, int B, int64_t Q, float* acc, float* iou, hipStream_t st);
int radar_cube_prepare(const float* raw, int B, int R, int A, int E, int Craw, int tA, int tE, int norm_i, float max_i, int norm_d,
                       float max_d, float* out, hipStream_t st);

// ---------------------------------------------------------------- query.hip
int query_uniform(const double* u, int64_t n, const double* pc_range, int aniso, int iso, float* out, hipStream_t st);
int query_uniform_cart(const double* u, int64_t n, const double* range_cart, const double* range_polar, int aniso, int iso, float* out,
                       int64_t* out_count, int* scratch, hipStream_t st);
int query_norm_points(const float* in, int64_t n, const double* pc_range, int aniso, int iso, float* out, hipStream_t st);
int query_refine(const float* pred, int64_t n_pred, int64_t aug_num, const int64_t* sel, const int64_t* scales, const double* u,
                 const double* pc_range, const double* voxel, int aniso, int iso, int normalise, float* out, hipStream_t st);

// ---------------------------------------------------------------- gemm_fp8.hip (MXFP8: e4m3 + e8m0 per 32 K-elements)
struct Mx8Args {
    GemmArgs g;                                    // shapes / C / bias / alpha as for gemm_nt; g.A, g.B unused; lda/ldb/strides in bytes
    const unsigned char *A8, *B8;                  // [batch][M][K], [batch][N][K] e4m3
    const unsigned char *SA, *SB;                  // [batch][M][K/32], [batch][N][K/32] e8m0
    int64_t strideSA, strideSB;
};
int gemm_mx8(const Mx8Args& a, int epi, hipStream_t st);
int quantize_mx8(const void* in, int in_is_bf16, int64_t ld_in, unsigned char* q, int64_t ld_q, unsigned char* scales, int64_t rows, int K,
                 hipStream_t st);
// small-M split-K residual GEMM + (optional) LayerNorm: see norm.hip.  splitk_for() = 0 when it does not pay.
int resid_splitk_ln(const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, float* x, bf16* h, const float* g, const float* b,
                    int64_t gstride, int rows_per_group, float add_one, float eps, int M, int K, int splits, float* scratch, hipStream_t st);
inline int splitk_max_rows() {                      // RALD_SPLITK_MAXM: A/B switch for the row count up to which split-K pays
    static const int v = RALD_PROBE_ENV("RALD_SPLITK_MAXM", 4096);
    return v;
}
inline int splitk_for(int M, int K) { return (K >= 2048 && M <= splitk_max_rows()) ? 4 : 0; }   // [M,512] outputs: few 128x128 tiles
int layernorm_mod_mx8(const float* x, unsigned char* q, unsigned char* scales, int64_t rows, int D, const float* gam, const float* bet,
                      int64_t gstride, int rows_per_group, float add_one, float eps, hipStream_t st);

// ---------------------------------------------------------------- train_kernels.hip (backward building blocks)
struct TransposeArgs {
    const void* in; int64_t ld_in, stride_in, stride_in2;     // [batch][batch2][rows][cols] f32 or bf16
    bf16* out;      int64_t ld_out, stride_out, stride_out2;  // [batch][batch2][cols][rows] bf16
    int rows, cols, batch, batch2;
};
int transpose_rows(const TransposeArgs& a, int in_is_bf16, hipStream_t st);
int ln_mod_bwd(const float* x, const float* dh, const float* s, int64_t gstride, int rows_per_group, float add_one, float eps, int64_t rows,
               int D, float* dx, float* ds, float* dt, hipStream_t st, bf16* dx_bf16 = nullptr);
int geglu_fwd(const bf16* u, bf16* hid, int64_t M, int I, hipStream_t st);
int geglu_bwd(const bf16* u, const bf16* dhid, bf16* du, int64_t M, int I, hipStream_t st);
int colsum(const void* X, int is_bf16, int64_t ld, int64_t M, int N, float* out, hipStream_t st);
int row_lse(const float* S, int64_t rows, int cols, float scale, float* lse, hipStream_t st);
int rowdot_heads(const bf16* dO, const bf16* O, int64_t M, int heads, int nq, float* delta, hipStream_t st);
int attn_bwd_elem(const float* S, const float* dP, const float* lse, const float* delta, int64_t batch, int R, int Cc, int64_t vbatch_stride,
                  int vstride, float scale, int by_col, bf16* P, bf16* dS, hipStream_t st);

int sgemm_acc(const float* A, int64_t lda, int trans_a, const float* B, int64_t ldb, int trans_b, float* Cm, int64_t ldc, int M, int N, int K,
              float alpha, hipStream_t st);
int silu_fwd(const float* x, float* y, int64_t n, hipStream_t st);
int silu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st);
int edm_loss_grad(const float* F, const float* xn, const float* y, const float* coef, int64_t per_sample, int64_t total, float* dF, float* D_out,
                  double* loss, hipStream_t st);

// ---------------------------------------------------------------- radar.hip / radar_train.hip (encoder, op level)
int conv3d_igemm(const bf16* in, const bf16* w_packed, const float* bias, const float* resid, float* out, int B, int ID, int IH, int IW,
                 int Cin, int Cout, int stride, int pad, hipStream_t st, bf16* out_bf16 = nullptr);
int groupnorm_fwd(const float* x, const float* gamma, const float* beta, bf16* y, double* stats, int B, int S, int C, int swish, hipStream_t st);
int conv_in_fwd(const float* cube, int cube_ch, int Cin, const float* W, const float* bias, float* out, int B, int D, int H, int Wd, int Cout,
                hipStream_t st);
int conv_pack_weights(const float* W, bf16* out, int Cout, int Cin, int pad_to, int dgrad, hipStream_t st);
int pad_channels(const float* x, bf16* out, int64_t M, int C, int Cpad, hipStream_t st);
int zero_insert2(const float* dy, bf16* out, int B, int OD, int OH, int OW, int C, hipStream_t st);
int im2col_t(const bf16* x, bf16* out, int B, int ID, int IH, int IW, int C, int stride, int pad, int64_t m0, int nchunk, hipStream_t st);
int conv_in_wgrad(const float* cube, int cube_ch, const float* dy, int B, int D, int H, int Wd, int Cout, float* dW, hipStream_t st);
int groupnorm_bwd(const float* x, const double* stats, const float* gamma, const float* beta, const float* da, float* dx, float* dgamma, float* dbeta,
                  double* gsum_scratch, int B, int S, int C, int swish, int accumulate, hipStream_t st, bf16* dx_bf16 = nullptr, int da_is_bf16 = 0);
int64_t groupnorm_bwd_scratch_bytes(int B, int S, int C);   // what gsum_scratch of groupnorm_bwd must hold
int groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, bf16* y, int B, int S, int C, int swish, hipStream_t st);
int rowdot(const bf16* a, const bf16* b, int64_t M, int C, float* out, hipStream_t st);

// ---------------------------------------------------------------- optim.hip
int optim_grad_sumsq(const float* g, int64_t n, double* out, hipStream_t st);
int optim_clip_coef(const double* sumsq, float pre_scale, float max_norm, float* out2, hipStream_t st);
int optim_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gscale, double lr, double beta1,
                    double beta2, double eps, double wd, int64_t step, double ema_rate, int write_grad, hipStream_t st);
int optim_ema(float* ema, const float* p, int64_t n, double rate, hipStream_t st);

}  // namespace rald
