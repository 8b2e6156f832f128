// extern "C" surface of librald_hip.so (include/rald_hip.h).  Thin: argument checks + dispatch.
#include <cstdlib>
#include <cstring>

#include "ae.h"
#include "dit.h"

using namespace rald;

struct rald_dit { Dit impl; };
struct rald_ae { Ae impl; };

extern "C" {

const char* rald_last_error(void) { return rald::last_error(); }
int rald_version(void) { return 2; }
int rald_build_flags(void) {
#ifdef RALD_PROBE
    return 1;
#else
    return 0;
#endif
}

int64_t rald_debug_f16_saturation_count(int32_t reset) {
    unsigned a = 0, b = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (f16_saturation_gemm(&a, reset != 0) || f16_saturation_attn(&b, reset != 0)) return -1;
    return (int64_t)a + (int64_t)b;
}

int rald_debug_poison_lds(void* stream) { return poison_lds((hipStream_t)stream); }

void rald_dit_default_config(rald_dit_config* c) {
    c->n_latents = 512; c->channels = 32; c->depth = 24; c->n_heads = 8; c->d_head = 64; c->t_channels = 256;
    c->context_dim = 512; c->n_cond_tokens = 64; c->with_radar_enc = 1; c->enc_hidden_ch = 64; c->enc_radar_ch = 16;
    c->radar_r = 128; c->radar_a = 64; c->radar_e = 32; c->sigma_data = 1.0f; c->qkv_dtype = 0;
}
int rald_dit_create(const rald_dit_config* cfg, rald_dit** out) {
    RALD_CHECK(cfg && out, "rald_dit_create: null argument");
    rald_dit* h = new rald_dit();
    h->impl.cfg = *cfg;
    int rc = h->impl.create();
    if (rc) { delete h; return rc; }
    *out = h;
    return 0;
}
void rald_dit_destroy(rald_dit* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    delete h;
}
int rald_dit_load_weight(rald_dit* h, const char* name, const float* data, int64_t nelem) {
    RALD_CHECK(h && name && data, "rald_dit_load_weight: null argument");
    return h->impl.load_weight(name, data, nelem);
}
int rald_dit_finalize(rald_dit* h) { RALD_CHECK(h, "null handle"); return h->impl.finalize(); }
int rald_dit_reserve(rald_dit* h, int32_t max_batch) {
    RALD_CHECK(h && max_batch >= 1, "rald_dit_reserve: bad argument");
    return h->impl.reserve(max_batch);
}
int64_t rald_dit_workspace_generation(const rald_dit* h) { return h ? h->impl.ws_generation : -1; }
int rald_dit_set_two_stream_min_batch(rald_dit* h, int32_t min_batch) {
    RALD_CHECK(h && min_batch >= 0, "rald_dit_set_two_stream_min_batch: bad argument");
    h->impl.split_min = min_batch;
    if (h->impl.ws_batch > 0) {                  // re-plan the workspace (the second half's buffers exist only when the split can happen)
        const int B = h->impl.ws_batch;
        h->impl.ws_batch = 0;
        return h->impl.reserve(B);
    }
    return 0;
}
int32_t rald_dit_two_stream_min_batch(const rald_dit* h) { return h ? h->impl.split_min : -1; }
int rald_dit_set_sigmas(rald_dit* h, const float* sigmas_host, int32_t n, void* stream) {
    RALD_CHECK(h && sigmas_host, "rald_dit_set_sigmas: null argument");
    return h->impl.set_sigmas(sigmas_host, n, (hipStream_t)stream);
}
int64_t rald_dit_cond_cache_bytes(const rald_dit* h, int32_t batch) { return h ? h->impl.cond_cache_bytes(batch) : -1; }
int rald_dit_encode_cond_tokens(rald_dit* h, const float* tokens, int32_t batch, void* cond_cache, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.encode_cond_tokens(tokens, batch, cond_cache, (hipStream_t)stream);
}
int rald_dit_encode_cond(rald_dit* h, const float* cube, int32_t batch, float* out_tokens, void* cond_cache, void* stream) {
    RALD_CHECK(h && cube && cond_cache && batch >= 1, "rald_dit_encode_cond: bad argument");
    return h->impl.encode_cond(cube, batch, out_tokens, cond_cache, (hipStream_t)stream);
}
int rald_dit_denoise(rald_dit* h, const float* x, int32_t batch, int32_t sigma_row, int32_t per_sample,
                     const void* cond_cache, float* out, int32_t raw_F, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.denoise(x, batch, sigma_row, per_sample, cond_cache, out, raw_F, (hipStream_t)stream);
}
int rald_dit_sample(rald_dit* h, const float* latents, int32_t batch, const void* cond_cache, int32_t num_steps,
                    float sigma_min, float sigma_max, float rho, float* out, void* stream) {
    RALD_CHECK(h && latents && cond_cache && out && batch >= 1, "rald_dit_sample: bad argument");
    return h->impl.sample(latents, batch, cond_cache, num_steps, sigma_min, sigma_max, rho, out, (hipStream_t)stream);
}

int rald_dit_profile_begin(rald_dit* h) { RALD_CHECK(h, "null handle"); h->impl.prof_mask = 0xf; return h->impl.profile_begin(); }
int rald_dit_profile_set_kinds(rald_dit* h, uint32_t kind_mask) {
    RALD_CHECK(h, "null handle");
    h->impl.prof_mask = kind_mask & 0xfu;
    return 0;
}
int rald_dit_profile_end(rald_dit* h, double* total_ms, int32_t* launches) {
    RALD_CHECK(h && total_ms && launches, "rald_dit_profile_end: null argument");
    int n = 0;
    int rc = h->impl.profile_end(total_ms, &n);
    *launches = n;
    return rc;
}

int rald_dit_profile_end_kinds(rald_dit* h, double* total_ms4, int32_t* launches4) {
    RALD_CHECK(h && total_ms4 && launches4, "rald_dit_profile_end_kinds: null argument");
    int n[Dit::PROF_KINDS];
    int rc = h->impl.profile_end_kinds(total_ms4, n);
    for (int k = 0; k < Dit::PROF_KINDS; ++k) launches4[k] = n[k];
    return rc;
}

// ---- autoencoder ---------------------------------------------------------------------------------
int rald_ae_create(const rald_ae_config* cfg, rald_ae** out) {
    RALD_CHECK(cfg && out, "rald_ae_create: null argument");
    rald_ae* h = new rald_ae();
    h->impl.cfg = *cfg;
    int rc = h->impl.create();
    if (rc) { delete h; return rc; }
    *out = h;
    return 0;
}
void rald_ae_destroy(rald_ae* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    delete h;
}
int rald_ae_load_weight(rald_ae* h, const char* name, const float* data, int64_t nelem) {
    RALD_CHECK(h && name && data, "rald_ae_load_weight: null argument");
    return h->impl.load_weight(name, data, nelem);
}
int rald_ae_finalize(rald_ae* h) { RALD_CHECK(h, "null handle"); return h->impl.finalize(); }
int rald_ae_encode(rald_ae* h, const float* pc, int32_t batch, const float* eps, float* out_mean, float* out_logvar,
                   float* out_z, float* out_kl, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.encode(pc, batch, eps, out_mean, out_logvar, out_z, out_kl, (hipStream_t)stream);
}
int64_t rald_ae_ctx_bytes(const rald_ae* h, int32_t batch) { return h ? h->impl.ctx_bytes(batch) : -1; }
int64_t rald_ae_workspace_generation(const rald_ae* h) { return h ? h->impl.ws_generation : -1; }
int rald_ae_decode_latents(rald_ae* h, const float* z, int32_t batch, void* ctx, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.decode_latents(z, batch, ctx, (hipStream_t)stream);
}
int rald_ae_decode_queries(rald_ae* h, const void* ctx, const float* queries, int32_t batch, int64_t n_queries,
                           float* out_logits, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.decode_queries(ctx, queries, batch, n_queries, out_logits, (hipStream_t)stream);
}

// tuning / test entry points of the streaming query decoder (ae_decode.hip)
int rald_op_ae_decode_queries_nw(rald_ae* h, const void* ctx, const float* queries, int32_t batch, int64_t n_queries, float* out_logits,
                                 int32_t waves_per_workgroup, void* stream) {
    RALD_CHECK(h, "null handle");
    return h->impl.decode_queries(ctx, queries, batch, n_queries, out_logits, (hipStream_t)stream, waves_per_workgroup);
}
int rald_op_ae_decode_tables(int32_t dim, const float* wq, const float* wk, const float* norm_w, const float* norm_b, const float* wpe,
                             const float* bpe, const float* wfold, float* t2aug_out, uint16_t* l_img_out) {
    RALD_CHECK(wq && wk && norm_w && norm_b && wpe && bpe && wfold && t2aug_out && l_img_out && dim >= 64, "rald_op_ae_decode_tables: bad argument");
    std::vector<float> t2;
    std::vector<unsigned short> li;
    RALD_TRY(rald::ae_decode_tables(dim, wq, wk, norm_w, norm_b, wpe, bpe, wfold, t2, li));
    memcpy(t2aug_out, t2.data(), t2.size() * 4);
    memcpy(l_img_out, li.data(), li.size() * 2);
    return 0;
}

// ---- standalone radar-spectrum encoder (RadarAutoencoder.encoder) -------------------------------
struct rald_radar { DeviceArena arena; Stager stager; RadarEncoder enc, dec; int R, A, E, cin, zc; bool dec_touched = false; };
int rald_radar_create(int32_t basic_channel, int32_t embed_dim, int32_t in_channels, int32_t R, int32_t A, int32_t E, rald_radar** out) {
    RALD_CHECK(out, "rald_radar_create: null argument");
    rald_radar* h = new rald_radar();
    h->R = R; h->A = A; h->E = E; h->cin = in_channels; h->zc = embed_dim;
    int rc = h->enc.create(basic_channel, embed_dim, R, A, E, 512, &h->arena, in_channels);
    if (!rc) rc = h->dec.create_decoder(basic_channel, embed_dim, 2, R, A, E, &h->arena);
    if (rc) { delete h; return rc; }
    *out = h;
    return 0;
}
void rald_radar_destroy(rald_radar* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    delete h;
}
int rald_radar_load_weight(rald_radar* h, const char* name, const float* data, int64_t nelem) {
    RALD_CHECK(h && name && data, "rald_radar_load_weight: null argument");
    return h->enc.load_weight(name, data, nelem, h->stager);
}
int rald_radar_load_decoder_weight(rald_radar* h, const char* name, const float* data, int64_t nelem) {
    RALD_CHECK(h && name && data, "rald_radar_load_decoder_weight: null argument");
    h->dec_touched = true;
    return h->dec.load_weight(name, data, nelem, h->stager);
}
int rald_radar_finalize(rald_radar* h) {
    RALD_CHECK(h, "null handle");
    std::string missing;
    RALD_CHECK(h->enc.all_loaded(&missing), "radar encoder: missing key '" + missing + "' (strict load)");
    // the decoder is optional (the generation path only encodes); once one of its tensors was loaded, all must be
    RALD_CHECK(!h->dec_touched || h->dec.all_loaded(&missing), "radar decoder: missing key '" + missing + "' (strict load)");
    return 0;
}
int rald_radar_decode(rald_radar* h, const float* z, int32_t batch, float* out_pred4, void* stream) {
    RALD_CHECK(h && z && out_pred4 && batch >= 1, "rald_radar_decode: bad argument");
    std::string missing;
    RALD_CHECK(h->dec.all_loaded(&missing), "radar decoder: weights not loaded ('" + missing + "')");
    return h->dec.decode(z, batch, out_pred4, (hipStream_t)stream);
}
int rald_radar_encode(rald_radar* h, const float* cube, int32_t batch, float* out_z, void* stream) {
    RALD_CHECK(h && cube && out_z && batch >= 1, "rald_radar_encode: bad argument");
    float* z = nullptr;
    RALD_TRY(h->enc.encode(cube, h->cin, batch, &z, (hipStream_t)stream));
    const size_t n = (size_t)batch * (h->R / 16) * (h->A / 16) * (h->E / 16) * h->zc;
    RALD_HIP(hipMemcpyAsync(out_z, z, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ---- decode post-processing -----------------------------------------------------
int64_t rald_post_scratch_bytes(int64_t n_queries) { return (int64_t)post_scratch_ints(n_queries) * 4; }
int rald_post_occupied_points(const float* logits, const float* queries, int64_t n_queries, const double* pc_range6_host,
                              int32_t norm_anisotropy, int32_t norm_isotropy, int32_t view_cone_mode, float threshold,
                              float* out_points, int64_t* out_index, int64_t* out_count, void* scratch, void* stream) {
    return post_occupied_points(logits, queries, n_queries, pc_range6_host, norm_anisotropy, norm_isotropy, view_cone_mode, threshold,
                                out_points, out_index, out_count, (int*)scratch, (hipStream_t)stream);
}
int rald_post_transform_points(const float* points, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy,
                               int32_t norm_isotropy, int32_t view_cone_mode, float* out_points, void* stream) {
    return post_transform_points(points, n, pc_range6_host, norm_anisotropy, norm_isotropy, view_cone_mode, out_points, (hipStream_t)stream);
}
int rald_post_chamfer_sums(const float* pred, int64_t n_pred, const float* gt, int64_t n_gt, double* out_sums2, void* stream) {
    return post_chamfer_sums(pred, n_pred, gt, n_gt, out_sums2, (hipStream_t)stream);
}
int rald_post_iou(const float* logits, const float* labels, int32_t batch, int64_t n_queries, float* out_accuracy, float* out_iou, void* stream) {
    return post_iou(logits, labels, batch, n_queries, out_accuracy, out_iou, (hipStream_t)stream);
}

int rald_radar_cube_prepare(const float* raw, int32_t batch, int32_t R, int32_t A, int32_t E, int32_t raw_channels, int32_t tgt_A,
                            int32_t tgt_E, int32_t norm_intensity, float max_intensity, int32_t norm_dopp, float max_dopp, float* out,
                            void* stream) {
    return radar_cube_prepare(raw, batch, R, A, E, raw_channels, tgt_A, tgt_E, norm_intensity, max_intensity, norm_dopp, max_dopp, out,
                              (hipStream_t)stream);
}

// ---- query generation + refine --------------------------------------------------
int rald_query_uniform(const double* u3n, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy, int32_t norm_isotropy,
                       float* out_queries, void* stream) {
    RALD_CHECK(pc_range6_host && (n == 0 || (u3n && out_queries)), "rald_query_uniform: null argument");
    return query_uniform(u3n, n, pc_range6_host, norm_anisotropy, norm_isotropy, out_queries, (hipStream_t)stream);
}
int rald_query_uniform_cart(const double* u3n, int64_t n, const double* pc_range_cart6_host, const double* pc_range6_host,
                            int32_t norm_anisotropy, int32_t norm_isotropy, float* out_queries, int64_t* out_count, void* scratch,
                            void* stream) {
    RALD_CHECK(pc_range_cart6_host && pc_range6_host && out_count && (n == 0 || (u3n && out_queries && scratch)),
               "rald_query_uniform_cart: null argument");
    return query_uniform_cart(u3n, n, pc_range_cart6_host, pc_range6_host, norm_anisotropy, norm_isotropy, out_queries, out_count,
                              (int*)scratch, (hipStream_t)stream);
}
int rald_query_norm_points(const float* points, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy, int32_t norm_isotropy,
                           float* out_points, void* stream) {
    RALD_CHECK(pc_range6_host && (n == 0 || (points && out_points)), "rald_query_norm_points: null argument");
    return query_norm_points(points, n, pc_range6_host, norm_anisotropy, norm_isotropy, out_points, (hipStream_t)stream);
}
int rald_query_refine(const float* helper_points, int64_t n_helper, int64_t aug_num, const int64_t* sel_index, const int64_t* aug_scales,
                      const double* u_bias, const double* pc_range6_host, const double* voxel_size3_host, int32_t norm_anisotropy,
                      int32_t norm_isotropy, int32_t normalise, float* out_points, void* stream) {
    RALD_CHECK(pc_range6_host && voxel_size3_host && (aug_num == 0 || (helper_points && out_points)), "rald_query_refine: null argument");
    return query_refine(helper_points, n_helper, aug_num, sel_index, aug_scales, u_bias, pc_range6_host, voxel_size3_host, norm_anisotropy,
                        norm_isotropy, normalise, out_points, (hipStream_t)stream);
}

// ---- optimizer step on flat parameter storage --------------------------------------
int rald_optim_grad_sumsq(const float* grads, int64_t n, double* out_sumsq, void* stream) {
    RALD_CHECK(out_sumsq && (n == 0 || grads), "rald_optim_grad_sumsq: null argument");
    return optim_grad_sumsq(grads, n, out_sumsq, (hipStream_t)stream);
}
int rald_optim_clip_coef(const double* sumsq, float pre_scale, float max_norm, float* out_norm_coef, void* stream) {
    RALD_CHECK(sumsq && out_norm_coef, "rald_optim_clip_coef: null argument");
    return optim_clip_coef(sumsq, pre_scale, max_norm, out_norm_coef, (hipStream_t)stream);
}
int rald_optim_adamw_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_params, int64_t n,
                         const float* grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                         double ema_rate, int32_t write_back_grads, void* stream) {
    RALD_CHECK(n == 0 || (params && grads && exp_avg && exp_avg_sq), "rald_optim_adamw_ema: null argument");
    return optim_adamw_ema(params, grads, exp_avg, exp_avg_sq, ema_params, n, grad_scale, lr, beta1, beta2, eps, weight_decay, step, ema_rate,
                           write_back_grads, (hipStream_t)stream);
}
int rald_optim_ema(float* ema_params, const float* params, int64_t n, double rate, void* stream) {
    RALD_CHECK(n == 0 || (ema_params && params), "rald_optim_ema: null argument");
    return optim_ema(ema_params, params, n, rate, (hipStream_t)stream);
}

// ---- kernel-level entry points -----------------------------------------------------------------
int rald_op_gemm_nt(const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb, int64_t strideB,
                    void* C, int64_t ldc, int64_t strideC, const float* bias, int32_t M, int32_t N, int32_t K,
                    int32_t batch, float alpha, int32_t epilogue, void* stream) {
    RALD_CHECK(A && B && C, "rald_op_gemm_nt: null pointer");
    GemmArgs g;
    g.A = (const bf16*)A; g.lda = lda; g.strideA = strideA; g.B = (const bf16*)B; g.ldb = ldb; g.strideB = strideB;
    g.C = C; g.ldc = ldc; g.strideC = strideC; g.bias = bias; g.M = M; g.N = N; g.K = K; g.batch = batch; g.alpha = alpha; g.alpha_ncols = 1 << 30; g.ablate = 0;
    g.ablate = RALD_PROBE_ENV("RALD_GEMM_ABLATE", 0);
    return gemm_nt(g, epilogue, (hipStream_t)stream);
}
int rald_op_gemm_nt2(const void* A, int64_t lda, int64_t strideA, int64_t strideA2, const void* B, int64_t ldb, int64_t strideB, int64_t strideB2,
                     void* C, int64_t ldc, int64_t strideC, int64_t strideC2, const float* bias, int32_t M, int32_t N, int32_t K, int32_t batch,
                     int32_t batch2, float alpha, int32_t epilogue, void* stream) {
    RALD_CHECK(A && B && C, "rald_op_gemm_nt2: null pointer");
    GemmArgs g = gemm_args((const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, bias, M, N, K);
    g.strideA = strideA; g.strideB = strideB; g.strideC = strideC; g.batch = batch; g.alpha = alpha;
    g.batch2 = batch2; g.strideA2 = strideA2; g.strideB2 = strideB2; g.strideC2 = strideC2;
    return gemm_nt(g, epilogue, (hipStream_t)stream);
}
int rald_op_gemm_tn(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, float* C, int64_t ldc, float* colsum, int32_t M, int32_t N1,
                    int32_t N2, void* stream) {
    return gemm_tn((const bf16*)A_bf16, lda, (const bf16*)B_bf16, ldb, C, ldc, colsum, M, N1, N2, (hipStream_t)stream);
}
int rald_op_conv3d_wgrad(const void* dy_bf16, const void* x_bf16, float* dW, float* dbias, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin,
                         int32_t Cout, int32_t stride, int32_t pad, void* stream) {
    return conv3d_wgrad_tn((const bf16*)dy_bf16, (const bf16*)x_bf16, dW, dbias, B, ID, IH, IW, Cin, Cout, stride, pad, (hipStream_t)stream);
}
int64_t rald_op_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2) { return 4 * gemm_tn_workspace_floats(M, N1, N2); }
int rald_op_gemm_tn_ws(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, float* C, int64_t ldc, float* colsum, int32_t M, int32_t N1,
                       int32_t N2, void* workspace, int64_t workspace_bytes, void* stream) {
    return gemm_tn((const bf16*)A_bf16, lda, (const bf16*)B_bf16, ldb, C, ldc, colsum, M, N1, N2, (hipStream_t)stream, (float*)workspace, workspace_bytes / 4);
}
int64_t rald_op_conv3d_wgrad_workspace_bytes(int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad) {
    return 4 * conv3d_wgrad_workspace_floats(B, ID, IH, IW, Cin, Cout, stride, pad);
}
int rald_op_conv3d_wgrad_ws(const void* dy_bf16, const void* x_bf16, float* dW, float* dbias, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin,
                            int32_t Cout, int32_t stride, int32_t pad, void* workspace, int64_t workspace_bytes, void* stream) {
    return conv3d_wgrad_tn((const bf16*)dy_bf16, (const bf16*)x_bf16, dW, dbias, B, ID, IH, IW, Cin, Cout, stride, pad, (hipStream_t)stream, (float*)workspace,
                           workspace_bytes / 4);
}
int rald_op_patches27(const float* cube, int32_t cube_ch, void* out_bf16, int32_t B, int32_t D, int32_t H, int32_t W, void* stream) {
    return patches27(cube, cube_ch, (bf16*)out_bf16, B, D, H, W, (hipStream_t)stream);
}
int rald_op_transpose(const void* in, int32_t in_is_bf16, int64_t ld_in, int64_t stride_in, int64_t stride_in2, void* out_bf16, int64_t ld_out,
                      int64_t stride_out, int64_t stride_out2, int32_t rows, int32_t cols, int32_t batch, int32_t batch2, void* stream) {
    TransposeArgs a;
    a.in = in; a.ld_in = ld_in; a.stride_in = stride_in; a.stride_in2 = stride_in2; a.out = (bf16*)out_bf16; a.ld_out = ld_out;
    a.stride_out = stride_out; a.stride_out2 = stride_out2; a.rows = rows; a.cols = cols; a.batch = batch; a.batch2 = batch2;
    return transpose_rows(a, in_is_bf16, (hipStream_t)stream);
}
int rald_op_ln_mod_bwd(const float* x, const float* dh, const float* scale, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                       int64_t rows, int32_t D, float* dx_accum, float* dscale_accum, float* dshift_accum, void* stream) {
    RALD_CHECK(x && dh && scale && dx_accum && dscale_accum && dshift_accum, "rald_op_ln_mod_bwd: null pointer");
    return ln_mod_bwd(x, dh, scale, gstride, rows_per_group, add_one, eps, rows, D, dx_accum, dscale_accum, dshift_accum, (hipStream_t)stream);
}
int rald_op_ln_mod_bwd_cast(const float* x, const float* dh, const float* scale, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                            int64_t rows, int32_t D, float* dx_accum, void* dx_bf16_out, float* dscale_accum, float* dshift_accum, void* stream) {
    RALD_CHECK(x && dh && scale && dx_accum && dx_bf16_out && dscale_accum && dshift_accum, "rald_op_ln_mod_bwd_cast: null pointer");
    return ln_mod_bwd(x, dh, scale, gstride, rows_per_group, add_one, eps, rows, D, dx_accum, dscale_accum, dshift_accum, (hipStream_t)stream, (bf16*)dx_bf16_out);
}
int rald_op_geglu_fwd(const void* u_bf16, void* hid_bf16, int64_t M, int32_t inner, void* stream) {
    RALD_CHECK(u_bf16 && hid_bf16, "rald_op_geglu_fwd: null pointer");
    return geglu_fwd((const bf16*)u_bf16, (bf16*)hid_bf16, M, inner, (hipStream_t)stream);
}
int rald_op_geglu_bwd(const void* u_bf16, const void* dhid_bf16, void* du_bf16, int64_t M, int32_t inner, void* stream) {
    RALD_CHECK(u_bf16 && dhid_bf16 && du_bf16, "rald_op_geglu_bwd: null pointer");
    return geglu_bwd((const bf16*)u_bf16, (const bf16*)dhid_bf16, (bf16*)du_bf16, M, inner, (hipStream_t)stream);
}
int rald_op_colsum(const void* X, int32_t is_bf16, int64_t ld, int64_t M, int32_t N, float* out_accum, void* stream) {
    return colsum(X, is_bf16, ld, M, N, out_accum, (hipStream_t)stream);
}
int rald_op_row_lse(const float* S, int64_t rows, int32_t cols, float scale, float* lse, void* stream) {
    RALD_CHECK(S && lse, "rald_op_row_lse: null pointer");
    return row_lse(S, rows, cols, scale, lse, (hipStream_t)stream);
}
int rald_op_rowdot_heads(const void* dO_bf16, const void* O_bf16, int64_t M, int32_t heads, int32_t nq, float* delta, void* stream) {
    RALD_CHECK(dO_bf16 && O_bf16 && delta, "rald_op_rowdot_heads: null pointer");
    return rowdot_heads((const bf16*)dO_bf16, (const bf16*)O_bf16, M, heads, nq, delta, (hipStream_t)stream);
}
int rald_op_attn_bwd_elem(const float* S, const float* dP, const float* lse, const float* delta, int64_t batch, int32_t R, int32_t Ccols,
                          int64_t vbatch_stride, int32_t vstride, float scale, int32_t by_col, void* P_bf16, void* dS_bf16, void* stream) {
    RALD_CHECK(S && dP && lse && delta && dS_bf16, "rald_op_attn_bwd_elem: null pointer");
    return attn_bwd_elem(S, dP, lse, delta, batch, R, Ccols, vbatch_stride, vstride, scale, by_col, (bf16*)P_bf16, (bf16*)dS_bf16, (hipStream_t)stream);
}
int rald_op_sgemm_acc(const float* A, int64_t lda, int32_t trans_a, const float* B, int64_t ldb, int32_t trans_b, float* C, int64_t ldc, int32_t M,
                      int32_t N, int32_t K, float alpha, void* stream) {
    return sgemm_acc(A, lda, trans_a, B, ldb, trans_b, C, ldc, M, N, K, alpha, (hipStream_t)stream);
}
int rald_op_silu_fwd(const float* x, float* y, int64_t n, void* stream) { return silu_fwd(x, y, n, (hipStream_t)stream); }
int rald_op_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) { return silu_bwd(x, dy, dx, n, (hipStream_t)stream); }
int rald_op_posemb(const float* t, float* out, int32_t n, int32_t channels, void* stream) {
    RALD_CHECK(t && out && n > 0, "rald_op_posemb: bad arguments");
    return positional_embedding(t, out, n, channels, (hipStream_t)stream);
}
int rald_op_edm_loss_grad(const float* F, const float* x_noised, const float* y, const float* coef3, int64_t per_sample, int64_t total, float* dF,
                          float* D_out, double* loss, void* stream) {
    RALD_CHECK(F && x_noised && y && coef3 && dF && loss, "rald_op_edm_loss_grad: null pointer");
    return edm_loss_grad(F, x_noised, y, coef3, per_sample, total, dF, D_out, loss, (hipStream_t)stream);
}
// ---- radar encoder, op level (forward pieces + backward building blocks) --------------------------------------
int rald_op_conv3d(const void* in_bf16, const void* w_packed_bf16, const float* bias, const float* resid, float* out, int32_t B, int32_t ID,
                   int32_t IH, int32_t IW, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, void* stream) {
    return conv3d_igemm((const bf16*)in_bf16, (const bf16*)w_packed_bf16, bias, resid, out, B, ID, IH, IW, Cin, Cout, stride, pad, (hipStream_t)stream);
}
int rald_op_conv3d_bf16(const void* in_bf16, const void* w_packed_bf16, const float* bias, void* out_bf16, int32_t B, int32_t ID, int32_t IH, int32_t IW,
                        int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, void* stream) {
    return conv3d_igemm((const bf16*)in_bf16, (const bf16*)w_packed_bf16, bias, nullptr, nullptr, B, ID, IH, IW, Cin, Cout, stride, pad, (hipStream_t)stream,
                        (bf16*)out_bf16);
}
int rald_op_groupnorm(const float* x, const float* gamma, const float* beta, void* y_bf16, double* stats, int32_t B, int32_t S, int32_t C,
                      int32_t swish, void* stream) {
    return groupnorm_fwd(x, gamma, beta, (bf16*)y_bf16, stats, B, S, C, swish, (hipStream_t)stream);
}
int rald_op_groupnorm_bwd(const float* x, const double* stats, const float* gamma, const float* beta, const float* da, float* dx, float* dgamma,
                          float* dbeta, double* gsum_scratch, int32_t B, int32_t S, int32_t C, int32_t swish, int32_t accumulate, void* stream) {
    return groupnorm_bwd(x, stats, gamma, beta, da, dx, dgamma, dbeta, gsum_scratch, B, S, C, swish, accumulate, (hipStream_t)stream);
}
int64_t rald_op_groupnorm_bwd_scratch_bytes(int32_t B, int32_t S, int32_t C) { return groupnorm_bwd_scratch_bytes(B, S, C); }
int rald_op_groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, void* y_bf16, int32_t B, int32_t S, int32_t C,
                            int32_t swish, void* stream) {
    return groupnorm_apply(x, stats, gamma, beta, (bf16*)y_bf16, B, S, C, swish, (hipStream_t)stream);
}
int rald_op_groupnorm_bwd_cast(const float* x, const double* stats, const float* gamma, const float* beta, const void* da, int32_t da_is_bf16, float* dx,
                               void* dx_bf16, float* dgamma, float* dbeta, double* gsum_scratch, int32_t B, int32_t S, int32_t C, int32_t swish,
                               int32_t accumulate, void* stream) {
    return groupnorm_bwd(x, stats, gamma, beta, (const float*)da, dx, dgamma, dbeta, gsum_scratch, B, S, C, swish, accumulate, (hipStream_t)stream,
                         (bf16*)dx_bf16, da_is_bf16);
}
int rald_op_conv_in(const float* cube, int32_t cube_ch, int32_t Cin, const float* W, const float* bias, float* out, int32_t B, int32_t D, int32_t H,
                    int32_t Wd, int32_t Cout, void* stream) {
    return conv_in_fwd(cube, cube_ch, Cin, W, bias, out, B, D, H, Wd, Cout, (hipStream_t)stream);
}
int rald_op_conv_in_wgrad(const float* cube, int32_t cube_ch, const float* dy, int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Cout, float* dW,
                          void* stream) {
    return conv_in_wgrad(cube, cube_ch, dy, B, D, H, Wd, Cout, dW, (hipStream_t)stream);
}
int rald_op_conv_pack_weights(const float* W, void* out_bf16, int32_t Cout, int32_t Cin, int32_t pad_to, int32_t dgrad, void* stream) {
    return conv_pack_weights(W, (bf16*)out_bf16, Cout, Cin, pad_to, dgrad, (hipStream_t)stream);
}
int rald_op_pad_channels(const float* x, void* out_bf16, int64_t M, int32_t C, int32_t Cpad, void* stream) {
    return pad_channels(x, (bf16*)out_bf16, M, C, Cpad, (hipStream_t)stream);
}
int rald_op_zero_insert2(const float* dy, void* out_bf16, int32_t B, int32_t OD, int32_t OH, int32_t OW, int32_t C, void* stream) {
    return zero_insert2(dy, (bf16*)out_bf16, B, OD, OH, OW, C, (hipStream_t)stream);
}
int rald_op_im2col_t(const void* x_bf16, void* out_bf16, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t C, int32_t stride, int32_t pad,
                     int64_t m0, int32_t nchunk, void* stream) {
    return im2col_t((const bf16*)x_bf16, (bf16*)out_bf16, B, ID, IH, IW, C, stride, pad, m0, nchunk, (hipStream_t)stream);
}
int rald_op_rowdot(const void* a_bf16, const void* b_bf16, int64_t M, int32_t C, float* out, void* stream) {
    return rowdot((const bf16*)a_bf16, (const bf16*)b_bf16, M, C, out, (hipStream_t)stream);
}
int rald_op_softmax_rows(const float* S, int64_t ld_s, void* P_bf16, int64_t ld_p, int32_t rows, int32_t n, void* stream) {
    RALD_CHECK(S && P_bf16, "rald_op_softmax_rows: null pointer");
    return softmax_rows(S, ld_s, (bf16*)P_bf16, ld_p, rows, n, (hipStream_t)stream);
}
int rald_op_gemm_mx8(const void* A8, const void* scaleA, int64_t lda, int64_t strideA, int64_t strideSA, const void* B8, const void* scaleB,
                     int64_t ldb, int64_t strideB, int64_t strideSB, void* C, int64_t ldc, int64_t strideC, const float* bias, int32_t M,
                     int32_t N, int32_t K, int32_t batch, float alpha, int32_t epilogue, void* stream) {
    RALD_CHECK(A8 && B8 && scaleA && scaleB && C, "rald_op_gemm_mx8: null pointer");
    Mx8Args a;
    a.A8 = (const unsigned char*)A8; a.B8 = (const unsigned char*)B8; a.SA = (const unsigned char*)scaleA; a.SB = (const unsigned char*)scaleB;
    a.strideSA = strideSA; a.strideSB = strideSB;
    GemmArgs& g = a.g;
    g.A = nullptr; g.lda = lda; g.strideA = strideA; g.B = nullptr; g.ldb = ldb; g.strideB = strideB;
    g.C = C; g.ldc = ldc; g.strideC = strideC; g.bias = bias; g.M = M; g.N = N; g.K = K; g.batch = batch; g.alpha = alpha; g.alpha_ncols = 1 << 30; g.ablate = 0;
    return gemm_mx8(a, epilogue, (hipStream_t)stream);
}
int rald_op_quantize_mx8(const void* in, int32_t in_is_bf16, int64_t ld_in, void* out_e4m3, int64_t ld_out, void* out_scales_e8m0, int64_t rows,
                         int32_t K, void* stream) {
    RALD_CHECK(rows == 0 || (in && out_e4m3 && out_scales_e8m0), "rald_op_quantize_mx8: null pointer");
    return quantize_mx8(in, in_is_bf16, ld_in, (unsigned char*)out_e4m3, ld_out, (unsigned char*)out_scales_e8m0, rows, K, (hipStream_t)stream);
}
int rald_op_layernorm_mx8(const float* x, void* out_e4m3, void* out_scales_e8m0, int64_t M, int32_t D, const float* g, const float* b,
                          int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream) {
    RALD_CHECK(M == 0 || (x && out_e4m3 && out_scales_e8m0 && g && b), "rald_op_layernorm_mx8: null pointer");
    return layernorm_mod_mx8(x, (unsigned char*)out_e4m3, (unsigned char*)out_scales_e8m0, M, D, g, b, gstride, rows_per_group, add_one, eps,
                             (hipStream_t)stream);
}
int rald_op_layernorm(const float* x, void* out_bf16, int32_t M, int32_t D, const float* g, const float* b,
                      int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream) {
    RALD_CHECK(x && out_bf16 && g && b, "rald_op_layernorm: null pointer");
    return layernorm_mod(x, (bf16*)out_bf16, M, D, g, b, gstride, rows_per_group, add_one, eps, (hipStream_t)stream);
}
int rald_op_attention(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                      const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                      int32_t nq, int32_t nk, int32_t k_rows, int32_t heads, int32_t batch, float scale, void* stream) {
    RALD_CHECK(Q && K && Vt && O, "rald_op_attention: null pointer");
    AttnArgs a;
    a.Q = (const bf16*)Q; a.ldq = ldq; a.strideQ = strideQ; a.K = (const bf16*)K; a.ldk = ldk; a.strideK = strideK;
    a.Vt = (const bf16*)Vt; a.ldvt = ldvt; a.strideVt = strideVt; a.O = (bf16*)O; a.ldo = ldo; a.strideO = strideO;
    a.nq = nq; a.nk = nk; a.k_rows = k_rows; a.heads = heads; a.batch = batch; a.scale = scale; a.q_prescaled = 0;
    a.q_prescaled = RALD_PROBE_ENV("RALD_ATTN_PRESCALED", a.q_prescaled);   // timing experiments (probe builds)
    return attention_d64(a, (hipStream_t)stream);
}
int64_t rald_op_attention_split_scratch_bytes(int32_t ksplit, int32_t nq, int32_t heads, int32_t batch) {
    return attention_split_scratch_bytes(ksplit, nq, heads, batch);
}
int rald_op_attention_split(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* Vt, int64_t ldvt,
                            int64_t strideVt, void* O, int64_t ldo, int64_t strideO, int32_t nq, int32_t nk, int32_t k_rows, int32_t heads,
                            int32_t batch, float scale, int32_t ksplit, void* scratch, void* stream) {
    RALD_CHECK(Q && K && Vt && O, "rald_op_attention_split: null pointer");
    AttnArgs a;
    a.Q = (const bf16*)Q; a.ldq = ldq; a.strideQ = strideQ; a.K = (const bf16*)K; a.ldk = ldk; a.strideK = strideK;
    a.Vt = (const bf16*)Vt; a.ldvt = ldvt; a.strideVt = strideVt; a.O = (bf16*)O; a.ldo = ldo; a.strideO = strideO;
    a.nq = nq; a.nk = nk; a.k_rows = k_rows; a.heads = heads; a.batch = batch; a.scale = scale; a.q_prescaled = 0;
    a.ksplit = ksplit > 0 ? ksplit : attention_pick_ksplit(nq, nk, heads, batch);
    a.part = (float*)scratch;
    return attention_d64(a, (hipStream_t)stream);
}
int rald_op_attention_vrow(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* V, int64_t ldv,
                           int64_t strideV, void* O, int64_t ldo, int64_t strideO, int32_t nq, int32_t nk, int32_t heads, int32_t batch, float scale,
                           void* stream) {
    RALD_CHECK(Q && K && V && O, "rald_op_attention_vrow: null pointer");
    AttnArgs a;
    a.Q = (const bf16*)Q; a.ldq = ldq; a.strideQ = strideQ; a.K = (const bf16*)K; a.ldk = ldk; a.strideK = strideK;
    a.Vt = nullptr; a.ldvt = 0; a.strideVt = 0; a.V = (const bf16*)V; a.ldv = ldv; a.strideV = strideV;
    a.O = (bf16*)O; a.ldo = ldo; a.strideO = strideO;
    a.nq = nq; a.nk = nk; a.k_rows = nk; a.heads = heads; a.batch = batch; a.scale = scale; a.q_prescaled = 0;
    return attention_d64(a, (hipStream_t)stream);
}
int rald_op_attention_bwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* V, int64_t ldv,
                          int64_t strideV, const void* O, int64_t ldo, int64_t strideO, const void* dO, int64_t lddo, int64_t strideDO,
                          void* dQ, int64_t lddq, int64_t strideDQ, void* dK, int64_t lddk, int64_t strideDK, void* dV, int64_t lddv, int64_t strideDV,
                          float* lse_scratch, float* delta_scratch, int32_t nq, int32_t nk, int32_t heads, int32_t batch, float scale, void* stream) {
    AttnBwdArgs a;
    a.Q = (const bf16*)Q; a.ldq = ldq; a.sq = strideQ; a.K = (const bf16*)K; a.ldk = ldk; a.sk = strideK; a.V = (const bf16*)V; a.ldv = ldv; a.sv = strideV;
    a.O = (const bf16*)O; a.ldo = ldo; a.so = strideO; a.dO = (const bf16*)dO; a.lddo = lddo; a.sdo = strideDO;
    a.dQ = (bf16*)dQ; a.lddq = lddq; a.sdq = strideDQ; a.dK = (bf16*)dK; a.lddk = lddk; a.sdk = strideDK; a.dV = (bf16*)dV; a.lddv = lddv; a.sdv = strideDV;
    a.lse = lse_scratch; a.delta = delta_scratch; a.nq = nq; a.nk = nk; a.heads = heads; a.batch = batch; a.scale = scale;
    return attention_bwd_d64(a, (hipStream_t)stream);
}
// fp16 shared-key form (the folded encoder attentions, ae_encode.hip): fp32 pre-scaled queries [batch?][nq][heads*64], ONE fp16 row per key
// [batch][k_rows][64] that is key and value of every head; rows nk .. k_rows-1 must be zero
int rald_op_attention_f16kv(const float* Q, int64_t ldq, int64_t strideQ, const void* KV_f16, void* O_bf16, int64_t ldo, int64_t strideO, int32_t nq,
                            int32_t nk, int32_t k_rows, int32_t heads, int32_t batch, int32_t ksplit, void* scratch, void* stream) {
    RALD_CHECK(Q && KV_f16 && O_bf16, "rald_op_attention_f16kv: null pointer");
    AttnArgs a;
    a.Q = nullptr; a.Qf = Q; a.ldq = ldq; a.strideQ = strideQ; a.f16 = 1;
    a.K = (const bf16*)KV_f16; a.ldk = 64; a.strideK = (int64_t)k_rows * 64;
    a.Vt = nullptr; a.ldvt = 0; a.strideVt = 0; a.V = (const bf16*)KV_f16; a.ldv = 64; a.strideV = (int64_t)k_rows * 64; a.v_padded = 1; a.hsk = 0;
    a.O = (bf16*)O_bf16; a.ldo = ldo; a.strideO = strideO;
    a.nq = nq; a.nk = nk; a.k_rows = k_rows; a.heads = heads; a.batch = batch; a.scale = 1.f; a.q_prescaled = 1;
    a.ksplit = ksplit >= 0 ? ksplit : attention_pick_ksplit(nq, nk, heads, batch);
    a.part = (float*)scratch;
    return attention_d64(a, (hipStream_t)stream);
}
int rald_op_ae_enc_features(const float* pc, const float* basis, const float* var_factor, void* F_f16, void* G_f16, int32_t batch, int32_t n_points,
                            int32_t rows_per_sample, void* stream) {
    return ae_enc_features(pc, basis, var_factor, F_f16, G_f16, batch, n_points, rows_per_sample, (hipStream_t)stream);
}
int rald_op_ae_encode_tables(int32_t dim, int32_t num_latents, int32_t heads, int32_t mix, const float* const* in, float* const* out) {
    RALD_CHECK(in && out && dim >= 64 && num_latents >= 1 && heads >= 1, "rald_op_ae_encode_tables: bad argument");
    for (int i = 0; i < 18; ++i) RALD_CHECK(in[i] || (!mix && i >= 2 && i <= 11 && i != 9), "rald_op_ae_encode_tables: null input tensor");
    const int I = heads * 64;
    std::vector<float> Rf, Q1, T4, X0, T1, T3, c3;
    RALD_TRY(rald::ae_encode_tables(dim, I, num_latents, heads, mix != 0, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[11],
                                    in[12], in[13], in[14], in[15], in[16], in[17], Rf, Q1, T4, X0, T1, T3, c3));
    const std::vector<float>* v[7] = {&Rf, &Q1, &T4, &X0, &T1, &T3, &c3};
    for (int i = 0; i < 7; ++i)
        if (out[i] && !v[i]->empty()) memcpy(out[i], v[i]->data(), v[i]->size() * 4);
    return 0;
}
int rald_op_proj_in(const float* xin, const float* W, float* x, int32_t M, int32_t C, int32_t D, const float* coef, int32_t coef_stride,
                    int32_t rows_per_group, void* stream) {
    RALD_CHECK(xin && W && x && coef && M >= 1 && rows_per_group >= 1, "rald_op_proj_in: bad argument");
    return proj_in(xin, W, x, M, C, D, coef, coef_stride, rows_per_group, (hipStream_t)stream);
}
int rald_op_final_norm_proj(const float* x, const float* gamma, const float* beta, const float* Wout, const float* xin, float* out, int32_t M,
                            int32_t D, int32_t C, const float* coef, int32_t coef_stride, int32_t rows_per_group, void* stream) {
    RALD_CHECK(x && gamma && beta && Wout && xin && out && coef && M >= 1 && rows_per_group >= 1, "rald_op_final_norm_proj: bad argument");
    return final_norm_proj(x, gamma, beta, Wout, xin, out, M, D, C, coef, coef_stride, rows_per_group, (hipStream_t)stream);
}
int rald_op_attn_self_proj(const void* qkv_bf16, int64_t ld, const void* Wo_bf16, float* part, int32_t n_latents, int32_t heads, int32_t batch,
                           void* stream) {
    return attn_self_proj((const bf16*)qkv_bf16, ld, (const bf16*)Wo_bf16, part, n_latents, heads, batch, (hipStream_t)stream);
}
int rald_op_xattn_q2_proj(const void* h_bf16, const void* Wq_bf16, const void* Kc_bf16, int64_t ldk, int64_t strideK, const void* Vt_bf16,
                          int64_t ldvt, int64_t strideVt, const void* Wo_bf16, float* part, int32_t M, int32_t n_latents, int32_t heads,
                          int32_t n_keys, float qscale, void* stream) {
    return xattn_q2_proj((const bf16*)h_bf16, (const bf16*)Wq_bf16, (const bf16*)Kc_bf16, ldk, strideK, (const bf16*)Vt_bf16, ldvt, strideVt,
                         (const bf16*)Wo_bf16, part, M, n_latents, heads, n_keys, qscale, (hipStream_t)stream);
}
int rald_op_reduce_resid_ln(const float* part, int32_t slabs, int64_t slab_stride, const float* bias, float* x, void* h_bf16, int32_t M,
                            const float* g, const float* b, int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream) {
    return reduce_resid_ln(part, slabs, slab_stride, bias, x, (bf16*)h_bf16, M, g, b, gstride, rows_per_group, add_one, eps, (hipStream_t)stream);
}
int rald_op_gemm_resid_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, void* h_bf16,
                          const float* g, const float* b, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                          int32_t M, int32_t K, void* stream) {
    GemmLnArgs a;
    a.A = (const bf16*)A; a.lda = lda; a.W = (const bf16*)W; a.ldw = ldw; a.bias = bias; a.x = x; a.h = (bf16*)h_bf16;
    a.g = g; a.b = b; a.gstride = gstride; a.rows_per_group = rows_per_group; a.add_one = add_one; a.eps = eps; a.M = M; a.K = K;
    return gemm_resid_ln(a, (hipStream_t)stream);
}
int rald_op_cast_bf16(const float* in, void* out_bf16, int64_t n, void* stream) {
    RALD_CHECK(in && out_bf16, "rald_op_cast_bf16: null pointer");
    return cast_f32_bf16(in, (bf16*)out_bf16, n, (hipStream_t)stream);
}

}  // extern "C"
