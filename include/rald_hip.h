/* rald_hip.h - C-ABI of librald_hip.so: the MI355X (gfx950) implementation of RaLD's hot path.
 *
 * The reference (RoyAPTX4869/RaLD) has no FFI of its own: its seam is the Python nn.Module API
 * (SURVEY.md 8b).  These entry points are what a binding for that seam calls; each one names
 * the reference interface (file:line under the reference root) it replaces.  INTEGRATION.md
 * shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; rald_last_error() gives the
 *     message for the calling thread.  Nothing is printed, nothing aborts.
 *   - all tensor pointers are DEVICE pointers (hipMalloc'd or torch-allocated), dense,
 *     row-major, fp32 unless stated; `stream` is a hipStream_t (NULL = default stream).
 *     Work is enqueued on `stream`; no entry point synchronises except *_create / *_destroy /
 *     *_load_weight / *_finalize / *_reserve (setup-time, blocking).
 *   - handles are re-entrant per handle (one stream at a time per handle); no global state.
 *   - weights are loaded by their reference checkpoint key (utils/misc.py:309-316), fp32,
 *     from host OR device memory; packing to the kernels' layouts (bf16, fused/permuted rows)
 *     happens inside the library.
 */
#ifndef RALD_HIP_H
#define RALD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* rald_last_error(void);
int rald_version(void);
/* bit 0: PROBE build (`make PROBE=1`): the only build whose kernels honour the RALD_* A/B and ablation environment
 * switches, some of which skip work (no DMA in a main loop, no epilogue stores).  0 for the shipped library, which reads no
 * environment variable at all; bench.py refuses to measure a library that reports anything else. */
int rald_build_flags(void);
/* Diagnostic (synchronises the device): how many times a lane clamped a value while writing an fp16 partial-sum slab since the last
 * reset.  The small-batch paths (<= 2 samples) pass per-head / split-K partial sums between kernels as fp16 x 2^-6, saturating at
 * +-4.19e6; a non-zero count means a result was clipped.  Returns -1 on a HIP error. */
int64_t rald_debug_f16_saturation_count(int32_t reset);
/* Diagnostic: one launch that leaves all 160 KiB of LDS of every CU filled with NaN patterns (0x7fc07fc0).  LDS is not cleared
 * between workgroups, so a kernel that reads a word it never wrote is only wrong when something else ran on that CU before it -
 * i.e. under concurrent streams.  The test suite poisons the LDS and requires bit-identical results. */
int rald_debug_poison_lds(void* stream);

/* ------------------------------------------------------------------------------------------
 * Denoiser: EDMPrecond + LatentArrayTransformer  (model/models_radar_generation.py:171-233,
 * :314-449)
 * ---------------------------------------------------------------------------------------- */
typedef struct rald_dit rald_dit;

typedef struct rald_dit_config {
    int32_t n_latents;      /* 512  (EDMPrecond n_latents, :316)                        */
    int32_t channels;       /* 32   latent channels C (:317)                            */
    int32_t depth;          /* 24   transformer blocks (:324, factories :452-482)       */
    int32_t n_heads;        /* 8                                                        */
    int32_t d_head;         /* 64   (only 64 is implemented)                            */
    int32_t t_channels;     /* 256  PositionalEmbedding width (:336)                    */
    int32_t context_dim;    /* 512  width of the condition tokens (:180-193)            */
    int32_t n_cond_tokens;  /* 64   = 8*4*2 radar tokens (:405)                         */
    int32_t with_radar_enc; /* 1: radar_enc.* / radar_*_emb / radar_token_project loaded */
    int32_t enc_hidden_ch;  /* 64   (configs.enc_hidden_ch, :348)                       */
    int32_t enc_radar_ch;   /* 16   (configs.enc_radar_ch, :349)                        */
    int32_t radar_r, radar_a, radar_e; /* 128, 64, 32 input cube (R,A,E)                */
    float sigma_data;       /* 1.0 (:321)                                               */
    int32_t qkv_dtype;      /* 0 = bf16 (default); 1 = MXFP8 e4m3 for the attention q/k/v projections (BASELINE config #5); 2 = also the GEGLU projection; 3 = also ff.net.2 */
} rald_dit_config;

void rald_dit_default_config(rald_dit_config* cfg);
int rald_dit_create(const rald_dit_config* cfg, rald_dit** out);
void rald_dit_destroy(rald_dit* h);
/* One tensor of the reference state_dict, by key (e.g. "model.transformer_blocks.3.attn1.to_q.weight");
 * `data` = fp32, host or device, `nelem` elements.  Unknown keys and wrong sizes are errors. */
int rald_dit_load_weight(rald_dit* h, const char* name, const float* data, int64_t nelem);
/* Checks that every key was loaded (strict=True semantics, utils/misc.py:346). */
int rald_dit_finalize(rald_dit* h);
/* Pre-allocates activation workspace for batches up to max_batch (otherwise grown on demand). */
int rald_dit_reserve(rald_dit* h, int32_t max_batch);
/* Counts the reallocations of handle-owned device buffers (activation workspace, noise-level tables).  A caller that
 * captured library calls into a hipGraph must re-capture when the value differs from the one read after capture: the
 * graph's kernels hold pointers into those buffers. */
int64_t rald_dit_workspace_generation(const rald_dit* h);
/* Two-stream schedule of an NFE (rald_dit_denoise / rald_dit_sample): from `min_batch` samples up (default 256; measured: +1.9 % there,
 * nothing at 128) the batch runs as
 * two half-batches on two HIP streams - the caller's and a handle-owned one, forked from and joined back into `stream` with events,
 * so the caller sees ordinary stream semantics.  Every CU of one launch runs the same phase at the same time (matrix loop, then the
 * store-heavy epilogue); a second independent half-batch fills those holes.  Each half runs exactly the kernels a batch of its size
 * runs alone: results are bit-identical to two sequential calls.  0 = never split.  Re-plans the workspace (blocking). */
int rald_dit_set_two_stream_min_batch(rald_dit* h, int32_t min_batch);
int32_t rald_dit_two_stream_min_batch(const rald_dit* h);

/* Noise-level table: for each of the n sigmas (HOST array) computes the EDM coefficients
 * (c_in, c_skip, c_out, c_noise; :422-425), the timestep embedding (:217-219) and all
 * depth*3 AdaLayerNorm modulations (:128-129) once; rald_dit_denoise refers to rows of it. */
int rald_dit_set_sigmas(rald_dit* h, const float* sigmas_host, int32_t n, void* stream);

/* Bytes of the condition cache FOR THIS BATCH SIZE (a cache is built for, and used with, one batch size; it starts with a 64-byte
 * header - magic, batch, layout flag, configuration hash - that rald_dit_denoise / rald_dit_sample check against their `batch`
 * argument BEFORE any launch: a cache this handle has not seen (a copy) is verified once by reading the header back, which
 * synchronises `stream` and cannot happen under graph capture): K and V^T of the condition
 * tokens for every block, and from 3 samples up (bf16 mode) also the folded forms K.to_q and to_out.V^T per sample and block
 * (1 MiB each) that turn the cross-attention sub-block into two GEMMs - not a linear function of `batch`, so always ask. */
int64_t rald_dit_cond_cache_bytes(const rald_dit* h, int32_t batch);
/* Condition tokens [B, n_cond_tokens, context_dim] -> cond cache (the K/V projections of
 * attn2 are step-invariant; CrossAttention.to_k/to_v :63-64). */
int rald_dit_encode_cond_tokens(rald_dit* h, const float* tokens, int32_t batch, void* cond_cache, void* stream);
/* EDMPrecond.process_radar_cond (:363-407): cube [B,R,A,E,2] -> tokens [B,64,C] (optional
 * output, may be NULL) and the cond cache.  Run ONCE per sample, outside the sampling loop. */
int rald_dit_encode_cond(rald_dit* h, const float* cube, int32_t batch, float* out_tokens, void* cond_cache, void* stream);

/* One NFE = EDMPrecond.forward (:412-430) with the condition already encoded:
 *   D_x = c_skip*x + c_out*F(c_in*x, c_noise, cond).   x,out: [B, n_latents, channels].
 * sigma_row selects the row of the table set by rald_dit_set_sigmas; per_sample != 0 means
 * sample b uses row sigma_row+b (training-style [B,1,1] sigmas, :419).
 * raw_F != 0 returns F(x, c_noise, cond) with no pre/post-conditioning
 * (= LatentArrayTransformer.forward, :215-233, c_noise = ln(sigma)/4 of the table row). */
int rald_dit_denoise(rald_dit* h, const float* x, int32_t batch, int32_t sigma_row, int32_t per_sample,
                     const void* cond_cache, float* out, int32_t raw_F, void* stream);

/* edm_sampler (:235-275) at S_churn=0: latents [B,n_latents,channels] ~ N(0,1) -> samples.
 * 2*num_steps-1 NFEs.  The sigma table is (re)built internally for the schedule. */
int rald_dit_sample(rald_dit* h, const float* latents, int32_t batch, const void* cond_cache, int32_t num_steps,
                    float sigma_min, float sigma_max, float rho, float* out, void* stream);

/* Live timing of the dominant kernel (the FF1 GEGLU GEMM, FeedForward :88-117): between begin
 * and end every launch of it inside rald_dit_denoise is bracketed by HIP events recorded on the
 * launch stream (up to 4096 launches); end synchronises those events and returns their sum. */
int rald_dit_profile_begin(rald_dit* h);
int rald_dit_profile_end(rald_dit* h, double* total_ms, int32_t* launches);
/* The same bracket also times the three fused residual + LayerNorm GEMMs of a block (the largest time share of an NFE); this form
 * returns all four kinds: [0] FF1 GEGLU GEMM, [1] attn1.to_out + residual + AdaLN (K = 512), [2] attn2 output projection + residual
 * + AdaLN (K = 512), [3] ff.net.2 + residual + AdaLN (K = 2048).  total_ms4 / launches4 point at 4 elements each.  Batches that
 * take another engine for a kind (small M) report 0 launches there. */
int rald_dit_profile_end_kinds(rald_dit* h, double* total_ms4, int32_t* launches4);
/* Between begin and end: which kinds are bracketed from now on (bit k = kind k; begin resets it to all four).  An event pair costs ~2.5 us of
 * stream time, 96 pairs per NFE at 24 blocks: a measurement that also reports the whole-job rate brackets the three residual + LayerNorm
 * kinds on a few NFEs only and the dominant kernel on all of them. */
int rald_dit_profile_set_kinds(rald_dit* h, uint32_t kind_mask);

/* ------------------------------------------------------------------------------------------
 * Set-latent autoencoder: KLAutoEncoder, query_type='mix' or 'learnable'  (model/models_ae.py:284-432)
 * ---------------------------------------------------------------------------------------- */
typedef struct rald_ae rald_ae;
typedef struct rald_ae_config {
    int32_t dim;          /* 512 (tiny config: 256)      create_autoencoder(dim=..) :434   */
    int32_t num_latents;  /* 512 (tiny: 128)             M                                  */
    int32_t latent_dim;   /* 32                          latent_dim                         */
    int32_t depth;        /* 24 (hard-coded :449)                                           */
    int32_t heads;        /* 8  (hard-coded :455)                                           */
    int32_t dim_head;     /* 64 (hard-coded :456)                                           */
    int32_t num_inputs;   /* P: encode asserts pc.shape[1] == num_inputs (:354)             */
    int32_t query_type;   /* 0 = 'mix' (:380-387, the shipped config), 1 = 'learnable' (:378-379: keys `latents.weight`
                             instead of s_latents / d_latents / mix_attn_layer / query_proj)   */
} rald_ae_config;

int rald_ae_create(const rald_ae_config* cfg, rald_ae** out);
void rald_ae_destroy(rald_ae* h);
int rald_ae_load_weight(rald_ae* h, const char* name, const float* data, int64_t nelem);
int rald_ae_finalize(rald_ae* h);
/* KLAutoEncoder.encode (:351-405): pc [B,P,3]; eps [B,M,latent_dim] = the posterior noise
 * (the reference draws it with torch.randn on the CPU global RNG, :153 - the caller passes it so
 * results are reproducible); outputs z [B,M,L], kl [B], and optionally mean / logvar [B,M,L]
 * (NULL to skip). */
int rald_ae_encode(rald_ae* h, const float* pc, int32_t batch, const float* eps, float* out_mean, float* out_logvar,
                   float* out_z, float* out_kl, void* stream);
/* decode (:408-424) split at its query-independent part: the latent stack (proj + depth x
 * [self-attention, GEGLU FF]) and the decoder context are computed ONCE per z into `ctx`
 * (rald_ae_ctx_bytes bytes, 16-byte aligned); any number of query sets can then be decoded
 * against it (the reference recomputes the 116-GFLOP stack for each of its <=4 decode calls per
 * sample, engine_generation.py:204, :275, :300). */
int64_t rald_ae_ctx_bytes(const rald_ae* h, int32_t batch);
int64_t rald_ae_workspace_generation(const rald_ae* h);   /* as rald_dit_workspace_generation */
int rald_ae_decode_latents(rald_ae* h, const float* z, int32_t batch, void* ctx, void* stream);
/* queries [B,Q,3] -> occupancy logits [B,Q] (the reference returns [B,Q,1]; occupied iff > 0) */
int rald_ae_decode_queries(rald_ae* h, const void* ctx, const float* queries, int32_t batch, int64_t n_queries,
                           float* out_logits, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radar-spectrum encoder alone: RadarAutoencoder.encoder / _encode (model/models_radar_encoder.py
 * :137-241, :390-393) - the frozen-encoder route of engine_generation.py:87, :191.  Keys are
 * those BELOW "encoder." in a RadarAutoencoder checkpoint.
 * ---------------------------------------------------------------------------------------- */
typedef struct rald_radar rald_radar;
int rald_radar_create(int32_t basic_channel, int32_t embed_dim, int32_t in_channels, int32_t R, int32_t A, int32_t E, rald_radar** out);
void rald_radar_destroy(rald_radar* h);
int rald_radar_load_weight(rald_radar* h, const char* name, const float* data, int64_t nelem);
int rald_radar_finalize(rald_radar* h);
/* cube [B,R,A,E,in_channels] -> z [B,R/16,A/16,E/16,embed_dim]  (= _encode's permuted output) */
int rald_radar_encode(rald_radar* h, const float* cube, int32_t batch, float* out_z, void* stream);
/* Decoder half (Decoder.forward :333-359, RadarAutoencoder.decode / forward :386-406): keys below "decoder." are loaded with
 * rald_radar_load_decoder_weight (optional: the generation path only encodes; once one is loaded rald_radar_finalize wants all).
 * z [B, R/16, A/16, E/16, embed_dim] (the layout rald_radar_encode returns) -> out_pred4 [B, R, A, E, 4] fp32: channels 0-1 are the
 * reconstruction (RadarAutoencoder.forward's 'pred' is out_pred4[..., :2]), channels 2-3 are zero padding of the convolution kernel. */
int rald_radar_load_decoder_weight(rald_radar* h, const char* name, const float* data, int64_t nelem);
int rald_radar_decode(rald_radar* h, const float* z, int32_t batch, float* out_pred4, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decode post-processing on the device (the host tail of engine_generation.evaluate, :229-243 and
 * :283-322; utils/utils.py:50-75 inverse_norm_points, :116-142 cal_metrics;
 * dataset_preprocessor/lidar.py:57-63 polar2cartesian)
 * ---------------------------------------------------------------------------------------- */
int64_t rald_post_scratch_bytes(int64_t n_queries);
/* np.where(logits > threshold) + grid[ind] + inverse_norm_points (+ polar2cartesian if view_cone_mode):
 * positives are written in ascending query index to out_points [<=Q,3] (out_index optional, may be
 * NULL), their number to *out_count (device int64).  pc_range6_host = [min0,min1,min2,max0,max1,max2] as DOUBLES
 * (the reference's ranges are Python floats; its isotropic branch adds a float64 offset). */
int rald_post_occupied_points(const float* logits, const float* queries, int64_t n_queries, const double* pc_range6_host,
                              int32_t norm_anisotropy, int32_t norm_isotropy, int32_t view_cone_mode, float threshold,
                              float* out_points, int64_t* out_index, int64_t* out_count, void* scratch, void* stream);
/* inverse_norm_points (+ polar2cartesian) of a whole array (the ground-truth surface, :290, :317) */
int rald_post_transform_points(const float* points, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy,
                               int32_t norm_isotropy, int32_t view_cone_mode, float* out_points, void* stream);
/* cal_metrics' two sums (exact nearest neighbour, fp64): out_sums2[0] = sum_pred min_gt ||.||,
 * out_sums2[1] = sum_gt min_pred ||.||;  chamfer = 0.5*out[0]/n_pred + 0.5*out[1]/n_gt */
int rald_post_chamfer_sums(const float* pred, int64_t n_pred, const float* gt, int64_t n_gt, double* out_sums2, void* stream);
/* pred = logits >= 0; accuracy[b] = mean(pred == labels); iou[b] = |pred & labels| / |pred | labels| + 1e-5 */
int rald_post_iou(const float* logits, const float* labels, int32_t batch, int64_t n_queries, float* out_accuracy, float* out_iou, void* stream);

/* ------------------------------------------------------------------------------------------
 * Query generation + refine on the device (the host numpy code between model.sample and vae.decode in
 * engine_generation.evaluate, :250-300).  Random draws are caller-supplied DEVICE arrays so that the
 * host mirror can replay numpy's global-RNG stream (bit-identical queries) or use a device generator.
 * ---------------------------------------------------------------------------------------- */
/* generate_query_points (utils/utils.py:147-175): u3n = [3,n] float64 uniforms in numpy's draw order
 * (all x, then all y, then all z); out[i,a] = float32(lo_a + (hi_a - lo_a) * u[a,i]), box = [-1,1]^3
 * (anisotropic) or +-scale_a/max_scale (isotropic). */
int rald_query_uniform(const double* u3n, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy, int32_t norm_isotropy,
                       float* out_queries, void* stream);
/* the use_cart_query branch (engine_generation.py:251-256): uniform in the cartesian box ->
 * inverse_norm_points(pc_range_cart) -> cartesian2polar (dataset_preprocessor/lidar.py:49-55) ->
 * norm_points(pc_range) -> remove_points_outside_fov (utils/utils.py:106-112), float64 throughout,
 * float32 on output; survivors keep their order, *out_count (device int64) = their number.
 * scratch: rald_post_scratch_bytes(n). */
int rald_query_uniform_cart(const double* u3n, int64_t n, const double* pc_range_cart6_host, const double* pc_range6_host,
                            int32_t norm_anisotropy, int32_t norm_isotropy, float* out_queries, int64_t* out_count, void* scratch,
                            void* stream);
/* norm_points (utils/utils.py:77-104) of a float32 array */
int rald_query_norm_points(const float* points, int64_t n, const double* pc_range6_host, int32_t norm_anisotropy, int32_t norm_isotropy,
                           float* out_points, void* stream);
/* aug_query_helper (datasets/utils/query_helper.py:3-42) [+ norm_points when normalise != 0, as
 * engine_generation.py:292-297 does]: out [aug_num,3] = the first min(n_helper, aug_num) helper points, then
 * for g < aug_num - n_helper: clip(helper[sel_index[g]] + (2*u_bias[g,:]-1) * voxel_size * aug_scales[g]).
 * sel_index / aug_scales: int64 [aug_num - n_helper]; u_bias: float64 [aug_num - n_helper, 3] (may be NULL when
 * n_helper >= aug_num). */
int rald_query_refine(const float* helper_points, int64_t n_helper, int64_t aug_num, const int64_t* sel_index, const int64_t* aug_scales,
                      const double* u_bias, const double* pc_range6_host, const double* voxel_size3_host, int32_t norm_anisotropy,
                      int32_t norm_isotropy, int32_t normalise, float* out_points, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step of train_one_epoch (engine_generation.py:96-110) on FLAT fp32 storage: the model's
 * parameters, gradients, Adam moments and the EMA copy are five arrays with one layout.
 *   clip_grad_norm_ (utils/misc.py:262) -> torch.optim.AdamW(lr) (main_generation.py:161; betas 0.9/0.999,
 *   eps 1e-8, weight_decay 0.01 defaults) -> update_ema(rate 0.999) (engine_generation.py:29-40, :110)
 * ---------------------------------------------------------------------------------------- */
/* *out_sumsq (device double) = sum g^2 (fp64 accumulation) */
int rald_optim_grad_sumsq(const float* grads, int64_t n, double* out_sumsq, void* stream);
/* out[0] = total_norm = sqrt(*sumsq) * pre_scale; out[1] = pre_scale * min(max_norm / (total_norm + 1e-6), 1)
 * (max_norm <= 0: no clipping).  pre_scale folds the 1/world of a SUM all-reduce.  Device in, device out: no sync. */
int rald_optim_clip_coef(const double* sumsq, float pre_scale, float max_norm, float* out_norm_coef, void* stream);
/* one fused pass: g *= *grad_scale (device scalar, NULL = 1); p *= 1 - lr*wd; m, v updates; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps);
 * ema = ema*rate + p*(1-rate) when ema_params != NULL.  step counts from 1.  write_back_grads != 0 stores the scaled gradient. */
int rald_optim_adamw_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_params, int64_t n,
                         const float* grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                         double ema_rate, int32_t write_back_grads, void* stream);
/* update_ema alone (the reference also runs it on gradient-accumulation iterations) */
int rald_optim_ema(float* ema_params, const float* params, int64_t n, double rate, void* stream);

/* ColoRadarDataset.process_radar_data (datasets/aligned_coloradar/Coloradar_dataset.py:432-475): raw cube
 * [B,R,A,E,raw_channels] (intensity dB, doppler, ..., validity mask last; the .bin layout of load_radarcube
 * :420-430) -> the network's input [B,R,tgt_A,tgt_E,2]: intensity clipped to [0,max] / max, doppler * mask
 * / max_dopp, bilinear (align_corners=True) upsampling over (A,E). */
int rald_radar_cube_prepare(const float* raw, int32_t batch, int32_t R, int32_t A, int32_t E, int32_t raw_channels, int32_t tgt_A,
                            int32_t tgt_E, int32_t norm_intensity, float max_intensity, int32_t norm_dopp, float max_dopp, float* out,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Kernel-level entry points (what the parity tests and microbenchmarks drive directly)
 * ---------------------------------------------------------------------------------------- */
/* C[b][m][n] = alpha * sum_k A[b][m][k]*B[b][n][k] (+bias[n]); A,B bf16 (K contiguous).
 * epilogue: 0 bf16 out, 1 f32 out, 2 f32 C += result, 3 GEGLU (packed B rows, bf16 out, N/2 cols) */
int rald_op_gemm_nt(const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb, int64_t strideB,
                    void* C, int64_t ldc, int64_t strideC, const float* bias, int32_t M, int32_t N, int32_t K,
                    int32_t batch, float alpha, int32_t epilogue, void* stream);
/* rald_op_gemm_nt with an inner batch (attention heads): grid z = batch * batch2, operand offset =
 * b1 * stride + b2 * stride2 */
int rald_op_gemm_nt2(const void* A, int64_t lda, int64_t strideA, int64_t strideA2, const void* B, int64_t ldb, int64_t strideB, int64_t strideB2,
                     void* C, int64_t ldc, int64_t strideC, int64_t strideC2, const float* bias, int32_t M, int32_t N, int32_t K, int32_t batch,
                     int32_t batch2, float alpha, int32_t epilogue, void* stream);
/* Backward building blocks of the transformer block (what autograd derives for
 * models_radar_generation.py:35-169).  dX = dY.W and dW = dY^T.X run on rald_op_gemm_nt with transposed operands. */
/* Weight gradient of a Linear without transposed copies (rald_amd/csrc/gemm_tn.hip): C[n1][n2] += sum_m A[m][n1] B[m][n2] for row-major bf16
 * A [M, N1] (= dY) and B [M, N2] (= X), fp32 C accumulated into with atomics; colsum (nullable) [N1] += column sums of A (the bias gradient).
 * N1, N2, lda, ldb multiples of 8. */
int rald_op_gemm_tn(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, float* C, int64_t ldc, float* colsum, int32_t M, int32_t N1,
                    int32_t N2, void* stream);
/* Weight gradient of a 3x3x3 Conv3d (Encoder :216-241 under autograd) with the patch matrix never formed: dW [Cout][Cin][27] +=
 * sum over output voxels of dy[v][co] x[v*stride - pad + offset(tap)][ci] (zero outside the volume); dbias (nullable) += column sums
 * of dy.  dy [B*OD*OH*OW][Cout] bf16, x [B][ID][IH][IW][Cin] bf16 channels-last, OD = ID / stride ... */
int rald_op_conv3d_wgrad(const void* dy_bf16, const void* x_bf16, float* dW, float* dbias, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin,
                         int32_t Cout, int32_t stride, int32_t pad, void* stream);
/* The two weight-gradient products above without atomics: with a caller-owned workspace of _workspace_bytes(...) bytes (16-byte aligned; 0 =
 * this shape keeps the atomic form and the workspace may be null) every row / voxel range stores its partial result there and a second launch
 * adds the ranges IN ORDER into C / dW (and colsum / dbias): bit-reproducible run to run, and several times faster for the convolution below
 * full resolution, whose atomics scatter over the parameter's tap-innermost layout. */
int64_t rald_op_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2);
int rald_op_gemm_tn_ws(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, float* C, int64_t ldc, float* colsum, int32_t M, int32_t N1,
                       int32_t N2, void* workspace, int64_t workspace_bytes, void* stream);
int64_t rald_op_conv3d_wgrad_workspace_bytes(int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad);
int rald_op_conv3d_wgrad_ws(const void* dy_bf16, const void* x_bf16, float* dW, float* dbias, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t Cin,
                            int32_t Cout, int32_t stride, int32_t pad, void* workspace, int64_t workspace_bytes, void* stream);
/* conv_in (one input channel) weight gradient, first half: the 27-neighbourhood of channel 0 of cube [B][D][H][W][cube_ch] fp32 per voxel as one
 * bf16 row of 32 (taps kd*9 + kh*3 + kw, zero outside the volume, 5 zero pads); dW = rald_op_gemm_tn(dy, patches). */
int rald_op_patches27(const float* cube, int32_t cube_ch, void* out_bf16, int32_t B, int32_t D, int32_t H, int32_t W, void* stream);
/* in [batch][batch2][rows][cols] (f32 or bf16) -> out [batch][batch2][cols][rows] bf16 */
int rald_op_transpose(const void* in, int32_t in_is_bf16, int64_t ld_in, int64_t stride_in, int64_t stride_in2, void* out_bf16, int64_t ld_out,
                      int64_t stride_out, int64_t stride_out2, int32_t rows, int32_t cols, int32_t batch, int32_t batch2, void* stream);
/* AdaLayerNorm :119-131 (add_one = 1) / LayerNorm (add_one = 0, scale = weight) backward, D = 512:
 * dx += ..., dscale[g] += sum_rows dh * xhat, dshift[g] += sum_rows dh  (g = row / rows_per_group, stride gstride) */
int rald_op_ln_mod_bwd(const float* x, const float* dh, const float* scale, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                       int64_t rows, int32_t D, float* dx_accum, float* dscale_accum, float* dshift_accum, void* stream);
/* the same, and the updated dx also as bf16 [rows][512] (what the next weight-/input-gradient GEMMs of the block read) */
int rald_op_ln_mod_bwd_cast(const float* x, const float* dh, const float* scale, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                            int64_t rows, int32_t D, float* dx_accum, void* dx_bf16_out, float* dscale_accum, float* dshift_accum, void* stream);
/* GEGLU :88-95 in the natural layout u = [a | g] (2*inner columns): hid = a * gelu_erf(g); and its backward */
int rald_op_geglu_fwd(const void* u_bf16, void* hid_bf16, int64_t M, int32_t inner, void* stream);
int rald_op_geglu_bwd(const void* u_bf16, const void* dhid_bf16, void* du_bf16, int64_t M, int32_t inner, void* stream);
/* bias gradient: out[n] += sum_m X[m][n] */
int rald_op_colsum(const void* X, int32_t is_bf16, int64_t ld, int64_t M, int32_t N, float* out_accum, void* stream);
/* attention backward, element-wise parts: lse[r] = log sum_c exp(scale*S[r][c]); delta[b][h][q] = <dO, O> over head h of row m = b*nq + q;
 * P = exp(scale*S - lse[i]), dS = P*(dP - delta[i])*scale with i = row (by_col 0) or column (by_col 1) */
int rald_op_row_lse(const float* S, int64_t rows, int32_t cols, float scale, float* lse, void* stream);
int rald_op_rowdot_heads(const void* dO_bf16, const void* O_bf16, int64_t M, int32_t heads, int32_t nq, float* delta, void* stream);
int rald_op_attn_bwd_elem(const float* S, const float* dP, const float* lse, const float* delta, int64_t batch, int32_t R, int32_t Ccols,
                          int64_t vbatch_stride, int32_t vstride, float scale, int32_t by_col, void* P_bf16, void* dS_bf16, void* stream);
/* thin fp32 products of the training step (timestep MLP, AdaLN linears, proj_in/out and their gradients):
 * C[m][n] += alpha * sum_k A(m,k)*B(n,k); A(m,k) = A[m*lda+k] or (trans_a) A[k*lda+m]; B(n,k) = B[n*ldb+k] or (trans_b) B[k*ldb+n] */
int rald_op_sgemm_acc(const float* A, int64_t lda, int32_t trans_a, const float* B, int64_t ldb, int32_t trans_b, float* C, int64_t ldc, int32_t M,
                      int32_t N, int32_t K, float alpha, void* stream);
int rald_op_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int rald_op_silu_bwd(const float* x_pre, const float* dy, float* dx, int64_t n, void* stream);
/* PositionalEmbedding :20-33: out [n, channels] = cat[cos, sin](outer(t, freqs)) */
int rald_op_posemb(const float* t, float* out, int32_t n, int32_t channels, void* stream);
/* EDMLoss :283-295 over EDMPrecond.forward's output mix :422-430: coef3[b] = {c_skip, c_out, weight};
 * D = c_skip*x_noised + c_out*F; *loss = mean(weight*(D - y)^2) (device double); dF = dloss/dF; D_out optional */
int rald_op_edm_loss_grad(const float* F, const float* x_noised, const float* y, const float* coef3, int64_t per_sample, int64_t total, float* dF,
                          float* D_out, double* loss, void* stream);
/* Radar-spectrum encoder at op level (model/models_radar_encoder.py:29-241), channels-last activations
 * [b][d][h][w][c]: the forward kernels of rald_radar_encode plus the backward building blocks (the shipped
 * configuration trains the encoder jointly with the denoiser). */
/* Conv3d k3 as implicit GEMM: in bf16 [B][ID][IH][IW][Cin], w packed bf16 [Cout][27][Cin] (rald_op_conv_pack_weights),
 * out f32 [B][ID/s][IH/s][IW/s][Cout] = conv + bias (+ resid).  Cin % 64 == 0, Cout % 4 == 0, stride 1|2. */
int rald_op_conv3d(const void* in_bf16, const void* w_packed_bf16, const float* bias, const float* resid, float* out, int32_t B, int32_t ID,
                   int32_t IH, int32_t IW, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, void* stream);
/* W [Cout][Cin][27] f32 (the parameter) -> packed bf16: dgrad = 0: [Cout][27][pad_to >= Cin]; dgrad = 1: the flipped,
 * transposed weights [Cin][27][pad_to >= Cout] that make rald_op_conv3d map dY to dX */
int rald_op_conv_pack_weights(const float* W, void* out_bf16, int32_t Cout, int32_t Cin, int32_t pad_to, int32_t dgrad, void* stream);
/* Normalize :9-12 (GroupNorm 32 groups, eps 1e-6) [+ swish :5-7]: y bf16; stats: B*64*(1 + ceil(S/512)) doubles - the first [B][32][2] =
 * {sum, sumsq} are kept for the backward, the rest is scratch for the per-block partials of the deterministic (atomic-free) reduction */
int rald_op_groupnorm(const float* x, const float* gamma, const float* beta, void* y_bf16, double* stats, int32_t B, int32_t S, int32_t C,
                      int32_t swish, void* stream);
/* its backward: da = gradient w.r.t. the (activated) output; dx written or accumulated; dgamma/dbeta accumulated; gsum_scratch: 8-byte aligned,
 * rald_op_groupnorm_bwd_scratch_bytes(B, S, C) bytes (group sums + the per-workgroup partial sums of the atomic-free, bit-reproducible reduction) */
int64_t rald_op_groupnorm_bwd_scratch_bytes(int32_t B, int32_t S, int32_t C);
int rald_op_groupnorm_bwd(const float* x, const double* stats, const float* gamma, const float* beta, const float* da, float* dx, float* dgamma,
                          float* dbeta, double* gsum_scratch, int32_t B, int32_t S, int32_t C, int32_t swish, int32_t accumulate, void* stream);
/* rald_op_groupnorm's normalisation alone, from the [B][32][2] statistics a forward call left (the training backward re-creates activations) */
int rald_op_groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, void* y_bf16, int32_t B, int32_t S, int32_t C,
                            int32_t swish, void* stream);
/* rald_op_groupnorm_bwd that also leaves the resulting dx rounded to bf16 in dx_bf16 (what the convolution gradients of the next layer read);
 * dx may be null when only the bf16 form is wanted (not with accumulate); da may hold bf16 (da_is_bf16 = 1) */
int rald_op_groupnorm_bwd_cast(const float* x, const double* stats, const float* gamma, const float* beta, const void* da, int32_t da_is_bf16, float* dx,
                               void* dx_bf16, float* dgamma, float* dbeta, double* gsum_scratch, int32_t B, int32_t S, int32_t C, int32_t swish,
                               int32_t accumulate, void* stream);
/* rald_op_conv3d with a bf16 result and no residual: the data-gradient convolutions of the training step, whose only reader is the
 * GroupNorm backward (da_is_bf16 = 1 above) */
int rald_op_conv3d_bf16(const void* in_bf16, const void* w_packed_bf16, const float* bias, void* out_bf16, int32_t B, int32_t ID, int32_t IH, int32_t IW,
                        int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, void* stream);
/* conv_in (Cin = 1 read in place from channel 0 of the cube) and its weight gradient dW [Cout][27] (accumulated) */
int rald_op_conv_in(const float* cube, int32_t cube_ch, int32_t Cin, const float* W, const float* bias, float* out, int32_t B, int32_t D, int32_t H,
                    int32_t Wd, int32_t Cout, void* stream);
int rald_op_conv_in_wgrad(const float* cube, int32_t cube_ch, const float* dy, int32_t B, int32_t D, int32_t H, int32_t Wd, int32_t Cout, float* dW,
                          void* stream);
/* x [M][C] f32 -> bf16 [M][Cpad] zero-filled;  dY [B][OD][OH][OW][C] f32 -> bf16 on the even positions of a 2x grid (Downsample dgrad) */
int rald_op_pad_channels(const float* x, void* out_bf16, int64_t M, int32_t C, int32_t Cpad, void* stream);
int rald_op_zero_insert2(const float* dy, void* out_bf16, int32_t B, int32_t OD, int32_t OH, int32_t OW, int32_t C, void* stream);
/* transposed im2col of output voxels [m0, m0+nchunk): out bf16 [C*27][nchunk], row ci*27 + tap (the weight tensor's own order) */
int rald_op_im2col_t(const void* x_bf16, void* out_bf16, int32_t B, int32_t ID, int32_t IH, int32_t IW, int32_t C, int32_t stride, int32_t pad,
                     int64_t m0, int32_t nchunk, void* stream);
int rald_op_rowdot(const void* a_bf16, const void* b_bf16, int64_t M, int32_t C, float* out, void* stream);
int rald_op_softmax_rows(const float* S, int64_t ld_s, void* P_bf16, int64_t ld_p, int32_t rows, int32_t n, void* stream);
/* Small-batch fused attention sub-blocks (rald_amd/csrc/attn_small.hip; CrossAttention :55-76 + the to_out Linear).
 * _self_proj: qkv [batch*n_latents][ld] bf16 = q (pre-multiplied by scale*log2e) | k | v of a fused projection, 8 heads x 64;
 *   part[h][row][512] = attention output of head h times Wo[:, 64h:64h+64]^T (fp32 partial of to_out).
 * _q2_proj: h [M][512] bf16 -> to_q (Wq, scaled by qscale) -> attention over 64 cached condition tokens (Kc[b*strideK + key*ldk +
 *   64h + d], Vt[b*strideVt + (64h + d)*ldvt + key]) -> part likewise.
 * _reduce_resid_ln: x[M][512] += bias + sum_s part[s] (fixed order); h = LN(x)*(add_one + g) + b as bf16 when h is given. */
int rald_op_attn_self_proj(const void* qkv_bf16, int64_t ld, const void* Wo_bf16, float* part, int32_t n_latents, int32_t heads, int32_t batch,
                           void* stream);
int rald_op_xattn_q2_proj(const void* h_bf16, const void* Wq_bf16, const void* Kc_bf16, int64_t ldk, int64_t strideK, const void* Vt_bf16,
                          int64_t ldvt, int64_t strideVt, const void* Wo_bf16, float* part, int32_t M, int32_t n_latents, int32_t heads,
                          int32_t n_keys, float qscale, void* stream);
/* The denoiser's first and last layers (LatentArrayTransformer.forward :221, :230-232, fused with the EDM coefficients :424-429; norm.hip),
 * fp32 throughout.  coef[s][coef_stride] = {c_in, c_skip, c_out, ...} of the sample s = row / rows_per_group.
 *   proj_in:         x[m][n] = c_in * sum_k xin[m][k] W[n][k]                                   W [D][C]
 *   final_norm_proj: out[m][c] = c_skip * xin[m][c] + c_out * sum_k LN(x[m]; gamma, beta)[k] Wout[c][k]     Wout [C][D], D = 512 */
int rald_op_proj_in(const float* xin, const float* W, float* x, int32_t M, int32_t C, int32_t D, const float* coef, int32_t coef_stride,
                    int32_t rows_per_group, void* stream);
int rald_op_final_norm_proj(const float* x, const float* gamma, const float* beta, const float* Wout, const float* xin, float* out, int32_t M,
                            int32_t D, int32_t C, const float* coef, int32_t coef_stride, int32_t rows_per_group, void* stream);
int rald_op_reduce_resid_ln(const float* part, int32_t slabs, int64_t slab_stride, const float* bias, float* x, void* h_bf16, int32_t M,
                            const float* g, const float* b, int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream);
/* Streaming query decoder (KLAutoEncoder.decode :417-424; rald_amd/csrc/ae_decode.hip).  _tables: the weight-only tables
 * it is built on, computed on the HOST in double from host tensors of decoder_cross_attn (to_q [d,d], k half of to_kv [d,d],
 * norm weight / bias [d]), point_embed.mlp (weight [d,51], bias [d]) and the folded value vector [d]:
 * t2aug_out [d][64] fp32, l_img_out [64][64] fp16 bits (no GPU needed: what the CPU tests check the folding with).
 * _queries_nw: rald_ae_decode_queries with the waves per workgroup given (8, 12, 16; 0 = default) for tuning runs. */
int rald_op_ae_decode_tables(int32_t dim, const float* wq, const float* wk, const float* norm_w, const float* norm_b, const float* wpe,
                             const float* bpe, const float* wfold, float* t2aug_out, uint16_t* l_img_out);
int rald_op_ae_decode_queries_nw(rald_ae* h, const void* ctx, const float* queries, int32_t batch, int64_t n_queries, float* out_logits,
                                 int32_t waves_per_workgroup, void* stream);
/* Folded encoder (KLAutoEncoder.encode :351-399; rald_amd/csrc/ae_encode.hip): both attentions of the latent queries over the
 * input points run with ONE fp16 row of 52 Fourier features per point as key and value (head dim 64).
 * _tables: the weight-only tables, computed on the HOST in double (no GPU needed).  in[18] = host fp32 tensors in the reference's
 *   layouts: point_embed.mlp weight [d,51], bias [d]; d_latents [M,d]; mix_attn_layer norm weight, bias [d], to_q [I,d], to_kv [2I,d],
 *   to_out weight [d,I], bias [d]; s_latents (mix) or latents (learnable) [M,d]; query_proj weight [d,d], bias [d];
 *   cross_attend_blocks.0 norm_context weight, bias [d], to_q [d,d], to_kv [2d,d], to_out weight [d,d], bias [d]   (I = heads*64; entries
 *   2-8, 10, 11 may be null when mix == 0).  out[7] (any may be null) = variance factor [52,52], mix queries [M,I], T4 [d,I], X0 [M,d],
 *   T1 [d,64], T3 [d,64], c3 [d] - see ae_encode.hip for what each multiplies.
 * _features: F, G fp16 [batch][rows_per_sample][64] from pc [batch][n_points][3] (rows_per_sample = n_points rounded up to 64).
 * rald_op_attention_f16kv: the attention kernel's fp16 form on such rows (fp32 queries already times scale*log2(e); ksplit < 0 = pick;
 *   scratch = rald_op_attention_split_scratch_bytes(16, ...) when the keys may be split). */
int rald_op_ae_encode_tables(int32_t dim, int32_t num_latents, int32_t heads, int32_t mix, const float* const* in, float* const* out);
int rald_op_ae_enc_features(const float* pc, const float* basis, const float* var_factor, void* F_f16, void* G_f16, int32_t batch, int32_t n_points,
                            int32_t rows_per_sample, void* stream);
int rald_op_attention_f16kv(const float* Q, int64_t ldq, int64_t strideQ, const void* KV_f16, void* O_bf16, int64_t ldo, int64_t strideO, int32_t nq,
                            int32_t nk, int32_t k_rows, int32_t heads, int32_t batch, int32_t ksplit, void* scratch, void* stream);
/* MXFP8 (OCP microscaling: e4m3 elements + one e8m0 scale per 32 consecutive K elements of a row), the
 * "fp8 MFMA QKV/proj path" of BASELINE config #5.  C = alpha * A . B^T + bias on
 * v_mfma_scale_f32_16x16x128_f8f6f4; epilogue 0 = bf16, 1 = f32, 2 = f32 residual accumulate.  K % 128 == 0;
 * lda/ldb/strides in bytes (= elements); scales [rows][K/32] contiguous per batch entry. */
int rald_op_gemm_mx8(const void* A8, const void* scaleA, int64_t lda, int64_t strideA, int64_t strideSA, const void* B8, const void* scaleB,
                     int64_t ldb, int64_t strideB, int64_t strideSB, void* C, int64_t ldc, int64_t strideC, const float* bias, int32_t M,
                     int32_t N, int32_t K, int32_t batch, float alpha, int32_t epilogue, void* stream);
/* rows of f32 (in_is_bf16 = 0) or bf16 -> MXFP8: block scale = the smallest power of two with amax / scale <= 448 */
int rald_op_quantize_mx8(const void* in, int32_t in_is_bf16, int64_t ld_in, void* out_e4m3, int64_t ld_out, void* out_scales_e8m0, int64_t rows,
                         int32_t K, void* stream);
/* rald_op_layernorm with an MXFP8 result (D = 512) */
int rald_op_layernorm_mx8(const float* x, void* out_e4m3, void* out_scales_e8m0, int64_t M, int32_t D, const float* g, const float* b,
                          int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream);
/* out_bf16 = LayerNorm(x_f32[M][D]) * (add_one + g[row/rows_per_group*gstride + c]) + b[...] */
int rald_op_layernorm(const float* x, void* out_bf16, int32_t M, int32_t D, const float* g, const float* b,
                      int64_t gstride, int32_t rows_per_group, float add_one, float eps, void* stream);
/* multi-head attention, head dim 64; Q[b][i][h*64+d], K[b][j][h*64+d], Vt[b][h*64+d][j] bf16 */
int rald_op_attention(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                      const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                      int32_t nq, int32_t nk, int32_t k_rows, int32_t heads, int32_t batch, float scale, void* stream);
/* rald_op_attention with the keys split over `ksplit` workgroups per query block (few queries x many keys, e.g. 512
 * latents x 10 000 points at batch 1): partial results go through `scratch` (rald_op_attention_split_scratch_bytes)
 * and a combine pass.  ksplit <= 0 picks a value from the shape. */
int64_t rald_op_attention_split_scratch_bytes(int32_t ksplit, int32_t nq, int32_t heads, int32_t batch);
int rald_op_attention_split(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* Vt, int64_t ldvt,
                            int64_t strideVt, void* O, int64_t ldo, int64_t strideO, int32_t nq, int32_t nk, int32_t k_rows, int32_t heads,
                            int32_t batch, float scale, int32_t ksplit, void* scratch, void* stream);
/* rald_op_attention with V row-major like K (V[b][j][h*64+d], e.g. a column slice of a fused q|k|v projection): the kernel
 * transposes it on the LDS read (ds_read_b64_tr_b16).  nk % 64 == 0. */
int rald_op_attention_vrow(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* V, int64_t ldv,
                           int64_t strideV, void* O, int64_t ldo, int64_t strideO, int32_t nq, int32_t nk, int32_t heads, int32_t batch, float scale,
                           void* stream);
/* gradients of rald_op_attention_vrow's O = softmax(Q K^T scale) V per head (torch autograd of CrossAttention,
 * model/models_radar_generation.py:66-75, in the training step engine_generation.py:74-98): dQ, dK, dV bf16 in the layouts of Q, K, V
 * (own leading dimensions and batch strides: column slices of fused buffers are fine).  Two launches, nothing score-shaped in memory.
 * lse_scratch / delta_scratch: fp32 [batch*heads*nq] each.  nq % 128 == 0, nk % 64 == 0. */
int rald_op_attention_bwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK, const void* V, int64_t ldv,
                          int64_t strideV, const void* O, int64_t ldo, int64_t strideO, const void* dO, int64_t lddo, int64_t strideDO,
                          void* dQ, int64_t lddq, int64_t strideDQ, void* dK, int64_t lddk, int64_t strideDK, void* dV, int64_t lddv, int64_t strideDV,
                          float* lse_scratch, float* delta_scratch, int32_t nq, int32_t nk, int32_t heads, int32_t batch, float scale, void* stream);
/* fused residual GEMM + next LayerNorm (N = 512): x[M][512] += A[M][K].W[512][K]^T + bias (fp32, in place);
 * h_bf16 = LayerNorm(x) * (add_one + g[row/rows_per_group*gstride + c]) + b[...] */
int rald_op_gemm_resid_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, void* h_bf16,
                          const float* g, const float* b, int64_t gstride, int32_t rows_per_group, float add_one, float eps,
                          int32_t M, int32_t K, void* stream);
int rald_op_cast_bf16(const float* in, void* out_bf16, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RALD_HIP_H */
