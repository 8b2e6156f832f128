// Host-side state of the denoiser handle (C++; the C-ABI wrappers live in api.hip).
#pragma once
#include <set>
#include <string>
#include <vector>

#include "../../include/rald_hip.h"
#include "common.h"
#include "kernels.h"

namespace rald {

const char* last_error();

struct DeviceArena {
    std::vector<void*> ptrs;
    void* alloc(size_t bytes, bool zero);
    void release(void* p);
    ~DeviceArena();
};

// fp32 host/device tensor -> packed device tensor (blocking; setup time only)
struct Stager {
    void* buf = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    int fetch(const float* data, int64_t nelem);
    int to_bf16(const float* data, bf16* dst, int rows, int cols, int64_t ld_dst, const int* rowmap);
    int to_f32(const float* data, float* dst, int rows, int cols, int64_t ld_dst, const int* rowmap);
    ~Stager();
};
std::vector<int> geglu_rowmap(int inner);

// ---- radar-spectrum encoder (radar.hip): model/models_radar_encoder.py Encoder + the
// tokeniser half of EDMPrecond.process_radar_cond
struct RadarEncoder {
    struct Impl;
    Impl* impl = nullptr;
    int create(int ch, int z_ch, int R, int A, int E, int token_ch, DeviceArena* arena, int in_channels = 1);
    bool all_loaded(std::string* missing) const;
    void expected_keys(const std::string& prefix, std::set<std::string>& out) const;
    int load_weight(const std::string& name, const float* data, int64_t nelem, Stager& st);        // keys below "radar_enc."
    int load_token_weight(const std::string& name, const float* data, int64_t nelem, Stager& st);  // radar_*_emb / radar_token_project
    // cube [B,R,A,E,2] -> *tokens (handle-owned, [B, R/16*A/16*E/16, token_ch] fp32)
    int tokens(const float* cube, int B, float** tokens, hipStream_t st);
    // cube channel 0 -> encoder latent z [B, r, a, e, z_ch] fp32 (RadarAutoencoder._encode layout)
    int encode(const float* cube, int cube_ch, int B, float** z, hipStream_t st);
    ~RadarEncoder();
};

struct Dit {
    rald_dit_config cfg;
    int D = 512;
    DeviceArena arena;
    Stager stager;
    struct Layer {
        bf16 *w_qk, *w_v, *w_o, *w_q2, *w_o2, *w_ff1, *w_ff2;
        bf16* w_q2t = nullptr;                                // attn2.to_q transposed [k][h*64+d]: operand of the folded condition keys (cond_fold)
        float *b_o, *b_o2, *b_ff1, *b_ff2;
        // MXFP8 copies of the attention projections (cfg.qkv_dtype == 1): e4m3 elements [rows][D] + e8m0 scales [rows][D/32]
        unsigned char *q8_qk = nullptr, *s8_qk = nullptr, *q8_v = nullptr, *s8_v = nullptr, *q8_q2 = nullptr, *s8_q2 = nullptr;
        unsigned char *q8_ff1 = nullptr, *s8_ff1 = nullptr;   // qkv_dtype >= 2: the GEGLU projection too (packed row order)
        unsigned char *q8_ff2 = nullptr, *s8_ff2 = nullptr;   // qkv_dtype == 3: and the feed-forward's output projection
    };
    std::vector<Layer> layers;
    bf16 *w_k2_all = nullptr, *w_v2_all = nullptr;      // attn2.to_k / to_v of every block stacked: [L*D, context_dim]
    float *w_mod = nullptr, *b_mod = nullptr;           // all AdaLN linears stacked: [(L*3)*2D, D] fp32
    float *w_t0 = nullptr, *b_t0 = nullptr, *w_t1 = nullptr, *b_t1 = nullptr;
    float *w_in = nullptr, *w_out = nullptr, *norm_g = nullptr, *norm_b = nullptr, *coef_raw = nullptr;
    bf16* w_out_hl = nullptr;                           // proj_out weight split into bf16 hi | lo (final_norm_proj's large-M form), built at finalize
    int* d_geglu_map = nullptr;
    RadarEncoder radar;
    std::set<std::string> expected, loaded;
    bool finalized = false;
    // noise-level tables: slot 0 = ad-hoc (rald_dit_set_sigmas / forward), slot 1 = the sampler's schedule.
    // Separate slots so that a captured hipGraph of the sampler keeps pointing at valid modulations
    // even if forward() is called with other sigmas in between.
    struct SigmaTable {
        std::string key;
        std::vector<float> host;
        int n = 0, cap = 0;
        float *sigma = nullptr, *coef = nullptr, *cnoise = nullptr, *pe = nullptr, *temb0 = nullptr, *temb = nullptr, *mod = nullptr;
    };
    SigmaTable tables[2];
    int build_table(SigmaTable& t, const float* sig, int n, hipStream_t st);
    int64_t mod_row() const { return (int64_t)cfg.depth * 3 * 2 * D; }
    // activation workspace.  ws_generation counts every reallocation of a buffer that a captured hipGraph may point
    // at (workspace, sigma tables): owners of captured graphs compare it with the value at capture time.
    int64_t ws_generation = 0;
    int ws_batch = 0;
    float* ws_part = nullptr;   // split-K partial sums of the small-batch FF2 (norm.hip resid_splitk_ln)
    float *ws_x = nullptr, *ws_xcur = nullptr, *ws_xeul = nullptr, *ws_den = nullptr, *ws_dcur = nullptr;
    unsigned char *ws_h8 = nullptr, *ws_hs = nullptr;   // MXFP8 AdaLN outputs feeding q/k/v (qkv_dtype >= 1)
    unsigned char *ws_g8 = nullptr, *ws_gs = nullptr;   // MXFP8 GEGLU output feeding ff.net.2 (qkv_dtype == 3)
    bf16 *ws_h = nullptr, *ws_qk = nullptr, *ws_vt = nullptr, *ws_o = nullptr, *ws_q2 = nullptr, *ws_g = nullptr, *ws_tok = nullptr;

    // live timing of the dominant kernel (the FF1 GEGLU GEMM) with HIP events on the launch stream
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    int prof_used = 0;
    int profile_begin();
    int profile_end(double* total_ms, int* launches);
    ~Dit();

    int create();
    int load_weight(const std::string& name, const float* data, int64_t nelem);
    int finalize();
    int reserve(int B);
    int set_sigmas(const float* sig, int n, hipStream_t st);
    int64_t cond_cache_bytes(int B) const;
    // Folded cross-attention (large batches, bf16): per sample and block the condition's keys absorb to_q and its values absorb to_out,
    //   Gt[(h,key)][k] = scale.log2e . sum_d K_h[key][d] Wq[h*64+d][k]      Ut[n][(h,key)] = sum_d Wo[n][h*64+d] V_h[key][d]
    // so the sub-block is  P = softmax_64(h.Gt^T)  (one GEMM with a softmax epilogue)  and  x += P.Ut^T + b  (the residual+LN GEMM with
    // per-sample weights): two launches instead of three, no attention kernel, no Q / O round trip.
    bool cond_fold(int B) const;
    int encode_cond_tokens(const float* tokens, int B, void* cache, hipStream_t st);
    int encode_cond(const float* cube, int B, float* out_tokens, void* cache, hipStream_t st);
    int denoise(const float* x, int B, int sigma_row, int per_sample, const void* cache, float* out, int raw_F, hipStream_t st, int slot = 0);
    int sample(const float* latents, int B, const void* cache, int num_steps, float smin, float smax, float rho, float* out,
               hipStream_t st);
};

}  // namespace rald
