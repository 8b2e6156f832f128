// Host-side state of the autoencoder handle (see ae.hip).
#pragma once
#include "dit.h"

namespace rald {

struct Ae {
    rald_ae_config cfg;
    int d = 512;     // model dim
    int I = 512;     // heads*dim_head of the multi-head blocks
    DeviceArena arena;
    Stager stager;
    struct AttnW { bf16 *w_q = nullptr, *w_k = nullptr, *w_v = nullptr, *w_o = nullptr; float *b_o = nullptr, *ng = nullptr, *nb = nullptr, *cg = nullptr, *cb = nullptr; };
    struct FfW { bf16 *w1 = nullptr, *w2 = nullptr; float *b1 = nullptr, *b2 = nullptr, *ng = nullptr, *nb = nullptr; };
    struct Layer { bf16 *w_qk = nullptr, *w_v = nullptr, *w_o = nullptr; float *b_o = nullptr, *ng = nullptr, *nb = nullptr; FfW ff; };
    AttnW cross, dec;                  // cross: only the query LayerNorm (ng / nb) lives on the device - the rest is folded (ae_encode.hip)
    FfW cross_ff;
    std::vector<Layer> layers;
    float *basis = nullptr, *w_proj = nullptr, *b_proj = nullptr, *b_ml = nullptr, *w_fold = nullptr;
    bf16* w_ml = nullptr;
    // folded encoder (ae_encode.hip): weight-only tables built at finalize() from host copies of the reference's tensors
    float *enc_r = nullptr, *enc_q1 = nullptr, *enc_x0 = nullptr, *enc_t1 = nullptr, *enc_c3 = nullptr;
    bf16 *enc_t4 = nullptr, *enc_t3 = nullptr;
    std::vector<float> h_lat, h_dlat, h_mix_ng, h_mix_nb, h_mix_wq, h_mix_wkv, h_mix_wo, h_mix_bo, h_wqp, h_bqp, h_cross_cg, h_cross_cb, h_cross_wq,
        h_cross_wkv, h_cross_wo, h_cross_bo;
    // streaming query decoder (ae_decode.hip): weight-only tables built at finalize()
    float* t2aug = nullptr;            // [d][64] fp32: LN_ctx(x) . t2aug = per-latent score coefficients (slot order) | h0 | hb | u
    unsigned short* l_img = nullptr;   // [64][64] fp16 image: |L.f~|^2 = variance of the query embedding
    int basis_diag = 0;                // PointEmbed basis is the reference's block-diagonal one (models_ae.py:115-124)
    std::vector<float> h_basis, h_wpe, h_bpe, h_dec_ng, h_dec_nb;
    int* d_geglu_map = nullptr;
    float c0 = 0.f;
    std::vector<float> h_dec_wq, h_dec_wkv, h_dec_wo, h_dec_bo, h_out_w, h_out_b;
    std::set<std::string> expected, loaded;
    bool finalized = false;
    int64_t ws_generation = 0;   // bumped by every workspace reallocation (captured hipGraphs point into the workspace)
    // encode workspace
    int enc_batch = 0;
    unsigned short *e_f = nullptr, *e_gk = nullptr;   // fp16 feature rows per point: F (mix layer) and G = rstd.F (cross_attend), [B][Pp][64]
    bf16 *e_o = nullptr, *e_o2 = nullptr, *e_h = nullptr, *e_g = nullptr;
    float *e_x = nullptr, *e_q2 = nullptr, *e_ml = nullptr;
    float* e_part = nullptr;   // key-split partials of the two attentions at small batch (attention.hip)
    std::vector<void**> enc_ptrs() {
        return {(void**)&e_f, (void**)&e_gk, (void**)&e_o, (void**)&e_o2, (void**)&e_h, (void**)&e_g, (void**)&e_x, (void**)&e_q2, (void**)&e_ml, (void**)&e_part};
    }
    // latent-stack workspace
    int dec_batch = 0;
    float *x_x = nullptr, *x_part = nullptr;   // x_part: split-K partial sums of the small-batch FF2 (norm.hip resid_splitk_ln)
    bf16 *x_h = nullptr, *x_qk = nullptr, *x_vt = nullptr, *x_o = nullptr, *x_g = nullptr;
    float* x_y = nullptr;                      // [B*M][64] fp32 context projection (ae_ctx_build)
    std::vector<void**> dec_ptrs() {
        return {(void**)&x_x, (void**)&x_part, (void**)&x_h, (void**)&x_qk, (void**)&x_vt, (void**)&x_o, (void**)&x_g, (void**)&x_y};
    }

    int create();
    int load_attn(AttnW& a, int inner, const std::string& t, const float* data, int64_t nelem, bool* handled);
    int load_folded_attn(bool is_mix, const std::string& t, const float* data, int64_t nelem, bool* handled);
    int load_ff(FfW& f, const std::string& t, const float* data, int64_t nelem, bool* handled);
    int load_weight(const std::string& name, const float* data, int64_t nelem);
    int finalize();
    int reserve_encode(int B);
    int reserve_decode(int B);
    int encode(const float* pc, int B, const float* eps, float* mean_o, float* logvar_o, float* z, float* kl, hipStream_t st);
    // decoder context: 64-byte header (BlobRegistry, dit.h) + one record per sample
    static constexpr uint32_t CTX_MAGIC = 0x52414358u;           // "RACX"
    BlobRegistry ctx_registry;
    BlobHeader ctx_header(int B) const;
    int64_t ctx_bytes(int B) const;
    int decode_latents(const float* z, int B, void* ctx, hipStream_t st);
    int decode_queries(const void* ctx, const float* q, int B, int64_t Q, float* out, hipStream_t st, int nw = 0);
};

}  // namespace rald
