// Set-latent autoencoder handle (KLAutoEncoder, query_type='mix'; model/models_ae.py:284-432).
//
//   encode          :351-405   PointEmbed -> mix query (8x64-head attention over the P points,
//                              no residual) -> query_proj -> 1-head d=dim cross-attention over the
//                              points + residual -> GEGLU FF + residual -> mean/logvar -> posterior
//                              (both attentions folded onto the points' 52 Fourier features: ae_encode.hip)
//   decode_latents  :410-414   proj -> depth x (self-attention, GEGLU FF), then the decoder context
//   decode_queries  :417-424   PointEmbed(queries) -> 1-head cross-attention over the latents -> Linear(dim,1)
//
// Decode is algebraically folded (exact in real arithmetic; the reference has no nonlinearity
// between these Linears): to_outputs(to_out(P.V)) = P.u + c with u = LN_ctx(x).(Wv^T.Wo^T.w_out), and the
// scores S = LN(PointEmbed(q)).Wq^T.K^T collapse onto the 51 Fourier features of the query (ae_decode.hip):
// per sample a [M x 64] coefficient table, per query a K = 64 contraction + a softmax-weighted dot with u,
// all inside ONE streaming kernel (12 B in, 4 B out per query).
#include "ae.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rald {

static int fetch_host(std::vector<float>& dst, const float* data, int64_t n) {
    dst.resize((size_t)n);
    RALD_HIP(hipMemcpy(dst.data(), data, (size_t)n * 4, hipMemcpyDefault));
    return 0;
}

int Ae::create() {
    const auto& c = cfg;
    d = c.dim;
    I = c.heads * c.dim_head;
    RALD_CHECK(c.dim_head == 64 && I == 512, "ae: heads*dim_head must be 8x64 (create_autoencoder hard-codes it, models_ae.py:447-458)");
    RALD_CHECK(d == 256 || d == 512, "ae: dim must be 256 or 512");
    RALD_CHECK(c.num_latents > 0 && c.num_latents % 64 == 0, "ae: num_latents must be a multiple of 64");
    RALD_CHECK(c.latent_dim >= 1 && c.latent_dim <= 64 && (2 * c.latent_dim) % 4 == 0, "ae: latent_dim must be in [2,64] and even");
    RALD_CHECK(c.depth >= 1 && c.depth <= 256, "ae: bad depth");
    RALD_CHECK(c.num_inputs >= 32, "ae: num_inputs too small");
    RALD_CHECK(c.query_type == 0 || c.query_type == 1, "ae: query_type must be 0 ('mix') or 1 ('learnable'); 'point' needs torch_cluster.fps");
    const bool mixq = c.query_type == 0;
    const int M = c.num_latents, L = c.latent_dim;
    auto B16 = [&](size_t n) { return (bf16*)arena.alloc(n * 2, true); };
    auto F32 = [&](size_t n) { return (float*)arena.alloc(n * 4, true); };
    basis = F32(72);
    auto mk_attn = [&](AttnW& a, int inner, bool ctx_norm) {
        a.w_q = B16((size_t)inner * d); a.w_k = B16((size_t)inner * d); a.w_v = B16((size_t)inner * d);
        a.w_o = B16((size_t)d * inner); a.b_o = F32(d); a.ng = F32(d); a.nb = F32(d);
        if (ctx_norm) { a.cg = F32(d); a.cb = F32(d); }
    };
    auto mk_ff = [&](FfW& f) {
        f.w1 = B16((size_t)8 * d * d); f.b1 = F32((size_t)8 * d); f.w2 = B16((size_t)d * 4 * d); f.b2 = F32(d);
        f.ng = F32(d); f.nb = F32(d);
    };
    cross.ng = F32(d); cross.nb = F32(d);
    mk_ff(cross_ff);
    mk_attn(dec, d, true);
    layers.resize(c.depth);
    for (auto& l : layers) {
        l.w_qk = B16((size_t)3 * I * d); l.w_v = l.w_qk + (size_t)2 * I * d;     // to_q | to_k | to_v stacked (w_v = alias of rows 2I..3I)
        l.w_o = B16((size_t)d * I); l.b_o = F32(d);
        l.ng = F32(d); l.nb = F32(d);
        mk_ff(l.ff);
    }
    enc_r = F32(52 * 52); enc_x0 = F32((size_t)M * d); enc_t1 = F32((size_t)d * 64); enc_c3 = F32(d);
    enc_t3 = B16((size_t)d * 64);
    if (mixq) { enc_q1 = F32((size_t)M * I); enc_t4 = B16((size_t)d * I); }
    w_proj = F32((size_t)d * L); b_proj = F32(d);
    w_ml = B16((size_t)2 * L * d); b_ml = F32((size_t)2 * L);
    t2aug = F32((size_t)d * 64);
    l_img = (unsigned short*)arena.alloc(64 * 64 * 2, true);
    w_fold = F32(d);
    std::vector<int> rm = geglu_rowmap(4 * d);
    d_geglu_map = (int*)arena.alloc(rm.size() * 4, false);
    RALD_CHECK(d_geglu_map && w_fold && t2aug && l_img, "ae: device allocation failed");
    RALD_HIP(hipMemcpy(d_geglu_map, rm.data(), rm.size() * 4, hipMemcpyHostToDevice));

    expected.clear();
    char buf[160];
    auto add_attn = [&](const std::string& p, bool ctx) {
        for (const char* n : {"fn.to_q.weight", "fn.to_kv.weight", "fn.to_out.weight", "fn.to_out.bias", "norm.weight", "norm.bias"})
            expected.insert(p + n);
        if (ctx) { expected.insert(p + "norm_context.weight"); expected.insert(p + "norm_context.bias"); }
    };
    auto add_ff = [&](const std::string& p) {
        for (const char* n : {"fn.net.0.weight", "fn.net.0.bias", "fn.net.2.weight", "fn.net.2.bias", "norm.weight", "norm.bias"})
            expected.insert(p + n);
    };
    add_attn("cross_attend_blocks.0.", true);
    add_ff("cross_attend_blocks.1.");
    for (const char* n : {"point_embed.basis", "point_embed.mlp.weight", "point_embed.mlp.bias", "to_outputs.weight", "to_outputs.bias",
                          "proj.weight", "proj.bias", "mean_fc.weight", "mean_fc.bias", "logvar_fc.weight", "logvar_fc.bias"})
        expected.insert(n);
    if (mixq) {
        for (const char* n : {"s_latents.weight", "d_latents.weight", "query_proj.weight", "query_proj.bias"}) expected.insert(n);
        add_attn("mix_attn_layer.", false);
    } else {
        expected.insert("latents.weight");
    }
    add_attn("decoder_cross_attn.", true);
    for (int i = 0; i < c.depth; ++i) {
        snprintf(buf, sizeof(buf), "layers.%d.0.", i); add_attn(buf, false);
        snprintf(buf, sizeof(buf), "layers.%d.1.", i); add_ff(buf);
    }
    return 0;
}

int Ae::load_attn(AttnW& a, int inner, const std::string& t, const float* data, int64_t nelem, bool* handled) {
    *handled = true;
    auto need = [&](int64_t n) -> int { RALD_CHECK(nelem == n, "ae: size mismatch for an attention tensor (" + t + ")"); return 0; };
    if (t == "fn.to_q.weight") { RALD_TRY(need((int64_t)inner * d)); return stager.to_bf16(data, a.w_q, inner, d, d, nullptr); }
    if (t == "fn.to_kv.weight") {     // first half of the rows is k, second half v (models_ae.py:89)
        RALD_TRY(need((int64_t)2 * inner * d));
        RALD_TRY(stager.fetch(data, nelem));
        RALD_TRY(pack_rows_bf16((const float*)stager.buf, a.w_k, inner, d, d, nullptr, nullptr));
        RALD_TRY(pack_rows_bf16((const float*)stager.buf + (size_t)inner * d, a.w_v, inner, d, d, nullptr, nullptr));
        RALD_HIP(hipDeviceSynchronize());
        return 0;
    }
    if (t == "fn.to_out.weight") { RALD_TRY(need((int64_t)d * inner)); return stager.to_bf16(data, a.w_o, d, inner, inner, nullptr); }
    if (t == "fn.to_out.bias") { RALD_TRY(need(d)); return stager.to_f32(data, a.b_o, 1, d, d, nullptr); }
    if (t == "norm.weight") { RALD_TRY(need(d)); return stager.to_f32(data, a.ng, 1, d, d, nullptr); }
    if (t == "norm.bias") { RALD_TRY(need(d)); return stager.to_f32(data, a.nb, 1, d, d, nullptr); }
    if (t == "norm_context.weight" && a.cg) { RALD_TRY(need(d)); return stager.to_f32(data, a.cg, 1, d, d, nullptr); }
    if (t == "norm_context.bias" && a.cb) { RALD_TRY(need(d)); return stager.to_f32(data, a.cb, 1, d, d, nullptr); }
    *handled = false;
    return 0;
}

// mix_attn_layer / cross_attend_blocks.0: kept on the host for the fold at finalize(); only cross_attend's query LayerNorm runs on the device
int Ae::load_folded_attn(bool is_mix, const std::string& t, const float* data, int64_t nelem, bool* handled) {
    *handled = true;
    const int inner = is_mix ? I : d;
    auto need = [&](int64_t n) -> int { RALD_CHECK(nelem == n, "ae: size mismatch for an encoder attention tensor (" + t + ")"); return 0; };
    if (t == "fn.to_q.weight") { RALD_TRY(need((int64_t)inner * d)); return fetch_host(is_mix ? h_mix_wq : h_cross_wq, data, nelem); }
    if (t == "fn.to_kv.weight") { RALD_TRY(need((int64_t)2 * inner * d)); return fetch_host(is_mix ? h_mix_wkv : h_cross_wkv, data, nelem); }
    if (t == "fn.to_out.weight") { RALD_TRY(need((int64_t)d * inner)); return fetch_host(is_mix ? h_mix_wo : h_cross_wo, data, nelem); }
    if (t == "fn.to_out.bias") { RALD_TRY(need(d)); return fetch_host(is_mix ? h_mix_bo : h_cross_bo, data, nelem); }
    if (t == "norm.weight") { RALD_TRY(need(d)); return is_mix ? fetch_host(h_mix_ng, data, nelem) : stager.to_f32(data, cross.ng, 1, d, d, nullptr); }
    if (t == "norm.bias") { RALD_TRY(need(d)); return is_mix ? fetch_host(h_mix_nb, data, nelem) : stager.to_f32(data, cross.nb, 1, d, d, nullptr); }
    if (!is_mix && t == "norm_context.weight") { RALD_TRY(need(d)); return fetch_host(h_cross_cg, data, nelem); }
    if (!is_mix && t == "norm_context.bias") { RALD_TRY(need(d)); return fetch_host(h_cross_cb, data, nelem); }
    *handled = false;
    return 0;
}

int Ae::load_ff(FfW& f, const std::string& t, const float* data, int64_t nelem, bool* handled) {
    *handled = true;
    auto need = [&](int64_t n) -> int { RALD_CHECK(nelem == n, "ae: size mismatch for a feed-forward tensor (" + t + ")"); return 0; };
    if (t == "fn.net.0.weight") { RALD_TRY(need((int64_t)8 * d * d)); return stager.to_bf16(data, f.w1, 8 * d, d, d, d_geglu_map); }
    if (t == "fn.net.0.bias") { RALD_TRY(need((int64_t)8 * d)); return stager.to_f32(data, f.b1, 8 * d, 1, 1, d_geglu_map); }
    if (t == "fn.net.2.weight") { RALD_TRY(need((int64_t)d * 4 * d)); return stager.to_bf16(data, f.w2, d, 4 * d, 4 * d, nullptr); }
    if (t == "fn.net.2.bias") { RALD_TRY(need(d)); return stager.to_f32(data, f.b2, 1, d, d, nullptr); }
    if (t == "norm.weight") { RALD_TRY(need(d)); return stager.to_f32(data, f.ng, 1, d, d, nullptr); }
    if (t == "norm.bias") { RALD_TRY(need(d)); return stager.to_f32(data, f.nb, 1, d, d, nullptr); }
    *handled = false;
    return 0;
}

int Ae::load_weight(const std::string& name, const float* data, int64_t nelem) {
    RALD_CHECK(expected.count(name), "ae: unexpected key '" + name + "'");
    const int M = cfg.num_latents, L = cfg.latent_dim;
    auto need = [&](int64_t n) -> int {
        RALD_CHECK(nelem == n, "ae: size mismatch for '" + name + "': got " + std::to_string(nelem) + ", expected " + std::to_string(n));
        return 0;
    };
    bool handled = false;
    int rc = 0, li = -1, sub = -1;
    char tail[128] = {0};
    if (name.rfind("cross_attend_blocks.0.", 0) == 0) rc = load_folded_attn(false, name.substr(22), data, nelem, &handled);
    else if (name.rfind("cross_attend_blocks.1.", 0) == 0) rc = load_ff(cross_ff, name.substr(22), data, nelem, &handled);
    else if (name.rfind("mix_attn_layer.", 0) == 0) rc = load_folded_attn(true, name.substr(15), data, nelem, &handled);
    else if (name.rfind("decoder_cross_attn.", 0) == 0) {
        const std::string t = name.substr(19);
        // host copies of the tensors that are folded at finalize()
        if (t == "fn.to_q.weight") RALD_TRY(fetch_host(h_dec_wq, data, nelem));
        if (t == "fn.to_kv.weight") RALD_TRY(fetch_host(h_dec_wkv, data, nelem));
        if (t == "fn.to_out.weight") RALD_TRY(fetch_host(h_dec_wo, data, nelem));
        if (t == "fn.to_out.bias") RALD_TRY(fetch_host(h_dec_bo, data, nelem));
        if (t == "norm.weight") RALD_TRY(fetch_host(h_dec_ng, data, nelem));
        if (t == "norm.bias") RALD_TRY(fetch_host(h_dec_nb, data, nelem));
        rc = load_attn(dec, d, t, data, nelem, &handled);
    } else if (sscanf(name.c_str(), "layers.%d.%d.%127s", &li, &sub, tail) == 3) {
        RALD_CHECK(li >= 0 && li < cfg.depth && (sub == 0 || sub == 1), "ae: bad layer index in '" + name + "'");
        Layer& l = layers[li];
        const std::string t(tail);
        if (sub == 1) rc = load_ff(l.ff, t, data, nelem, &handled);
        else {
            handled = true;
            if (t == "fn.to_q.weight") { RALD_TRY(need((int64_t)I * d)); rc = stager.to_bf16(data, l.w_qk, I, d, d, nullptr); }
            else if (t == "fn.to_kv.weight") {
                RALD_TRY(need((int64_t)2 * I * d));
                RALD_TRY(stager.fetch(data, nelem));
                RALD_TRY(pack_rows_bf16((const float*)stager.buf, l.w_qk + (size_t)I * d, I, d, d, nullptr, nullptr));
                RALD_TRY(pack_rows_bf16((const float*)stager.buf + (size_t)I * d, l.w_v, I, d, d, nullptr, nullptr));
                RALD_HIP(hipDeviceSynchronize());
            }
            else if (t == "fn.to_out.weight") { RALD_TRY(need((int64_t)d * I)); rc = stager.to_bf16(data, l.w_o, d, I, I, nullptr); }
            else if (t == "fn.to_out.bias") { RALD_TRY(need(d)); rc = stager.to_f32(data, l.b_o, 1, d, d, nullptr); }
            else if (t == "norm.weight") { RALD_TRY(need(d)); rc = stager.to_f32(data, l.ng, 1, d, d, nullptr); }
            else if (t == "norm.bias") { RALD_TRY(need(d)); rc = stager.to_f32(data, l.nb, 1, d, d, nullptr); }
            else handled = false;
        }
    } else {
        handled = true;
        if (name == "point_embed.basis") { RALD_TRY(need(72)); RALD_TRY(fetch_host(h_basis, data, nelem)); rc = stager.to_f32(data, basis, 1, 72, 72, nullptr); }
        else if (name == "point_embed.mlp.weight") { RALD_TRY(need((int64_t)d * 51)); rc = fetch_host(h_wpe, data, nelem); }
        else if (name == "point_embed.mlp.bias") { RALD_TRY(need(d)); rc = fetch_host(h_bpe, data, nelem); }
        else if (name == "s_latents.weight" || name == "latents.weight") { RALD_TRY(need((int64_t)M * d)); rc = fetch_host(h_lat, data, nelem); }
        else if (name == "d_latents.weight") { RALD_TRY(need((int64_t)M * d)); rc = fetch_host(h_dlat, data, nelem); }
        else if (name == "query_proj.weight") { RALD_TRY(need((int64_t)d * d)); rc = fetch_host(h_wqp, data, nelem); }
        else if (name == "query_proj.bias") { RALD_TRY(need(d)); rc = fetch_host(h_bqp, data, nelem); }
        else if (name == "to_outputs.weight") { RALD_TRY(need(d)); rc = fetch_host(h_out_w, data, nelem); }
        else if (name == "to_outputs.bias") { RALD_TRY(need(1)); rc = fetch_host(h_out_b, data, nelem); }
        else if (name == "proj.weight") { RALD_TRY(need((int64_t)d * L)); rc = stager.to_f32(data, w_proj, d, L, L, nullptr); }
        else if (name == "proj.bias") { RALD_TRY(need(d)); rc = stager.to_f32(data, b_proj, 1, d, d, nullptr); }
        else if (name == "mean_fc.weight") { RALD_TRY(need((int64_t)L * d)); rc = stager.to_bf16(data, w_ml, L, d, d, nullptr); }
        else if (name == "logvar_fc.weight") { RALD_TRY(need((int64_t)L * d)); rc = stager.to_bf16(data, w_ml + (size_t)L * d, L, d, d, nullptr); }
        else if (name == "mean_fc.bias") { RALD_TRY(need(L)); rc = stager.to_f32(data, b_ml, 1, L, L, nullptr); }
        else if (name == "logvar_fc.bias") { RALD_TRY(need(L)); rc = stager.to_f32(data, b_ml + L, 1, L, L, nullptr); }
        else handled = false;
    }
    if (rc) return rc;
    RALD_CHECK(handled, "ae: unknown key '" + name + "'");
    loaded.insert(name);
    finalized = false;
    return 0;
}

int Ae::finalize() {
    for (const auto& k : expected) RALD_CHECK(loaded.count(k), "ae: missing key '" + k + "' (strict load)");
    const int M = cfg.num_latents;
    // (1) the folded encoder's weight-only tables (ae_encode.hip), in double on the host
    {
        const bool mixq = cfg.query_type == 0;
        std::vector<float> Rf, Q1, T4, X0, T1, T3, c3;
        RALD_TRY(ae_encode_tables(d, I, M, cfg.heads, mixq, h_wpe.data(), h_bpe.data(), h_dlat.data(), h_mix_ng.data(), h_mix_nb.data(), h_mix_wq.data(),
                                  h_mix_wkv.data(), h_mix_wo.data(), h_mix_bo.data(), h_lat.data(), h_wqp.data(), h_bqp.data(), h_cross_cg.data(),
                                  h_cross_cb.data(), h_cross_wq.data(), h_cross_wkv.data(), h_cross_wo.data(), h_cross_bo.data(), Rf, Q1, T4, X0, T1, T3, c3));
        RALD_HIP(hipMemcpy(enc_r, Rf.data(), Rf.size() * 4, hipMemcpyHostToDevice));
        RALD_HIP(hipMemcpy(enc_x0, X0.data(), X0.size() * 4, hipMemcpyHostToDevice));
        RALD_HIP(hipMemcpy(enc_t1, T1.data(), T1.size() * 4, hipMemcpyHostToDevice));
        RALD_HIP(hipMemcpy(enc_c3, c3.data(), c3.size() * 4, hipMemcpyHostToDevice));
        RALD_TRY(stager.to_bf16(T3.data(), enc_t3, d, 64, 64, nullptr));
        if (mixq) {
            RALD_HIP(hipMemcpy(enc_q1, Q1.data(), Q1.size() * 4, hipMemcpyHostToDevice));
            RALD_TRY(stager.to_bf16(T4.data(), enc_t4, d, I, I, nullptr));
        }
    }
    // (2) decoder folding, in double on the host:
    //     wo' = Wo^T.w_out [d];  w_fold = Wv^T.wo' [d];  c0 = b_o.w_out + b_out;  WqT for G = K.Wq
    RALD_CHECK((int64_t)h_dec_wq.size() == (int64_t)d * d && (int64_t)h_dec_wkv.size() == (int64_t)2 * d * d &&
               (int64_t)h_dec_wo.size() == (int64_t)d * d && (int)h_dec_bo.size() == d && (int)h_out_w.size() == d && h_out_b.size() == 1,
               "ae: decoder tensors missing for folding");
    std::vector<double> wo1(d, 0.0);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) wo1[j] += (double)h_dec_wo[(size_t)i * d + j] * h_out_w[i];
    std::vector<float> wf(d);
    const float* Wv = h_dec_wkv.data() + (size_t)d * d;
    for (int c = 0; c < d; ++c) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += (double)Wv[(size_t)j * d + c] * wo1[j];
        wf[c] = (float)s;
    }
    double c = h_out_b[0];
    for (int i = 0; i < d; ++i) c += (double)h_dec_bo[i] * h_out_w[i];
    c0 = (float)c;
    RALD_HIP(hipMemcpy(w_fold, wf.data(), (size_t)d * 4, hipMemcpyHostToDevice));
    // (3) the streaming query decoder's weight-only tables (ae_decode.hip): score coefficients and the variance factor
    RALD_CHECK((int)h_dec_ng.size() == d && (int)h_dec_nb.size() == d && (int64_t)h_wpe.size() == (int64_t)d * 51 && (int)h_bpe.size() == d &&
               h_basis.size() == 72, "ae: decoder tensors missing for the query-decode tables");
    std::vector<float> t2;
    std::vector<unsigned short> limg;
    RALD_TRY(ae_decode_tables(d, h_dec_wq.data(), h_dec_wkv.data(), h_dec_ng.data(), h_dec_nb.data(), h_wpe.data(), h_bpe.data(), wf.data(), t2, limg));
    RALD_HIP(hipMemcpy(t2aug, t2.data(), t2.size() * 4, hipMemcpyHostToDevice));
    RALD_HIP(hipMemcpy(l_img, limg.data(), limg.size() * 2, hipMemcpyHostToDevice));
    basis_diag = 1;                    // block-diagonal basis (x -> columns 0-7, y -> 8-15, z -> 16-23): one multiply per projection
    for (int a = 0; a < 3; ++a)
        for (int e = 0; e < 24; ++e)
            if (e / 8 != a && h_basis[(size_t)a * 24 + e] != 0.f) basis_diag = 0;
    finalized = true;
    return 0;
}

int Ae::reserve_encode(int B) {
    if (B <= enc_batch) return 0;
    RALD_HIP(hipDeviceSynchronize());
    ++ws_generation;
    for (void** p : enc_ptrs()) if (*p) { arena.release(*p); *p = nullptr; }
    const size_t P = cfg.num_inputs, Pp = round_up(P, 64), M = cfg.num_latents, L = cfg.latent_dim;
    const size_t b = B;
    e_f = (unsigned short*)arena.alloc(b * Pp * 64 * 2, true);
    e_gk = (unsigned short*)arena.alloc(b * Pp * 64 * 2, true);
    e_o = (bf16*)arena.alloc(b * M * I * 2, true);
    e_o2 = (bf16*)arena.alloc(b * M * 64 * 2, true);
    e_x = (float*)arena.alloc(b * M * d * 4, true);
    e_q2 = (float*)arena.alloc(b * M * 64 * 4, true);
    e_h = (bf16*)arena.alloc(b * M * d * 2, true);
    e_g = (bf16*)arena.alloc(b * M * 4 * d * 2, true);
    e_ml = (float*)arena.alloc(b * M * 2 * L * 4, true);
    {   // key-split partials of the larger of the two attentions (any batch up to B: the split count falls as the batch grows)
        int64_t need = 0;
        for (int bb = 1; bb <= B; ++bb) {
            const int64_t n1 = cfg.query_type == 0 ? attention_split_scratch_bytes(attention_pick_ksplit((int)M, (int)P, cfg.heads, bb), (int)M, cfg.heads, bb) : 0;
            const int64_t n2 = attention_split_scratch_bytes(attention_pick_ksplit((int)M, (int)P, 1, bb), (int)M, 1, bb);
            need = std::max(need, std::max(n1, n2));
        }
        e_part = (float*)arena.alloc((size_t)need, true);
    }
    for (void** p : enc_ptrs()) RALD_CHECK(*p, "ae: encode workspace allocation failed");
    enc_batch = B;
    return 0;
}

int Ae::encode(const float* pc, int B, const float* eps, float* mean_o, float* logvar_o, float* z, float* kl, hipStream_t st) {
    RALD_CHECK(finalized, "ae: weights not finalized");
    RALD_CHECK(pc && eps && z && kl && B >= 1, "ae: bad arguments");
    RALD_TRY(reserve_encode(B));
    const int P = cfg.num_inputs, Pp = (int)round_up(P, 64), M = cfg.num_latents, L = cfg.latent_dim;
    const int BM = B * M;
    // ---- per point: the Fourier features of PointEmbed (:355) and 1/std of its embedding, as fp16 key = value rows
    RALD_TRY(ae_enc_features(pc, basis, enc_r, e_f, e_gk, B, P, Pp, st));
    auto folded_attention = [&](const float* Q, int64_t ldq, int64_t strideQ, const unsigned short* KV, bf16* O, int64_t ldo, int heads) -> int {
        AttnArgs a;
        a.Q = nullptr; a.Qf = Q; a.ldq = ldq; a.strideQ = strideQ; a.f16 = 1;
        a.K = (const bf16*)KV; a.ldk = 64; a.strideK = (int64_t)Pp * 64;
        a.Vt = nullptr; a.ldvt = 0; a.strideVt = 0;
        a.V = (const bf16*)KV; a.ldv = 64; a.strideV = (int64_t)Pp * 64; a.v_padded = 1;
        a.hsk = 0;                                                         // every head reads the same 64 feature columns
        a.O = O; a.ldo = ldo; a.strideO = (int64_t)M * ldo;
        a.nq = M; a.nk = P; a.k_rows = Pp; a.heads = heads; a.batch = B; a.scale = 1.f; a.q_prescaled = 1;
        a.ksplit = attention_pick_ksplit(M, P, heads, B);                  // 512 queries x 10 000 keys: few workgroups at small B without it
        a.part = e_part;
        return attention_d64(a, st);
    };
    const float* xin = nullptr;
    if (cfg.query_type == 0) {
        // ---- mix query (:380-386): O1[m][64h + j] = sum_p softmax_p(Q1_h[m].f_p) f_p[j];  x = X0 + O1.T4^T
        RALD_TRY(folded_attention(enc_q1, I, 0, e_f, e_o, I, cfg.heads));
        GemmArgs g4 = gemm_args(e_o, I, enc_t4, I, e_x, d, nullptr, BM, d, I);
        RALD_TRY(gemm_nt(g4, EPI_F32, st));
        xin = e_x;
    }
    // ---- x = cross_attn(x, context=pc_embeddings) + x (:395): Q' = LN(x).T1; O' = softmax(Q'.g_p) g_p; x += O'.T3^T + c3
    RALD_TRY(ae_enc_qproj(xin, enc_x0, e_x, cross.ng, cross.nb, enc_t1, e_q2, BM, M, d, st));
    RALD_TRY(folded_attention(e_q2, 64, (int64_t)M * 64, e_gk, e_o2, 64, 1));
    GemmArgs g3 = gemm_args(e_o2, 64, enc_t3, 64, e_x, d, enc_c3, BM, d, 64);
    RALD_TRY(gemm_nt(g3, EPI_RESID, st));
    // ---- x = cross_ff(x) + x                                                      (:396)
    RALD_TRY(layernorm_mod(e_x, e_h, BM, d, cross_ff.ng, cross_ff.nb, 0, 1 << 30, 0.f, 1e-5f, st));
    GemmArgs f1 = gemm_args(e_h, d, cross_ff.w1, d, e_g, 4 * d, cross_ff.b1, BM, 8 * d, d);
    RALD_TRY(gemm_nt(f1, EPI_GEGLU, st));
    GemmArgs f2 = gemm_args(e_g, 4 * d, cross_ff.w2, 4 * d, e_x, d, cross_ff.b2, BM, d, 4 * d);
    RALD_TRY(gemm_nt(f2, EPI_RESID, st));
    // ---- mean / logvar (:398-399) and the posterior (:401-403)
    RALD_TRY(cast_f32_bf16(e_x, e_h, (int64_t)BM * d, st));
    GemmArgs ml = gemm_args(e_h, d, w_ml, d, e_ml, 2 * L, b_ml, BM, 2 * L, d);
    RALD_TRY(gemm_nt(ml, EPI_F32, st));
    RALD_TRY(posterior(e_ml, eps, mean_o, logvar_o, z, kl, B, M, L, st));
    return 0;
}

int Ae::reserve_decode(int B) {
    if (B <= dec_batch) return 0;
    RALD_HIP(hipDeviceSynchronize());
    ++ws_generation;
    for (void** p : dec_ptrs()) if (*p) { arena.release(*p); *p = nullptr; }
    const size_t M = cfg.num_latents, b = B;
    x_x = (float*)arena.alloc(b * M * d * 4, true);
    {   // split-K partials of the small-batch FF2 (4 slabs) / per-head partials of the fused attention sub-block (heads slabs, <= 2048 rows)
        const size_t rows = b * M, r4 = rows < (size_t)splitk_max_rows() ? rows : (size_t)splitk_max_rows(), r8 = rows < 2048 ? rows : 2048;
        x_part = (float*)arena.alloc((4 * r4 > (size_t)cfg.heads * r8 ? 4 * r4 : (size_t)cfg.heads * r8) * 512 * 4, true);
    }
    x_h = (bf16*)arena.alloc(b * M * d * 2, true);
    x_qk = (bf16*)arena.alloc(b * M * 3 * I * 2, true);       // q | k | v
    x_vt = (bf16*)arena.alloc(b * I * M * 2, true);
    x_o = (bf16*)arena.alloc(b * M * I * 2, true);
    x_g = (bf16*)arena.alloc(b * M * 4 * d * 2, true);
    x_y = (float*)arena.alloc(b * M * 64 * 4 + b * 4 + 16, true);        // projection [B*M][64] + one |max| word per sample
    for (void** p : dec_ptrs()) RALD_CHECK(*p, "ae: decode workspace allocation failed");
    dec_batch = B;
    return 0;
}

int64_t Ae::ctx_bytes(int B) const {
    // 64-byte header, then per sample: fp16 coefficient image [M][64] + u [M] f32 + the image's scale (ae_decode.hip)
    return BLOB_HEADER_BYTES + (int64_t)B * ae_ctx_stride(cfg.num_latents);
}
BlobHeader Ae::ctx_header(int B) const {
    BlobHeader hd{};
    uint32_t h = 2166136261u;
    for (int v : {cfg.dim, cfg.num_latents, cfg.latent_dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.query_type}) { h ^= (uint32_t)v; h *= 16777619u; }
    hd.magic = CTX_MAGIC; hd.batch = B; hd.flag = 0; hd.cfg_hash = h; hd.bytes = ctx_bytes(B);
    return hd;
}

int Ae::decode_latents(const float* z, int B, void* ctx, hipStream_t st) {
    RALD_CHECK(finalized, "ae: weights not finalized");
    RALD_CHECK(z && ctx && B >= 1 && (uintptr_t)ctx % 16 == 0, "ae: bad arguments");
    RALD_TRY(reserve_decode(B));
    RALD_TRY(ctx_registry.stamp(ctx, ctx_header(B), st));
    ctx = (char*)ctx + BLOB_HEADER_BYTES;
    const int M = cfg.num_latents, L = cfg.latent_dim, BM = B * M;
    const float scale = 1.0f / sqrtf((float)cfg.dim_head);
    RALD_TRY(small_k_linear(z, w_proj, b_proj, x_x, BM, L, d, st));                     // x = proj(z)  (:410)
    static const bool fuse_env = RALD_PROBE_ENV("RALD_FUSE_LN", 1) != 0;
    const bool fuse_ok = fuse_env && d == 512;                              // the fused epilogue owns whole 512-wide rows
    auto resid_ln = [&](const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, int K, const float* ng, const float* nb) -> int {
        if (d == 512 && splitk_for(BM, K))
            return resid_splitk_ln(A, lda, W, ldw, bias, x_x, x_h, ng, nb, 0, 1 << 30, 0.f, 1e-5f, BM, K, splitk_for(BM, K), x_part, st);
        if (fuse_ok && gemm_resid_ln_pays(BM, K)) {
            GemmLnArgs g;
            g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.x = x_x; g.h = x_h;
            g.g = ng; g.b = nb; g.gstride = 0; g.rows_per_group = 1 << 30; g.add_one = 0.f; g.eps = 1e-5f; g.M = BM; g.K = K;
            return gemm_resid_ln(g, st);
        }
        GemmArgs o = gemm_args(A, lda, W, ldw, x_x, d, bias, BM, d, K);
        RALD_TRY(gemm_nt(o, EPI_RESID, st));
        return layernorm_mod(x_x, x_h, BM, d, ng, nb, 0, 1 << 30, 0.f, 1e-5f, st);
    };
    RALD_TRY(layernorm_mod(x_x, x_h, BM, d, layers[0].ng, layers[0].nb, 0, 1 << 30, 0.f, 1e-5f, st));
    for (size_t li = 0; li < layers.size(); ++li) {
        const Layer& l = layers[li];
        // x = self_attn(x) + x   (:413); LN(x) is already in x_h (prologue / previous layer's FF2 epilogue)
        // q | k | v in one projection; V stays row-major and is transposed on the attention kernel's LDS read
        AttnArgs a;
        if (M % 64 == 0) {
            GemmArgs qkv = gemm_args(x_h, d, l.w_qk, d, x_qk, 3 * I, nullptr, BM, 3 * I, d);
            qkv.alpha = scale * 1.4426950408889634f; qkv.alpha_ncols = I;
            RALD_TRY(gemm_nt(qkv, EPI_BF16, st));
            a.Q = x_qk; a.ldq = 3 * I; a.strideQ = (int64_t)M * 3 * I;
            a.K = x_qk + I; a.ldk = 3 * I; a.strideK = (int64_t)M * 3 * I;
            a.Vt = nullptr; a.ldvt = 0; a.strideVt = 0;
            a.V = x_qk + 2 * I; a.ldv = 3 * I; a.strideV = (int64_t)M * 3 * I;
        } else {
            GemmArgs qk = gemm_args(x_h, d, l.w_qk, d, x_qk, 2 * I, nullptr, BM, 2 * I, d);
            qk.alpha = scale * 1.4426950408889634f; qk.alpha_ncols = I;
            RALD_TRY(gemm_nt(qk, EPI_BF16, st));
            GemmArgs vt = gemm_args(l.w_v, d, x_h, d, x_vt, M, nullptr, I, M, d);
            vt.batch = B; vt.strideB = (int64_t)M * d; vt.strideC = (int64_t)I * M;
            RALD_TRY(gemm_nt(vt, EPI_BF16, st));
            a.Q = x_qk; a.ldq = 2 * I; a.strideQ = (int64_t)M * 2 * I;
            a.K = x_qk + I; a.ldk = 2 * I; a.strideK = (int64_t)M * 2 * I;
            a.Vt = x_vt; a.ldvt = M; a.strideVt = (int64_t)I * M;
        }
        if (d == 512 && M % 64 == 0 && small_m_fused(BM, M, cfg.heads, I, 64)) {
            // small batches: attention + per-head slice of to_out in one kernel, partials summed with the residual + the FF's PreNorm
            RALD_TRY(attn_self_proj(x_qk, 3 * I, l.w_o, x_part, M, cfg.heads, B, st, true));
            RALD_TRY(reduce_resid_ln(x_part, cfg.heads, (int64_t)BM * d, l.b_o, x_x, x_h, BM, l.ff.ng, l.ff.nb, 0, 1 << 30, 0.f, 1e-5f, st, true));
        } else {
        a.O = x_o; a.ldo = I; a.strideO = (int64_t)M * I;
        a.nq = M; a.nk = M; a.k_rows = M; a.heads = cfg.heads; a.batch = B; a.scale = scale; a.q_prescaled = 1;
        RALD_TRY(attention_d64(a, st));
        RALD_TRY(resid_ln(x_o, I, l.w_o, I, l.b_o, I, l.ff.ng, l.ff.nb));             // + the FF's PreNorm
        }
        // x = self_ff(x) + x                                                            (:414)
        GemmArgs f1 = gemm_args(x_h, d, l.ff.w1, d, x_g, 4 * d, l.ff.b1, BM, 8 * d, d);
        RALD_TRY(gemm_nt(f1, EPI_GEGLU, st));
        // FF2 + the next consumer's LayerNorm: next layer's attention PreNorm, or the decoder's norm_context
        const float* ng = (li + 1 < layers.size()) ? layers[li + 1].ng : dec.cg;
        const float* nb = (li + 1 < layers.size()) ? layers[li + 1].nb : dec.cb;
        RALD_TRY(resid_ln(x_g, 4 * d, l.ff.w2, 4 * d, l.ff.b2, 4 * d, ng, nb));
    }
    // decoder context (ae_decode.hip): LN_ctx(x) . t2aug -> per-latent score coefficients, h0, hb, u; one fp16 image per sample.
    // (x_h holds LN_ctx(x) in bf16 from the last epilogue; the context is projected from the fp32 residual stream instead.)
    RALD_TRY(ae_ctx_build(x_x, dec.cg, dec.cb, t2aug, x_y, ctx, B, M, d, st));
    return 0;
}

int Ae::decode_queries(const void* ctx, const float* q, int B, int64_t Q, float* out, hipStream_t st, int nw) {
    RALD_CHECK(finalized, "ae: weights not finalized");
    RALD_CHECK(ctx && q && out && B >= 1 && Q >= 1, "ae: bad arguments");
    RALD_CHECK((uintptr_t)ctx % 16 == 0, "ae: decoder context must be 16-byte aligned");
    RALD_TRY(ctx_registry.check(ctx, ctx_header(B), st, "decoder context"));
    ctx = (const char*)ctx + BLOB_HEADER_BYTES;
    return ae_decode_stream(ctx, l_img, q, out, basis, basis_diag, B, Q, cfg.num_latents, c0, st, nw);
}

}  // namespace rald
