// Denoiser handle: weight packing, noise-level table, condition cache, one NFE, the Heun sampler.
// Reference: model/models_radar_generation.py (LatentArrayTransformer :171-233, EDMPrecond
// :314-449, edm_sampler :235-275).  See include/rald_hip.h for the C-ABI contract.
#include "dit.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rald {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

// ---------------------------------------------------------------------------------------------
// DeviceArena: owns every hipMalloc of a handle
// ---------------------------------------------------------------------------------------------
void* DeviceArena::alloc(size_t bytes, bool zero) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (zero) {
        // the memset runs on the null stream, which callers' non-blocking streams (torch side streams) do not wait for:
        // finish it before anyone can launch work that writes this buffer
        (void)hipMemset(p, 0, bytes);
        (void)hipStreamSynchronize(nullptr);
    }
    ptrs.push_back(p);
    return p;
}
void DeviceArena::release(void* p) {
    for (size_t i = 0; i < ptrs.size(); ++i)
        if (ptrs[i] == p) {
            (void)hipFree(p);
            ptrs.erase(ptrs.begin() + i);
            return;
        }
}
DeviceArena::~DeviceArena() {
    for (void* p : ptrs) (void)hipFree(p);
}

// ---------------------------------------------------------------------------------------------
// Weight staging: fp32 (host or device) -> packed device tensors
// ---------------------------------------------------------------------------------------------
__global__ void pack_rows_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                     int64_t ld_dst, const int* __restrict__ rowmap) {
    const int r = blockIdx.x;
    const int dr = rowmap ? rowmap[r] : r;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(int64_t)dr * ld_dst + c] = src[(int64_t)r * cols + c];
}

int Stager::ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    cap = 0;
    RALD_HIP(hipMalloc(&buf, bytes));
    cap = bytes;
    return 0;
}
Stager::~Stager() {
    if (buf) (void)hipFree(buf);
}
int Stager::fetch(const float* data, int64_t nelem) {
    RALD_TRY(ensure((size_t)nelem * 4));
    RALD_HIP(hipMemcpy(buf, data, (size_t)nelem * 4, hipMemcpyDefault));
    return 0;
}
int Stager::to_bf16(const float* data, bf16* dst, int rows, int cols, int64_t ld_dst, const int* rowmap) {
    RALD_TRY(fetch(data, (int64_t)rows * cols));
    RALD_TRY(pack_rows_bf16((const float*)buf, dst, rows, cols, ld_dst, rowmap, nullptr));
    RALD_HIP(hipDeviceSynchronize());
    return 0;
}
int Stager::to_f32(const float* data, float* dst, int rows, int cols, int64_t ld_dst, const int* rowmap) {
    RALD_TRY(fetch(data, (int64_t)rows * cols));
    hipLaunchKernelGGL(pack_rows_f32_kernel, dim3(rows), dim3(256), 0, nullptr, (const float*)buf, dst, rows, cols, ld_dst, rowmap);
    RALD_HIP(hipGetLastError());
    RALD_HIP(hipDeviceSynchronize());
    return 0;
}

// GEGLU row packing (gemm.hip): x column c -> packed row 32*(c/16) + c%16, gate column c -> +16
std::vector<int> geglu_rowmap(int inner) {
    std::vector<int> m(2 * inner);
    for (int c = 0; c < inner; ++c) {
        m[c] = 32 * (c / 16) + c % 16;
        m[inner + c] = 32 * (c / 16) + 16 + c % 16;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// Dit
// ---------------------------------------------------------------------------------------------
int Dit::create() {
    const auto& c = cfg;
    D = c.n_heads * c.d_head;
    RALD_CHECK(c.d_head == 64, "dit: only d_head = 64 is implemented");
    RALD_CHECK(c.qkv_dtype >= 0 && c.qkv_dtype <= 3, "dit: qkv_dtype must be 0 (bf16), 1 (MXFP8 q/k/v), 2 (+ GEGLU projection) or 3 (+ feed-forward output projection)");
    RALD_CHECK(D == 512, "dit: inner dim (n_heads*d_head) must be 512");
    RALD_CHECK(c.n_latents > 0 && c.n_latents % 64 == 0, "dit: n_latents must be a positive multiple of 64");
    RALD_CHECK(c.channels >= 1 && c.channels <= 64, "dit: channels must be in [1,64]");
    RALD_CHECK(c.depth >= 1 && c.depth <= 256, "dit: bad depth");
    RALD_CHECK(c.context_dim % 64 == 0 && c.context_dim > 0, "dit: context_dim must be a multiple of 64");
    RALD_CHECK(c.n_cond_tokens > 0 && c.n_cond_tokens % 64 == 0, "dit: n_cond_tokens must be a multiple of 64");
    RALD_CHECK(c.t_channels % 4 == 0, "dit: t_channels must be a multiple of 4");
    const int L = c.depth;
    layers.resize(L);
    auto B16 = [&](size_t n) { return (bf16*)arena.alloc(n * 2, true); };
    auto F32 = [&](size_t n) { return (float*)arena.alloc(n * 4, true); };
    for (auto& l : layers) {
        l.w_qk = B16((size_t)3 * D * D);                 // to_q | to_k | to_v stacked: one [1536, 512] projection
        l.w_v = l.w_qk + (size_t)2 * D * D;              // (alias: rows 1024..1535)
        l.w_o = B16((size_t)D * D);
        l.b_o = F32(D);
        l.w_q2 = B16((size_t)D * D);
        l.w_q2t = B16((size_t)D * D);
        l.w_o2 = B16((size_t)D * D);
        l.b_o2 = F32(D);
        l.w_ff1 = B16((size_t)8 * D * D);
        l.b_ff1 = F32((size_t)8 * D);
        l.w_ff2 = B16((size_t)D * 4 * D);
        l.b_ff2 = F32(D);
        if (c.qkv_dtype >= 1) {
            auto U8 = [&](size_t n) { return (unsigned char*)arena.alloc(n, true); };
            l.q8_qk = U8((size_t)3 * D * D); l.s8_qk = U8((size_t)3 * D * D / 32);   // q | k | v stacked like the bf16 weights
            l.q8_v = l.q8_qk + (size_t)2 * D * D; l.s8_v = l.s8_qk + (size_t)2 * D * D / 32;
            l.q8_q2 = U8((size_t)D * D);     l.s8_q2 = U8((size_t)D * D / 32);
            if (c.qkv_dtype >= 2) { l.q8_ff1 = U8((size_t)8 * D * D); l.s8_ff1 = U8((size_t)8 * D * D / 32); RALD_CHECK(l.q8_ff1 && l.s8_ff1, "dit: device allocation failed"); }
            if (c.qkv_dtype == 3) { l.q8_ff2 = U8((size_t)4 * D * D); l.s8_ff2 = U8((size_t)4 * D * D / 32); RALD_CHECK(l.q8_ff2 && l.s8_ff2, "dit: device allocation failed"); }
            RALD_CHECK(l.q8_qk && l.s8_qk && l.q8_v && l.s8_v && l.q8_q2 && l.s8_q2,
                       "dit: device allocation failed");
        }
    }
    w_k2_all = B16((size_t)L * D * c.context_dim);
    w_v2_all = B16((size_t)L * D * c.context_dim);
    w_mod = F32((size_t)L * 3 * 2 * D * D);
    b_mod = F32((size_t)L * 3 * 2 * D);
    w_t0 = F32((size_t)D * c.t_channels);
    b_t0 = F32(D);
    w_t1 = F32((size_t)D * D);
    b_t1 = F32(D);
    w_in = F32((size_t)D * c.channels);
    w_out = F32((size_t)c.channels * D);
    w_out_hl = B16((size_t)2 * c.channels * D);
    norm_g = F32(D);
    norm_b = F32(D);
    coef_raw = F32(4);
    std::vector<int> rm = geglu_rowmap(4 * D);
    d_geglu_map = (int*)arena.alloc(rm.size() * 4, false);
    RALD_CHECK(w_mod && d_geglu_map && coef_raw, "dit: device allocation failed");
    RALD_HIP(hipMemcpy(d_geglu_map, rm.data(), rm.size() * 4, hipMemcpyHostToDevice));
    const float raw[4] = {1.f, 0.f, 1.f, 0.f};   // c_in = 1, c_skip = 0, c_out = 1: plain F(x)
    RALD_HIP(hipMemcpy(coef_raw, raw, sizeof(raw), hipMemcpyHostToDevice));
    if (c.with_radar_enc) RALD_TRY(radar.create(c.enc_hidden_ch, c.enc_radar_ch, c.radar_r, c.radar_a, c.radar_e, D, &arena));
    // expected keys
    expected.clear();
    char buf[160];
    expected.insert("model.proj_in.weight");
    for (int i = 0; i < L; ++i) {
        static const char* names[] = {"attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight", "attn1.to_out.0.weight",
                                      "attn1.to_out.0.bias", "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff.net.2.weight",
                                      "ff.net.2.bias", "attn2.to_q.weight", "attn2.to_k.weight", "attn2.to_v.weight",
                                      "attn2.to_out.0.weight", "attn2.to_out.0.bias", "norm1.linear.weight", "norm1.linear.bias",
                                      "norm2.linear.weight", "norm2.linear.bias", "norm3.linear.weight", "norm3.linear.bias"};
        for (const char* n : names) {
            snprintf(buf, sizeof(buf), "model.transformer_blocks.%d.%s", i, n);
            expected.insert(buf);
        }
    }
    for (const char* n : {"model.norm.weight", "model.norm.bias", "model.proj_out.weight", "model.map_layer0.weight",
                          "model.map_layer0.bias", "model.map_layer1.weight", "model.map_layer1.bias"})
        expected.insert(n);
    if (c.with_radar_enc) radar.expected_keys("radar_enc.", expected);
    if (c.with_radar_enc)
        for (const char* n : {"radar_r_emb.weight", "radar_a_emb.weight", "radar_e_emb.weight", "radar_token_project.weight",
                              "radar_token_project.bias"})
            expected.insert(n);
    return 0;
}

int Dit::load_weight(const std::string& name, const float* data, int64_t nelem) {
    RALD_CHECK(expected.count(name), "dit: unexpected key '" + name + "'");
    const auto& c = cfg;
    auto need = [&](int64_t n) -> int {
        RALD_CHECK(nelem == n, "dit: size mismatch for '" + name + "': got " + std::to_string(nelem) + ", expected " + std::to_string(n));
        return 0;
    };
    int rc = 0;
    int li = -1;
    char tail[128] = {0};
    if (name.rfind("radar_enc.", 0) == 0) {
        rc = radar.load_weight(name.substr(10), data, nelem, stager);
    } else if (name == "radar_r_emb.weight" || name == "radar_a_emb.weight" || name == "radar_e_emb.weight" ||
               name == "radar_token_project.weight" || name == "radar_token_project.bias") {
        rc = radar.load_token_weight(name, data, nelem, stager);
    } else if (sscanf(name.c_str(), "model.transformer_blocks.%d.%127s", &li, tail) == 2) {
        RALD_CHECK(li >= 0 && li < c.depth, "dit: block index out of range in '" + name + "'");
        Layer& l = layers[li];
        const std::string t(tail);
        const int Cd = c.context_dim;
        if (t == "attn1.to_q.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_bf16(data, l.w_qk, D, D, D, nullptr); }
        else if (t == "attn1.to_k.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_bf16(data, l.w_qk + (size_t)D * D, D, D, D, nullptr); }
        else if (t == "attn1.to_v.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_bf16(data, l.w_v, D, D, D, nullptr); }
        else if (t == "attn1.to_out.0.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_bf16(data, l.w_o, D, D, D, nullptr); }
        else if (t == "attn1.to_out.0.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, l.b_o, 1, D, D, nullptr); }
        else if (t == "ff.net.0.proj.weight") { RALD_TRY(need((int64_t)8 * D * D)); rc = stager.to_bf16(data, l.w_ff1, 8 * D, D, D, d_geglu_map); }
        else if (t == "ff.net.0.proj.bias") { RALD_TRY(need((int64_t)8 * D)); rc = stager.to_f32(data, l.b_ff1, 8 * D, 1, 1, d_geglu_map); }
        else if (t == "ff.net.2.weight") { RALD_TRY(need((int64_t)D * 4 * D)); rc = stager.to_bf16(data, l.w_ff2, D, 4 * D, 4 * D, nullptr); }
        else if (t == "ff.net.2.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, l.b_ff2, 1, D, D, nullptr); }
        else if (t == "attn2.to_q.weight") {
            RALD_TRY(need((int64_t)D * D));
            RALD_TRY(stager.to_bf16(data, l.w_q2, D, D, D, nullptr));
            std::vector<float> hw((size_t)D * D), ht((size_t)D * D);                  // transposed copy for cond_fold (host: once per load)
            RALD_HIP(hipMemcpy(hw.data(), data, hw.size() * 4, hipMemcpyDefault));
            for (int r = 0; r < D; ++r)
                for (int c2 = 0; c2 < D; ++c2) ht[(size_t)c2 * D + r] = hw[(size_t)r * D + c2];
            rc = stager.to_bf16(ht.data(), l.w_q2t, D, D, D, nullptr);
        }
        else if (t == "attn2.to_k.weight") { RALD_TRY(need((int64_t)D * Cd)); rc = stager.to_bf16(data, w_k2_all + (size_t)li * D * Cd, D, Cd, Cd, nullptr); }
        else if (t == "attn2.to_v.weight") { RALD_TRY(need((int64_t)D * Cd)); rc = stager.to_bf16(data, w_v2_all + (size_t)li * D * Cd, D, Cd, Cd, nullptr); }
        else if (t == "attn2.to_out.0.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_bf16(data, l.w_o2, D, D, D, nullptr); }
        else if (t == "attn2.to_out.0.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, l.b_o2, 1, D, D, nullptr); }
        else {
            int j = -1;
            char kind[16] = {0};
            RALD_CHECK(sscanf(tail, "norm%d.linear.%15s", &j, kind) == 2 && j >= 1 && j <= 3, "dit: unknown key '" + name + "'");
            const size_t slot = (size_t)li * 3 + (j - 1);
            if (!strcmp(kind, "weight")) { RALD_TRY(need((int64_t)2 * D * D)); rc = stager.to_f32(data, w_mod + slot * 2 * D * D, 2 * D, D, D, nullptr); }
            else { RALD_TRY(need((int64_t)2 * D)); rc = stager.to_f32(data, b_mod + slot * 2 * D, 1, 2 * D, 2 * D, nullptr); }
        }
    } else if (name == "model.proj_in.weight") { RALD_TRY(need((int64_t)D * c.channels)); rc = stager.to_f32(data, w_in, D, c.channels, c.channels, nullptr); }
    else if (name == "model.proj_out.weight") { RALD_TRY(need((int64_t)c.channels * D)); rc = stager.to_f32(data, w_out, c.channels, D, D, nullptr); }
    else if (name == "model.norm.weight") { RALD_TRY(need(D)); rc = stager.to_f32(data, norm_g, 1, D, D, nullptr); }
    else if (name == "model.norm.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, norm_b, 1, D, D, nullptr); }
    else if (name == "model.map_layer0.weight") { RALD_TRY(need((int64_t)D * c.t_channels)); rc = stager.to_f32(data, w_t0, D, c.t_channels, c.t_channels, nullptr); }
    else if (name == "model.map_layer0.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, b_t0, 1, D, D, nullptr); }
    else if (name == "model.map_layer1.weight") { RALD_TRY(need((int64_t)D * D)); rc = stager.to_f32(data, w_t1, D, D, D, nullptr); }
    else if (name == "model.map_layer1.bias") { RALD_TRY(need(D)); rc = stager.to_f32(data, b_t1, 1, D, D, nullptr); }
    else RALD_CHECK(false, "dit: unknown key '" + name + "'");
    if (rc) return rc;
    loaded.insert(name);
    tables[0].key.clear();   // any cached table depends on the weights
    tables[1].key.clear();
    return 0;
}

int Dit::finalize() {
    for (const auto& k : expected) RALD_CHECK(loaded.count(k), "dit: missing key '" + k + "' (strict load)");
    if (cfg.qkv_dtype >= 1) {                  // MXFP8 copies of the attention projections, from the bf16 weights
        for (auto& l : layers) {
            RALD_TRY(quantize_mx8(l.w_qk, 1, D, l.q8_qk, D, l.s8_qk, 3 * D, D, nullptr));      // q | k | v (q8_v / s8_v alias its last third)
            RALD_TRY(quantize_mx8(l.w_q2, 1, D, l.q8_q2, D, l.s8_q2, D, D, nullptr));
            if (cfg.qkv_dtype >= 2) RALD_TRY(quantize_mx8(l.w_ff1, 1, D, l.q8_ff1, D, l.s8_ff1, 8 * D, D, nullptr));
            if (cfg.qkv_dtype == 3) RALD_TRY(quantize_mx8(l.w_ff2, 1, 4 * D, l.q8_ff2, 4 * D, l.s8_ff2, D, 4 * D, nullptr));
        }
        RALD_HIP(hipDeviceSynchronize());
    }
    RALD_TRY(split_hi_lo(w_out, w_out_hl, cfg.channels * D, nullptr));      // proj_out for the MFMA form of the output layer (norm.hip)
    RALD_HIP(hipDeviceSynchronize());
    finalized = true;
    return 0;
}

void Dit::free_work(Work& w) {
    for (void* p : {(void*)w.x, (void*)w.part, (void*)w.h, (void*)w.qk, (void*)w.vt, (void*)w.o, (void*)w.q2, (void*)w.g, (void*)w.h8, (void*)w.hs,
                    (void*)w.g8, (void*)w.gs})
        if (p) arena.release(p);
    w = Work();
}
int Dit::alloc_work(Work& w, size_t M) {
    const size_t B = M / cfg.n_latents;
    w.x = (float*)arena.alloc(M * D * 4, true);
    w.h = (bf16*)arena.alloc(M * D * 2, true);
    w.qk = (bf16*)arena.alloc(M * 3 * D * 2, true);       // q | k | v rows of 3*D
    w.vt = (bf16*)arena.alloc(B * D * cfg.n_latents * 2, true);
    w.o = (bf16*)arena.alloc(M * D * 2, true);
    w.q2 = (bf16*)arena.alloc(M * D * 2, true);
    w.g = (bf16*)arena.alloc(M * 4 * D * 2, true);
    // split-K partials of the small-batch FF2 (4 slabs of up to splitk_max_rows() rows) / per-head partials of the fused
    // attention sub-blocks (n_heads slabs of up to 2048 rows, kernels.h small_m_fused)
    {
        const size_t r4 = M < (size_t)splitk_max_rows() ? M : (size_t)splitk_max_rows();
        const size_t r8 = M < 2048 ? M : 2048;
        const size_t slabs_rows = 4 * r4 > (size_t)cfg.n_heads * r8 ? 4 * r4 : (size_t)cfg.n_heads * r8;
        w.part = (float*)arena.alloc(slabs_rows * 512 * 4, true);
    }
    RALD_CHECK(w.x && w.h && w.qk && w.vt && w.o && w.q2 && w.g && w.part, "dit: workspace allocation failed");
    if (cfg.qkv_dtype >= 1) {
        w.h8 = (unsigned char*)arena.alloc(M * D, true);
        w.hs = (unsigned char*)arena.alloc(M * D / 32, true);
        RALD_CHECK(w.h8 && w.hs, "dit: workspace allocation failed");
        if (cfg.qkv_dtype == 3) {
            w.g8 = (unsigned char*)arena.alloc(M * 4 * D, true);
            w.gs = (unsigned char*)arena.alloc(M * 4 * D / 32, true);
            RALD_CHECK(w.g8 && w.gs, "dit: workspace allocation failed");
        }
    }
    w.rows = (int)M;
    return 0;
}

int Dit::reserve(int B) {
    if (B <= ws_batch) return 0;
    RALD_HIP(hipDeviceSynchronize());
    ++ws_generation;
    free_work(wk[0]);
    free_work(wk[1]);
    for (void* p : {(void*)ws_tok, (void*)ws_xcur, (void*)ws_xeul, (void*)ws_den, (void*)ws_dcur})
        if (p) arena.release(p);
    const size_t M = (size_t)B * cfg.n_latents;
    const size_t nl = (size_t)B * cfg.n_latents * cfg.channels;
    RALD_TRY(alloc_work(wk[0], M));
    int b0 = 0;
    if (split_sizes(B, b0)) {                                   // the second half of a two-stream NFE works on buffers of its own
        RALD_TRY(alloc_work(wk[1], (size_t)(B - b0 > b0 ? B - b0 : b0) * cfg.n_latents));
        if (!side) {
            RALD_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            RALD_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            RALD_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
    }
    ws_tok = (bf16*)arena.alloc((size_t)B * cfg.n_cond_tokens * cfg.context_dim * 2, true);
    ws_xcur = (float*)arena.alloc(nl * 4, true);
    ws_xeul = (float*)arena.alloc(nl * 4, true);
    ws_den = (float*)arena.alloc(nl * 4, true);
    ws_dcur = (float*)arena.alloc(nl * 4, true);
    RALD_CHECK(ws_tok && ws_xcur && ws_xeul && ws_den && ws_dcur, "dit: workspace allocation failed");
    ws_batch = B;
    return 0;
}

// coef[s] = {c_in, c_skip, c_out, c_noise}  (EDMPrecond.forward :422-425, fp32 like the reference)
__global__ void edm_coef_kernel(const float* __restrict__ sigma, float* __restrict__ coef, float* __restrict__ c_noise,
                                int n, float sd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sigma[i];
    const float den = s * s + sd * sd;
    coef[4 * i + 0] = 1.0f / sqrtf(den);
    coef[4 * i + 1] = sd * sd / den;
    coef[4 * i + 2] = s * sd / sqrtf(den);
    const float cn = logf(s) / 4.0f;
    coef[4 * i + 3] = cn;
    c_noise[i] = cn;
}

int Dit::build_table(SigmaTable& t, const float* sig, int n, hipStream_t st) {
    RALD_CHECK(finalized, "dit: weights not finalized");
    RALD_CHECK(n >= 1 && n <= 4096, "dit: sigma table must have 1..4096 rows");
    std::string key((const char*)sig, (size_t)n * 4);
    if (key == t.key) return 0;
    if (n > t.cap) {
        RALD_HIP(hipDeviceSynchronize());
        ++ws_generation;
        for (void* p : {(void*)t.sigma, (void*)t.coef, (void*)t.cnoise, (void*)t.pe, (void*)t.temb0, (void*)t.temb, (void*)t.mod})
            if (p) arena.release(p);
        const int cap = n < 64 ? 64 : n;
        t.sigma = (float*)arena.alloc((size_t)cap * 4, true);
        t.coef = (float*)arena.alloc((size_t)cap * 16, true);
        t.cnoise = (float*)arena.alloc((size_t)cap * 4, true);
        t.pe = (float*)arena.alloc((size_t)cap * cfg.t_channels * 4, true);
        t.temb0 = (float*)arena.alloc((size_t)cap * D * 4, true);
        t.temb = (float*)arena.alloc((size_t)cap * D * 4, true);
        t.mod = (float*)arena.alloc((size_t)cap * mod_row() * 4, true);
        RALD_CHECK(t.sigma && t.coef && t.cnoise && t.pe && t.temb0 && t.temb && t.mod, "dit: sigma table allocation failed");
        t.cap = cap;
    }
    t.key.clear();
    t.host.assign(sig, sig + n);   // the async copy below must not read a caller buffer that may go away
    RALD_HIP(hipMemcpyAsync(t.sigma, t.host.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(edm_coef_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, t.sigma, t.coef, t.cnoise, n, cfg.sigma_data);
    RALD_HIP(hipGetLastError());
    RALD_TRY(positional_embedding(t.cnoise, t.pe, n, cfg.t_channels, st));
    RALD_TRY(skinny_linear(t.pe, w_t0, b_t0, t.temb0, n, D, cfg.t_channels, ACT_SILU, st));
    RALD_TRY(skinny_linear(t.temb0, w_t1, b_t1, t.temb, n, D, D, ACT_SILU, st));
    RALD_TRY(skinny_linear(t.temb, w_mod, b_mod, t.mod, n, (int)mod_row(), D, ACT_NONE, st));
    t.key = key;
    t.n = n;
    return 0;
}

int Dit::set_sigmas(const float* sig, int n, hipStream_t st) { return build_table(tables[0], sig, n, st); }

bool Dit::cond_fold(int B) const {
    static const bool on = RALD_PROBE_ENV("RALD_COND_FOLD", 1) != 0;
    return on && cfg.qkv_dtype == 0 && cfg.n_cond_tokens == 64 && cfg.d_head == 64 && cfg.n_heads * 64 == D && cfg.n_latents % 128 == 0 &&
           !small_m_fused(B * cfg.n_latents, cfg.n_latents, cfg.n_heads, D, cfg.n_cond_tokens);
}

int64_t Dit::cond_cache_bytes(int B) const {
    // 64-byte header, then Kc [B*T][L*D] bf16  +  Vtc [B][L*D][T] bf16;  with cond_fold also Vc [B*T][L*D], Gt [B][L][D][D], Ut [B][L][D][D]
    const int64_t kv = (int64_t)B * cfg.n_cond_tokens * cfg.depth * D * 2;
    return COND_HEADER_BYTES + (cond_fold(B) ? 3 * kv + (int64_t)2 * B * cfg.depth * D * D * 2 : 2 * kv);
}

uint32_t Dit::cfg_hash() const {
    uint32_t h = 2166136261u;
    for (int v : {cfg.n_latents, cfg.channels, cfg.depth, cfg.n_heads, cfg.d_head, cfg.context_dim, cfg.n_cond_tokens, cfg.qkv_dtype}) {
        h ^= (uint32_t)v; h *= 16777619u;
    }
    return h;
}

// writes the header with a kernel (a host struct handed to an async copy would have to outlive the call)
__global__ void blob_header_kernel(uint32_t* dst, uint32_t magic, int batch, int flag, uint32_t hash, int64_t bytes) {
    if (threadIdx.x == 0) {
        dst[0] = magic; dst[1] = (uint32_t)batch; dst[2] = (uint32_t)flag; dst[3] = hash;
        dst[4] = (uint32_t)(bytes & 0xffffffffu); dst[5] = (uint32_t)((uint64_t)bytes >> 32);
        for (int i = 6; i < 16; ++i) dst[i] = 0;
    }
}
int BlobRegistry::stamp(void* blob, const BlobHeader& hd, hipStream_t st) {
    RALD_CHECK(blob && (uintptr_t)blob % 16 == 0, "blob must be a 16-byte aligned device pointer");
    hipLaunchKernelGGL(blob_header_kernel, dim3(1), dim3(64), 0, st, (uint32_t*)blob, hd.magic, hd.batch, hd.flag, hd.cfg_hash, hd.bytes);
    RALD_HIP(hipGetLastError());
    if (known.size() > 4096) known.clear();                    // bounded: entries of freed blobs are only ever replaced
    known[blob] = hd;
    return 0;
}
int BlobRegistry::check(const void* blob, const BlobHeader& want, hipStream_t st, const char* what) {
    const std::string w(what);
    RALD_CHECK(blob && (uintptr_t)blob % 16 == 0, w + " must be a 16-byte aligned device pointer");
    auto same = [&](const BlobHeader& h) {
        return h.magic == want.magic && h.batch == want.batch && h.flag == want.flag && h.cfg_hash == want.cfg_hash && h.bytes == want.bytes;
    };
    auto it = known.find(blob);
    if (it != known.end() && same(it->second)) return 0;
    // unknown pointer (a copy, a recycled address) or a registered blob of another batch: the header in device memory decides
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    RALD_CHECK(cs == hipStreamCaptureStatusNone, w + " is not known to this handle for this batch size; use it once outside graph capture first");
    BlobHeader hd;
    RALD_HIP(hipStreamSynchronize(st));            // the header may still be in flight on this stream (a copy enqueued by the caller)
    RALD_HIP(hipMemcpy(&hd, blob, sizeof(hd), hipMemcpyDeviceToHost));
    RALD_CHECK(hd.magic == want.magic, w + ": no header found (not produced by this library's encode / decode_latents call)");
    RALD_CHECK(hd.cfg_hash == want.cfg_hash, w + " was built by a handle with another configuration");
    RALD_CHECK(hd.batch == want.batch, w + " was built for batch " + std::to_string(hd.batch) + ", used with batch " + std::to_string(want.batch));
    RALD_CHECK(hd.flag == want.flag && hd.bytes == want.bytes, w + ": layout does not match this batch size");
    known[blob] = hd;
    return 0;
}

BlobHeader Dit::cond_header(int B) const {
    BlobHeader hd{};
    hd.magic = COND_MAGIC; hd.batch = B; hd.flag = cond_fold(B) ? 1 : 0; hd.cfg_hash = cfg_hash(); hd.bytes = cond_cache_bytes(B);
    return hd;
}

int Dit::encode_cond_tokens(const float* tokens, int B, void* cache, hipStream_t st) {
    RALD_CHECK(finalized, "dit: weights not finalized");
    RALD_CHECK(B >= 1 && tokens && cache, "dit: bad arguments");
    RALD_CHECK((uintptr_t)cache % 16 == 0, "dit: cond cache must be 16-byte aligned");
    RALD_TRY(reserve(B));
    const int T = cfg.n_cond_tokens, Cd = cfg.context_dim, L = cfg.depth;
    RALD_TRY(cond_registry.stamp(cache, cond_header(B), st));
    RALD_TRY(cast_f32_bf16(tokens, ws_tok, (int64_t)B * T * Cd, st));
    bf16* Kc = (bf16*)((char*)cache + COND_HEADER_BYTES);
    bf16* Vtc = Kc + (size_t)B * T * L * D;
    // K for all blocks at once: [B*T, Cd] x [L*D, Cd]^T
    GemmArgs g = gemm_args(ws_tok, Cd, w_k2_all, Cd, Kc, (int64_t)L * D, nullptr, B * T, L * D, Cd);
    RALD_TRY(gemm_nt(g, EPI_BF16, st));
    // V^T for all blocks: per sample [L*D, Cd] x [T, Cd]^T -> [L*D][T]
    GemmArgs v = gemm_args(w_v2_all, Cd, ws_tok, Cd, Vtc, T, nullptr, L * D, T, Cd);
    v.batch = B;
    v.strideB = (int64_t)T * Cd;
    v.strideC = (int64_t)L * D * T;
    RALD_TRY(gemm_nt(v, EPI_BF16, st));
    if (cond_fold(B)) {
        bf16* Vc = Vtc + (size_t)B * T * L * D;
        bf16* Gt = Vc + (size_t)B * T * L * D;
        bf16* Ut = Gt + (size_t)B * L * D * D;
        GemmArgs vr = gemm_args(ws_tok, Cd, w_v2_all, Cd, Vc, (int64_t)L * D, nullptr, B * T, L * D, Cd);      // V row-major, like K
        RALD_TRY(gemm_nt(vr, EPI_BF16, st));
        const float qscale = (1.0f / sqrtf((float)cfg.d_head)) * 1.4426950408889634f;
        const int H = cfg.n_heads;
        for (int li = 0; li < L; ++li) {
            // Gt[b][li][h*64 + key][k] = qscale . sum_d Kc[b][key][li*D + h*64 + d] . Wq^T[k][h*64 + d]
            GemmArgs gg = gemm_args(Kc + (size_t)li * D, (int64_t)L * D, layers[li].w_q2t, D, Gt + (size_t)li * D * D, D, nullptr, T, D, 64);
            gg.batch = B; gg.strideA = (int64_t)T * L * D; gg.strideB = 0; gg.strideC = (int64_t)L * D * D;
            gg.batch2 = H; gg.strideA2 = 64; gg.strideB2 = 64; gg.strideC2 = (int64_t)T * D;
            gg.alpha = qscale;
            RALD_TRY(gemm_nt(gg, EPI_BF16, st));
            // Ut[b][li][n][h*64 + key] = sum_d Wo[n][h*64 + d] . Vc[b][key][li*D + h*64 + d]
            GemmArgs gu = gemm_args(layers[li].w_o2, D, Vc + (size_t)li * D, (int64_t)L * D, Ut + (size_t)li * D * D, D, nullptr, D, T, 64);
            gu.batch = B; gu.strideA = 0; gu.strideB = (int64_t)T * L * D; gu.strideC = (int64_t)L * D * D;
            gu.batch2 = H; gu.strideA2 = 64; gu.strideB2 = 64; gu.strideC2 = T;
            RALD_TRY(gemm_nt(gu, EPI_BF16, st));
        }
    }
    return 0;
}

int Dit::encode_cond(const float* cube, int B, float* out_tokens, void* cache, hipStream_t st) {
    RALD_CHECK(cfg.with_radar_enc, "dit: handle was created without the radar encoder");
    RALD_CHECK(finalized, "dit: weights not finalized");
    float* tok = nullptr;
    RALD_TRY(radar.tokens(cube, B, &tok, st));
    if (out_tokens) RALD_HIP(hipMemcpyAsync(out_tokens, tok, (size_t)B * cfg.n_cond_tokens * D * 4, hipMemcpyDeviceToDevice, st));
    return encode_cond_tokens(tok, B, cache, st);
}

int Dit::denoise(const float* x, int B, int sigma_row, int per_sample, const void* cache, float* out, int raw_F, hipStream_t st, int slot) {
    RALD_CHECK(finalized, "dit: weights not finalized");
    RALD_CHECK(B >= 1 && x && out && cache, "dit: bad arguments");
    const SigmaTable& tb = tables[slot];
    RALD_CHECK(!tb.key.empty(), "dit: rald_dit_set_sigmas has not been called");
    RALD_CHECK(sigma_row >= 0 && sigma_row + (per_sample ? B : 1) <= tb.n, "dit: sigma_row out of range of the sigma table");
    RALD_TRY(cond_registry.check(cache, cond_header(B), st, "condition cache"));
    RALD_TRY(reserve(B));
    int b0 = 0;
    if (split_sizes(B, b0) && wk[1].rows >= (B - b0) * cfg.n_latents && wk[0].rows >= b0 * cfg.n_latents && side) {
        // two half-batches on two streams: fork from the caller's stream, join back into it
        // The second half starts when the first one reaches its first feed-forward (about half a block later): launched together the
        // two halves would run the same kernel at the same time and compete for the same unit; half a block apart the matrix-bound
        // feed-forward GEMMs of one half meet the attention / residual + LayerNorm kernels of the other (measured: no gain without
        // the offset).
        fork_pending = true;
        RALD_TRY(denoise_range(x, B, 0, b0, sigma_row, per_sample, cache, out, raw_F, st, slot, wk[0], true));
        if (fork_pending) { RALD_HIP(hipEventRecord(ev_fork, st)); fork_pending = false; }      // (depth-0 models never reach the mark)
        RALD_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        RALD_TRY(denoise_range(x, B, b0, B - b0, sigma_row, per_sample, cache, out, raw_F, side, slot, wk[1], false));
        RALD_HIP(hipEventRecord(ev_join, side));
        RALD_HIP(hipStreamWaitEvent(st, ev_join, 0));
        return 0;
    }
    return denoise_range(x, B, 0, B, sigma_row, per_sample, cache, out, raw_F, st, slot, wk[0], true);
}

int Dit::denoise_range(const float* x, int Bfull, int b0, int B, int sigma_row, int per_sample, const void* cache, float* out, int raw_F,
                       hipStream_t st, int slot, Work& w, bool timed_ok) {
    const SigmaTable& tb = tables[slot];
    const int NL = cfg.n_latents, T = cfg.n_cond_tokens, L = cfg.depth, C = cfg.channels;
    const int M = B * NL;
    RALD_CHECK(w.rows >= M, "dit: workspace smaller than the batch (internal)");
    float* const ws_x = w.x; float* const ws_part = w.part;
    bf16 *const ws_h = w.h, *const ws_qk = w.qk, *const ws_vt = w.vt, *const ws_o = w.o, *const ws_q2 = w.q2, *const ws_g = w.g;
    unsigned char *const ws_h8 = w.h8, *const ws_hs = w.hs, *const ws_g8 = w.g8, *const ws_gs = w.gs;
    x += (size_t)b0 * NL * C;
    out += (size_t)b0 * NL * C;
    const int64_t mrow = mod_row();
    const int srow = sigma_row + (per_sample ? b0 : 0);
    const float* mod = tb.mod + (int64_t)srow * mrow;
    const int64_t gstride = per_sample ? mrow : 0;
    const float* coef = raw_F ? coef_raw : tb.coef + 4 * (int64_t)srow;
    const int cstride = (per_sample && !raw_F) ? 4 : 0;
    // the cache is laid out for the FULL batch; this range's rows start at sample b0
    const bf16* Kc0 = (const bf16*)((const char*)cache + COND_HEADER_BYTES);
    const bf16* Vtc0 = Kc0 + (size_t)Bfull * T * L * D;
    const bf16* Kc = Kc0 + (size_t)b0 * T * L * D;
    const bf16* Vtc = Vtc0 + (size_t)b0 * L * D * T;
    const float scale = 1.0f / sqrtf((float)cfg.d_head);
    const float qscale = scale * 1.4426950408889634f;     // softmax scale and log2(e) folded into q by the projection epilogue

    // live timing (rald_dit_profile_begin / _end): events around the launches of one kind on the launch stream
    auto timed_launch = [&](int kind, auto&& launch) -> int {
        const bool timed = timed_ok && prof_on && ((prof_mask >> kind) & 1u) && prof_used + 2 <= (int)prof_ev.size();
        if (timed) RALD_HIP(hipEventRecord(prof_ev[prof_used], st));
        RALD_TRY(launch());
        if (timed) { RALD_HIP(hipEventRecord(prof_ev[prof_used + 1], st)); prof_kind[prof_used / 2] = kind; prof_used += 2; }
        return 0;
    };
    // probe builds: RALD_FUSE_LN=0 falls back to separate LayerNorm launches (A/B and debugging)
    static const bool fuse_ln = RALD_PROBE_ENV("RALD_FUSE_LN", 1) != 0;
    auto resid_ln = [&](const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, int K, const float* mnext) -> int {
        // x += A.W^T + bias, then (if mnext) h = AdaLN(x; mnext) for the next sub-block
        if (splitk_for(M, K))
            return resid_splitk_ln(A, lda, W, ldw, bias, ws_x, mnext ? ws_h : nullptr, mnext, mnext ? mnext + D : nullptr, gstride, NL, 1.0f, 1e-5f,
                                   M, K, splitk_for(M, K), ws_part, st);
        if (fuse_ln && mnext && gemm_resid_ln_pays(M, K)) {
            GemmLnArgs g;
            g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.x = ws_x; g.h = ws_h;
            g.g = mnext; g.b = mnext + D; g.gstride = gstride; g.rows_per_group = NL; g.add_one = 1.0f; g.eps = 1e-5f;
            g.M = M; g.K = K;
            return gemm_resid_ln(g, st);
        }
        GemmArgs o = gemm_args(A, lda, W, ldw, ws_x, D, bias, M, D, K);
        RALD_TRY(gemm_nt(o, EPI_RESID, st));
        if (mnext) RALD_TRY(layernorm_mod(ws_x, ws_h, M, D, mnext, mnext + D, gstride, NL, 1.0f, 1e-5f, st));
        return 0;
    };
    // the same with one weight matrix per sample (W + sample * strideW): the folded cross-attention's output projection
    auto resid_ln_w = [&](const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, int K, const float* mnext, int64_t strideW) -> int {
        if (fuse_ln && mnext && gemm_resid_ln_pays(M, K)) {
            GemmLnArgs g;
            g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.x = ws_x; g.h = ws_h;
            g.g = mnext; g.b = mnext + D; g.gstride = gstride; g.rows_per_group = NL; g.add_one = 1.0f; g.eps = 1e-5f;
            g.M = M; g.K = K; g.strideW = strideW; g.w_rows = NL;
            return gemm_resid_ln(g, st);
        }
        GemmArgs o = gemm_args(A, lda, W, ldw, ws_x, D, bias, NL, D, K);
        o.batch = B; o.strideA = (int64_t)NL * lda; o.strideB = strideW; o.strideC = (int64_t)NL * D;
        RALD_TRY(gemm_nt(o, EPI_RESID, st));
        if (mnext) RALD_TRY(layernorm_mod(ws_x, ws_h, M, D, mnext, mnext + D, gstride, NL, 1.0f, 1e-5f, st));
        return 0;
    };
    const bool fold = cond_fold(Bfull);
    RALD_CHECK(!fold || !small_m_fused(M, NL, cfg.n_heads, D, T), "dit: a folded condition cache cannot feed the small-batch kernels (internal)");
    const bf16* Gt0 = Vtc0 + (size_t)2 * Bfull * T * L * D;         // (behind Vc; only there when fold)
    const bf16* Gt = Gt0 + (size_t)b0 * L * D * D;
    const bf16* Ut = Gt0 + (size_t)Bfull * L * D * D + (size_t)b0 * L * D * D;
    RALD_TRY(proj_in(x, w_in, ws_x, M, C, D, coef, cstride, NL, st));
    if (cfg.qkv_dtype >= 1) {
        // ---- MXFP8 q/k/v projections (BASELINE config #5; qkv_dtype 2 adds the GEGLU projection of the feed-forward).  The AdaLN outputs that feed to_q / to_k / to_v (norm1,
        // norm2) are produced directly in e4m3 + e8m0/32 by the fused residual+LayerNorm GEMM epilogue (or by
        // layernorm_mod_mx8 where that kernel does not pay) and multiplied on v_mfma_scale_f32_16x16x128_f8f6f4;
        // to_out and the feed-forward stay bf16 (their A operands - attention output, GEGLU output - would need
        // quantising epilogues of their own before fp8 pays there).
        auto ln8 = [&](const float* m) { return layernorm_mod_mx8(ws_x, ws_h8, ws_hs, M, D, m, m + D, gstride, NL, 1.0f, 1e-5f, st); };
        auto mx = [&](const unsigned char* A8, const unsigned char* SA, const unsigned char* B8, const unsigned char* SB, void* Cp, int64_t ldc,
                      const float* bias, int m, int n) {
            Mx8Args a;
            a.A8 = A8; a.SA = SA; a.B8 = B8; a.SB = SB; a.strideSA = 0; a.strideSB = 0;
            a.g = gemm_args(nullptr, D, nullptr, D, Cp, ldc, bias, m, n, D);
            return a;
        };
        // x += A.W^T + bias (bf16 GEMM), then the next AdaLN as MXFP8 into ws_h8 / ws_hs
        auto resid_ln8 = [&](const bf16* A, int64_t lda, const bf16* W, int64_t ldw, const float* bias, int K, const float* mnext) -> int {
            if (fuse_ln && gemm_resid_ln_pays(M, K)) {
                GemmLnArgs g;
                g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.x = ws_x; g.h = nullptr; g.h8 = ws_h8; g.hs = ws_hs;
                g.g = mnext; g.b = mnext + D; g.gstride = gstride; g.rows_per_group = NL; g.add_one = 1.0f; g.eps = 1e-5f;
                g.M = M; g.K = K;
                return gemm_resid_ln(g, st);
            }
            GemmArgs o = gemm_args(A, lda, W, ldw, ws_x, D, bias, M, D, K);
            RALD_TRY(gemm_nt(o, EPI_RESID, st));
            return ln8(mnext);
        };
        RALD_TRY(ln8(mod));                                                               // norm1 of block 0
        for (int li = 0; li < L; ++li) {
            const Layer& l = layers[li];
            const float* m2 = mod + (int64_t)(li * 3 + 1) * 2 * D;
            const float* m3 = mod + (int64_t)(li * 3 + 2) * 2 * D;
            AttnArgs a1;
            if (NL % 64 == 0) {                                                            // fused q|k|v projection, row-major V (see the bf16 path)
                Mx8Args qkv = mx(ws_h8, ws_hs, l.q8_qk, l.s8_qk, ws_qk, 3 * D, nullptr, M, 3 * D);
                qkv.g.alpha = qscale; qkv.g.alpha_ncols = D;
                RALD_TRY(gemm_mx8(qkv, EPI_BF16, st));
                a1.Q = ws_qk; a1.ldq = 3 * D; a1.strideQ = (int64_t)NL * 3 * D;
                a1.K = ws_qk + D; a1.ldk = 3 * D; a1.strideK = (int64_t)NL * 3 * D;
                a1.Vt = nullptr; a1.ldvt = 0; a1.strideVt = 0;
                a1.V = ws_qk + 2 * D; a1.ldv = 3 * D; a1.strideV = (int64_t)NL * 3 * D;
            } else {
                Mx8Args qk = mx(ws_h8, ws_hs, l.q8_qk, l.s8_qk, ws_qk, 2 * D, nullptr, M, 2 * D);
                qk.g.alpha = qscale; qk.g.alpha_ncols = D;
                RALD_TRY(gemm_mx8(qk, EPI_BF16, st));
                Mx8Args vt = mx(l.q8_v, l.s8_v, ws_h8, ws_hs, ws_vt, NL, nullptr, D, NL);    // V^T = Wv . h^T per sample
                vt.g.batch = B; vt.g.strideB = (int64_t)NL * D; vt.strideSB = (int64_t)NL * D / 32; vt.g.strideC = (int64_t)D * NL;
                RALD_TRY(gemm_mx8(vt, EPI_BF16, st));
                a1.Q = ws_qk; a1.ldq = 2 * D; a1.strideQ = (int64_t)NL * 2 * D;
                a1.K = ws_qk + D; a1.ldk = 2 * D; a1.strideK = (int64_t)NL * 2 * D;
                a1.Vt = ws_vt; a1.ldvt = NL; a1.strideVt = (int64_t)D * NL;
            }
            a1.O = ws_o; a1.ldo = D; a1.strideO = (int64_t)NL * D;
            a1.nq = NL; a1.nk = NL; a1.k_rows = NL; a1.heads = cfg.n_heads; a1.batch = B; a1.scale = scale; a1.q_prescaled = 1;
            RALD_TRY(attention_d64(a1, st));
            RALD_TRY(resid_ln8(ws_o, D, l.w_o, D, l.b_o, D, m2));                        // + norm2 (MXFP8) for to_q of attn2
            Mx8Args q2 = mx(ws_h8, ws_hs, l.q8_q2, l.s8_q2, ws_q2, D, nullptr, M, D);
            q2.g.alpha = qscale;
            RALD_TRY(gemm_mx8(q2, EPI_BF16, st));
            AttnArgs a2;
            a2.Q = ws_q2; a2.ldq = D; a2.strideQ = (int64_t)NL * D;
            a2.K = Kc + (size_t)li * D; a2.ldk = (int64_t)L * D; a2.strideK = (int64_t)T * L * D;
            a2.Vt = Vtc + (size_t)li * D * T; a2.ldvt = T; a2.strideVt = (int64_t)L * D * T;
            a2.O = ws_o; a2.ldo = D; a2.strideO = (int64_t)NL * D;
            a2.nq = NL; a2.nk = T; a2.k_rows = T; a2.heads = cfg.n_heads; a2.batch = B; a2.scale = scale; a2.q_prescaled = 1;
            RALD_TRY(attention_d64(a2, st));
            // qkv_dtype 3: the GEGLU output leaves the FF1 epilogue as MXFP8 and ff.net.2 (+ residual + next AdaLN) consumes it
            if (fork_pending && timed_ok) { RALD_HIP(hipEventRecord(ev_fork, st)); fork_pending = false; }   // two-stream schedule: the other half starts here
            const bool ff2_mx = cfg.qkv_dtype == 3 && M % 256 == 0 && M >= 4096 && fuse_ln && gemm_resid_ln_pays(M, 4 * D);
            if (cfg.qkv_dtype >= 2) {
                RALD_TRY(resid_ln8(ws_o, D, l.w_o2, D, l.b_o2, D, m3));                  // + norm3 (MXFP8)
                Mx8Args f1 = mx(ws_h8, ws_hs, l.q8_ff1, l.s8_ff1, ws_g, 4 * D, l.b_ff1, M, 8 * D);
                if (ff2_mx) { f1.g.out8 = ws_g8; f1.g.outs = ws_gs; }
                RALD_TRY(gemm_mx8(f1, EPI_GEGLU, st));
                if (ff2_mx) {
                    const float* mn = (li + 1 < L) ? mod + (int64_t)((li + 1) * 3) * 2 * D : m3;      // last block: the LN output is unused
                    GemmLnArgs g;
                    g.A = nullptr; g.W = nullptr; g.A8 = ws_g8; g.SA = ws_gs; g.W8 = l.q8_ff2; g.SW = l.s8_ff2; g.lda = 4 * D; g.ldw = 4 * D;
                    g.bias = l.b_ff2; g.x = ws_x; g.h = nullptr; g.h8 = ws_h8; g.hs = ws_hs;
                    g.g = mn; g.b = mn + D; g.gstride = gstride; g.rows_per_group = NL; g.add_one = 1.0f; g.eps = 1e-5f; g.M = M; g.K = 4 * D;
                    RALD_TRY(gemm_resid_ln(g, st));
                    continue;
                }
            } else {
                RALD_TRY(resid_ln(ws_o, D, l.w_o2, D, l.b_o2, D, m3));                   // + norm3 (bf16) for the feed-forward
                GemmArgs f1 = gemm_args(ws_h, D, l.w_ff1, D, ws_g, 4 * D, l.b_ff1, M, 8 * D, D);
                RALD_TRY(gemm_nt(f1, EPI_GEGLU, st));
            }
            if (li + 1 < L) RALD_TRY(resid_ln8(ws_g, 4 * D, l.w_ff2, 4 * D, l.b_ff2, 4 * D, mod + (int64_t)((li + 1) * 3) * 2 * D));
            else RALD_TRY(resid_ln(ws_g, 4 * D, l.w_ff2, 4 * D, l.b_ff2, 4 * D, nullptr));
        }
        RALD_TRY(final_norm_proj(ws_x, norm_g, norm_b, w_out, x, out, M, D, C, coef, cstride, NL, st, w_out_hl));
        return 0;
    }
    RALD_TRY(layernorm_mod(ws_x, ws_h, M, D, mod, mod + D, gstride, NL, 1.0f, 1e-5f, st));      // norm1 of block 0
    for (int li = 0; li < L; ++li) {
        const Layer& l = layers[li];
        const float* m1 = mod + (int64_t)(li * 3 + 0) * 2 * D;
        const float* m2 = mod + (int64_t)(li * 3 + 1) * 2 * D;
        const float* m3 = mod + (int64_t)(li * 3 + 2) * 2 * D;
        // ---- x += attn1(norm1(x, t))                                               (:166)
        // (norm1(x) is already in ws_h: produced by the previous block's FF2 epilogue / the prologue)
        (void)m1;
        // one projection for q | k | v (N = 1536); the attention kernel reads V row-major through ds_read_b64_tr_b16, so no
        // transposed copy of V and no separate V^T GEMM (RALD_ATTN_VROW=0: the two-GEMM form, for A/B runs)
        static const bool vrow_env = RALD_PROBE_ENV("RALD_ATTN_VROW", 1) != 0;
        const bool vrow = vrow_env && NL % 64 == 0;
        AttnArgs a1;
        if (vrow) {
            GemmArgs qkv = gemm_args(ws_h, D, l.w_qk, D, ws_qk, 3 * D, nullptr, M, 3 * D, D);
            qkv.alpha = qscale; qkv.alpha_ncols = D;                                // q columns only
            RALD_TRY(gemm_nt(qkv, EPI_BF16, st));
            a1.Q = ws_qk; a1.ldq = 3 * D; a1.strideQ = (int64_t)NL * 3 * D;
            a1.K = ws_qk + D; a1.ldk = 3 * D; a1.strideK = (int64_t)NL * 3 * D;
            a1.Vt = nullptr; a1.ldvt = 0; a1.strideVt = 0;
            a1.V = ws_qk + 2 * D; a1.ldv = 3 * D; a1.strideV = (int64_t)NL * 3 * D;
        } else {
            GemmArgs qk = gemm_args(ws_h, D, l.w_qk, D, ws_qk, 2 * D, nullptr, M, 2 * D, D);
            qk.alpha = qscale; qk.alpha_ncols = D;
            RALD_TRY(gemm_nt(qk, EPI_BF16, st));
            GemmArgs vt = gemm_args(l.w_v, D, ws_h, D, ws_vt, NL, nullptr, D, NL, D);   // V^T = Wv . h^T per sample
            vt.batch = B; vt.strideB = (int64_t)NL * D; vt.strideC = (int64_t)D * NL;
            RALD_TRY(gemm_nt(vt, EPI_BF16, st));
            a1.Q = ws_qk; a1.ldq = 2 * D; a1.strideQ = (int64_t)NL * 2 * D;
            a1.K = ws_qk + D; a1.ldk = 2 * D; a1.strideK = (int64_t)NL * 2 * D;
            a1.Vt = ws_vt; a1.ldvt = NL; a1.strideVt = (int64_t)D * NL;
        }
        if (vrow && small_m_fused(M, NL, cfg.n_heads, D, T)) {
            // small batches (attn_small.hip): attention + that head's slice of to_out in one kernel per (head, 32-query block),
            // the 8 per-head partials summed into the residual stream together with the next AdaLN; then to_q + the 64-key
            // radar cross-attention + to_out slice likewise.  8 launches per block instead of 12.
            RALD_TRY(attn_self_proj(ws_qk, 3 * D, l.w_o, ws_part, NL, cfg.n_heads, B, st, true));
            RALD_TRY(reduce_resid_ln(ws_part, cfg.n_heads, (int64_t)M * D, l.b_o, ws_x, ws_h, M, m2, m2 + D, gstride, NL, 1.0f, 1e-5f, st, true));
            RALD_TRY(xattn_q2_proj(ws_h, l.w_q2, Kc + (size_t)li * D, (int64_t)L * D, (int64_t)T * L * D, Vtc + (size_t)li * D * T, T, (int64_t)L * D * T,
                                   l.w_o2, ws_part, M, NL, cfg.n_heads, T, qscale, st, true));
            RALD_TRY(reduce_resid_ln(ws_part, cfg.n_heads, (int64_t)M * D, l.b_o2, ws_x, ws_h, M, m3, m3 + D, gstride, NL, 1.0f, 1e-5f, st, true));
        } else {
        a1.O = ws_o; a1.ldo = D; a1.strideO = (int64_t)NL * D;
        a1.nq = NL; a1.nk = NL; a1.k_rows = NL; a1.heads = cfg.n_heads; a1.batch = B; a1.scale = scale; a1.q_prescaled = 1;
        RALD_TRY(attention_d64(a1, st));
        RALD_TRY(timed_launch(1, [&] { return resid_ln(ws_o, D, l.w_o, D, l.b_o, D, m2); }));   // + norm2 for the next sub-block
        // ---- x += attn2(norm2(x, t), context)                                      (:167)
        if (fold) {
            // folded form (see cond_fold in dit.h): P = softmax over each head's 64 keys of h.Gt^T, then x += P.Ut^T + b_o (+ norm3)
            GemmArgs p1 = gemm_args(ws_h, D, Gt + (size_t)li * D * D, D, ws_q2, D, nullptr, NL, D, D);
            p1.batch = B; p1.strideA = (int64_t)NL * D; p1.strideB = (int64_t)L * D * D; p1.strideC = (int64_t)NL * D;
            RALD_TRY(gemm_nt(p1, EPI_SOFTMAX64, st));
            RALD_TRY(timed_launch(2, [&] { return resid_ln_w(ws_q2, D, Ut + (size_t)li * D * D, D, l.b_o2, D, m3, (int64_t)L * D * D); }));
        } else {
        GemmArgs q2 = gemm_args(ws_h, D, l.w_q2, D, ws_q2, D, nullptr, M, D, D);
        q2.alpha = qscale;
        RALD_TRY(gemm_nt(q2, EPI_BF16, st));
        AttnArgs a2;
        a2.Q = ws_q2; a2.ldq = D; a2.strideQ = (int64_t)NL * D;
        a2.K = Kc + (size_t)li * D; a2.ldk = (int64_t)L * D; a2.strideK = (int64_t)T * L * D;
        a2.Vt = Vtc + (size_t)li * D * T; a2.ldvt = T; a2.strideVt = (int64_t)L * D * T;
        a2.O = ws_o; a2.ldo = D; a2.strideO = (int64_t)NL * D;
        a2.nq = NL; a2.nk = T; a2.k_rows = T; a2.heads = cfg.n_heads; a2.batch = B; a2.scale = scale; a2.q_prescaled = 1;
        RALD_TRY(attention_d64(a2, st));
        RALD_TRY(resid_ln(ws_o, D, l.w_o2, D, l.b_o2, D, m3));                     // + norm3
        }
        }
        // ---- x += ff(norm3(x, t))                                                   (:168)
        if (fork_pending && timed_ok) { RALD_HIP(hipEventRecord(ev_fork, st)); fork_pending = false; }   // two-stream schedule: the other half starts here
        GemmArgs f1 = gemm_args(ws_h, D, l.w_ff1, D, ws_g, 4 * D, l.b_ff1, M, 8 * D, D);
        RALD_TRY(timed_launch(0, [&] { return gemm_nt(f1, EPI_GEGLU, st); }));
        const float* m1_next = (li + 1 < L) ? mod + (int64_t)((li + 1) * 3) * 2 * D : nullptr;   // norm1 of the next block
        RALD_TRY(timed_launch(3, [&] { return resid_ln(ws_g, 4 * D, l.w_ff2, 4 * D, l.b_ff2, 4 * D, m1_next); }));
    }
    RALD_TRY(final_norm_proj(ws_x, norm_g, norm_b, w_out, x, out, M, D, C, coef, cstride, NL, st, w_out_hl));
    return 0;
}

int Dit::profile_begin() {
    if (prof_ev.empty()) {
        prof_ev.resize(2 * 16384);
        prof_kind.assign(16384, 0);
        for (auto& e : prof_ev) RALD_HIP(hipEventCreate(&e));
    }
    prof_used = 0;
    prof_on = true;
    return 0;
}
int Dit::profile_end_kinds(double* total_ms, int* launches) {
    prof_on = false;
    for (int k = 0; k < PROF_KINDS; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
    for (int i = 0; i + 1 < prof_used; i += 2) {
        RALD_HIP(hipEventSynchronize(prof_ev[i + 1]));
        float ms = 0.f;
        RALD_HIP(hipEventElapsedTime(&ms, prof_ev[i], prof_ev[i + 1]));
        const int k = prof_kind[i / 2];
        total_ms[k] += ms;
        launches[k] += 1;
    }
    prof_used = 0;
    return 0;
}
int Dit::profile_end(double* total_ms, int* launches) {
    double ms[PROF_KINDS];
    int n[PROF_KINDS];
    RALD_TRY(profile_end_kinds(ms, n));
    *total_ms = ms[0];
    *launches = n[0];
    return 0;
}
Dit::~Dit() {
    for (auto& e : prof_ev) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
}

int Dit::sample(const float* latents, int B, const void* cache, int num_steps, float smin, float smax, float rho,
                float* out, hipStream_t st) {
    RALD_CHECK(num_steps >= 2 && num_steps <= 2048, "dit: num_steps must be in [2,2048]");
    RALD_TRY(cond_registry.check(cache, cond_header(B), st, "condition cache"));
    // Karras schedule in fp32, as the reference computes it (edm_sampler :246-249); t_N = 0.
    std::vector<float> t(num_steps + 1);
    const float a = powf(smax, 1.0f / rho), b = powf(smin, 1.0f / rho);
    for (int i = 0; i < num_steps; ++i) t[i] = powf(a + (float)i / (float)(num_steps - 1) * (b - a), rho);
    t[num_steps] = 0.f;
    RALD_TRY(reserve(B));
    RALD_TRY(build_table(tables[1], t.data(), num_steps, st));
    const int64_t n = (int64_t)B * cfg.n_latents * cfg.channels;
    RALD_TRY(scale_f32(latents, ws_xcur, t[0], n, st));                              // x_next = latents * t_0
    for (int i = 0; i < num_steps; ++i) {
        const float tc = t[i], tn = t[i + 1];
        RALD_TRY(denoise(ws_xcur, B, i, 0, cache, ws_den, 0, st, 1));                // Euler step (:263-266)
        const bool last = i == num_steps - 1;
        RALD_TRY(heun_euler(ws_xcur, ws_den, tc, tn, ws_dcur, last ? out : ws_xeul, n, st));
        if (!last) {                                                                 // 2nd-order correction (:269-273)
            RALD_TRY(denoise(ws_xeul, B, i + 1, 0, cache, ws_den, 0, st, 1));
            RALD_TRY(heun_correct(ws_xcur, ws_xeul, ws_den, ws_dcur, tc, tn, ws_xcur, n, st));   // in place, elementwise
        }
    }
    return 0;
}

}  // namespace rald
