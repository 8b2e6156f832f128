// Host-side state of the denoiser handle (C++; the C-ABI wrappers live in api.hip).
#pragma once
#include <map>
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

// Caller-owned opaque device blobs (the denoiser's condition cache, the autoencoder's decoder context) have a layout that depends
// on the batch they were built for.  Each starts with a 64-byte header (magic, batch, layout flag, configuration hash, size),
// mirrored in a host-side registry keyed by the device pointer.  A call that consumes a blob checks the registry BEFORE any launch;
// an unknown pointer (a copy of a blob, a recycled address) is verified once by reading its header back (synchronises the stream;
// impossible under graph capture, where the call fails instead).
struct BlobHeader { uint32_t magic; int32_t batch; int32_t flag; uint32_t cfg_hash; int64_t bytes; uint32_t pad[10]; };
static_assert(sizeof(BlobHeader) == 64, "blob header is 64 bytes");
constexpr int BLOB_HEADER_BYTES = 64;
struct BlobRegistry {
    std::map<const void*, BlobHeader> known;
    int stamp(void* blob, const BlobHeader& hd, hipStream_t st);                                  // writes the header (a kernel on st) and registers it
    int check(const void* blob, const BlobHeader& want, hipStream_t st, const char* what);       // 0 when blob carries exactly `want`
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
    // the Decoder half of the RadarAutoencoder (models_radar_encoder.py:243-359) on the same kernels: a second object of this class
    int create_decoder(int ch, int z_ch, int out_ch, int R, int A, int E, DeviceArena* arena);
    // z [B][R/16][A/16][E/16][z_ch] -> pred4 [B][R][A][E][4] fp32 (the first out_ch channels are the reconstruction)
    int decode(const float* z, int B, float* pred4, hipStream_t st);
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
    // the per-NFE activations of one batch (or of one half of a batch, see two-stream schedule below)
    struct Work {
        float* x = nullptr;                 // residual stream [M][D] fp32
        float* part = nullptr;              // split-K partial sums of the small-batch FF2 (norm.hip resid_splitk_ln) / per-head slabs
        bf16 *h = nullptr, *qk = nullptr, *vt = nullptr, *o = nullptr, *q2 = nullptr, *g = nullptr;
        unsigned char *h8 = nullptr, *hs = nullptr;   // MXFP8 AdaLN outputs feeding q/k/v (qkv_dtype >= 1)
        unsigned char *g8 = nullptr, *gs = nullptr;   // MXFP8 GEGLU output feeding ff.net.2 (qkv_dtype == 3)
        int rows = 0;
    };
    Work wk[2];                             // wk[0]: the whole batch (or its first half), wk[1]: the second half on the side stream
    int alloc_work(Work& w, size_t M);
    void free_work(Work& w);
    float *ws_xcur = nullptr, *ws_xeul = nullptr, *ws_den = nullptr, *ws_dcur = nullptr;
    bf16* ws_tok = nullptr;
    // Two-stream schedule: from `split_min` samples up an NFE runs as two half-batches on two HIP streams (the caller's and a
    // handle-owned one, fork / join by events).  Inside one launch every CU runs the same phase at the same time (MFMA loop,
    // then the store-heavy epilogue; an HBM-bound kernel leaves the matrix cores idle): a second, independent half-batch fills
    // those holes.  Each half runs exactly the kernels a batch of that size runs alone, so results are bit-identical to two
    // sequential calls.  split_min = 0 turns it off.
    int split_min = 256;     // measured on MI355X (tools/ab_two_stream.py): B = 128 22.35 vs 22.08 ms unsplit (no gain: a batch of 128 already
                             // runs two rounds of tiles per kernel and is out of step by itself), B = 256 44.5 vs 45.3 ms (+1.9 %)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_pending = false;
    // (the first half is a multiple of 64 samples: the GEMMs' per-tile k-offsets repeat every 256 row tiles, so both halves see the
    //  offsets they would see inside the whole batch)
    int split_sizes(int B, int& b0) const { b0 = ((B / 2 + 32) / 64) * 64; return split_min > 0 && B >= split_min && b0 >= 64 && b0 < B; }
    // condition cache header + registry (BlobRegistry above)
    static constexpr uint32_t COND_MAGIC = 0x52414c44u;          // "RALD"
    static constexpr int COND_HEADER_BYTES = BLOB_HEADER_BYTES;
    BlobRegistry cond_registry;
    uint32_t cfg_hash() const;
    BlobHeader cond_header(int B) const;

    // live timing of the dominant kernel (the FF1 GEGLU GEMM) with HIP events on the launch stream
    // kinds: 0 = FF1 GEGLU GEMM, 1 = attn1.to_out + residual + AdaLN (K = 512), 2 = attn2 output + residual + AdaLN (K = 512, per-sample
    // weights when folded), 3 = FF2 + residual + AdaLN (K = 2048)
    static constexpr int PROF_KINDS = 4;
    bool prof_on = false;
    unsigned prof_mask = 0xf;          // bit k: launches of kind k are bracketed while prof_on (an event pair costs ~2.5 us of stream time: 96 pairs per NFE
                                       // were 0.5 ms of a 22 ms NFE - a caller that wants the whole-job rate beside the dominant kernel's time narrows the mask)
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_kind;
    int prof_used = 0;
    int profile_begin();
    int profile_end(double* total_ms, int* launches);                 // kind 0 only (the round-1 entry point)
    int profile_end_kinds(double* total_ms, int* launches);           // arrays of PROF_KINDS
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
    // samples [b0, b0 + Bn) of a batch of Bfull (the condition cache is laid out for Bfull) on workspace w
    int denoise_range(const float* x, int Bfull, int b0, int Bn, int sigma_row, int per_sample, const void* cache, float* out, int raw_F,
                      hipStream_t st, int slot, Work& w, bool timed_ok);
    int sample(const float* latents, int B, const void* cache, int num_steps, float smin, float smax, float rho, float* out,
               hipStream_t st);
};

}  // namespace rald
