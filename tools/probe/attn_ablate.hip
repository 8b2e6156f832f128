// Which unit bounds attention_d64_kernel?  Build with -DRALD_ATTN_ABLATE=<bits> (see attention.hip) and time
// the self-attention shape of the denoiser at B = 64 (fused q|k|v buffer, row-major V, prescaled q).
#include "../../rald_amd/csrc/attention.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace rald { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    using namespace rald;
    const int B = 64, N = 512, H = 8, D = 512;
    const int NK = argc > 1 ? atoi(argv[1]) : N;   // keys per sample (rows of the buffer reused modulo N via stride 0 trick is not possible: allocate NK rows)
    const int R = NK > N ? NK : N;
    std::vector<unsigned short> h((size_t)B * R * 3 * D);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16 *qkv, *o;
    hipMalloc(&qkv, h.size() * 2); hipMalloc(&o, (size_t)B * N * D * 2);
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    AttnArgs a{};
    a.Q = qkv; a.ldq = 3 * D; a.strideQ = (int64_t)R * 3 * D;
    a.K = qkv + D; a.ldk = 3 * D; a.strideK = (int64_t)R * 3 * D;
    a.V = qkv + 2 * D; a.ldv = 3 * D; a.strideV = (int64_t)R * 3 * D; a.Vt = nullptr;
    a.O = o; a.ldo = D; a.strideO = (int64_t)N * D;
    a.nq = N; a.nk = NK; a.heads = H; a.batch = B; a.k_rows = R; a.scale = 0.125f; a.q_prescaled = 1;
    float* part; hipMalloc(&part, 64); hipMemset(part, 0, 64); a.part = part;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) attention_d64(a, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) attention_d64(a, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned short> ho((size_t)B * N * D);
    hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < ho.size(); ++i) { unsigned u = (unsigned)ho[i] << 16; float f; memcpy(&f, &u, 4); cs += f * ((i % 7) + 1); }
    float hp[2]; hipMemcpy(hp, part, 8, hipMemcpyDeviceToHost);
    if (hp[1] > 0) printf("WG 1000: %.0f shader clocks in %.0f ticks of the 100 MHz clock -> %.2f GHz, %.2f us\n", hp[0], hp[1], hp[0] / hp[1] * 0.1, hp[1] * 0.01);
    printf("stages=%d checksum=%.6e ", RALD_ATTN_STAGES, cs);
    printf("nk=%d ablate=%d  %.2f us / launch  (%.0f TFLOP/s nominal)\n", NK, RALD_ATTN_ABLATE, ms * 1000 / 50, 4.0 * N * NK * 64 * H * B / (ms / 50 * 1e-3) / 1e12);
    return 0;
}
