// Where do the small-batch fused attention kernels (rald_amd/csrc/attn_small.hip) spend their time?  One wave stamps the
// shader clock at the phase boundaries; B = 1 shapes, random operands, 20 back-to-back launches each.
#define RALD_SMALL_STAMPS 1
#include "../../rald_amd/csrc/attn_small.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace rald { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main() {
    using namespace rald;
    const int NL = 512, D = 512, T = 64, L = 2;
    unsigned s = 777;
    auto rnd = [&](size_t n) {
        std::vector<unsigned short> v(n);
        for (auto& x : v) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.5f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
        bf16* d; hipMalloc(&d, n * 2); hipMemcpy(d, v.data(), n * 2, hipMemcpyHostToDevice); return d;
    };
    bf16 *qkv = rnd((size_t)NL * 3 * D), *Wo = rnd((size_t)D * D), *Wq = rnd((size_t)D * D), *hin = rnd((size_t)NL * D), *Kc = rnd((size_t)T * L * D), *Vt = rnd((size_t)L * D * T);
    float* part; hipMalloc(&part, (size_t)8 * NL * D * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 2; ++k) {
        auto run = [&]() { return k == 0 ? attn_self_proj(qkv, 3 * D, Wo, part, NL, 8, 1, 0) : xattn_q2_proj(hin, Wq, Kc, L * D, (int64_t)T * L * D, Vt, T, (int64_t)L * D * T, Wo, part, NL, NL, 8, T, 0.18f, 0); };
        for (int i = 0; i < 5; ++i) if (run()) return 1;
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[2][16]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_small_stamps), sizeof(c));
        printf("%s: %.2f us per launch back to back; shader clocks since entry:", k == 0 ? "attn_self_proj" : "xattn_q2_proj", ms * 1000 / 20);
        for (int i = 1; i < (k == 0 ? 6 : 7); ++i) printf(" [%d] %lld", i, c[k][i] - c[k][0]);
        if (k == 0) printf("  passes over the same code: %lld %lld %lld clocks", c[0][12] - c[0][8], c[0][13] - c[0][9], c[0][14] - c[0][10]);
        printf("\n");
    }
    return 0;
}
