// Where does the fused residual + LayerNorm GEMM (128 x 512 tiles) spend its k-loop: waiting at the tile hand-overs (DMA not landed /
// barrier skew) or issuing?  Wave 0 of every workgroup sums the shader clocks of its hand-over waits and of the whole loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probe/ln_timeline.hip -o tools/probe/ln_timeline
#define RALD_LN_STAMPS 1
#include "../../rald_amd/csrc/gemm_ln.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
namespace rald { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    using namespace rald;
    const int K = argc > 1 ? atoi(argv[1]) : 2048;
    const int M = 32768, N = 512;
    std::vector<unsigned short> h((size_t)M * K), w((size_t)N * K);
    unsigned s = 777;
    auto fill = [&](std::vector<unsigned short>& v, float sc) {
        for (auto& x : v) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) & 0xffff) / 65536.f - 0.5f) * sc; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    };
    fill(h, 1.f); fill(w, 0.05f);
    bf16 *A, *W, *H; float *x, *bias, *g;
    hipMalloc(&A, h.size() * 2); hipMalloc(&W, w.size() * 2); hipMalloc(&H, (size_t)M * N * 2); hipMalloc(&x, (size_t)M * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&g, N * 8);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4); hipMemset(g, 0, N * 8); hipMemset(x, 0, (size_t)M * N * 4);
    GemmLnArgs a;
    a.A = A; a.lda = K; a.W = W; a.ldw = K; a.bias = bias; a.x = x; a.h = H; a.g = g; a.b = g + N; a.gstride = 0; a.rows_per_group = 512; a.add_one = 1.f; a.eps = 1e-5f; a.M = M; a.K = K;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) if (gemm_resid_ln(a, 0)) { fprintf(stderr, "launch failed\n"); return 1; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < 30; ++i) { hipMemsetAsync(x, 0, 64, 0); gemm_resid_ln(a, 0); }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(1024 * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_ln_stamps), st.size() * 8);
    std::vector<double> wait, loop, epi;
    for (int i = 0; i < M / 128; ++i) { wait.push_back((double)st[i * 4]); loop.push_back((double)st[i * 4 + 1]); epi.push_back((st[i * 4 + 3] - st[i * 4 + 2]) * 0.01); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("K %d: %.1f us / launch (with stamps) = %.0f TFLOP/s; per workgroup (median): k-loop %.0f shader clocks, of which %.0f (%.0f %%) waiting at the %d hand-overs; MFMA issue alone would be %d clocks; epilogue %.2f us\n",
           K, ms * 1000 / 30, 2.0 * M * N * K / (ms * 1000 / 30) / 1e6, med(loop), med(wait), 100.0 * med(wait) / med(loop), K / 64, K / 64 * 2048, med(epi));
    return 0;
}
