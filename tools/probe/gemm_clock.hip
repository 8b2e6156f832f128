// What shader clock does the chip hold while the FF1 GEMM (256x256 tiles, GEGLU epilogue) runs?  One workgroup in the middle
// of the grid samples clock64() (shader clocks) and wall_clock64() (100 MHz) over its life; random operands, B = 64 shape.
#define RALD_GEMM_CLOCK 1
#include "../../rald_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace rald { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    using namespace rald;
    const int M = 32768, K = 512, N = 4096;
    const bool zeros = argc > 1 && atoi(argv[1]) == 1;          // all-zero operands: same instruction stream, far fewer bit flips
    std::vector<unsigned short> h((size_t)M * K), w((size_t)N * K);
    unsigned s = 777;
    auto fill = [&](std::vector<unsigned short>& v) {
        for (auto& x : v) { s = s * 1664525u + 1013904223u; float f = zeros ? 0.f : ((s >> 8) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    };
    fill(h); fill(w);
    bf16 *A, *W, *C; float* bias;
    hipMalloc(&A, h.size() * 2); hipMalloc(&W, w.size() * 2); hipMalloc(&C, (size_t)M * (N / 2) * 2); hipMalloc(&bias, N * 4);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    GemmArgs g = gemm_args(A, K, W, K, C, N / 2, bias, M, N, K);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) if (gemm_nt(g, EPI_GEGLU, 0)) return 1;
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) gemm_nt(g, EPI_GEGLU, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_gemm_clk), 16);
    const double us = ms * 1000 / 50;
    printf("%s operands: %.1f us / launch = %.0f TFLOP/s; one workgroup: %lld shader clocks in %.2f us -> %.2f GHz (MFMA peak at that clock %.0f TFLOP/s)\n",
           zeros ? "zero" : "random", us, 2.0 * M * N * K / us / 1e6, c[0], c[1] * 0.01, c[0] / (c[1] * 10.0), 2500.0 * c[0] / (c[1] * 10.0) / 2.4);
    return 0;
}
