// Where does a workgroup of the 256x256 LDS-DMA GEMM spend its life, and what happens on a CU between two workgroups?
// Every workgroup stamps the 100 MHz wall clock at entry (0), after its first stage has landed (1), after the k-loop (2),
// after the epilogue's last store was ISSUED (3) and after the stores were acknowledged (4), plus HW_ID / XCC_ID.
// FF1 shape of the denoiser at B = 64 (M = 32768, N = 4096, K = 512, GEGLU epilogue), random operands.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probe/gemm_timeline.hip -o tools/probe/gemm_timeline
#define RALD_GEMM_STAMPS 1
#include "../../rald_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
namespace rald { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    using namespace rald;
    const int epi = argc > 1 ? atoi(argv[1]) : EPI_GEGLU;
    const int N = argc > 2 ? atoi(argv[2]) : 4096;
    const int M = 32768, K = 512;
    std::vector<unsigned short> h((size_t)M * K), w((size_t)N * K);
    unsigned s = 777;
    auto fill = [&](std::vector<unsigned short>& v) {
        for (auto& x : v) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    };
    fill(h); fill(w);
    bf16 *A, *W, *C; float* bias;
    hipMalloc(&A, h.size() * 2); hipMalloc(&W, w.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    GemmArgs g = gemm_args(A, K, W, K, C, epi == EPI_GEGLU ? N / 2 : N, bias, M, N, K);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) if (gemm_nt(g, epi, 0)) return 1;
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) gemm_nt(g, epi, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nwg = (M / 256) * (N / 256);
    std::vector<long long> st((size_t)8192 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_gemm_stamps), st.size() * 8);
    printf("epi %d N %d: %.1f us / launch = %.0f TFLOP/s (with stamps), %d workgroups\n", epi, N, ms * 1000 / 50, 2.0 * M * N * K / (ms * 1000 / 50) / 1e6, nwg);
    // per-phase medians (units of 10 ns)
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    std::vector<double> d01, d12, d23, d34, d04;
    long long tmin = 1LL << 62, tmax = 0;
    struct Rec { long long t0, t4; int lin; };
    std::map<long long, std::vector<Rec>> per_cu;
    for (int i = 0; i < nwg && i < 8192; ++i) {
        const long long* r = &st[(size_t)i * 8];
        d01.push_back((r[1] - r[0]) * 0.01); d12.push_back((r[2] - r[1]) * 0.01); d23.push_back((r[3] - r[2]) * 0.01);
        d34.push_back((r[4] - r[3]) * 0.01); d04.push_back((r[4] - r[0]) * 0.01);
        tmin = std::min(tmin, r[0]); tmax = std::max(tmax, r[4]);
        // HW_ID gfx9: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        const long long cu = ((r[7] & 0xf) << 16) | ((r[6] >> 8) & 0xff);
        per_cu[cu].push_back({r[0], r[4], i});
    }
    printf("last launch: first entry -> last end %.1f us; distinct (xcc, se/sh/cu) ids %zu\n", (tmax - tmin) * 0.01, per_cu.size());
    printf("median us per workgroup: entry->stage0 landed %.2f | k-loop %.2f | epilogue (stores issued) %.2f | stores acked %.2f | whole %.2f\n",
           med(d01), med(d12), med(d23), med(d34), med(d04));
    std::vector<double> gaps, first, per_cu_n;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.t0 < b.t0; });
        first.push_back((v[0].t0 - tmin) * 0.01);
        per_cu_n.push_back((double)v.size());
        for (size_t j = 1; j < v.size(); ++j) gaps.push_back((v[j].t0 - v[j - 1].t4) * 0.01);
    }
    std::sort(gaps.begin(), gaps.end());
    if (!gaps.empty())
        printf("gap on one CU between a workgroup's last store ack and the next workgroup's entry: median %.2f us, p10 %.2f, p90 %.2f (n = %zu)\n",
               med(gaps), gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], gaps.size());
    printf("workgroups per CU: median %.0f; first entry after launch start: median %.2f us\n", med(per_cu_n), med(first));
    // phase alignment: how many workgroups are in their epilogue at the same time (histogram over the launch in 1-us bins)
    const int bins = (int)((tmax - tmin) / 100) + 1;
    std::vector<int> in_epi(bins, 0), in_loop(bins, 0);
    for (int i = 0; i < nwg && i < 8192; ++i) {
        const long long* r = &st[(size_t)i * 8];
        for (long long t = r[1]; t < r[2]; t += 100) in_loop[(t - tmin) / 100]++;
        for (long long t = r[2]; t < r[4]; t += 100) in_epi[(t - tmin) / 100]++;
    }
    printf("workgroups in k-loop / in epilogue per 1-us bin:\n");
    for (int b = 0; b < bins; ++b) printf("%3d:%3d/%3d%s", b, in_loop[b], in_epi[b], (b % 8 == 7) ? "\n" : "  ");
    printf("\n");
    return 0;
}
