// Probe: operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (gfx950).
// Measured (mfma_fp8_scale.hip): lane l (r = l&15, g = l>>4) holds A[r][16g..16g+15] in VGPRs 0-3 and
// A[r][64+16g..64+16g+15] in VGPRs 4-7, B likewise (B[col][k]); byte[opsel] of lane l's scale operand is
// the e8m0 scale (2^(s-127)) of row r, K-block g = K 32g..32g+31; C/D: col = l&15, row = 4*(l>>4) + reg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C, int opsel) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    // measured layout: bytes 0-15 of the lane = K 16g..16g+15, bytes 16-31 = K 64+16g..64+16g+15;
    // the scale of lane (r, g) applies to K-block g = K 32g..32g+31 of row r
    const int* pa = reinterpret_cast<const int*>(A + r * 128 + g * 16);
    const int* pb = reinterpret_cast<const int*>(B + r * 128 + g * 16);
    for (int i = 0; i < 4; ++i) { a[i] = pa[i]; b[i] = pb[i]; a[4 + i] = pa[16 + i]; b[4 + i] = pb[16 + i]; }
    // scale bytes: put the lane's scale into byte `opsel` of the dword, garbage (0x55) elsewhere
    const unsigned fill = 0x55555555u;
    unsigned wa = (fill & ~(0xFFu << (8 * opsel))) | ((unsigned)sa[r * 4 + g] << (8 * opsel));
    unsigned wb = (fill & ~(0xFFu << (8 * opsel))) | ((unsigned)sb[r * 4 + g] << (8 * opsel));
    v4f c = {0.f, 0.f, 0.f, 0.f};
    switch (opsel) {
        case 0: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)wa, 0, (int)wb); break;
        case 1: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, (int)wa, 1, (int)wb); break;
        case 2: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, (int)wa, 2, (int)wb); break;
        default: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, (int)wa, 3, (int)wb); break;
    }
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];
}

static uint8_t enc(int v) {   // small non-negative / negative ints exactly representable in e4m3fn
    static const uint8_t t[] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0 1 2 3 4
    return v >= 0 ? t[v] : (uint8_t)(t[-v] | 0x80);
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128], hsa[64], hsb[64];
    int iA[16][128], iB[16][128];
    srand(7);
    int bad_total = 0;
    for (int trial = 0; trial < 8; ++trial) {
        const int opsel = trial & 3;
        for (int r = 0; r < 16; ++r)
            for (int k = 0; k < 128; ++k) {
                iA[r][k] = rand() % 9 - 4; iB[r][k] = rand() % 9 - 4;
                hA[r * 128 + k] = enc(iA[r][k]); hB[r * 128 + k] = enc(iB[r][k]);
            }
        for (int i = 0; i < 64; ++i) { hsa[i] = trial < 4 ? 127 : 124 + rand() % 7; hsb[i] = trial < 4 ? 127 : 124 + rand() % 7; }
        uint8_t *dA, *dB, *dsa, *dsb; float* dC;
        hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 1024);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, opsel);
        float hC[256];
        hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) s += iA[i][32 * g + k] * iB[j][32 * g + k];
                    ref += s * std::ldexp(1.0, hsa[i * 4 + g] - 127) * std::ldexp(1.0, hsb[j * 4 + g] - 127);
                }
                if (std::fabs(hC[i * 16 + j] - ref) > 1e-3 * (1 + std::fabs(ref))) { if (bad < 4) printf("  trial %d [%d][%d] got %g want %g\n", trial, i, j, hC[i * 16 + j], ref); ++bad; }
            }
        printf("trial %d opsel %d scales %s: %d / 256 mismatches\n", trial, opsel, trial < 4 ? "unit" : "random", bad);
        bad_total += bad;
        hipFree(dA); hipFree(dB); hipFree(dsa); hipFree(dsb); hipFree(dC);
    }
    printf(bad_total ? "PROBE FAILED\n" : "PROBE OK: assumed layout confirmed\n");
    return bad_total ? 1 : 0;
}
