// Probe 2: which lane's scale byte applies to which (row, k-block) in v_mfma_scale_f32_16x16x128_f8f6f4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const int* A, const int* B, const unsigned* wa, const unsigned* wb, float* C) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)wa[l], 0, (int)wb[l]);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];
}

int main() {
    int hA[512], hB[512]; unsigned hwa[64], hwb[64]; float hC[256];
    int *dA, *dB; unsigned *dwa, *dwb; float* dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dwa, 256); hipMalloc(&dwb, 256); hipMalloc(&dC, 1024);
    for (int which = 0; which < 2; ++which) {          // 0: probe A scales, 1: probe B scales
        printf("---- %s scales: lane -> affected (%s, data lane group)\n", which ? "B" : "A", which ? "col" : "row");
        for (int ls = 0; ls < 64; ++ls) {
            printf("lane %2d:", ls);
            for (int gb = 0; gb < 4; ++gb) {
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 8; ++i) {
                        const int one = 0x38383838;
                        const int in_blk = (l >> 4) == gb ? one : 0;
                        hA[l * 8 + i] = which == 0 ? in_blk : one;
                        hB[l * 8 + i] = which == 1 ? in_blk : one;
                    }
                for (int l = 0; l < 64; ++l) { hwa[l] = 0x7F7F7F7Fu; hwb[l] = 0x7F7F7F7Fu; }
                (which ? hwb : hwa)[ls] = 0x7F7F7F80u;      // byte 0 = 128 -> x2
                hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
                hipMemcpy(dwa, hwa, 256, hipMemcpyHostToDevice); hipMemcpy(dwb, hwb, 256, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dwa, dwb, dC);
                hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 16; ++i) {
                    const float v = which ? hC[0 * 16 + i] : hC[i * 16 + 0];   // B scale -> look along a row; A scale -> along a column
                    if (v != 32.f) printf(" (%d,g%d)=%g", i, gb, v);
                }
            }
            printf("\n");
        }
    }
    return 0;
}
