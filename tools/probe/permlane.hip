// what v_permlane32_swap / v_permlane16_swap return per lane (gfx950): prints r[0], r[1] for v[lane] = lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned v = threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    u2 q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    o[threadIdx.x * 4 + 0] = r[0]; o[threadIdx.x * 4 + 1] = r[1]; o[threadIdx.x * 4 + 2] = q[0]; o[threadIdx.x * 4 + 3] = q[1];
}
int main() {
    int* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: swap32 (%2d, %2d)  swap16 (%2d, %2d)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
