#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(lane, lane, 0x101, 0xf, 0xf, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xf, 0xf, false);
    out[128 + lane] = __builtin_amdgcn_update_dpp(lane, lane, 0x111, 0xf, 0xf, false);
}
int main() {
    int* d; hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; ++r) { for (int i = 0; i < 20; ++i) printf("%d ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
