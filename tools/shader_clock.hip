// Effective shader clock under a VALU-saturating load (gfx950): s_memtime (shader clock) against s_memrealtime (100 MHz) around a long
// fp32 add chain on every SIMD of the chip.   hipcc --offload-arch=gfx950 -O3 tools/shader_clock.hip -o tools/shader_clock.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_load(float* out, unsigned long long* t, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] + 1.000001f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* t;
    const int grid = 256 * 6;                      // 6 workgroups of 4 wavefronts per CU
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&t, grid * 16);
    for (int iters : {2000, 20000, 200000}) {
        hipLaunchKernelGGL(k_load, dim3(grid), dim3(256), 0, 0, out, t, iters);
        (void)hipDeviceSynchronize();
        static unsigned long long h[2 * 256 * 6];
        (void)hipMemcpy(h, t, grid * 16, hipMemcpyDeviceToHost);
        double c = 0, r = 0;
        for (int i = 0; i < grid; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
        printf("iters %6d: %.0f shader clocks in %.2f us -> %.0f MHz; %.2f clocks per wave64 add per SIMD (6 waves/SIMD)\n", iters, c / grid, r / grid / 100.0,
               c / r * 100.0, (c / grid) / (iters * 8.0 * 6));
    }
    return 0;
}
