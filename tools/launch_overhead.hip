// Fixed cost of a kernel boundary on one stream (gfx950): back-to-back dependent launches of small kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_overhead.hip -o tools/launch_overhead.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_write(float4* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_copy(const float4* a, float4* b, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}
__global__ void k_read(const float4* a, float4* b, int n) {       // reads, (practically) never writes
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { float4 v = a[i]; if (v.x == 123456.f) b[i] = v; }
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f(i);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f(i);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}
int main() {
    const int reps = 2000;
    float4 *a, *b, *c, *fa, *fb;
    CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20)); CK(hipMalloc(&c, 64 << 20));
    CK(hipExtMallocWithFlags((void**)&fa, 64 << 20, hipDeviceMallocFinegrained)); CK(hipExtMallocWithFlags((void**)&fb, 64 << 20, hipDeviceMallocFinegrained));
    CK(hipMemset(a, 0, 64 << 20)); CK(hipMemset(b, 0, 64 << 20)); CK(hipMemset(c, 0, 64 << 20)); CK(hipMemset(fa, 0, 64 << 20)); CK(hipMemset(fb, 0, 64 << 20));
    for (int grid : {64, 504, 3360})
        printf("empty  grid %5d x 512                         : %6.2f us / launch\n", grid, timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, 0); }, reps));
    for (int kb : {64, 1024, 10240}) {
        const int n = kb * 1024 / 16;
        const int grid = n / 256 < 504 ? (n / 256 > 0 ? n / 256 : 1) : 504;
        printf("--- %d KB, grid %d x 256 / 512\n", kb, grid);
        printf("write (same buffer)                             : %6.2f\n", timeit([&](int) { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n); }, reps));
        printf("read never-written buffer                       : %6.2f\n", timeit([&](int) { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, c, b, n); }, reps));
        printf("copy c -> a (source never rewritten)            : %6.2f\n", timeit([&](int) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, c, a, n); }, reps));
        printf("copy ping-pong a <-> b (read after write)       : %6.2f\n", timeit([&](int i) { if (i & 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, b, a, n); else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, reps));
        printf("copy ping-pong, fine-grained allocations        : %6.2f\n", timeit([&](int i) { if (i & 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, fb, fa, n); else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, fa, fb, n); }, reps));
        printf("write a ; read a (alternating)                  : %6.2f (per pair of launches)\n", 2 * timeit([&](int i) { if (i & 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n); else hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n); }, reps));
    }
    return 0;
}
