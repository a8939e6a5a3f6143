#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n4; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    float4* d; float* o; hipMalloc(&d, (size_t)1 << 30); hipMalloc(&o, 4); hipMemset(d, 0, (size_t)1 << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512}) {
        const size_t n4 = (mb << 20) / 16;
        for (int w = 0; w < 3; ++w) k_read<<<2048, 256>>>(d, n4, o);
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) k_read<<<2048, 256>>>(d, n4, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("re-read %4zu MB x20: %.1f us per pass -> %.2f TB/s\n", mb, ms * 50, (double)(mb << 20) / (ms * 1e-3 / 20) / 1e12);
    }
    return 0;
}
