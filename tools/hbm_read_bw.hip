#include <hip/hip_runtime.h>
#include <stdio.h>
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
// contiguous chunk per block (like a per-k streaming kernel)
__global__ __launch_bounds__(256) void k_read_chunk(const float4* __restrict__ p, size_t n4, size_t chunk4, float* out) {
    const size_t b0 = (size_t)blockIdx.x * chunk4;
    float acc = 0.f;
    for (size_t i = b0 + threadIdx.x; i < b0 + chunk4 && i < n4; i += 256 * 4) {
        float4 v0 = p[i], v1 = (i + 256 < b0 + chunk4) ? p[i + 256] : make_float4(0,0,0,0), v2 = (i + 512 < b0 + chunk4) ? p[i + 512] : make_float4(0,0,0,0), v3 = (i + 768 < b0 + chunk4) ? p[i + 768] : make_float4(0,0,0,0);
        acc += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y + v2.z + v2.w + v3.x + v3.y + v3.z + v3.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)270 << 20; const size_t n4 = bytes / 16;
    float4* d; float* o; hipMalloc(&d, bytes * 2); hipMalloc(&o, 4); hipMemset(d, 0, bytes * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) k_read<4><<<blocks, 256>>>(d + (it & 1) * n4, n4, o);   // alternate buffers: defeat MALL reuse
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read unroll4 grid=%6d : %.1f us per 270MB -> %.2f TB/s\n", blocks, ms * 100, bytes / (ms * 1e-4) / 1e12);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) k_read<8><<<blocks, 256>>>(d + (it & 1) * n4, n4, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("read unroll8 grid=%6d : %.1f us per 270MB -> %.2f TB/s\n", blocks, ms * 100, bytes / (ms * 1e-4) / 1e12);
    }
    for (size_t chunkKB : {64, 128, 512}) {
        const size_t chunk4 = chunkKB * 1024 / 16; const int blocks = (int)((n4 + chunk4 - 1) / chunk4);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) k_read_chunk<<<blocks, 256>>>(d + (it & 1) * n4, n4, chunk4, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read chunk %zuKB blocks=%d : %.1f us -> %.2f TB/s\n", chunkKB, blocks, ms * 100, bytes / (ms * 1e-4) / 1e12);
    }
    return 0;
}
