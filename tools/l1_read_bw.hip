// L1-hit read bandwidth of one CU / the chip for 16-byte and 4-byte loads (wave64): every wavefront re-reads a small window that stays in
// its CU's 32 KB vector L1.   hipcc --offload-arch=gfx950 -O3 tools/l1_read_bw.hip -o tools/l1_read_bw.bin && tools/l1_read_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ __launch_bounds__(256) void k(const T* __restrict__ buf, int window_elems, int iters, float* out) {
    const T* base = buf + (size_t)(blockIdx.x % 64) * window_elems;
    float acc = 0.f;
    int idx = threadIdx.x % window_elems;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const T v = base[idx];
            acc += reinterpret_cast<const float*>(&v)[0];
            idx += 256; if (idx >= window_elems) idx -= window_elems;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <typename T>
void run(const char* name, int window_bytes, int blocks) {
    const int we = window_bytes / sizeof(T);
    T* buf; float* out;
    hipMalloc(&buf, (size_t)64 * window_bytes); hipMemset(buf, 0, (size_t)64 * window_bytes); hipMalloc(&out, 4);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, buf, we, 10, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, buf, we, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 256 * iters * 8 * sizeof(T);
    printf("%-10s window %5d B, %5d blocks: %.2f TB/s chip = %.1f B/clk/CU at 2.1 GHz over 256 CUs\n", name, window_bytes, blocks, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 256 / 2.1e9);
    hipFree(buf); hipFree(out);
}
int main() {
    for (int blocks : {256 * 4, 256 * 8}) {
        run<float4>("dwordx4", 8192, blocks);
        run<float4>("dwordx4", 16384, blocks);
        run<float2>("dwordx2", 8192, blocks);
        run<float>("dword", 4096, blocks);
    }
    return 0;
}
