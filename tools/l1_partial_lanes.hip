// Cost of a 16-byte vector load (L1 hits) when only some lanes of the wavefront are active: does the texture path charge per active lane,
// per active group of lanes, or per instruction?   hipcc --offload-arch=gfx950 -O3 tools/l1_partial_lanes.hip -o /tmp/l1pl && /tmp/l1pl
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ buf, int window_elems, int iters, float* out) {
    const float4* base = buf + (size_t)(blockIdx.x % 64) * window_elems;
    const int lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? true : MODE == 1 ? (lane & 15) == 15 : MODE == 2 ? (lane & 3) == 3 : MODE == 3 ? (lane * 37 % 64) < 7 : MODE == 4 ? lane < 16 : (lane & 1) == 0;
    float acc = 0.f;
    int idx = threadIdx.x % window_elems;
    if (on)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = base[idx];
                acc += v.x;
                idx += 256; if (idx >= window_elems) idx -= window_elems;
            }
        }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE>
void run(const char* name, int blocks) {
    const int window_bytes = 8192, we = window_bytes / 16;
    float4* buf; float* out;
    hipMalloc(&buf, (size_t)64 * window_bytes); hipMemset(buf, 0, (size_t)64 * window_bytes); hipMalloc(&out, 4);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, we, 10, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, we, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters * 8;           // wave-level load instructions
    printf("%-44s %6.1f clk per load instruction and CU (2.1 GHz, 256 CUs)\n", name, ms * 1e-3 * 2.1e9 / (instr / 256));
    hipFree(buf); hipFree(out);
}
int main() {
    run<0>("all 64 lanes", 2048);
    run<5>("every second lane (32)", 2048);
    run<4>("lanes 0..15 (16, one row)", 2048);
    run<2>("one lane per group of 4 (16)", 2048);
    run<1>("one lane per row of 16 (4)", 2048);
    run<3>("7 scattered lanes", 2048);
    return 0;
}
