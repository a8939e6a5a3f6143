// Can the second half of ONE grid wait for the first half?  (the adjoint + update tiles of channel c followed, in the same
// launch, by the forward-march chunks of channel c.)  Producers (low block indices) copy a slice of memory and bump a counter; consumers
// (high block indices) spin -- bounded -- until the counter is complete, then run a dependent chain that uses little bandwidth.
// Reported: time of the fused launch against the two kernels back to back, consumers that timed out (in-order dispatch of a grid is not
// a documented guarantee), and how long consumers waited.     hipcc --offload-arch=gfx950 -O3 inlaunch_dependency.hip -o /tmp/ild && /tmp/ild
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void producer(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, int blk, int nblk, unsigned* done) {
    const size_t per = (n4 + nblk - 1) / nblk, lo = (size_t)blk * per, hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { float4 v = src[i]; v.x += 1.0f; dst[i] = v; }
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); atomicAdd(done, 1u); }
}
__device__ __forceinline__ void consumer(const float4* __restrict__ dst, float* __restrict__ out, int blk, int steps) {
    // a dependent chain: every step reads one element written by the producers, then spins a little (the march's barrier + adds)
    float acc = 0.f;
    size_t idx = (size_t)blk * 977 + threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        acc += dst[idx].x;
        idx = (idx * 31 + 7) & ((1u << 20) - 1);
        for (int k = 0; k < 60; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-7f);
        __syncthreads();
    }
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_fused(const float4* src, float4* dst, size_t n4, int nprod, int steps, unsigned* done, unsigned* stats, float* out, int mode) {
    if ((int)blockIdx.x < nprod) { producer(src, dst, n4, blockIdx.x, nprod, done); return; }
    __shared__ unsigned ok;
    if (threadIdx.x == 0 && mode == 0) ok = 1;
    if (threadIdx.x == 0 && mode != 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nprod && spins < 20000000u) { ++spins; if (mode == 2) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); } else __builtin_amdgcn_s_sleep(20); }
        ok = spins < 20000000u;
        if (!ok) atomicAdd(&stats[0], 1u);
        atomicAdd(&stats[1], spins > 0 ? 1u : 0u);
        atomicMax(&stats[2], spins);
    }
    __syncthreads();
    __threadfence();
    consumer(dst, out, blockIdx.x - nprod, steps);
}
__global__ __launch_bounds__(512) void k_prod(const float4* src, float4* dst, size_t n4, int nprod, unsigned* done) { producer(src, dst, n4, blockIdx.x, nprod, done); }
__global__ __launch_bounds__(512) void k_cons(const float4* dst, float* out, int steps) { consumer(dst, out, blockIdx.x, steps); }

int main() {
    const size_t n4 = (size_t)40 << 20 >> 4 << 0;             // 40 MB read + 40 MB written: ~15 us
    float4 *src, *dst; unsigned *done, *stats; float* out;
    hipMalloc(&src, n4 * 16); hipMalloc(&dst, n4 * 16); hipMalloc(&done, 4); hipMalloc(&stats, 16); hipMalloc(&out, 4);
    hipMemset(src, 0, n4 * 16); hipMemset(dst, 0, n4 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nprod : {640, 1024}) for (int ncons : {504}) for (int steps : {20}) for (int mode : {0, 1, 2}) {
        float ms_seq = 0, ms_fused = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(done, 0, 4);
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_prod, dim3(nprod), dim3(512), 0, 0, src, dst, n4, nprod, done); hipLaunchKernelGGL(k_cons, dim3(ncons), dim3(512), 0, 0, dst, out, steps); }
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms_seq, a, b);
            hipMemset(stats, 0, 16);
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) { hipMemsetAsync(done, 0, 4, 0); hipLaunchKernelGGL(k_fused, dim3(nprod + ncons), dim3(512), 0, 0, src, dst, n4, nprod, steps, done, stats, out, mode); }
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms_fused, a, b);
        }
        unsigned st[4]; hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost);
        printf("mode %d (0 no wait, 1 poll, 2 poll with long sleep) producers %4d consumers %3d steps %2d: two kernels %.1f us, one grid %.1f us (incl. a 4-byte memset); consumers timed out %u, had to wait %u of %d, longest wait %u polls\n",
               mode, nprod, ncons, steps, ms_seq * 50, ms_fused * 50, st[0], st[1], 20 * ncons, st[2]);
    }
    return 0;
}
