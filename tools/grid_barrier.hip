// Cost and correctness of a hand-rolled device-wide barrier inside one persistent kernel on gfx950 (8 XCDs, one L2 each):
// every workgroup writes a slab, release-fences, joins an agent-scope counter barrier, acquire-fences and checks the slab another
// workgroup (on another XCD: consecutive workgroup ids go to consecutive XCDs) wrote before the barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o tools/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: every thread fences (release before, acquire after); 1: only thread 0 fences (the workgroup barrier orders the others);
//      2: no fences at all (raw barrier cost; data exchange then needs scoped loads / stores); 3: like 1, acquire by one thread per wave
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    if (MODE == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 2) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (agent scope by default in HIP: buffer_inv sc1)
        }
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    if (MODE == 3 && (threadIdx.x & 63) == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int PER_THREAD, int MODE>
__global__ __launch_bounds__(512) void k_persist(unsigned* ctr, int nbar, float* data, unsigned* errors) {
    const int nthr = gridDim.x * blockDim.x, me = blockIdx.x * blockDim.x + threadIdx.x;
    const int other = ((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x;      // a thread of the next workgroup (next XCD)
    unsigned bad = 0;
    for (int i = 0; i < nbar; ++i) {
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) data[(size_t)k * nthr + me] = (float)(i + k);
        grid_barrier<MODE>(ctr, (unsigned)(i + 1) * gridDim.x);
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) bad += data[(size_t)k * nthr + other] != (float)(i + k);
        grid_barrier<MODE>(ctr + 32, (unsigned)(i + 1) * gridDim.x);                          // (before the slabs are overwritten)
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    unsigned *ctr, *err; float* data;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&data, (size_t)64 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const void* kern[4] = {(const void*)k_persist<8, 0>, (const void*)k_persist<8, 1>, (const void*)k_persist<8, 2>, (const void*)k_persist<8, 3>};
    for (int mode = 0; mode < 4; ++mode) {
        for (int grid : {256, 512}) {
            int nbar = 300;
            CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(err, 0, 4));
            void* args[] = {&ctr, &nbar, &data, &err};
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel(kern[mode], dim3(grid), dim3(512), args, 0, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("mode %d grid %3d x 512, %4d x (write 8 floats/thread, barrier, read, barrier): %7.2f us per barrier pair, %6.2f per barrier, stale reads %u\n", mode, grid, nbar,
                   ms / nbar * 1e3, ms / nbar * 1e3 / 2, h);
        }
    }
    return 0;
}
