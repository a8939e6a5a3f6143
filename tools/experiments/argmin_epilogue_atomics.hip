// "measure, do not argue, a per-workgroup min epilogue in k_corr_fused (one atomicMin key per voxel and item)
// against the 2 x 53 us k_argmin4".  This is the epilogue's memory side ALONE: 507 workgroups (the fused kernel's work items, two per
// CU) walk the 26 planes of the coarse grid and issue one 64-bit atomicMin per voxel and plane to the shared key array (30 784 keys,
// 246 KB) -- 15.6 M atomics per direction, the same addresses from every workgroup, plane by plane as the box-2 stage would emit them.
// No SSD arithmetic, no LDS, no cost-volume stores: a LOWER bound for what the epilogue adds to the kernel.
//   hipcc --offload-arch=gfx950 -O3 argmin_epilogue_atomics.hip -o /tmp/aea && /tmp/aea
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(320) void k_epilogue(unsigned long long* __restrict__ keys, int planes, int per_plane, int spread) {
    // thread = one quad of a 32 x 37 plane (320 quads, 296 used); values differ per item so that some atomics win and most lose
    const int q = threadIdx.x;
    if (4 * q >= per_plane) return;
    for (int z = 0; z < planes; ++z) {
        // (spread: items start at different planes, as the two dispatch rounds and the per-item skew would make them)
        const int zz = spread ? (z + (int)blockIdx.x) % planes : z;
        unsigned long long* kp = keys + (size_t)zz * per_plane + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < per_plane) {
                const unsigned v = 0x3f800000u + (((unsigned)blockIdx.x * 2654435761u) ^ (unsigned)(zz * per_plane + 4 * q + j) * 40503u) % 100000u;
                atomicMin(kp + j, ((unsigned long long)v << 32) | blockIdx.x);
            }
    }
}
// the same traffic with one atomic per voxel and WORKGROUP-PAIR merged in LDS first is not possible: a work item owns its displacements
// for ALL voxels, there is nothing to merge inside an item.  What CAN shrink it: one key per voxel and (dH, dW) pair = 169 x 30 784
// = 5.2 M atomics if the three D-shift groups of a pair shared a workgroup (they do not: 15 wavefronts per group).
int main() {
    const int planes = 26, per_plane = 32 * 37, items = 507;
    unsigned long long* keys;
    hipMalloc(&keys, sizeof(unsigned long long) * planes * per_plane);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int spread = 0; spread < 2; ++spread)
        for (int n : {items, 169}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(keys, 0xff, sizeof(unsigned long long) * planes * per_plane);
                hipEventRecord(a);
                hipLaunchKernelGGL(k_epilogue, dim3(n), dim3(320), 0, 0, keys, planes, per_plane, spread);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("%3d items x %d voxels = %.1f M 64-bit atomicMin, %s: %.1f us  (k_argmin4 re-reads the 270 MB volume in 45-53 us)\n", n, planes * per_plane,
                   (double)n * planes * per_plane / 1e6, spread ? "items out of phase" : "items in phase", best * 1e3);
        }
    return 0;
}
