// mind_common.h -- definitions shared by the two MIND-SSC stencil kernels (mind.hip: tiled, mindmarch.hip: z-marching)
#pragma once
#include "cvx_common.h"

namespace cvx {

// shift pairs in the reference's PRE-permutation channel order (derived by executing convex_adam_utils.py:31-47)
struct MindOffsets {
    int o1[12][3] = {{0,0,-1},{0,-1,0},{0,-1,0},{0,0,1},{0,0,1},{1,0,0},
                     {1,0,0},{1,0,0},{0,1,0},{0,1,0},{0,1,0},{0,1,0}};
    int o2[12][3] = {{-1,0,0},{-1,0,0},{0,0,-1},{-1,0,0},{0,-1,0},{0,0,-1},
                     {0,-1,0},{0,0,1},{-1,0,0},{0,0,-1},{0,0,1},{1,0,0}};
};
// final channel j holds pre-permutation channel PERM[j], PERM = {6,8,1,11,2,10,0,7,9,4,5,3}
// (convex_adam_utils.py:66); the stores use its inverse: destination channel of pre-permutation channel c
__device__ constexpr int MIND_INV[12] = {6, 2, 4, 11, 9, 10, 0, 7, 1, 8, 5, 3};

struct MindStats {
    double m1, m2, m3;     // split grids (see oracle orc_split_make)
    double a1, a2, a3;     // exact partial sums
    float imin, imax;
    float mean_override;   // reference-bits mode (option mind_mean_threads): torch's own mean, see k_torch_sum_* in mind.hip
    int use_override;
};

// z-marching stencil (mindmarch.hip): radius 1, dilation 2, rows of a multiple of 4 voxels, 16-byte aligned pointers
bool mind_march_supported(const float* img, const float* out, int H, int W, int D, int radius, int dilation);
void launch_mind_march(const float* img, int H, int W, int D, MindStats* st, float* out, hipStream_t s);

}  // namespace cvx
