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
    unsigned n_repair;     // single-pass pooled path: blocks whose pooled cells k_mind_repair recomputed
};

// z-marching stencil (mindmarch.hip): radius 1, dilation 2, rows of a multiple of 4 voxels, 16-byte aligned pointers
bool mind_march_supported(const float* img, const float* out, int H, int W, int D, int radius, int dilation);
// Where the stencil pass puts the raw patch SSDs.  T == 0: planar [12][H][W][D].  T > 0 (pipeline, mind.hip::launch_mind_pooled): blocked by the
// tiles of k_mind_finish_pool -- [tile (z, y, x)][12][T][T][24] -- so that a tile is 12 T^2 24 contiguous floats: the second pass then reads whole
// 128-byte lines in order instead of 96-byte row pieces shared with the neighbouring tile (it runs at the depth of the L1 miss queue: half as many
// requests for the same bytes)
struct MindRawLayout {
    int T, ntx, nty;                 // tile edge (z, y), tiles along x (of 24 voxels) and y
    size_t tile_floats, chan_floats; // 12 T T 24 and T T 24 (planar: unused / H W D)
};
void launch_mind_march(const float* img, int H, int W, int D, MindStats* st, float* out, hipStream_t s, MindRawLayout lay = MindRawLayout{0, 0, 0, 0, 0});
// single-pass pooled descriptor (mindmarch.hip): stencil + unclamped normalisation + both poolings, block statistics for k_mind_repair (mind.hip)
bool mind_single_supported(int ga, int gb);
void launch_mind_march_pool(const float* img, int H, int W, int D, int ga, float* out1, int gb, void* out2, int records, MindStats* st, unsigned* blk, hipStream_t s);

}  // namespace cvx
