// mind.hip -- MIND-SSC descriptors (reference: src/convexAdam/convex_adam_utils.py:24-68).
//
// out_c(x) = exp( -(D_c(x) - min_c D_c(x)) / clamp(mean_c(D - min), 0.001*mu, 1000*mu) )
// D_c(x)   = box_{(2r+1)^3}[ (I(P + o1_c*d) - I(P + o2_c*d))^2 ](x)      replicate borders twice
// mu       = mean over the volume of mean_c(D - min)
//
// Two launches (the global mean mu is a grid-wide dependency):
//   k_mind        : tiled stencil -- the 12 patch-SSDs D_c(x) -> out (raw), per-voxel variance -> order-independent
//                   exact sum (three power-of-two split grids, double atomics)   [reads V*4, writes 12*V*4 B]
//   k_mind_finish : streaming, in place -- min, mean, clamp, exp per voxel        [reads + writes 12*V*4 B]
// (recomputing the stencil in the second pass instead costs 2.2 x the time of streaming the 12 channels once)
// Tile: 4 x 8 x 64 voxels (H x W x D) per 512-thread workgroup, image tile with halo r+d staged in
// LDS once, squared-difference tile per channel double-buffered in LDS, each thread owns 4
// consecutive D-voxels and keeps 12 x 4 results in registers.  Roofline: HBM (385 MB per image when
// the full-resolution descriptor is materialised); the 27-tap raster-order sums (ATen avg_pool3d
// order, one exact division) make it VALU/LDS-bound in practice -- see DESIGN.md.
#include "cvx_common.h"

namespace cvx {

// shift pairs in the reference's PRE-permutation channel order (derived by executing :31-47)
struct MindOffsets {
    int o1[12][3] = {{0,0,-1},{0,-1,0},{0,-1,0},{0,0,1},{0,0,1},{1,0,0},
                     {1,0,0},{1,0,0},{0,1,0},{0,1,0},{0,1,0},{0,1,0}};
    int o2[12][3] = {{-1,0,0},{-1,0,0},{0,0,-1},{-1,0,0},{0,-1,0},{0,0,-1},
                     {0,-1,0},{0,0,1},{-1,0,0},{0,0,-1},{0,0,1},{1,0,0}};
};
// final channel j holds pre-permutation channel PERM[j], PERM = {6,8,1,11,2,10,0,7,9,4,5,3}
// (convex_adam_utils.py:66); the store below uses its inverse.

// destination channel of pre-permutation channel c (inverse of PERM)
__device__ constexpr int MIND_INV[12] = {6, 2, 4, 11, 9, 10, 0, 7, 1, 8, 5, 3};

struct MindStats {
    double m1, m2, m3;     // split grids (see oracle orc_split_make)
    double a1, a2, a3;     // exact partial sums
    float lo, hi, mean;    // clamp bounds
    float imin, imax;
};

constexpr int TZ = 4, TY = 8, TX = 64, RUN = 4, NT = 512;

// ---- min / max of the image (bound for the exact accumulation) -----------------------------------
__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ img, size_t V, float* part) {
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x) {
        const float v = img[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o));
        mx = fmaxf(mx, __shfl_down(mx, o));
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        part[2 * blockIdx.x] = mn;
        part[2 * blockIdx.x + 1] = mx;
    }
}
__global__ __launch_bounds__(256) void k_mind_stats_init(const float* part, int nparts, double count, MindStats* st) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_down(mn, o)); mx = fmaxf(mx, __shfl_down(mx, o)); }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
    const double range = (double)mx - (double)mn;
    double bound = range * range;
    if (!(bound > 0.0)) bound = 1e-300;
    int e;
    (void)frexp(bound * count, &e);
    const double top = ldexp(1.0, e + 1);
    st->m1 = 1.5 * top;
    st->m2 = st->m1 * 0x1p-30;
    st->m3 = st->m2 * 0x1p-30;
    st->a1 = st->a2 = st->a3 = 0.0;
    st->imin = mn;
    st->imax = mx;
}
__global__ void k_mind_stats_finish(MindStats* st, double count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float gm = (float)((st->a1 + (st->a2 + st->a3)) / count);
    st->mean = gm;
    st->lo = (float)((double)gm * 0.001);      // python: mind_var.mean().item()*0.001   (:61)
    st->hi = (float)((double)gm * 1000.0);
}

// ---- the tiled stencil ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(NT) void k_mind(const float* __restrict__ img, int H, int W, int D, int dil, int nbuf,
                                              MindStats* __restrict__ st, float* __restrict__ out) {
    constexpr int K = 2 * R + 1;
    constexpr int SZ = TZ + 2 * R, SY = TY + 2 * R, SX = TX + 2 * R;
    constexpr int SXP = (SX + 3) / 4 * 4 + 2;        // row pitch = 2 (mod 4) floats: the 8-byte reads of lanes 16 B apart in
                                                     // adjacent rows land on disjoint banks
    const int halo = R + dil;
    const int IZ = TZ + 2 * halo, IY = TY + 2 * halo, IX = TX + 2 * halo;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* simg = smem;
    float* ssq = smem + ((IZ * IY * IX + 3) / 4) * 4;   // two buffers of SZ*SY*SXP

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;

    // image tile with replicate (clamp) addressing: simg[q] = I(clamp(origin - halo + q))
    for (int i = tid; i < IZ * IY * IX; i += NT) {
        const int ix = i % IX, iy = (i / IX) % IY, iz = i / (IX * IY);
        const int gz = clampi(z0 - halo + iz, 0, H - 1), gy = clampi(y0 - halo + iy, 0, W - 1),
                  gx = clampi(x0 - halo + ix, 0, D - 1);
        simg[i] = img[((size_t)gz * W + gy) * D + gx];
    }
    __syncthreads();

    const int trun = tid % (TX / RUN), ty = (tid / (TX / RUN)) % TY, tz = tid / ((TX / RUN) * TY);
    const int tx0 = trun * RUN;
    float res[12][RUN];

    // squared-difference stage: every thread owns NSQ fixed positions of the (tile + R) region; their clamped
    // source index in the image tile and their destination index are the same for all 12 channels
    constexpr int NSQ = (SZ * SY * SX + NT - 1) / NT;
    int sq_src[NSQ], sq_dst[NSQ];
#pragma unroll
    for (int e = 0; e < NSQ; ++e) {
        const int i = tid + e * NT;
        const int sx = i % SX, sy = (i / SX) % SY, sz = i / (SX * SY);
        // the box sees the clamped POSITION (rpad2), the shifts clamp again (rpad1): I(clamp(clamp(P)+o*d))
        const int pz = clampi(z0 - R + sz, 0, H - 1), py = clampi(y0 - R + sy, 0, W - 1), px = clampi(x0 - R + sx, 0, D - 1);
        sq_src[e] = ((pz - (z0 - halo)) * IY + (py - (y0 - halo))) * IX + (px - (x0 - halo));
        sq_dst[e] = (i < SZ * SY * SX) ? (sz * SY + sy) * SXP + sx : -1;
        if (i >= SZ * SY * SX) sq_src[e] = (halo * IY + halo) * IX + halo;      // any in-range position; never stored
    }

    constexpr MindOffsets MO{};
#pragma unroll
    for (int c = 0; c < 12; ++c) {          // fully unrolled: res[c][] must stay in registers
        float* sq = ssq + (c & (nbuf - 1)) * (SZ * SY * SXP);
        const int off1 = ((MO.o1[c][0] * IY + MO.o1[c][1]) * IX + MO.o1[c][2]) * dil;
        const int off2 = ((MO.o2[c][0] * IY + MO.o2[c][1]) * IX + MO.o2[c][2]) * dil;
#pragma unroll
        for (int e = 0; e < NSQ; ++e) {
            const float df = simg[sq_src[e] + off1] - simg[sq_src[e] + off2];
            if (sq_dst[e] >= 0) sq[sq_dst[e]] = df * df;
        }
        __syncthreads();
        // raster-order box sums (z slowest, x fastest) for 4 adjacent outputs from aligned 8-byte LDS reads,
        // one exact division by K^3
        float s[RUN];
#pragma unroll
        for (int j = 0; j < RUN; ++j) s[j] = 0.0f;
#pragma unroll 1
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
                const float* row = sq + ((tz + a) * SY + (ty + b)) * SXP + tx0;
                float rv[RUN + 2 * R];
#pragma unroll
                for (int j = 0; j < (RUN + 2 * R) / 2; ++j) {
                    const f32x2 q = lds_load2(row + 2 * j);
                    rv[2 * j] = q.x; rv[2 * j + 1] = q.y;
                }
#pragma unroll
                for (int j = 0; j < RUN; ++j)
#pragma unroll
                    for (int cc = 0; cc < K; ++cc) s[j] += rv[j + cc];
            }
#pragma unroll
        for (int j = 0; j < RUN; ++j) res[c][j] = div_exact<K * K * K>(s[j]);
        // double-buffered sq (nbuf = 2): the next channel writes the other buffer, one barrier per channel;
        // large radius/dilation tiles only fit one buffer and need a second barrier
        if (nbuf == 1) __syncthreads();
    }

    const int gz = z0 + tz, gy = y0 + ty;     // threads of an overhanging tile still join the reduction
    const size_t V = (size_t)H * W * D;
    const size_t tail_from = (V / 32) * 32;          // ATen outer-sum tail columns (interleaved order)
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double m1 = st->m1, m2 = st->m2, m3 = st->m3;

#pragma unroll
    for (int j = 0; j < RUN; ++j) {
        const int gx = x0 + tx0 + j;
        const bool valid = (gz < H) && (gy < W) && (gx < D);
        float mc[12];
        float mn = res[0][j];
#pragma unroll
        for (int c = 1; c < 12; ++c) mn = fminf(mn, res[c][j]);   // no NaN handling needed: NaN stays NaN below
#pragma unroll
        for (int c = 0; c < 12; ++c) mc[c] = res[c][j] - mn;
        const size_t lin = ((size_t)gz * W + gy) * D + gx;
        const float sum = (lin >= tail_from) ? outer_sum_ilp<12>(mc) : cascade_seq<12>(mc);
        const float var = fdiv(sum, 12.0f);
        if (valid) {
            const double v = (double)var;
            const double q1 = (v + m1) - m1, r1 = v - q1;
            const double q2 = (r1 + m2) - m2, r2 = r1 - q2;
            const double q3 = (r2 + m3) - m3;
            a1 += q1; a2 += q2; a3 += q3;
        }
    }

    // every partial sum is exactly representable -> any reduction order gives the same bits
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o); a3 += __shfl_down(a3, o);
    }
    __shared__ double red[3][NT / 64];
    if ((tid & 63) == 0) { red[0][tid >> 6] = a1; red[1][tid >> 6] = a2; red[2][tid >> 6] = a3; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < NT / 64; ++i) { a1 += red[0][i]; a2 += red[1][i]; a3 += red[2][i]; }
        atomicAdd(&st->a1, a1); atomicAdd(&st->a2, a2); atomicAdd(&st->a3, a3);
    }
    // raw patch-SSDs -> out, already in the final channel order (normalised in place by k_mind_finish)
    if (gz < H && gy < W) {
        const int gx0 = x0 + tx0;
        const bool vec = ((D & 3) == 0) && (gx0 + RUN <= D);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            float* dst = out + (size_t)MIND_INV[c] * V + ((size_t)gz * W + gy) * D + gx0;
            if (vec) {
                *reinterpret_cast<float4*>(dst) = make_float4(res[c][0], res[c][1], res[c][2], res[c][3]);
            } else {
#pragma unroll
                for (int j = 0; j < RUN; ++j)
                    if (gx0 + j < D) dst[j] = res[c][j];
            }
        }
    }
}

// out_c(x) = exp(-(D_c - min_c D) / clamp(mean_c(D - min), lo, hi)) in place; NV voxels per thread (4 = 16-byte access).
// The channel mean runs over the reference's PRE-permutation channel order (the permutation is applied last, :66).
template <int NV>
__global__ __launch_bounds__(256) void k_mind_finish(float* __restrict__ out, size_t V, const MindStats* __restrict__ st) {
    const size_t x = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * NV;
    if (x >= V) return;
    const float lo = st->lo, hi = st->hi;
    const size_t tail_from = (V / 32) * 32;
    float r[12][NV];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const float* src = out + (size_t)MIND_INV[c] * V + x;
        if (NV == 4) {
            const float4 q = *reinterpret_cast<const float4*>(src);
            r[c][0] = q.x; r[c][1 % NV] = q.y; r[c][2 % NV] = q.z; r[c][3 % NV] = q.w;
        } else r[c][0] = src[0];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float mc[12];
        float mn = r[0][j];
#pragma unroll
        for (int c = 1; c < 12; ++c) mn = fminf(mn, r[c][j]);
#pragma unroll
        for (int c = 0; c < 12; ++c) mc[c] = r[c][j] - mn;
        const float sum = (x + j >= tail_from) ? outer_sum_ilp<12>(mc) : cascade_seq<12>(mc);
        float var = fdiv(sum, 12.0f);
        var = var < lo ? lo : var;
        var = var > hi ? hi : var;
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c][j] = cvx_expf(-fdiv(mc[c], var));
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        float* dst = out + (size_t)MIND_INV[c] * V + x;
        if (NV == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[c][0], r[c][1 % NV], r[c][2 % NV], r[c][3 % NV]);
        else dst[0] = r[c][0];
    }
}

static size_t mind_lds_bytes(int R, int dil, int nbuf) {
    const int halo = R + dil;
    const int IZ = TZ + 2 * halo, IY = TY + 2 * halo, IX = TX + 2 * halo;
    const int SZ = TZ + 2 * R, SY = TY + 2 * R, SX = TX + 2 * R, SXP = (SX + 3) / 4 * 4 + 2;
    return sizeof(float) * ((size_t)((IZ * IY * IX + 3) / 4) * 4 + (size_t)nbuf * SZ * SY * SXP);
}

template <int R>
static int mind_launch_r(const float* img, int H, int W, int D, int dil, MindStats* st, float* out, hipStream_t s) {
    const dim3 grid(cdiv(D, TX), cdiv(W, TY), cdiv(H, TZ));
    const int nbuf = mind_lds_bytes(R, dil, 2) <= 160 * 1024 ? 2 : 1;
    const size_t lds = mind_lds_bytes(R, dil, nbuf);
    static size_t granted0 = 0;
    ensure_dynamic_lds(&k_mind<R>, lds, granted0);
    const double count = (double)H * W * D;
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL((k_mind<R>), grid, dim3(NT), lds, s, img, H, W, D, dil, nbuf, st, out);
    hipLaunchKernelGGL(k_mind_stats_finish, dim3(1), dim3(1), 0, s, st, count);
    if (V % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(k_mind_finish<4>, dim3((unsigned)cdiv64((int64_t)(V / 4), 256)), dim3(256), 0, s, out, V, st);
    else
        hipLaunchKernelGGL(k_mind_finish<1>, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, s, out, V, st);
    return check_last("mindssc");
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_mindssc_workspace_bytes(int H, int W, int D, int radius, int dilation) {
    (void)H; (void)W; (void)D; (void)radius; (void)dilation;
    return 256 + 2 * 1024 * sizeof(float) + 256 + sizeof(MindStats) + 256;
}

extern "C" int cvx_mindssc_f32(const float* img, int H, int W, int D, int radius, int dilation, float* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(img && out && workspace, "cvx_mindssc_f32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_mindssc_f32: bad extent %dx%dx%d", H, W, D);
    CVX_REQUIRE(radius >= 1 && radius <= 3, "cvx_mindssc_f32: radius %d not in 1..3", radius);
    CVX_REQUIRE(dilation >= 1 && dilation <= 4, "cvx_mindssc_f32: dilation %d not in 1..4", dilation);
    if (workspace_bytes < cvx_mindssc_workspace_bytes(H, W, D, radius, dilation))
        return fail(CVX_ERR_WORKSPACE, "cvx_mindssc_f32: workspace too small");
    if (mind_lds_bytes(radius, dilation, 1) > 160 * 1024)
        return fail(CVX_ERR_UNSUPPORTED, "cvx_mindssc_f32: radius %d dilation %d exceeds the LDS tile", radius, dilation);
    hipStream_t s = as_stream(stream);
    Carver cv(workspace, workspace_bytes);
    float* part = cv.take<float>(2 * 1024);
    MindStats* st = cv.take<MindStats>(1);
    const size_t V = (size_t)H * W * D;
    const int nb = (int)(V / 4096 + 1 < 1024 ? V / 4096 + 1 : 1024);
    hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(256), 0, s, img, V, part);
    hipLaunchKernelGGL(k_mind_stats_init, dim3(1), dim3(256), 0, s, part, nb, (double)V, st);
    switch (radius) {
        case 1: return mind_launch_r<1>(img, H, W, D, dilation, st, out, s);
        case 2: return mind_launch_r<2>(img, H, W, D, dilation, st, out, s);
        default: return mind_launch_r<3>(img, H, W, D, dilation, st, out, s);
    }
}
