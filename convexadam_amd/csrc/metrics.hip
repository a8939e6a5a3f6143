// metrics.hip -- evaluation operators around the hot path: what the self-configuring sweep
// computes for every (setting, pair) after a registration, on the device, so that only a handful of scalars returns
// to the host.
//   k_jacobian_det      hyper_util:86-108   central differences (Conv3d taps -0.5, 0, 0.5, zero pad) + identity,
//                                           3x3 determinant in the reference's operation order, cropped by 2 voxels
//   k_jacobian_stats    convex_run_withconfig.py:148-150   sum / sum of squares of log(clamp(det + 3)) and the number
//                                           of negative determinants (float64 accumulation; torch's log is not restated)
//   k_warp_nearest      convex_run_withconfig.py:141       F.grid_sample(seg, grid0 + disp/scale, mode='nearest')
//   k_label_overlap     hyper_util:53-60    per-label |A|, |B|, |A and B| (exact integer counts behind dice_coeff)
//   k_map_linear_f64    apply_convex.py:13-24              scipy.ndimage.map_coordinates(order=1, mode='constant')
// All HBM-bound single passes over 3-12 bytes per voxel; none is on the timed path of bench.py.
#include <math.h>

#include "cvx_common.h"

namespace cvx {

__device__ __forceinline__ float ctr(const float* __restrict__ p, int i, int n, size_t stride) {
    // 0.5*x[i+1] + (-0.5)*x[i-1] with zero padding; powers of two: exact in any order
    const float nx = i + 1 < n ? p[stride] : 0.0f, pv = i > 0 ? p[-(ptrdiff_t)stride] : 0.0f;
    return 0.5f * nx + -0.5f * pv;
}

__global__ __launch_bounds__(256) void k_jacobian_det(const float* __restrict__ flow, int H, int W, int D, float sH, float sW,
                                                      float sD, int convert1, float* __restrict__ out) {
    const int Ho = H - 4, Wo = W - 4, Do = D - 4;
    const size_t n = (size_t)Ho * Wo * Do;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % Do) + 2, y = (int)((i / Do) % Wo) + 2, z = (int)(i / ((size_t)Do * Wo)) + 2;
    const size_t V = (size_t)H * W * D, p = ((size_t)z * W + y) * D + x;
    const float sc[3] = {sH, sW, sD};
    float J[3][3];                                   // J[i][c] = d pix_c / d axis_i
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* f = flow + (size_t)c * V + p;
        float nb[6] = {z + 1 < H ? f[(size_t)W * D] : 0.0f, z > 0 ? f[-(ptrdiff_t)((size_t)W * D)] : 0.0f,
                       y + 1 < W ? f[D] : 0.0f, y > 0 ? f[-D] : 0.0f, x + 1 < D ? f[1] : 0.0f, x > 0 ? f[-1] : 0.0f};
        if (convert1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) nb[k] = nb[k] * sc[c];                 // dense_flow * (size-1)/2
        }
        J[0][c] = 0.5f * nb[0] + -0.5f * nb[1];
        J[1][c] = 0.5f * nb[2] + -0.5f * nb[3];
        J[2][c] = 0.5f * nb[4] + -0.5f * nb[5];
    }
    J[0][0] += 1.0f; J[1][1] += 1.0f; J[2][2] += 1.0f;
    const float t0 = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]);
    const float t1 = J[1][0] * (J[0][1] * J[2][2] - J[0][2] * J[2][1]);
    const float t2 = J[2][0] * (J[0][1] * J[1][2] - J[0][2] * J[1][1]);
    out[i] = (t0 - t1) + t2;
}

__device__ __forceinline__ double jac_log(float j) {
    float a = j + 3.0f;
    a = a < 0.000000001f ? 0.000000001f : a;
    a = a > 1000000000.0f ? 1000000000.0f : a;
    return log((double)a);
}
// sums of (l - l0) and (l - l0)^2 with l0 = the first element's value: a shift keeps the variance free of cancellation
__global__ __launch_bounds__(256) void k_jacobian_stats(const float* __restrict__ jac, size_t n, double* __restrict__ acc) {
    const double l0 = jac_log(jac[0]);
    double s = 0.0, s2 = 0.0, neg = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float j = jac[i];
        const double l = jac_log(j) - l0;
        s += l; s2 += l * l;
        neg += j < 0.0f ? 1.0 : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); s2 += __shfl_down(s2, o); neg += __shfl_down(neg, o); }
    // one set of atomics per workgroup (round 4: 24 K double atomics on three addresses took 0.3 ms of a 0.36 ms call)
    __shared__ double part[4][3];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = s; part[threadIdx.x >> 6][1] = s2; part[threadIdx.x >> 6][2] = neg; }
    cvx_barrier();
    if (threadIdx.x < 3) atomicAdd(&acc[threadIdx.x], (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void k_warp_nearest(const float* __restrict__ seg, const float* __restrict__ disp, int H, int W,
                                                      int D, const float* __restrict__ bh, const float* __restrict__ bw,
                                                      const float* __restrict__ bd, float* __restrict__ out) {
    const size_t V = (size_t)H * W * D;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= V) return;
    const int x = (int)(p % D), y = (int)((p / D) % W), z = (int)(p / ((size_t)D * W));
    const float scH = (float)(H - 1) / 2.0f, scW = (float)(W - 1) / 2.0f, scD = (float)(D - 1) / 2.0f;   // scale1 (:135)
    const float gz = bh[z] + fdiv(disp[p], scH), gy = bw[y] + fdiv(disp[V + p], scW), gx = bd[x] + fdiv(disp[2 * V + p], scD);
    // grid_sampler_unnormalize (align_corners=False), std::nearbyint (half to even), zeros padding
    const float fz = rintf(unnormalize(gz, H)), fy = rintf(unnormalize(gy, W)), fx = rintf(unnormalize(gx, D));
    float v = 0.0f;
    if (fz >= 0.0f && fz <= (float)(H - 1) && fy >= 0.0f && fy <= (float)(W - 1) && fx >= 0.0f && fx <= (float)(D - 1))
        v = seg[((size_t)(int)fz * W + (int)fy) * D + (int)fx];
    out[p] = v;
}

// counts[0][l] = |a == l|, counts[1][l] = |b == l|, counts[2][l] = |a == l and b == l| for l < nlab (workgroup-local
// histograms in LDS, one global atomic per label and workgroup)
__global__ __launch_bounds__(256) void k_label_overlap(const float* __restrict__ a, const float* __restrict__ b, size_t n, int nlab,
                                                       unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int hist[];           // [3][nlab]
    for (int i = threadIdx.x; i < 3 * nlab; i += blockDim.x) hist[i] = 0u;
    cvx_barrier();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float fa = a[i], fb = b[i];
        const int la = (fa >= 0.0f && fa < (float)nlab && fa == floorf(fa)) ? (int)fa : -1;
        const int lb = (fb >= 0.0f && fb < (float)nlab && fb == floorf(fb)) ? (int)fb : -1;
        if (la >= 0) atomicAdd(&hist[la], 1u);
        if (lb >= 0) atomicAdd(&hist[nlab + lb], 1u);
        if (la >= 0 && la == lb) atomicAdd(&hist[2 * nlab + la], 1u);
    }
    cvx_barrier();
    for (int i = threadIdx.x; i < 3 * nlab; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

// scipy.ndimage.map_coordinates(moving, disp + identity, order=1, mode='constant', cval=0): float64 arithmetic,
// 0 when a coordinate leaves [0, n-1], else sum over the 8 taps (axis 0 slowest) of ((v * w0) * w1) * w2
__global__ __launch_bounds__(256) void k_map_linear_f64(const double* __restrict__ mov, const double* __restrict__ disp, int H, int W,
                                                        int D, double* __restrict__ out) {
    const size_t V = (size_t)H * W * D;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= V) return;
    const int x = (int)(p % D), y = (int)((p / D) % W), z = (int)(p / ((size_t)D * W));
    const double cz = disp[3 * p] + (double)z, cy = disp[3 * p + 1] + (double)y, cx = disp[3 * p + 2] + (double)x;
    double r = 0.0;
    if (cz >= 0.0 && cz <= (double)(H - 1) && cy >= 0.0 && cy <= (double)(W - 1) && cx >= 0.0 && cx <= (double)(D - 1)) {
        const double fz = floor(cz), fy = floor(cy), fx = floor(cx);
        const double tz = cz - fz, ty = cy - fy, tx = cx - fx;
        const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
        const int z1 = z0 + 1 < H ? z0 + 1 : H - 1, y1 = y0 + 1 < W ? y0 + 1 : W - 1, x1 = x0 + 1 < D ? x0 + 1 : D - 1;
        const int zz[2] = {z0, z1}, yy[2] = {y0, y1}, xx[2] = {x0, x1};
        // scipy completes the spline weights so that they sum to exactly one: w1 = 1 - w0 (not t)
        const double wz[2] = {1.0 - tz, 1.0 - (1.0 - tz)}, wy[2] = {1.0 - ty, 1.0 - (1.0 - ty)}, wx[2] = {1.0 - tx, 1.0 - (1.0 - tx)};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k) r += ((mov[((size_t)zz[i] * W + yy[j]) * D + xx[k]] * wz[i]) * wy[j]) * wx[k];
    }
    out[p] = r;
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_jacobian_det_f32(const float* flow, int H, int W, int D, int convert1, float* out, void* stream) {
    CVX_REQUIRE(flow && out, "cvx_jacobian_det_f32: null pointer");
    CVX_REQUIRE(H > 4 && W > 4 && D > 4, "cvx_jacobian_det_f32: the field must exceed the 2-voxel crop (%dx%dx%d)", H, W, D);
    const size_t n = (size_t)(H - 4) * (W - 4) * (D - 4);
    // torch.Tensor([H-1, W-1, D-1]) / 2 in float32 (hyper_util:89)
    hipLaunchKernelGGL(k_jacobian_det, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, as_stream(stream), flow, H, W, D,
                       (float)(H - 1) / 2.0f, (float)(W - 1) / 2.0f, (float)(D - 1) / 2.0f, convert1, out);
    return check_last("jacobian_det");
}

extern "C" int cvx_jacobian_stats_f64(const float* jac, int64_t n, double* acc3, void* stream) {
    CVX_REQUIRE(jac && acc3 && n > 0, "cvx_jacobian_stats_f64: bad arguments");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(acc3, 0, 3 * sizeof(double), s) != hipSuccess) return fail(CVX_ERR_LAUNCH, "jacobian_stats: memset failed");
    const int nb = (int)(cdiv64(n, 256 * 8) < 1024 ? cdiv64(n, 256 * 8) : 1024);
    hipLaunchKernelGGL(k_jacobian_stats, dim3(nb), dim3(256), 0, s, jac, (size_t)n, acc3);
    return check_last("jacobian_stats");
}

extern "C" int cvx_warp_labels_nearest_f32(const float* seg, const float* disp, int H, int W, int D, const float* base_h,
                                           const float* base_w, const float* base_d, float* out, void* stream) {
    CVX_REQUIRE(seg && disp && base_h && base_w && base_d && out && seg != out, "cvx_warp_labels_nearest_f32: bad pointers");
    CVX_REQUIRE(H > 1 && W > 1 && D > 1, "cvx_warp_labels_nearest_f32: bad extent");
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_warp_nearest, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, as_stream(stream), seg, disp, H, W, D,
                       base_h, base_w, base_d, out);
    return check_last("warp_nearest");
}

extern "C" int cvx_label_overlap_i64(const float* a, const float* b, int64_t n, int num_labels, int64_t* counts, void* stream) {
    CVX_REQUIRE(a && b && counts && n > 0, "cvx_label_overlap_i64: bad arguments");
    CVX_REQUIRE(num_labels > 0 && num_labels <= 4096, "cvx_label_overlap_i64: num_labels %d not in 1..4096", num_labels);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(counts, 0, 3 * sizeof(int64_t) * (size_t)num_labels, s) != hipSuccess)
        return fail(CVX_ERR_LAUNCH, "label_overlap: memset failed");
    const int nb = (int)(cdiv64(n, 256 * 16) < 1024 ? cdiv64(n, 256 * 16) : 1024);
    hipLaunchKernelGGL(k_label_overlap, dim3(nb), dim3(256), 3 * sizeof(unsigned) * (size_t)num_labels, s, a, b, (size_t)n, num_labels,
                       reinterpret_cast<unsigned long long*>(counts));
    return check_last("label_overlap");
}

extern "C" int cvx_map_coordinates_linear_f64(const double* moving, const double* disp, int H, int W, int D, double* out,
                                              void* stream) {
    CVX_REQUIRE(moving && disp && out && moving != out, "cvx_map_coordinates_linear_f64: bad pointers");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_map_coordinates_linear_f64: bad extent");
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_map_linear_f64, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, as_stream(stream), moving, disp, H, W, D,
                       out);
    return check_last("map_coordinates");
}
