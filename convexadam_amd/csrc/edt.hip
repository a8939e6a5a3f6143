// edt.hip -- exact Euclidean feature transform on the device: for every voxel the coordinates of the nearest
// ZERO voxel, with the tie-breaking of scipy.ndimage.distance_transform_edt(return_indices=True), which the reference's
// masked feature path calls on the host (convex_adam_MIND.py:44,49).  Algorithm: Maurer, Qi & Raghavan (TPAMI 2003) as in
// scipy's ni_measure.c (_VoronoiFT / _ComputeFT): three passes, one per axis; in a pass every line along that axis is
// independent -- one thread per line, its site stack g[] and the line's copy f[][3] in a thread-interleaved scratch.
// Integer / comparison work on a few MB: latency-bound, three short launches.
#include "cvx_common.h"

namespace cvx {

// obj != 0 -> no site (-1), else the voxel's own coordinates
__global__ __launch_bounds__(256) void k_ft_init(const float* __restrict__ obj, int H, int W, int D, int* __restrict__ feat) {
    const size_t V = (size_t)H * W * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const bool site = obj[i] == 0.0f;
    const int x = (int)(i % D), y = (int)((i / D) % W), z = (int)(i / ((size_t)D * W));
    feat[i] = site ? z : -1;
    feat[V + i] = site ? y : -1;
    feat[2 * V + i] = site ? x : -1;
}

// one thread = one line along AXIS; scratch element e of thread t lives at scr[e * nlines + t]
template <int AXIS>
__global__ __launch_bounds__(128) void k_ft_pass(int* __restrict__ feat, int H, int W, int D, int* __restrict__ scr) {
    const int len = AXIS == 0 ? H : (AXIS == 1 ? W : D);
    const int n1 = AXIS == 0 ? W : H, n2 = AXIS == 2 ? W : D;            // the two fixed coordinates (slow, fast)
    const int nlines = n1 * n2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    const int c1 = t / n2, c2 = t % n2;
    int coor[3];
    size_t base, stride;
    const size_t V = (size_t)H * W * D;
    if (AXIS == 0) { coor[0] = 0; coor[1] = c1; coor[2] = c2; base = (size_t)c1 * D + c2; stride = (size_t)W * D; }
    else if (AXIS == 1) { coor[0] = c1; coor[1] = 0; coor[2] = c2; base = (size_t)c1 * W * D + c2; stride = D; }
    else { coor[0] = c1; coor[1] = c2; coor[2] = 0; base = ((size_t)c1 * W + c2) * D; stride = 1; }
    int* pf = feat + base;
    // scratch views: f[ii][jj] and g[l]
    auto F = [&](int ii, int jj) -> int& { return scr[((size_t)(ii * 3 + jj)) * nlines + t]; };
    auto G = [&](int l) -> int& { return scr[((size_t)(3 * len + l)) * nlines + t]; };
    for (int ii = 0; ii < len; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) F(ii, jj) = pf[ii * stride + jj * V];
    int l = -1;
    for (int ii = 0; ii < len; ++ii) {
        if (F(ii, 0) < 0) continue;
        const double fd = F(ii, AXIS);
        double wR = 0.0;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            if (jj != AXIS) { const double tw = F(ii, jj) - coor[jj]; wR += tw * tw; }
        while (l >= 1) {
            const int idx1 = G(l), idx2 = G(l - 1);
            const double f1 = F(idx1, AXIS), a = f1 - F(idx2, AXIS), b = fd - f1, c = a + b;
            double uR = 0.0, vR = 0.0;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                if (jj != AXIS) {
                    const double cc = coor[jj], tu = F(idx2, jj) - cc, tv = F(idx1, jj) - cc;
                    uR += tu * tu; vR += tv * tv;
                }
            if (c * vR - b * uR - a * wR - a * b * c <= 0.0) break;      // integer-valued doubles: exact
            --l;
        }
        G(++l) = ii;
    }
    const int maxl = l;
    if (maxl < 0) return;
    l = 0;
    for (int ii = 0; ii < len; ++ii) {
        auto dist2 = [&](int site) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) { const double tt = jj == AXIS ? F(site, jj) - ii : F(site, jj) - coor[jj]; s += tt * tt; }
            return s;
        };
        double delta1 = dist2(G(l));
        while (l < maxl) {
            const double delta2 = dist2(G(l + 1));
            if (delta1 <= delta2) break;
            delta1 = delta2;
            ++l;
        }
        const int idx1 = G(l);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) pf[ii * stride + jj * V] = F(idx1, jj);
    }
}

// the reference's flat index into the half-resolution volume: idx[0]*D//2*W//2 + idx[1]*D//2 + idx[2] with the FULL-resolution
// W, D (convex_adam_MIND.py:45,50), i.e. ((f0*D)//2*W)//2 + (f1*D)//2 + f2 in Python's left-to-right integer arithmetic
__global__ __launch_bounds__(256) void k_ft_flat_index(const int* __restrict__ feat, size_t V, int Wfull, int Dfull,
                                                       int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int64_t f0 = feat[i], f1 = feat[V + i], f2 = feat[2 * V + i];
    auto fdiv2 = [](int64_t a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); };    // Python floor division
    out[i] = fdiv2(fdiv2(f0 * Dfull) * Wfull) + fdiv2(f1 * Dfull) + f2;
}


// ---- Hausdorff-95 building blocks (reference: cupy_hd95, self_configuring/convexAdam_hyper_util.py:32-51) ----------
// inside = (nearest-upsampled label map == label), outside = 1 - inside; nearest index as in ATen's upsample_nearest3d with a
// given scale factor: src = min(floor(dst * (1.0f / p)), in - 1) in float32 (:33-34)
__global__ __launch_bounds__(256) void k_label_mask(const float* __restrict__ seg, int H, int W, int D, float label, int p,
                                                    float* __restrict__ inside, float* __restrict__ outside,
                                                    unsigned long long* __restrict__ count) {
    const int Ho = H * p, Wo = W * p, Do = D * p;
    const size_t Vo = (size_t)Ho * Wo * Do;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < Vo) {
        const int x = (int)(i % Do), y = (int)((i / Do) % Wo), z = (int)(i / ((size_t)Do * Wo));
        const float sc = 1.0f / (float)p;
        const int sz = min((int)floorf((float)z * sc), H - 1), sy = min((int)floorf((float)y * sc), W - 1),
                  sx = min((int)floorf((float)x * sc), D - 1);
        in = seg[((size_t)sz * W + sy) * D + sx] == label;
        inside[i] = in ? 1.0f : 0.0f;
        outside[i] = in ? 0.0f : 1.0f;
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

// squared Euclidean distance to the nearest zero voxel of obj (0 on the zero voxels themselves), from the feature transform
__global__ __launch_bounds__(256) void k_edt_sqdist(const float* __restrict__ obj, const int* __restrict__ feat, int H, int W, int D,
                                                    int* __restrict__ d2) {
    const size_t V = (size_t)H * W * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    int r = 0;
    if (obj[i] != 0.0f) {
        const int x = (int)(i % D), y = (int)((i / D) % W), z = (int)(i / ((size_t)D * W));
        const int dz = feat[i] - z, dy = feat[V + i] - y, dx = feat[2 * V + i] - x;
        r = dz * dz + dy * dy + dx * dx;
    }
    d2[i] = r;
}

// histogram of the squared distances (a_in2 + a_out2: one of the two is 0 at every voxel) over the surface of b (b_in2 == 1)
__global__ __launch_bounds__(256) void k_surface_hist(const int* __restrict__ a_in2, const int* __restrict__ a_out2,
                                                      const int* __restrict__ b_in2, size_t n, int nbins,
                                                      unsigned long long* __restrict__ hist, int* __restrict__ overflow) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || b_in2[i] != 1) return;
    const int bin = a_in2[i] + a_out2[i];
    if (bin < 0 || bin >= nbins) { *overflow = 1; return; }
    atomicAdd(&hist[bin], 1ull);
}

// out[0], out[1] = value (bin index) of the k0-th and k1-th smallest entry (0-based), out[2] = number of entries; -1 if out of range
__global__ __launch_bounds__(1024) void k_hist_order_stats(const unsigned long long* __restrict__ hist, int nbins, long long k0,
                                                           long long k1, long long* __restrict__ out) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (nbins + 1023) / 1024;
    const int lo = min(t * per, nbins), hi = min(lo + per, nbins);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        out[0] = -1; out[1] = -1; out[2] = (long long)run;
    }
    __syncthreads();
    unsigned long long before = part[t];
    for (int i = lo; i < hi; ++i) {
        const unsigned long long c = hist[i];
        if (c) {
            if (k0 >= 0 && (unsigned long long)k0 >= before && (unsigned long long)k0 < before + c) out[0] = i;
            if (k1 >= 0 && (unsigned long long)k1 >= before && (unsigned long long)k1 < before + c) out[1] = i;
        }
        before += c;
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_feature_transform_workspace_bytes(int H, int W, int D) {
    // per line: 3*len coordinates + len stack entries, for the pass with the largest (lines x length) = 4 * V ints
    return 256 + sizeof(int) * 4 * (size_t)H * W * D;
}

extern "C" int cvx_feature_transform_i32(const float* obj, int H, int W, int D, int* feat, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    CVX_REQUIRE(obj && feat && workspace, "cvx_feature_transform_i32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_feature_transform_i32: bad extent %dx%dx%d", H, W, D);
    if (workspace_bytes < cvx_feature_transform_workspace_bytes(H, W, D))
        return fail(CVX_ERR_WORKSPACE, "cvx_feature_transform_i32: workspace too small");
    hipStream_t s = as_stream(stream);
    Carver cv(workspace, workspace_bytes);
    const size_t V = (size_t)H * W * D;
    int* scr = cv.take<int>(4 * V);
    hipLaunchKernelGGL(k_ft_init, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, s, obj, H, W, D, feat);
    hipLaunchKernelGGL(k_ft_pass<0>, dim3(cdiv(W * D, 128)), dim3(128), 0, s, feat, H, W, D, scr);
    hipLaunchKernelGGL(k_ft_pass<1>, dim3(cdiv(H * D, 128)), dim3(128), 0, s, feat, H, W, D, scr);
    hipLaunchKernelGGL(k_ft_pass<2>, dim3(cdiv(H * W, 128)), dim3(128), 0, s, feat, H, W, D, scr);
    return check_last("feature_transform");
}

extern "C" int cvx_feature_flat_index_i64(const int* feat, int H, int W, int D, int W_full, int D_full, int64_t* out, void* stream) {
    CVX_REQUIRE(feat && out, "cvx_feature_flat_index_i64: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0 && W_full > 0 && D_full > 0, "cvx_feature_flat_index_i64: bad extent");
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_ft_flat_index, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, as_stream(stream), feat, V, W_full, D_full, out);
    return check_last("feature_flat_index");
}

extern "C" int cvx_label_mask_f32(const float* seg, int H, int W, int D, int label, int precision, float* inside, float* outside,
                                  int64_t* count, void* stream) {
    CVX_REQUIRE(seg && inside && outside && count, "cvx_label_mask_f32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_label_mask_f32: bad extent %dx%dx%d", H, W, D);
    CVX_REQUIRE(precision >= 1 && precision <= 8, "cvx_label_mask_f32: precision %d not in 1..8", precision);
    hipStream_t s = as_stream(stream);
    const size_t Vo = (size_t)H * W * D * precision * precision * precision;
    if (hipMemsetAsync(count, 0, sizeof(int64_t), s) != hipSuccess) return fail(CVX_ERR_LAUNCH, "cvx_label_mask_f32: memset failed");
    hipLaunchKernelGGL(k_label_mask, dim3((unsigned)cdiv64((int64_t)Vo, 256)), dim3(256), 0, s, seg, H, W, D, (float)label, precision,
                       inside, outside, reinterpret_cast<unsigned long long*>(count));
    return check_last("label_mask");
}

extern "C" int cvx_edt_sqdist_i32(const float* obj, const int* feat, int H, int W, int D, int* d2, void* stream) {
    CVX_REQUIRE(obj && feat && d2, "cvx_edt_sqdist_i32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_edt_sqdist_i32: bad extent %dx%dx%d", H, W, D);
    CVX_REQUIRE((double)H * H + (double)W * W + (double)D * D < 2147483647.0, "cvx_edt_sqdist_i32: extent too large for int32 distances");
    const size_t V = (size_t)H * W * D;
    hipLaunchKernelGGL(k_edt_sqdist, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, as_stream(stream), obj, feat, H, W, D, d2);
    return check_last("edt_sqdist");
}

extern "C" int cvx_surface_hist_i64(const int* a_in2, const int* a_out2, const int* b_in2, int64_t n, int nbins, int64_t* hist,
                                    int* overflow, void* stream) {
    CVX_REQUIRE(a_in2 && a_out2 && b_in2 && hist && overflow, "cvx_surface_hist_i64: null pointer");
    CVX_REQUIRE(n > 0 && nbins > 0, "cvx_surface_hist_i64: bad size");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(hist, 0, sizeof(int64_t) * (size_t)nbins, s) != hipSuccess || hipMemsetAsync(overflow, 0, sizeof(int), s) != hipSuccess)
        return fail(CVX_ERR_LAUNCH, "cvx_surface_hist_i64: memset failed");
    hipLaunchKernelGGL(k_surface_hist, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, a_in2, a_out2, b_in2, (size_t)n, nbins,
                       reinterpret_cast<unsigned long long*>(hist), overflow);
    return check_last("surface_hist");
}

extern "C" int cvx_hist_order_stats_i64(const int64_t* hist, int nbins, int64_t k0, int64_t k1, int64_t* out3, void* stream) {
    CVX_REQUIRE(hist && out3, "cvx_hist_order_stats_i64: null pointer");
    CVX_REQUIRE(nbins > 0, "cvx_hist_order_stats_i64: bad size");
    hipLaunchKernelGGL(k_hist_order_stats, dim3(1), dim3(1024), 0, as_stream(stream), reinterpret_cast<const unsigned long long*>(hist),
                       nbins, (long long)k0, (long long)k1, reinterpret_cast<long long*>(out3));
    return check_last("hist_order_stats");
}
