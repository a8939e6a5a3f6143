// pool.hip -- pooling / box smoothing / trilinear resize / grid_sample / label features.
//
// Reference call sites: F.avg_pool3d(g, stride=g) convex_adam_MIND.py:118-119,149-150;
// F.avg_pool3d(k, stride=1, padding=k/2) convex_adam_utils.py:96,107 and convex_adam_MIND.py:166,191;
// F.interpolate(trilinear) convex_adam_MIND.py:141,153,182; F.grid_sample convex_adam_utils.py:126-127;
// label one-hot features convex_adam_nnUNet.py:19-38.
// All of them are HBM/L2-bound gathers; arithmetic follows the ATen CPU kernels (raster-order sums,
// one division; FMA exactly where the ATen build fuses).
#include <hip/hip_fp16.h>
#include <math.h>
#include <string.h>

#include "cvx_common.h"

namespace cvx {

// ---- avg_pool3d(g, stride g): one thread per output, raster sum of g^3 taps, one division -------
__global__ __launch_bounds__(256) void k_avgpool(const float* __restrict__ in, int C, int H, int W, int D, int g,
                                                 float* __restrict__ out) {
    const int Ho = H / g, Wo = W / g, Do = D / g;
    const size_t n = (size_t)C * Ho * Wo * Do;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % Do), w = (int)((i / Do) % Wo), h = (int)((i / ((size_t)Do * Wo)) % Ho);
    const int c = (int)(i / ((size_t)Do * Wo * Ho));
    const float* base = in + (((size_t)c * H + (size_t)h * g) * W + (size_t)w * g) * D + (size_t)d * g;
    float s = 0.0f;
    for (int z = 0; z < g; ++z)
        for (int y = 0; y < g; ++y) {
            const float* row = base + ((size_t)z * W + y) * D;
            for (int x = 0; x < g; ++x) s += row[x];
        }
    out[i] = fdiv(s, (float)(g * g * g));
}

// Same result with the window size known at compile time: the G rows of one z-slice are fetched as 8-byte loads
// before they are added (G*G/2.. loads in flight per thread instead of one), which is what this HBM-bound pass needs.
// Requires even G and even D (8-byte aligned row segments).
template <int G>
__global__ __launch_bounds__(256) void k_avgpool_even(const float* __restrict__ in, int C, int H, int W, int D,
                                                      float* __restrict__ out) {
    static_assert(G % 2 == 0, "even windows only");
    const int Ho = H / G, Wo = W / G, Do = D / G;
    const size_t n = (size_t)C * Ho * Wo * Do;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % Do), w = (int)((i / Do) % Wo), h = (int)((i / ((size_t)Do * Wo)) % Ho);
    const int c = (int)(i / ((size_t)Do * Wo * Ho));
    const float* base = in + (((size_t)c * H + (size_t)h * G) * W + (size_t)w * G) * D + (size_t)d * G;
    float s = 0.0f;
#pragma unroll
    for (int z = 0; z < G; ++z) {
        float2 v[G][G / 2];
#pragma unroll
        for (int y = 0; y < G; ++y)
#pragma unroll
            for (int x = 0; x < G / 2; ++x) v[y][x] = *reinterpret_cast<const float2*>(base + ((size_t)z * W + y) * D + 2 * x);
#pragma unroll
        for (int y = 0; y < G; ++y)
#pragma unroll
            for (int x = 0; x < G / 2; ++x) { s += v[y][x].x; s += v[y][x].y; }
    }
    out[i] = fdiv(s, (float)(G * G * G));
}

// ---- avg_pool3d(k, stride 1, pad k/2), forward and ATen-ordered backward (global-memory version)
template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_box_zero(const float* __restrict__ in, float* __restrict__ out, int C, int H,
                                                  int W, int D, int k) {
    const size_t n = (size_t)C * H * W * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D), w = (int)((i / D) % W), h = (int)((i / ((size_t)D * W)) % H);
    const int c = (int)(i / ((size_t)D * W * H));
    const int p = k / 2;
    const float div = (float)(k * k * k);
    const int h0 = max(h - p, 0), h1 = min(h + p, H - 1), w0 = max(w - p, 0), w1 = min(w + p, W - 1),
              d0 = max(d - p, 0), d1 = min(d + p, D - 1);
    const float* ic = in + (size_t)c * H * W * D;
    float s = 0.0f;
    for (int z = h0; z <= h1; ++z)
        for (int y = w0; y <= w1; ++y)
            for (int x = d0; x <= d1; ++x) {
                const float v = ic[((size_t)z * W + y) * D + x];
                s += BACKWARD ? fdiv(v, div) : v;     // backward: every output adds gradOut/k^3
            }
    out[i] = BACKWARD ? s : fdiv(s, div);
}

// the same with an EVEN kernel: padding k/2 on both sides of a window of k taps -> (H+1, W+1, D+1); output o covers inputs
// o-k/2 .. o+k/2-1 (the reference's even `selected_smooth`, convex_adam_MIND.py:184-191: its "+1" is overwritten at :189)
__global__ __launch_bounds__(256) void k_box_grow(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int D, int k) {
    const int Ho = H + 1, Wo = W + 1, Do = D + 1;
    const size_t n = (size_t)C * Ho * Wo * Do;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % Do), w = (int)((i / Do) % Wo), h = (int)((i / ((size_t)Do * Wo)) % Ho);
    const int c = (int)(i / ((size_t)Do * Wo * Ho));
    const int p = k / 2;
    const int h0 = max(h - p, 0), h1 = min(h - p + k, H), w0 = max(w - p, 0), w1 = min(w - p + k, W), d0 = max(d - p, 0), d1 = min(d - p + k, D);
    const float* ic = in + (size_t)c * H * W * D;
    float s = 0.0f;
    for (int z = h0; z < h1; ++z)
        for (int y = w0; y < w1; ++y)
            for (int x = d0; x < d1; ++x) s += ic[((size_t)z * W + y) * D + x];
    out[i] = fdiv(s, (float)(k * k * k));
}

// ---- the same forward filter as a z-walk: every tap is read once per thread and plane ----------------------------------------------
// k_box_zero reads k^3 taps per output through a dependent chain of global loads (27 / 125 loads per output: 0.3-1.9 ms per filter of
// the sweep's kovesi chains).  Here a thread owns CPT consecutive columns of one row and a segment of L planes and walks along z: per
// plane it loads its k rows x (CPT + 2R) columns once and adds them, in ATen's raster order, to a ring of k running sums -- the sum of
// output plane o starts at plane o - R from +0.0 and is complete after plane o + R, i.e. the very chain of additions of the reference's
// loop (z slowest, then y, then x; taps outside the volume contribute nothing: +0.0, which leaves every partial sum unchanged).
// k^3 additions per output + one exact division; a segment pays 2R extra planes of additions for its L outputs.
template <int R, int CPT>
__global__ __launch_bounds__(256) void k_box_walk(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int D, int L, int nq) {
    constexpr int K = 2 * R + 1, WC = CPT + 2 * R;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;         // (c, y, q)
    if (item >= (size_t)C * W * nq) return;
    const int lane = threadIdx.x & 63;
    const int q = (int)(item % nq), y = (int)((item / nq) % W), c = (int)(item / ((size_t)nq * W));
    const int x0 = q * CPT, z0 = (int)blockIdx.y * L, zend = min(z0 + L, H);
    const size_t plane = (size_t)W * D;
    const float* ic = in + (size_t)c * H * plane;
    float* oc = out + (size_t)c * H * plane;
    bool rok[K];
    unsigned roff[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { const int yy = y - R + i; rok[i] = yy >= 0 && yy < W; roff[i] = (unsigned)((rok[i] ? yy : 0) * D + x0); }
    bool cl[R], cr[R];                                                        // halo columns inside the row?
#pragma unroll
    for (int t = 0; t < R; ++t) { cl[t] = x0 - R + t >= 0; cr[t] = x0 + CPT + t < D; }
    float sum[K][CPT];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int j = 0; j < CPT; ++j) sum[r][j] = 0.0f;
    constexpr float DIVF = (float)(K * K * K);
    for (int p = z0 - R; p < zend + R; ++p) {
        if (p >= 0 && p < H) {                                                // (uniform) a plane outside the volume adds nothing
            const float* pp = ic + (size_t)p * plane;
            float w[K][WC];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float* rp = pp + roff[i];
                if (CPT == 4) {
                    const float4 a = rok[i] ? *reinterpret_cast<const float4*>(rp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    w[i][R] = a.x; w[i][R + 1] = a.y; w[i][R + 2] = a.z; w[i][R + 3] = a.w;
                } else {
                    const float2 a = rok[i] ? *reinterpret_cast<const float2*>(rp) : make_float2(0.f, 0.f);
                    w[i][R] = a.x; w[i][R + 1] = a.y;
                }
                if (R <= CPT) {
                    // the halo columns are the neighbour lanes' own columns (items are (c, y, q) with q fastest: lane - 1 holds the quad to the left
                    // unless this lane starts a row): DPP wave shifts instead of 2 R scalar loads per row; the first / last lane of a wavefront
                    // has no neighbour lane and loads (a one-lane load costs the texture path a lane, not a wavefront)
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        const float fromL = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[i][CPT + t]), 0x138, 0xf, 0xf, true));      // wave_shr:1: lane - 1's column CPT - R + t
                        const float fromR = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[i][R + t]), 0x130, 0xf, 0xf, true));        // wave_shl:1: lane + 1's column t
                        float hl = fromL, hr = fromR;
                        if (lane == 0 && rok[i] && cl[t]) hl = rp[t - R];
                        if (lane == 63 && rok[i] && cr[t]) hr = rp[CPT + t];
                        w[i][t] = cl[t] ? hl : 0.0f;
                        w[i][R + CPT + t] = cr[t] ? hr : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        w[i][t] = (rok[i] && cl[t]) ? rp[t - R] : 0.0f;
                        w[i][R + CPT + t] = (rok[i] && cr[t]) ? rp[CPT + t] : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int t = 0; t < K; ++t)
#pragma unroll
                    for (int r = 0; r < K; ++r)
#pragma unroll
                        for (int j = 0; j < CPT; ++j) sum[r][j] += w[i][j + t];
        }
        const int o = p - R;                                                  // the output plane this plane completes
        if (o >= z0 && o < zend) {
            float* op = oc + ((size_t)o * W + y) * D + x0;
            float v[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) v[j] = (K == 3 || K == 5 || K == 7) ? div_exact<(K == 3 || K == 5 || K == 7) ? K * K * K : 27>(sum[0][j]) : fdiv(sum[0][j], DIVF);
            if (CPT == 4) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[CPT - 2], v[CPT - 1]);
            else *reinterpret_cast<float2*>(op) = make_float2(v[0], v[1]);
        }
#pragma unroll
        for (int r = 0; r + 1 < K; ++r)
#pragma unroll
            for (int j = 0; j < CPT; ++j) sum[r][j] = sum[r + 1][j];
#pragma unroll
        for (int j = 0; j < CPT; ++j) sum[K - 1][j] = 0.0f;
    }
}

template <int R>
static int launch_box_walk_r(const float* in, float* out, int C, int H, int W, int D, hipStream_t s) {
    // columns per thread and planes per segment from the volume: enough wavefronts for the chip first, then long segments
    const size_t quads4 = (size_t)C * W * (D / 4);
    const bool cpt2 = quads4 * H < ((size_t)1 << 22);                         // small grids (the Adam control grid at grid_sp_adam 2): two columns per thread
    const int cpt = cpt2 ? 2 : 4, nq = D / cpt;
    const size_t items = (size_t)C * W * nq;
    int L = 16;
    while (L > 4 && items * (size_t)cdiv(H, L) < (size_t)256 * 256 * 6) L -= 4;            // ~6 wavefronts per SIMD wanted
    const dim3 grid((unsigned)cdiv64((int64_t)items, 256), (unsigned)cdiv(H, L));
    if (cpt2) hipLaunchKernelGGL((k_box_walk<R, 2>), grid, dim3(256), 0, s, in, out, C, H, W, D, L, nq);
    else hipLaunchKernelGGL((k_box_walk<R, 4>), grid, dim3(256), 0, s, in, out, C, H, W, D, L, nq);
    return check_last("box_walk");
}

int launch_box_zero(const float* in, float* out, int C, int H, int W, int D, int k, bool backward, hipStream_t s) {
    const size_t n = (size_t)C * H * W * D;
    const bool al = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (!backward && al && D % 4 == 0 && in != out && options().box_walk != 0) {
        if (k == 3) return launch_box_walk_r<1>(in, out, C, H, W, D, s);
        if (k == 5) return launch_box_walk_r<2>(in, out, C, H, W, D, s);
        if (k == 7) return launch_box_walk_r<3>(in, out, C, H, W, D, s);
    }
    const dim3 grid((unsigned)cdiv64((int64_t)n, 256));
    if (backward) hipLaunchKernelGGL(k_box_zero<true>, grid, dim3(256), 0, s, in, out, C, H, W, D, k);
    else hipLaunchKernelGGL(k_box_zero<false>, grid, dim3(256), 0, s, in, out, C, H, W, D, k);
    return check_last("box_zero");
}

// ---- GaussianSmoothing: 5-tap convolution along one axis, replicate padding (hyper_util.py:423-437) -------------
// forward  (oneDNN conv, pinned on torch 2.10 CPU): acc = w0*x0 ; acc = fma(w_t, x_t, acc), t = 1..4
// backward (conv backward-data + replication_pad3d_backward): gxp[j] = w0*g[j], then t = 1..4 ascending
//          fma(w_t, g[j-t], acc) for multi-channel tensors (a single-channel tensor takes another oneDNN kernel that
//          rounds the products: acc + w_t*g) on the padded index range (g = 0 outside), then the pad gradient adds
//          the border entries of gxp in ascending order into the first / last element.
struct GaussW { float w[5]; };
template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_gauss1d(const float* __restrict__ in, float* __restrict__ out, size_t total, int n,
                                                 size_t stride, GaussW gw, bool fused_bwd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = (int)((i / stride) % n);                 // coordinate along the filtered axis
    const float* base = in + (i - (size_t)a * stride);     // element 0 of this line
    if (!BACKWARD) {
        float acc = gw.w[0] * base[(size_t)clampi(a - 2, 0, n - 1) * stride];
#pragma unroll
        for (int t = 1; t < 5; ++t) acc = __builtin_fmaf(gw.w[t], base[(size_t)clampi(a + t - 2, 0, n - 1) * stride], acc);
        out[i] = acc;
    } else {
        float r = 0.0f;
        const int j0 = a == 0 ? 0 : a + 2, j1 = a == n - 1 ? n + 3 : a + 2;     // padded positions that clamp onto a
        for (int j = j0; j <= j1; ++j) {
            float g0 = (j >= 0 && j < n) ? base[(size_t)j * stride] : 0.0f;
            float acc = gw.w[0] * g0;
#pragma unroll
            for (int t = 1; t < 5; ++t) {
                const int q = j - t;
                const float gq = (q >= 0 && q < n) ? base[(size_t)q * stride] : 0.0f;
                acc = fused_bwd ? __builtin_fmaf(gw.w[t], gq, acc) : acc + gw.w[t] * gq;
            }
            r += acc;
        }
        out[i] = r;
    }
}
static int launch_gauss1d(const float* in, float* out, int C, int H, int W, int D, int axis, const float* w5, bool backward,
                          hipStream_t s) {
    const size_t total = (size_t)C * H * W * D;
    const int n = axis == 0 ? H : (axis == 1 ? W : D);
    const size_t stride = axis == 0 ? (size_t)W * D : (axis == 1 ? (size_t)D : 1);
    GaussW gw;
    for (int t = 0; t < 5; ++t) gw.w[t] = w5[t];
    const dim3 grid((unsigned)cdiv64((int64_t)total, 256));
    if (backward) hipLaunchKernelGGL(k_gauss1d<true>, grid, dim3(256), 0, s, in, out, total, n, stride, gw, C > 1);
    else hipLaunchKernelGGL(k_gauss1d<false>, grid, dim3(256), 0, s, in, out, total, n, stride, gw, C > 1);
    return check_last("gauss1d");
}

// applies a smoother (or its adjoint) with ping-pong through `tmp`; the last stage lands in `out`
int launch_smoother(const float* in, float* out, float* tmp, int C, int H, int W, int D, const cvx_smoother& sm, bool backward,
                    hipStream_t s) {
    const int nst = sm.kind == 1 ? 3 : sm.n_boxes;
    const float* src = in;
    for (int i = 0; i < nst; ++i) {
        float* dst = ((nst - 1 - i) & 1) ? tmp : out;
        const int st = backward ? nst - 1 - i : i;          // the adjoint runs the stages in reverse order
        int rc;
        if (sm.kind == 1) rc = launch_gauss1d(src, dst, C, H, W, D, st, sm.gauss_w, backward, s);
        else rc = launch_box_zero(src, dst, C, H, W, D, sm.box_k[st], backward, s);
        if (rc) return rc;
        src = dst;
    }
    return CVX_OK;
}

// ---- masked-feature helpers (convex_adam_MIND.py:36-54) ------------------------------------------------------
// (ReplicationPad3d(1) + AvgPool3d(3, stride 1))(mask) > 0.9 : raster sum of 27 clamped taps, one exact division
__global__ __launch_bounds__(256) void k_mask_erode(const float* __restrict__ mask, int H, int W, int D, float thr,
                                                    float* __restrict__ out) {
    const size_t n = (size_t)H * W * D;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D), w = (int)((i / D) % W), h = (int)(i / ((size_t)D * W));
    float s = 0.0f;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b)
            for (int c = -1; c <= 1; ++c)
                s += mask[((size_t)clampi(h + a, 0, H - 1) * W + clampi(w + b, 0, W - 1)) * D + clampi(d + c, 0, D - 1)];
    out[i] = fdiv(s, 27.0f) > thr ? 1.0f : 0.0f;
}
__global__ __launch_bounds__(256) void k_gather_index(const float* __restrict__ src, const int64_t* __restrict__ idx, size_t n,
                                                      float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ m, const float* __restrict__ a, const float* __restrict__ b,
                                                size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = m[i] != 0.0f ? a[i] : b[i];
}

// ---- trilinear resize (ATen upsample_trilinear3d, align_corners=False) ----------------------------
//   src = max(fma(in/out, dst + 0.5, -0.5), 0); i0 = min(floor(src), in-1); l1 = clamp(src - i0, 0, 1)
//   l0 = 1 - l1; i1 = i0 + (i0 < in-1); per level (last dim first): r = fma(v0, l0, v1 * l1)
__device__ __forceinline__ void lin_coef(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
    if (in == out) { i0 = o; i1 = o; l0 = 1.0f; l1 = 0.0f; return; }
    const float ratio = fdiv((float)in, (float)out);
    float src = __builtin_fmaf(ratio, (float)o + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int a = (int)floorf(src);
    a = a > in - 1 ? in - 1 : a;
    float l = src - (float)a;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
    i0 = a;
    i1 = a + ((a < in - 1) ? 1 : 0);
    l1 = l;
    l0 = 1.0f - l;
}
// One thread per output voxel, all channels.  CT > 0: the channel count at compile time (fields: 3) -- the channel loop is
// unrolled and the 8 * CT taps are in flight together; with a run-time count the loop runs one memory round trip per channel
// and the kernel is bound by that latency (59 vs 3x us for the 82 MB field of the final up-sampling).
template <int CT>
__global__ __launch_bounds__(256) void k_resize(const float* __restrict__ in, int C, int h, int w, int d,
                                                float* __restrict__ out, int H, int W, int D, float pre_mul,
                                                float post_div) {
    // grid = (pieces of a row, W, H): no integer division by run-time extents, z and y coefficients are wave-uniform
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y, z = (int)blockIdx.z;
    if (x >= D) return;
    const size_t n = (size_t)H * W * D;
    const size_t i = ((size_t)z * W + y) * D + x;
    int z0, z1, y0, y1, x0, x1;
    float lz0, lz1, ly0, ly1, lx0, lx1;
    lin_coef(z, h, H, z0, z1, lz0, lz1);
    lin_coef(y, w, W, y0, y1, ly0, ly1);
    lin_coef(x, d, D, x0, x1, lx0, lx1);
    const size_t o00 = ((size_t)z0 * w + y0) * d, o01 = ((size_t)z0 * w + y1) * d, o10 = ((size_t)z1 * w + y0) * d,
                 o11 = ((size_t)z1 * w + y1) * d;
    const size_t cs = (size_t)h * w * d;
    auto one = [&](const float (&v)[8]) {
        const float a0 = __builtin_fmaf(v[0] * pre_mul, lx0, (v[1] * pre_mul) * lx1);   // pre_mul = 1: exact no-op
        const float a1 = __builtin_fmaf(v[2] * pre_mul, lx0, (v[3] * pre_mul) * lx1);
        const float b0 = __builtin_fmaf(v[4] * pre_mul, lx0, (v[5] * pre_mul) * lx1);
        const float b1 = __builtin_fmaf(v[6] * pre_mul, lx0, (v[7] * pre_mul) * lx1);
        const float l0 = __builtin_fmaf(a0, ly0, a1 * ly1);
        const float l1 = __builtin_fmaf(b0, ly0, b1 * ly1);
        float r = __builtin_fmaf(l0, lz0, l1 * lz1);
        if (post_div != 1.0f) r = fdiv(r, post_div);
        return r;
    };
    auto fetch = [&](const float* ic, float (&v)[8]) {
        v[0] = ic[o00 + x0]; v[1] = ic[o00 + x1]; v[2] = ic[o01 + x0]; v[3] = ic[o01 + x1];
        v[4] = ic[o10 + x0]; v[5] = ic[o10 + x1]; v[6] = ic[o11 + x0]; v[7] = ic[o11 + x1];
    };
    if (CT > 0) {
        float v[CT > 0 ? CT : 1][8];
#pragma unroll
        for (int c = 0; c < CT; ++c) fetch(in + (size_t)c * cs, v[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) out[(size_t)c * n + i] = one(v[c]);
    } else {
        for (int c = 0; c < C; ++c) {
            float v[8];
            fetch(in + (size_t)c * cs, v);
            out[(size_t)c * n + i] = one(v);
        }
    }
}
// Exact factor-2 up-sampling (the final up-sampling of the Adam grid, convex_adam_MIND.py:182): one thread per SOURCE voxel k = (kz, ky, kx)
// produces the 2 x 2 x 2 outputs (2k .. 2k+1).  Their corners all lie in the 3 x 3 x 3 neighbourhood of k (clamped), which is loaded once
// (27 instead of 64 taps per channel), and ATen's chain of three 1-D interpolations is shared level by level: 18 x-level values, 12
// y-level values, 8 results -- every output goes through exactly the operations k_resize performs for it (lin_coef per output index decides
// which two of the three taps it blends, so the clamped borders and a one-voxel axis are covered by the same code).
template <int CT>
__global__ __launch_bounds__(256) void k_resize_up2(const float* __restrict__ in, int h, int w, int d, float* __restrict__ out, float pre_mul,
                                                    float post_div) {
    const int kx = (int)(blockIdx.x * blockDim.x + threadIdx.x), ky = (int)blockIdx.y, kz = (int)blockIdx.z;
    if (kx >= d) return;
    const int H = 2 * h, W = 2 * w, D = 2 * d;
    struct Ax { int p0[2], p1[2]; float l0[2], l1[2]; };
    auto axis = [](int k, int n_in, Ax& a) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int i0, i1;
            lin_coef(2 * k + t, n_in, 2 * n_in, i0, i1, a.l0[t], a.l1[t]);
            a.p0[t] = i0 - k + 1;           // position inside the triple (k-1, k, k+1): 0, 1 or 2
            a.p1[t] = i1 - k + 1;
        }
    };
    Ax az, ay, ax;
    axis(kz, h, az); axis(ky, w, ay); axis(kx, d, ax);
    const int zi[3] = {kz > 0 ? kz - 1 : 0, kz, kz < h - 1 ? kz + 1 : h - 1}, yi[3] = {ky > 0 ? ky - 1 : 0, ky, ky < w - 1 ? ky + 1 : w - 1},
              xi[3] = {kx > 0 ? kx - 1 : 0, kx, kx < d - 1 ? kx + 1 : d - 1};
    auto sel = [](float v0, float v1, float v2, int p) { return p == 0 ? v0 : (p == 1 ? v1 : v2); };
    const size_t cs = (size_t)h * w * d, n = (size_t)H * W * D;
    float v[CT][3][3][3];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int e = 0; e < 3; ++e) v[c][a][b][e] = in[(size_t)c * cs + ((size_t)zi[a] * w + yi[b]) * d + xi[e]];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float X[3][3][2], Y[3][2][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float u0 = sel(v[c][a][b][0], v[c][a][b][1], v[c][a][b][2], ax.p0[t]), u1 = sel(v[c][a][b][0], v[c][a][b][1], v[c][a][b][2], ax.p1[t]);
                    X[a][b][t] = __builtin_fmaf(u0 * pre_mul, ax.l0[t], (u1 * pre_mul) * ax.l1[t]);
                }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float u0 = sel(X[a][0][t], X[a][1][t], X[a][2][t], ay.p0[ty]), u1 = sel(X[a][0][t], X[a][1][t], X[a][2][t], ay.p1[ty]);
                    Y[a][ty][t] = __builtin_fmaf(u0, ay.l0[ty], u1 * ay.l1[ty]);
                }
#pragma unroll
        for (int tz = 0; tz < 2; ++tz)
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                float r[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float u0 = sel(Y[0][ty][t], Y[1][ty][t], Y[2][ty][t], az.p0[tz]), u1 = sel(Y[0][ty][t], Y[1][ty][t], Y[2][ty][t], az.p1[tz]);
                    r[t] = __builtin_fmaf(u0, az.l0[tz], u1 * az.l1[tz]);
                    if (post_div != 1.0f) r[t] = fdiv(r[t], post_div);
                }
                *reinterpret_cast<float2*>(out + (size_t)c * n + ((size_t)(2 * kz + tz) * W + (2 * ky + ty)) * D + 2 * kx) = make_float2(r[0], r[1]);
            }
    }
}
int launch_resize(const float* in, int C, int h, int w, int d, float* out, int H, int W, int D, float pre_mul,
                  float post_div, hipStream_t s) {
    if (H > 65535 || W > 65535) return fail(CVX_ERR_UNSUPPORTED, "resize_trilinear: output extent %dx%d exceeds the grid limits", H, W);
    if (C == 3 && H == 2 * h && W == 2 * w && D == 2 * d && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && options().resize_up2 != 0) {
        const int bx = d > 128 ? 256 : (d > 64 ? 128 : 64);
        hipLaunchKernelGGL(k_resize_up2<3>, dim3((unsigned)cdiv(d, bx), (unsigned)w, (unsigned)h), dim3(bx), 0, s, in, h, w, d, out, pre_mul, post_div);
        return check_last("resize_trilinear");
    }
    const int bx = D > 128 ? 256 : (D > 64 ? 128 : 64);              // short rows: do not pad them to 256 lanes
    const dim3 grid((unsigned)cdiv(D, bx), (unsigned)W, (unsigned)H), block(bx);
    if (C == 3) hipLaunchKernelGGL(k_resize<3>, grid, block, 0, s, in, C, h, w, d, out, H, W, D, pre_mul, post_div);
    else if (C == 1) hipLaunchKernelGGL(k_resize<1>, grid, block, 0, s, in, C, h, w, d, out, H, W, D, pre_mul, post_div);
    else hipLaunchKernelGGL(k_resize<0>, grid, block, 0, s, in, C, h, w, d, out, H, W, D, pre_mul, post_div);
    return check_last("resize_trilinear");
}

// Two chained resizes without the (H,W,D) intermediate: out = resize(resize(in, (H,W,D)), (h2,w2,d2)) / post_div (the pipeline's
// disp_hr -> disp_lr, convex_adam_MIND.py:141,153: 82 MB written and read back at OASIS size).  ATen's trilinear kernel is a
// chain of three 1-D interpolations (x, then y, then z), each rounded to float, so the first resize factors exactly:
//   k_resize_yx : T[c][zc][Y][X] = the y-level value for COARSE plane zc at fine (Y, X)            (h x W x D, 13 MB)
//   k_resize2   : an output needs 2 x 2 x 2 intermediate values; each is fma(T[z0], lz0, T[z1] * lz1) -- the z level of the
//                 first resize -- followed by the three levels of the second resize, all with k_resize's operations and
//                 roundings: bit-identical to the two-pass form.
template <int CT>
__global__ __launch_bounds__(256) void k_resize_yx(const float* __restrict__ in, int C, int h, int w, int d, float* __restrict__ T,
                                                   int W, int D) {
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y, z = (int)blockIdx.z;     // z: coarse plane
    if (x >= D) return;
    const size_t n = (size_t)h * W * D;
    const size_t i = ((size_t)z * W + y) * D + x;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_coef(y, w, W, y0, y1, ly0, ly1);
    lin_coef(x, d, D, x0, x1, lx0, lx1);
    auto one = [&](int c) {
        const float* r0 = in + (((size_t)c * h + z) * w + y0) * d;
        const float* r1 = in + (((size_t)c * h + z) * w + y1) * d;
        const float a = __builtin_fmaf(r0[x0], lx0, r0[x1] * lx1);
        const float b = __builtin_fmaf(r1[x0], lx0, r1[x1] * lx1);
        T[(size_t)c * n + i] = __builtin_fmaf(a, ly0, b * ly1);
    };
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) one(c);
    } else
        for (int c = 0; c < C; ++c) one(c);
}
template <int CT>
__global__ __launch_bounds__(256) void k_resize2(const float* __restrict__ T, int C, int h, int H, int W, int D, float* __restrict__ out,
                                                 int h2, int w2, int d2, float post_div) {
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y, z = (int)blockIdx.z;
    if (x >= d2) return;
    const size_t n = (size_t)h2 * w2 * d2;
    const size_t i = ((size_t)z * w2 + y) * d2 + x;
    int Z[2], Y[2], X[2], cz[2][2];
    float LZ[2], LY[2], LX[2], wz[2][2];
    lin_coef(z, H, h2, Z[0], Z[1], LZ[0], LZ[1]);
    lin_coef(y, W, w2, Y[0], Y[1], LY[0], LY[1]);
    lin_coef(x, D, d2, X[0], X[1], LX[0], LX[1]);
#pragma unroll
    for (int a = 0; a < 2; ++a) lin_coef(Z[a], h, H, cz[a][0], cz[a][1], wz[a][0], wz[a][1]);
    const size_t plane = (size_t)W * D;
    // (wave-uniform: every lane's two x taps adjacent and 8-byte aligned; T is the 256-byte aligned scratch of launch_resize2)
    const bool pairs = (D & 1) == 0 && (reinterpret_cast<uintptr_t>(T) & 7) == 0 && __all(X[1] == X[0] + 1 && (X[0] & 1) == 0);
    auto one = [&](int c) {
        const float* Tc = T + (size_t)c * h * plane;
        float v[2][2][2][2];                                        // [a][level-1 plane][b][e]: all 16 taps in flight
        if (pairs) {                                                // the two x taps are one aligned 8-byte piece (factor-2 reduction: X = 2x, 2x + 1)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float2 q = *reinterpret_cast<const float2*>(Tc + (size_t)cz[a][p] * plane + (size_t)Y[b] * D + X[0]);
                        v[a][p][b][0] = q.x; v[a][p][b][1] = q.y;
                    }
        } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 2; ++e) v[a][p][b][e] = Tc[(size_t)cz[a][p] * plane + (size_t)Y[b] * D + X[e]];
        }
        float lev1[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float lev2[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float m0 = __builtin_fmaf(v[a][0][b][0], wz[a][0], v[a][1][b][0] * wz[a][1]);   // intermediate (Z[a], Y[b], X[0])
                const float m1 = __builtin_fmaf(v[a][0][b][1], wz[a][0], v[a][1][b][1] * wz[a][1]);
                lev2[b] = __builtin_fmaf(m0, LX[0], m1 * LX[1]);
            }
            lev1[a] = __builtin_fmaf(lev2[0], LY[0], lev2[1] * LY[1]);
        }
        float r = __builtin_fmaf(lev1[0], LZ[0], lev1[1] * LZ[1]);
        if (post_div != 1.0f) r = fdiv(r, post_div);
        out[(size_t)c * n + i] = r;
    };
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) one(c);
    } else
        for (int c = 0; c < C; ++c) one(c);
}
// `scratch`: C * h * W * D floats
int launch_resize2(const float* in, int C, int h, int w, int d, int H, int W, int D, float* scratch, float* out, int h2, int w2, int d2,
                   float post_div, hipStream_t s) {
    if (W > 65535 || h > 65535 || w2 > 65535 || h2 > 65535) return fail(CVX_ERR_UNSUPPORTED, "resize_trilinear2: extent exceeds the grid limits");
    auto bx = [](int n) { return n > 128 ? 256 : (n > 64 ? 128 : 64); };
    const dim3 g1((unsigned)cdiv(D, bx(D)), (unsigned)W, (unsigned)h), g2((unsigned)cdiv(d2, bx(d2)), (unsigned)w2, (unsigned)h2);
    if (C == 3) {
        hipLaunchKernelGGL(k_resize_yx<3>, g1, dim3(bx(D)), 0, s, in, C, h, w, d, scratch, W, D);
        hipLaunchKernelGGL(k_resize2<3>, g2, dim3(bx(d2)), 0, s, scratch, C, h, H, W, D, out, h2, w2, d2, post_div);
    } else {
        hipLaunchKernelGGL(k_resize_yx<0>, g1, dim3(bx(D)), 0, s, in, C, h, w, d, scratch, W, D);
        hipLaunchKernelGGL(k_resize2<0>, g2, dim3(bx(d2)), 0, s, scratch, C, h, H, W, D, out, h2, w2, d2, post_div);
    }
    return check_last("resize_trilinear2");
}

// ---- generic grid_sample ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_sample(const float* __restrict__ vol, int C, int h, int w, int d,
                                                     const float* __restrict__ grid, size_t vo, float* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= vo) return;
    Tri t;
    tri_setup(t, grid[3 * p], grid[3 * p + 1], grid[3 * p + 2], h, w, d);
    const size_t vi = (size_t)h * w * d;
    for (int c = 0; c < C; ++c) out[(size_t)c * vo + p] = tri_sample(t, vol + (size_t)c * vi, h, w, d);
}

// ---- label features ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_label_hist(const float* __restrict__ lab, int64_t V, int max_label,
                                                    unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int sh[];
    for (int i = threadIdx.x; i <= max_label; i += blockDim.x) sh[i] = 0;
    cvx_barrier();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)lab[i];
        if (l >= 0 && l <= max_label) atomicAdd(&sh[l], 1u);
    }
    cvx_barrier();
    for (int i = threadIdx.x; i <= max_label; i += blockDim.x)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
__global__ __launch_bounds__(256) void k_label_features(const float* __restrict__ lab, int64_t V, int C,
                                                        const int* __restrict__ present,
                                                        const float* __restrict__ weights, float mult,
                                                        float* __restrict__ feat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int l = (int)lab[i];
    for (int c = 0; c < C; ++c) {
        const float oh = (l == present[c]) ? 1.0f : 0.0f;
        feat[(size_t)c * V + i] = mult * (oh * weights[c]);      // 10*(onehot*weight), convex_adam_nnUNet.py:35
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_avgpool_f32(const float* in, int C, int H, int W, int D, int g, float* out, void* stream) {
    CVX_REQUIRE(in && out, "cvx_avgpool_f32: null pointer");
    CVX_REQUIRE(C > 0 && H > 0 && W > 0 && D > 0 && g > 0, "cvx_avgpool_f32: bad arguments");
    CVX_REQUIRE(H / g > 0 && W / g > 0 && D / g > 0, "cvx_avgpool_f32: pooling window %d larger than the volume", g);
    const size_t n = (size_t)C * (H / g) * (W / g) * (D / g);
    const dim3 grid((unsigned)cdiv64((int64_t)n, 256));
    const bool even = (D % 2 == 0) && (reinterpret_cast<uintptr_t>(in) & 7) == 0;
    if (even && g == 2) hipLaunchKernelGGL(k_avgpool_even<2>, grid, dim3(256), 0, as_stream(stream), in, C, H, W, D, out);
    else if (even && g == 4) hipLaunchKernelGGL(k_avgpool_even<4>, grid, dim3(256), 0, as_stream(stream), in, C, H, W, D, out);
    else if (even && g == 6) hipLaunchKernelGGL(k_avgpool_even<6>, grid, dim3(256), 0, as_stream(stream), in, C, H, W, D, out);
    else if (even && g == 8) hipLaunchKernelGGL(k_avgpool_even<8>, grid, dim3(256), 0, as_stream(stream), in, C, H, W, D, out);
    else hipLaunchKernelGGL(k_avgpool, grid, dim3(256), 0, as_stream(stream), in, C, H, W, D, g, out);
    return check_last("avgpool");
}

static int check_smoother(const cvx_smoother* sm) {
    CVX_REQUIRE(sm, "smoother: null");
    CVX_REQUIRE(sm->kind == 0 || sm->kind == 1, "smoother: kind must be 0 (box chain) or 1 (gaussian)");
    if (sm->kind == 0) {
        CVX_REQUIRE(sm->n_boxes >= 1 && sm->n_boxes <= 4, "smoother: n_boxes must be 1..4");
        for (int i = 0; i < sm->n_boxes; ++i) CVX_REQUIRE(sm->box_k[i] >= 1 && (sm->box_k[i] & 1), "smoother: box size must be odd");
    }
    return CVX_OK;
}
extern "C" size_t cvx_smooth_workspace_bytes(int C, int H, int W, int D) { return 256 + sizeof(float) * (size_t)C * H * W * D; }
extern "C" int cvx_smooth_f32(const float* in, int C, int H, int W, int D, const cvx_smoother* sm, int backward, float* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(in && out && in != out && workspace, "cvx_smooth_f32: bad pointers");
    CVX_REQUIRE(C > 0 && H > 0 && W > 0 && D > 0, "cvx_smooth_f32: bad extent");
    int rc = check_smoother(sm);
    if (rc) return rc;
    if (workspace_bytes < cvx_smooth_workspace_bytes(C, H, W, D)) return fail(CVX_ERR_WORKSPACE, "cvx_smooth_f32: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float* tmp = cv.take<float>((size_t)C * H * W * D);
    return launch_smoother(in, out, tmp, C, H, W, D, *sm, backward != 0, as_stream(stream));
}

extern "C" int cvx_mask_erode_f32(const float* mask, int H, int W, int D, float threshold, float* out, void* stream) {
    CVX_REQUIRE(mask && out && H > 0 && W > 0 && D > 0, "cvx_mask_erode_f32: bad arguments");
    const size_t n = (size_t)H * W * D;
    hipLaunchKernelGGL(k_mask_erode, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, as_stream(stream), mask, H, W, D, threshold, out);
    return check_last("mask_erode");
}
extern "C" int cvx_gather_f32(const float* src, const int64_t* index, int64_t n, float* out, void* stream) {
    CVX_REQUIRE(src && index && out && n > 0, "cvx_gather_f32: bad arguments");
    hipLaunchKernelGGL(k_gather_index, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, as_stream(stream), src, index, (size_t)n, out);
    return check_last("gather");
}
extern "C" int cvx_select_f32(const float* mask, const float* a, const float* b, int64_t n, float* out, void* stream) {
    CVX_REQUIRE(mask && a && b && out && n > 0, "cvx_select_f32: bad arguments");
    hipLaunchKernelGGL(k_select, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, as_stream(stream), mask, a, b, (size_t)n, out);
    return check_last("select");
}

extern "C" size_t cvx_box_smooth_workspace_bytes(int C, int H, int W, int D, int passes) {
    return passes > 1 ? 256 + sizeof(float) * (size_t)C * H * W * D : 0;
}
extern "C" int cvx_box_smooth_f32(const float* in, int C, int H, int W, int D, int k, int passes, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(in && out, "cvx_box_smooth_f32: null pointer");
    CVX_REQUIRE(C > 0 && H > 0 && W > 0 && D > 0, "cvx_box_smooth_f32: bad extent");
    CVX_REQUIRE(k >= 1 && (k & 1), "cvx_box_smooth_f32: kernel %d must be odd (the reference's even-kernel path changes the "
                "volume size, convex_adam_MIND.py:185-191)", k);
    CVX_REQUIRE(passes >= 1, "cvx_box_smooth_f32: passes must be >= 1");
    if (workspace_bytes < cvx_box_smooth_workspace_bytes(C, H, W, D, passes))
        return fail(CVX_ERR_WORKSPACE, "cvx_box_smooth_f32: workspace too small");
    CVX_REQUIRE(in != out, "cvx_box_smooth_f32: in-place not supported");
    hipStream_t s = as_stream(stream);
    Carver cv(workspace, workspace_bytes);
    float* tmp = passes > 1 ? cv.take<float>((size_t)C * H * W * D) : nullptr;
    // ping-pong so that the last pass lands in `out`
    const float* src = in;
    for (int p = 0; p < passes; ++p) {
        float* dst = ((passes - 1 - p) & 1) ? tmp : out;
        int rc = launch_box_zero(src, dst, C, H, W, D, k, false, s);
        if (rc) return rc;
        src = dst;
    }
    return CVX_OK;
}

extern "C" int cvx_box_grow_f32(const float* in, int C, int H, int W, int D, int k, float* out, void* stream) {
    CVX_REQUIRE(in && out && in != out, "cvx_box_grow_f32: null pointer or in-place");
    CVX_REQUIRE(C > 0 && H > 0 && W > 0 && D > 0, "cvx_box_grow_f32: bad extent");
    CVX_REQUIRE(k >= 2 && !(k & 1) && k <= 64, "cvx_box_grow_f32: kernel %d must be even, 2 .. 64 (odd kernels: cvx_box_smooth_f32)", k);
    CVX_REQUIRE(k / 2 <= H && k / 2 <= W && k / 2 <= D, "cvx_box_grow_f32: padding %d exceeds the extent (avg_pool3d: pad <= kernel / 2 and input >= pad)", k / 2);
    const size_t n = (size_t)C * (H + 1) * (W + 1) * (D + 1);
    hipLaunchKernelGGL(k_box_grow, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, as_stream(stream), in, out, C, H, W, D, k);
    return check_last("box_grow");
}

extern "C" int cvx_resize_trilinear_f32(const float* in, int C, int h, int w, int d, float* out, int H, int W, int D,
                                        void* stream) {
    CVX_REQUIRE(in && out, "cvx_resize_trilinear_f32: null pointer");
    CVX_REQUIRE(C > 0 && h > 0 && w > 0 && d > 0 && H > 0 && W > 0 && D > 0, "cvx_resize_trilinear_f32: bad extent");
    return launch_resize(in, C, h, w, d, out, H, W, D, 1.0f, 1.0f, as_stream(stream));
}

extern "C" int cvx_grid_sample_f32(const float* vol, int C, int h, int w, int d, const float* grid, int ho, int wo,
                                   int dd, float* out, void* stream) {
    CVX_REQUIRE(vol && grid && out, "cvx_grid_sample_f32: null pointer");
    CVX_REQUIRE(C > 0 && h > 0 && w > 0 && d > 0 && ho > 0 && wo > 0 && dd > 0, "cvx_grid_sample_f32: bad extent");
    const size_t vo = (size_t)ho * wo * dd;
    hipLaunchKernelGGL(k_grid_sample, dim3((unsigned)cdiv64((int64_t)vo, 256)), dim3(256), 0, as_stream(stream), vol, C,
                       h, w, d, grid, vo, out);
    return check_last("grid_sample");
}

extern "C" int cvx_label_histogram_i64(const float* lab, int64_t V, int max_label, int64_t* hist, void* stream) {
    CVX_REQUIRE(lab && hist && V > 0, "cvx_label_histogram_i64: bad arguments");
    CVX_REQUIRE(max_label >= 0 && max_label < 8192, "cvx_label_histogram_i64: max_label %d out of range", max_label);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(hist, 0, sizeof(int64_t) * (size_t)(max_label + 1), s) != hipSuccess)
        return fail(CVX_ERR_LAUNCH, "cvx_label_histogram_i64: memset failed");
    const int nb = (int)(cdiv64(V, 256 * 16) < 1024 ? cdiv64(V, 256 * 16) : 1024);
    hipLaunchKernelGGL(k_label_hist, dim3(nb), dim3(256), sizeof(unsigned) * (size_t)(max_label + 1), s, lab, V, max_label,
                       reinterpret_cast<unsigned long long*>(hist));
    return check_last("label_histogram");
}

// weight = 1/((n_fix + n_mov) + eps).float().pow(.3); weight /= weight.mean()   (convex_adam_nnUNet.py:32-33)
// ---- host-side restatement of torch.pow(float32 tensor, scalar) and torch.sum for the label weights ----------------------------------
// (the same arithmetic as oracle/cvx_oracle.c sp_sleef_powf / torch_inner_sum, whose header explains how it was pinned: ATen evaluates
// the leading blocks of 32 elements with Sleef's powf -- double-float log and exp, fused multiply-adds -- and the trailing n mod 32
// elements with the scalar std::pow(float, double exponent))
namespace {
typedef struct { float x, y; } lw_f2;
static inline float lw_i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t lw_f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline lw_f2 lw_mk(float x, float y) { lw_f2 r = {x, y}; return r; }
static inline lw_f2 lw_dfadd2_f_f(float x, float y) { lw_f2 r; r.x = x + y; float v = r.x - x; r.y = (x - (r.x - v)) + (y - v); return r; }
static inline lw_f2 lw_dfadd2_f2_f(lw_f2 x, float y) { lw_f2 r; r.x = x.x + y; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y - v); r.y = r.y + x.y; return r; }
static inline lw_f2 lw_dfadd2_f2_f2(lw_f2 x, lw_f2 y) { lw_f2 r; r.x = x.x + y.x; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y.x - v); r.y = r.y + (x.y + y.y); return r; }
static inline lw_f2 lw_dfadd_f2_f2(lw_f2 x, lw_f2 y) { lw_f2 r; r.x = x.x + y.x; r.y = x.x - r.x + y.x + x.y + y.y; return r; }
static inline lw_f2 lw_dfadd_f_f2(float x, lw_f2 y) { lw_f2 r; r.x = x + y.x; r.y = x - r.x + y.x + y.y; return r; }
static inline lw_f2 lw_dfmul_f2_f(lw_f2 x, float y) { lw_f2 r; r.x = x.x * y; r.y = fmaf(x.y, y, fmaf(x.x, y, -r.x)); return r; }
static inline lw_f2 lw_dfmul_f2_f2(lw_f2 x, lw_f2 y) { lw_f2 r; r.x = x.x * y.x; r.y = fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -r.x))); return r; }
static inline lw_f2 lw_dfsqu(lw_f2 x) { lw_f2 r; r.x = x.x * x.x; r.y = fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -r.x)); return r; }
static inline lw_f2 lw_dfdiv(lw_f2 n, lw_f2 d) {
    float t = 1.0f / d.x; lw_f2 q; q.x = n.x * t;
    float u = fmaf(t, n.x, -q.x);
    q.y = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
    q.y = fmaf(q.x, q.y, fmaf(n.y, t, u));
    return q;
}
static inline lw_f2 lw_dfscale(lw_f2 d, float s) { return lw_mk(d.x * s, d.y * s); }
static inline lw_f2 lw_dfnormalize(lw_f2 t) { lw_f2 s; s.x = t.x + t.y; s.y = t.x - s.x + t.y; return s; }
static lw_f2 lw_logkf(float d) {
    int o = d < 1.17549435e-38f;
    if (o) d = d * (float)(1LL << 32) * (float)(1LL << 32);
    int e = ((lw_f2i(d * (1.0f / 0.75f)) >> 23) & 0xff) - 0x7f;
    float m = lw_i2f(lw_f2i(d) + ((-e) << 23));
    if (o) e -= 64;
    lw_f2 x = lw_dfdiv(lw_dfadd2_f_f(-1.0f, m), lw_dfadd2_f_f(1.0f, m));
    lw_f2 x2 = lw_dfsqu(x);
    float t = 0.240320354700088500976562f;
    t = fmaf(t, x2.x, 0.285112679004669189453125f);
    t = fmaf(t, x2.x, 0.400007992982864379882812f);
    lw_f2 c = lw_mk(0.66666662693023681640625f, 3.69183861259614332084311e-09f);
    lw_f2 s = lw_dfmul_f2_f(lw_mk(0.69314718246459960938f, -1.904654323148236017e-09f), (float)e);
    s = lw_dfadd_f2_f2(s, lw_dfscale(x, 2.0f));
    s = lw_dfadd_f2_f2(s, lw_dfmul_f2_f2(lw_dfmul_f2_f2(x2, x), lw_dfadd2_f2_f2(lw_dfmul_f2_f(x2, t), c)));
    return s;
}
static float lw_ldexpkf(float x, int q) {
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    m += 127; m = m < 0 ? 0 : m; m = m > 255 ? 255 : m;
    float u = lw_i2f(m << 23);
    x = x * u * u * u * u;
    u = lw_i2f((q + 0x7f) << 23);
    return x * u;
}
static float lw_expkf(lw_f2 d) {
    float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    int q = (int)rintf(u);
    lw_f2 s = lw_dfadd2_f2_f(d, (float)q * -0.693145751953125f);
    s = lw_dfadd2_f2_f(s, (float)q * -1.428606765330187045e-06f);
    s = lw_dfnormalize(s);
    float t = 0.00136324646882712841033936f;
    t = fmaf(t, s.x, 0.00836596917361021041870117f);
    t = fmaf(t, s.x, 0.0416710823774337768554688f);
    t = fmaf(t, s.x, 0.166665524244308471679688f);
    t = fmaf(t, s.x, 0.499999850988388061523438f);
    lw_f2 tt = lw_dfadd_f2_f2(s, lw_dfmul_f2_f(lw_dfsqu(s), t));
    tt = lw_dfadd_f_f2(1.0f, tt);
    u = tt.x + tt.y;
    u = lw_ldexpkf(u, q);
    if (d.x < -104.0f) u = 0.0f;
    return u;
}
static float lw_sleef_powf(float x, float y) {          // x > 0 only
    return lw_expkf(lw_dfmul_f2_f(lw_logkf(fabsf(x)), y));
}
// torch.sum of n <= 255 contiguous floats on one thread (ATen SumKernel, 8-float vectors): the scalar tail first, then per lane four
// interleaved partial sums of the vectors (vector i -> partial i mod 4, leftover vectors -> partial 0, ((p0 + p1) + p2) + p3), lanes in
// order; fewer than 8 elements: the same scheme on scalars (the cascade levels start at 16 values per partial: never reached here).
static float torch_sum_small(const float* x, int n) {
    auto strided = [](const float* v, int stride, int size) { float a = 0.0f; for (int i = 0; i < size; ++i) a += v[i * stride]; return a; };
    if (n < 8) {
        const int n4 = n / 4;
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = strided(x + k, 4, n4);
        for (int i = n4 * 4; i < n; ++i) p[0] += x[i];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        return 0.0f + p[0];
    }
    const int nv = n / 8, nv4 = nv / 4;
    float fin = 0.0f;
    for (int k = nv * 8; k < n; ++k) fin += x[k];
    for (int lane = 0; lane < 8; ++lane) {
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = strided(x + k * 8 + lane, 32, nv4);
        for (int i = nv4 * 4; i < nv; ++i) p[0] += x[i * 8 + lane];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        fin += p[0];
    }
    return 0.0f + fin;
}
}  // namespace

extern "C" int cvx_label_weights_host(const int64_t* hist_fix_host, const int64_t* hist_mov_host, int max_label,
                                      int* present_host, float* weights_host) {
    int C = 0;
    for (int l = 0; l <= max_label; ++l)
        if (hist_fix_host[l] + hist_mov_host[l] > 0) present_host[C++] = l;
    // weight = 1 / (n_fix + n_mov + eps).float().pow(.3) ; weight /= weight.mean()          (convex_adam_nnUNet.py:31-32)
    for (int c = 0; c < C; ++c) {
        const float cnt = (float)(hist_fix_host[present_host[c]] + hist_mov_host[present_host[c]]) + 1e-32f;
        const int blk = options().label_pow_block > 0 ? (int)options().label_pow_block : 32;      // 2 x the vector width of the reference host's ATen build
        const float pw = c < (C / blk) * blk ? lw_sleef_powf(cnt, 0.3f) : (float)pow((double)cnt, 0.3);
        weights_host[c] = 1.0f / pw;
    }
    const float mean = torch_sum_small(weights_host, C) / (float)C;
    for (int c = 0; c < C; ++c) weights_host[c] = weights_host[c] / mean;
    return C;
}

extern "C" int cvx_label_features_f32(const float* lab, int64_t V, int C, const int* present, const float* weights,
                                      float mult, float* feat, void* stream) {
    CVX_REQUIRE(lab && present && weights && feat && V > 0 && C > 0, "cvx_label_features_f32: bad arguments");
    hipLaunchKernelGGL(k_label_features, dim3((unsigned)cdiv64(V, 256)), dim3(256), 0, as_stream(stream), lab, V, C, present,
                       weights, mult, feat);
    return check_last("label_features");
}


// fp16 storage of a float32 buffer: x = float(half(x)), round to nearest even
namespace cvx {
__global__ __launch_bounds__(256) void k_round_f16(float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __half2float(__float2half_rn(x[i]));
}
}  // namespace cvx
extern "C" int cvx_round_f16_f32(float* x, int64_t n, void* stream) {
    CVX_REQUIRE(x && n >= 0, "cvx_round_f16_f32: bad arguments");
    if (n > 0) hipLaunchKernelGGL(cvx::k_round_f16, dim3((unsigned)cvx::cdiv64(n, 256)), dim3(256), 0, cvx::as_stream(stream), x, n);
    return cvx::check_last("round_f16");
}


// ---- output packing of convex_adam_pt (reference convex_adam_MIND.py:198-202) ---------------------------------
//   x = disp_hr[0,0].cpu().to(dtype).numpy() ... ; displacements = np.stack((x,y,z),3).astype(float)
// field [3][H][W][D] float32 -> out [H][W][D][3] float64, every value passed through `dtype` first (quantize: 0 = float32, 1 = float16
// round to nearest even).  `out` may be DEVICE memory or PINNED HOST memory mapped into the device's address space (hipHostMalloc /
// torch pin_memory): the kernel then writes the 24 bytes of a voxel straight across PCIe -- no device copy of the permuted field, no
// pageable download, no single-threaded widening on the host.  One thread per voxel: three coalesced 4-byte reads (one per channel),
// one contiguous 24-byte write; a wavefront writes 1536 contiguous bytes.
namespace cvx {
// One workgroup packs 256 voxels = 768 consecutive doubles of the output: the three channel values of a voxel meet in LDS, and every
// thread then stores three 8-byte values at lane-consecutive addresses (a wavefront writes 512 contiguous bytes per store: whole
// lines for the PCIe write combiner when `out` is host memory).
__global__ __launch_bounds__(256) void k_pack_field_f64(const float* __restrict__ f, size_t V, int quantize, double* __restrict__ out) {
    __shared__ float sv[768];
    const size_t p0 = (size_t)blockIdx.x * 256, p = p0 + threadIdx.x;
    if (p < V) {
        float a = f[p], b = f[V + p], c = f[2 * V + p];
        if (quantize == 1) { a = __half2float(__float2half_rn(a)); b = __half2float(__float2half_rn(b)); c = __half2float(__float2half_rn(c)); }
        sv[3 * threadIdx.x] = a; sv[3 * threadIdx.x + 1] = b; sv[3 * threadIdx.x + 2] = c;
    }
    cvx_barrier();
    const size_t n = min((size_t)768, 3 * (V - p0));
    double* o = out + 3 * p0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const unsigned i = threadIdx.x + 256 * r;
        if (i < n) o[i] = (double)sv[i];
    }
}
}  // namespace cvx
extern "C" int cvx_pack_field_f64(const float* field, int H, int W, int D, int quantize, double* out, void* stream) {
    CVX_REQUIRE(field && out && H > 0 && W > 0 && D > 0, "cvx_pack_field_f64: bad arguments");
    CVX_REQUIRE(quantize == 0 || quantize == 1, "cvx_pack_field_f64: quantize must be 0 (float32) or 1 (float16)");
    const size_t V = (size_t)H * W * D;
    // (a persistent grid of 256 / 1024 workgroups was measured too: 4.4-4.6 ms instead of 3.0 ms for the 165 MB of the benchmark field -- a
    // full grid keeps more PCIe writes in flight)
    hipLaunchKernelGGL(cvx::k_pack_field_f64, dim3((unsigned)cvx::cdiv64((int64_t)V, 256)), dim3(256), 0, cvx::as_stream(stream), field, V, quantize, out);
    return cvx::check_last("pack_field");
}

