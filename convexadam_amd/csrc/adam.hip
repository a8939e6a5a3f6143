// adam.hip -- Adam instance optimisation (reference: convex_adam_MIND.py:155-182).
//
// Per iteration (all on one stream; P, m, v, U, G are [3][h][w][d] float32):
//   U  = box3(box3(box3(P)))                  zero pad, raster 27-tap sums, /27            (:166)
//   gU = d/dU [ mean_x(mean_c((warp(M2)(x) - F2(x))^2) * cost_scale) + lambda * diffusion(U) ]
//        -- k_warp_grad: one thread per control point gathers 8 corners x C channels of M2, evaluates
//        ATen's grid_sampler_3d_backward expressions channel by channel, divides by the
//        normalisation scale, and adds the six one-sided regulariser terms in autograd's arrival
//        order (data, D[:-1], D[1:], H[:-1], H[1:], W[:-1], W[1:])                        (:167-178)
//   G  = box3^T(box3^T(box3^T(gU)))           ATen avg_pool3d_backward order (sum of tap/27)
//   Adam(lr=1, betas=(.9,.999), eps=1e-8): m = fma(.1, g-m, m); v = fma(.001*g, g, v*.999);
//        P += (-(1/bc1) * m) / (sqrt(v)/sqrt(bc2) + eps)                                   (:179)
// The loop returns U of the LAST forward pass (:181), i.e. parameters after niter-1 updates.
// Roofline: HBM/L2 -- 185.8 MB algorithmic traffic per iteration at OASIS size; the
// working set (F2, M2 = 2 x 41 MB) stays resident in the 256 MiB Infinity Cache across iterations.
#include <math.h>

#include "cvx_common.h"

namespace cvx {

// ---- three chained 3^3 box filters in one launch ---------------------------------------------------------
// One workgroup = one channel x one 8x8x32 output tile.  The input tile (+3 halo rows/planes, columns x0-4 ..
// x0+35, aligned float4 global loads) is staged in LDS; the passes shrink the region by one voxel each in z,y
// (14 -> 12 -> 10 -> 8 rows) and along x (38 -> 36 -> 34 columns needed).  Every intermediate is zero outside the
// VOLUME (each avg_pool3d zero-pads its own input).
//   forward  (ATen avg_pool3d):           out = (raster sum of 27 taps) / 27            at every pass
//   backward (ATen avg_pool3d_backward):  out = raster sum of (tap / 27): taps are divided once when they
//                                          are staged, the last pass stores the plain sum
// With ADAM the last backward pass applies the Adam update to P, m, v in place instead of storing G.
// One thread evaluates a PAIR of adjacent columns from 9 aligned 16-byte windows [c-1 .. c+2], read as two
// ds_read_b64: per tap row two packed adds + two scalar adds, no cross-lane traffic, no bank conflicts.  Each
// pass stores its result shifted by one more index so that the next pass's windows are aligned again:
//   "column" c = x - x0 + 8;  A holds the input at index c, B holds pass 1 at index c+1, A then pass 2 at c+2.
constexpr int BT_Z = 8, BT_Y = 8, BT_X = 32, BT_NT = 512, BT_PX = 48;

// PASS 1: pairs (c, c+1), c = 5 + 2p, p < 19  (needs columns 6..41)   src shift 0 -> dst shift 1
// PASS 2: pairs (c, c+1), c = 6 + 2p, p < 18  (needs columns 7..40)   src shift 1 -> dst shift 2
// PASS 3: pairs (c, c+1), c = 7 + 2p, p < 17  (outputs columns 8..39) src shift 2 -> global
template <int PASS, bool BACKWARD, bool ADAM>
__device__ __forceinline__ void box_pass(const float* __restrict__ src, int sy, float* __restrict__ dst, int dz, int dy,
                                         int gz0, int gy0, int x0, int h, int w, int d, float* __restrict__ gout,
                                         float* __restrict__ P, float* __restrict__ m, float* __restrict__ v, AdamConsts ac,
                                         float* __restrict__ gsave) {
    constexpr int NP = 20 - PASS, C0 = 4 + PASS, SH = PASS - 1;         // pairs per row, first column, source shift
    const int nr = dz * dy * NP;
    for (int r = threadIdx.x; r < nr; r += BT_NT) {
        const int p = r % NP, row = r / NP;
        const int y = row % dy, z = row / dy;
        const int c = C0 + 2 * p;                                        // outputs: columns c, c+1
        const float* win = src + ((z + 1) * sy + (y + 1)) * BT_PX + (c - 1 + SH);   // window = columns c-1 .. c+2
        f32x2 s = {0.0f, 0.0f};
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int b = -1; b <= 1; ++b) {
                const float* q = win + (a * sy + b) * BT_PX;
                const f32x2 lo = lds_load2(q), hi = lds_load2(q + 2);
                s += lo;                 // s0 += t0 ; s1 += t1
                s.x += lo.y;             // s0 += t1
                s.y += hi.x;             // s1 += t2
                s += hi;                 // s0 += t2 ; s1 += t3
            }
        const int gz = gz0 + z, gy = gy0 + y, gx = x0 - 8 + c;
        const bool rowin = gz >= 0 && gz < h && gy >= 0 && gy < w;
        if (PASS < 3) {
            f32x2 o;
            o.x = (rowin && gx >= 0 && gx < d) ? div_exact<27>(s.x) : 0.0f;
            o.y = (rowin && gx + 1 >= 0 && gx + 1 < d) ? div_exact<27>(s.y) : 0.0f;
            lds_store2(dst + (z * dy + y) * BT_PX + (c + SH + 1), o);
        } else if (rowin) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cc = c + j;
                if (cc < 8 || cc > 39 || gx + j >= d) continue;           // columns 8..39 are the output tile
                const size_t i = ((size_t)gz * w + gy) * d + gx + j;
                const float sj = j ? s.y : s.x;
                if (!BACKWARD) gout[i] = div_exact<27>(sj);
                else if (!ADAM) gout[i] = sj;
                else {
                    const float g = sj;
                    const float mo = m[i];
                    const float mm = __builtin_fmaf(ac.w1, g - mo, mo);          // exp_avg.lerp_(grad, 1-beta1)
                    float vv = v[i] * ac.b2;                                      // exp_avg_sq.mul_(beta2)
                    vv = __builtin_fmaf(ac.omb2 * g, g, vv);                      // .addcmul_(grad, grad, value=1-beta2)
                    const float den = fdiv(adam_sqrt(vv, ac.sqrt_tbl), ac.bc2s) + 1e-8f;   // (sqrt / bias_correction2_sqrt).add_(eps)
                    P[i] = P[i] + fdiv(ac.neg_step * mm, den);                    // addcdiv_(exp_avg, denom, value=-step_size)
                    m[i] = mm;
                    v[i] = vv;
                    if (gsave) gsave[i] = g;
                }
            }
        }
    }
}

template <bool BACKWARD, bool ADAM>
__global__ __launch_bounds__(BT_NT) void k_box3x3(const float* __restrict__ in, float* __restrict__ out, int h, int w, int d,
                                                  float* __restrict__ P, float* __restrict__ m, float* __restrict__ v,
                                                  AdamConsts ac, float* __restrict__ gsave) {
    __shared__ __attribute__((aligned(16))) float A[(BT_Z + 6) * (BT_Y + 6) * BT_PX];
    __shared__ __attribute__((aligned(16))) float B[(BT_Z + 4) * (BT_Y + 4) * BT_PX];
    const int ntx = (d + BT_X - 1) / BT_X, nty = (w + BT_Y - 1) / BT_Y, ntz = (h + BT_Z - 1) / BT_Z;
    int b = blockIdx.x;
    const int tx = b % ntx; b /= ntx;
    const int ty = b % nty; b /= nty;
    const int tz = b % ntz; const int c = b / ntz;
    const int x0 = tx * BT_X, y0 = ty * BT_Y, z0 = tz * BT_Z;
    const size_t V = (size_t)h * w * d;
    const float* ic = in + (size_t)c * V;
    // stage columns x0-4 .. x0+35 (LDS columns 4..43) of rows z0-3.., y0-3..; zero outside the volume
    constexpr int AZ = BT_Z + 6, AY = BT_Y + 6, NCH = (BT_X + 8) / 4;
    const bool vec = (d & 3) == 0;
    for (int i = threadIdx.x; i < AZ * AY * NCH; i += BT_NT) {
        const int ch = i % NCH, y = (i / NCH) % AY, z = i / (NCH * AY);
        const int gz = z0 - 3 + z, gy = y0 - 3 + y, gx = x0 - 4 + 4 * ch;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gz >= 0 && gz < h && gy >= 0 && gy < w) {
            const float* rowp = ic + ((size_t)gz * w + gy) * d;
            if (vec && gx >= 0 && gx + 3 < d) q = *reinterpret_cast<const float4*>(rowp + gx);
            else {
                if (gx >= 0 && gx < d) q.x = rowp[gx];
                if (gx + 1 >= 0 && gx + 1 < d) q.y = rowp[gx + 1];
                if (gx + 2 >= 0 && gx + 2 < d) q.z = rowp[gx + 2];
                if (gx + 3 >= 0 && gx + 3 < d) q.w = rowp[gx + 3];
            }
            // backward: every tap is gradOut / 27 (the only place where the dividend may be -0.0 -> IEEE division)
            if (BACKWARD) { q.x = fdiv(q.x, 27.0f); q.y = fdiv(q.y, 27.0f); q.z = fdiv(q.z, 27.0f); q.w = fdiv(q.w, 27.0f); }
        }
        *reinterpret_cast<float4*>(A + (z * AY + y) * BT_PX + 4 + 4 * ch) = q;
    }
    cvx_barrier();
    float* oc = out ? out + (size_t)c * V : nullptr;
    float* Pc = P ? P + (size_t)c * V : nullptr;
    float* mc = m ? m + (size_t)c * V : nullptr;
    float* vc = v ? v + (size_t)c * V : nullptr;
    float* gs = gsave ? gsave + (size_t)c * V : nullptr;
    // pass 1: A (14x14 rows) -> B (12x12 rows); pass 2: B -> A (10x10 rows); pass 3: A -> tile
    box_pass<1, BACKWARD, ADAM>(A, AY, B, BT_Z + 4, BT_Y + 4, z0 - 2, y0 - 2, x0, h, w, d, nullptr, nullptr, nullptr, nullptr, ac, nullptr);
    cvx_barrier();
    box_pass<2, BACKWARD, ADAM>(B, BT_Y + 4, A, BT_Z + 2, BT_Y + 2, z0 - 1, y0 - 1, x0, h, w, d, nullptr, nullptr, nullptr, nullptr, ac, nullptr);
    cvx_barrier();
    box_pass<3, BACKWARD, ADAM>(A, BT_Y + 2, nullptr, BT_Z, BT_Y, z0, y0, x0, h, w, d, oc, Pc, mc, vc, ac, gs);
}

__global__ __launch_bounds__(256) void k_adam_update(const float* __restrict__ G, float* __restrict__ P, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, AdamConsts ac) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = G[i];
    const float mo = m[i];
    const float mm = __builtin_fmaf(ac.w1, g - mo, mo);          // exp_avg.lerp_(grad, 1-beta1)
    float vv = v[i] * ac.b2;                                      // exp_avg_sq.mul_(beta2)
    vv = __builtin_fmaf(ac.omb2 * g, g, vv);                      // .addcmul_(grad, grad, value=1-beta2)
    const float den = fdiv(adam_sqrt(vv, ac.sqrt_tbl), ac.bc2s) + 1e-8f;   // (sqrt / bias_correction2_sqrt).add_(eps)
    P[i] = P[i] + fdiv(ac.neg_step * mm, den);                    // addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mm;
    v[i] = vv;
}

static int launch_box3x3(const float* in, float* out, int h, int w, int d, bool backward, float* P, float* m, float* v,
                         AdamConsts ac, float* gsave, hipStream_t s) {
    // rows of up to 126 voxels: z-marching pipeline (boxmarch.hip); longer rows: the tiled kernel below
    const bool force_tiled = options().box_tiled != 0;
    if (!force_tiled && (backward ? (out || P) && box3_tile_supported(in, out, h, w, d, P, m, v, gsave) : box3_tile_fwd_supported(in, out, h, w, d))) {
        long long ft = backward ? options().box_bwd_tile : options().box_fwd_tile;
        if (ft < 0) ft = box3_march_supported(d) ? box3_tile_fwd_auto(h, w, d) : 2000;      // (rows beyond the marching kernel's range: always tiles)
        if (ft >= 1000) return launch_box3_tile(in, out, h, w, d, (int)ft, backward, P, m, v, ac, gsave, s);
    }
    if (!force_tiled && box3_march_supported(d)) return launch_box3_march(in, out, h, w, d, backward, P, m, v, ac, gsave, s);
    const int nb = cdiv(d, BT_X) * cdiv(w, BT_Y) * cdiv(h, BT_Z) * 3;
    if (!backward) hipLaunchKernelGGL((k_box3x3<false, false>), dim3(nb), dim3(BT_NT), 0, s, in, out, h, w, d, P, m, v, ac, gsave);
    else if (!P) hipLaunchKernelGGL((k_box3x3<true, false>), dim3(nb), dim3(BT_NT), 0, s, in, out, h, w, d, P, m, v, ac, gsave);
    else hipLaunchKernelGGL((k_box3x3<true, true>), dim3(nb), dim3(BT_NT), 0, s, in, out, h, w, d, P, m, v, ac, gsave);
    return check_last("box3x3");
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_adam_workspace_bytes(int C, int h, int w, int d) {
    const size_t V = (size_t)h * w * d, CP = (size_t)(C + 3) / 4 * 4;
    return 3 * (256 + sizeof(float) * 3 * V) + 2 * (256 + sizeof(float) * CP * (V + 1)) + 256;
}

extern "C" int cvx_adam_run_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                const float* base_w, const float* base_d, float* U, float* grad_out,
                                const int* snapshot_iters_host, int n_snap, float* snapshots, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return cvx_adam_run_smoother_f32(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U,
                                     grad_out, snapshot_iters_host, n_snap, snapshots, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cvx_adam_run_smoother_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                         float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                         const float* base_w, const float* base_d, float* U, float* grad_out,
                                         const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return cvx::adam_run_impl(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U, grad_out,
                              snapshot_iters_host, n_snap, snapshots, sm, true, false, 0, workspace, workspace_bytes, stream);
}

extern "C" int cvx_adam_run_fast_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                     float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                     const float* base_w, const float* base_d, float* U, float* grad_out,
                                     const int* snapshot_iters_host, int n_snap, float* snapshots, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return cvx::adam_run_impl(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U, grad_out,
                              snapshot_iters_host, n_snap, snapshots, nullptr, true, false, 1, workspace, workspace_bytes, stream);
}

extern "C" int cvx_adam_run_fast_all_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                         float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                         const float* base_w, const float* base_d, float* U, float* grad_out,
                                         const int* snapshot_iters_host, int n_snap, float* snapshots, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    return cvx::adam_run_impl(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U, grad_out,
                              snapshot_iters_host, n_snap, snapshots, nullptr, true, false, 2, workspace, workspace_bytes, stream);
}

extern "C" int cvx_adam_run_mode_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                     float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                     const float* base_w, const float* base_d, float* U, float* grad_out,
                                     const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm, int mode,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE((mode & ~16) >= 0 && (mode & ~16) <= 2, "cvx_adam_run_mode_f32: mode must be 0 (exact), 1 (fast) or 2 (fast_all), + 16 for half-precision feature records");
    return cvx::adam_run_impl(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U, grad_out,
                              snapshot_iters_host, n_snap, snapshots, sm, true, (mode & 16) != 0, mode & 3, workspace, workspace_bytes, stream);
}

extern "C" int cvx_smooth_fast_f32(const float* in, int h, int w, int d, const cvx_smoother* sm, int backward, float* out, void* stream) {
    CVX_REQUIRE(in && out && sm && h > 0 && w > 0 && d > 0, "cvx_smooth_fast_f32: bad arguments");
    CVX_REQUIRE(sm->kind == 0, "cvx_smooth_fast_f32: box chains only (a Gaussian is three short 1-D convolutions already: cvx_smooth_f32)");
    return cvx::launch_boxchain_fast(in, out, h, w, d, *sm, backward != 0, as_stream(stream));
}

extern "C" int cvx_box3_fast_f32(const float* in, int h, int w, int d, float* out, void* stream) {
    CVX_REQUIRE(in && out && in != out && h > 0 && w > 0 && d > 0, "cvx_box3_fast_f32: bad arguments");
    return cvx::launch_box3_fast(in, out, h, w, d, nullptr, nullptr, nullptr, 1.0, 1.0, nullptr, as_stream(stream));
}

extern "C" int cvx_adam_run_ex_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                   float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                   const float* base_w, const float* base_d, float* U, float* grad_out,
                                   const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm,
                                   int feature_storage, void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(feature_storage == 0 || feature_storage == 1, "cvx_adam_run_ex_f32: feature_storage must be 0 (float32) or 1 (fp16)");
    return cvx::adam_run_impl(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, base_h, base_w, base_d, U, grad_out,
                              snapshot_iters_host, n_snap, snapshots, sm, true, feature_storage == 1, 0, workspace, workspace_bytes, stream);
}

// keep_state = false (whole-pair pipeline): P, m, v are scratch there and the result is U of the LAST forward pass
// (convex_adam_MIND.py:181), so the gradient and the Adam step of the final iteration are never observed and are skipped.
int cvx::adam_run_impl(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                       float lambda_weight, int niter, int step0, float cost_scale, const float* base_h, const float* base_w,
                       const float* base_d, float* U, float* grad_out, const int* snapshot_iters_host, int n_snap,
                       float* snapshots, const cvx_smoother* sm, bool keep_state, bool f16_features, int fast, void* workspace,
                       size_t workspace_bytes, void* stream, bool features_are_records) {
    CVX_REQUIRE(F2 && M2 && P && m && v && U && base_h && base_w && base_d, "cvx_adam_run_f32: null pointer");
    CVX_REQUIRE(C > 0 && h > 1 && w > 1 && d > 1, "cvx_adam_run_f32: bad extent C=%d %dx%dx%d", C, h, w, d);
    CVX_REQUIRE(niter >= 0 && step0 >= 0, "cvx_adam_run_f32: negative iteration count");
    CVX_REQUIRE(n_snap == 0 || (snapshot_iters_host && snapshots), "cvx_adam_run_f32: snapshot buffers missing");
    if (!workspace || workspace_bytes < cvx_adam_workspace_bytes(C, h, w, d))
        return fail(CVX_ERR_WORKSPACE, "cvx_adam_run_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const size_t V = (size_t)h * w * d;
    Carver cv(workspace, workspace_bytes);
    float* gU = cv.take<float>(3 * V);
    float* t1 = cv.take<float>(3 * V);
    float* t2 = cv.take<float>(3 * V);
    // generic smoother path unless it is the packaged chain of three 3^3 boxes (fused LDS kernels)
    const bool fused = !sm || (sm->kind == 0 && sm->n_boxes == 3 && sm->box_k[0] == 3 && sm->box_k[1] == 3 && sm->box_k[2] == 3);
    if (sm) {
        CVX_REQUIRE(sm->kind == 0 || sm->kind == 1, "cvx_adam_run_smoother_f32: smoother kind must be 0 or 1");
        if (sm->kind == 0) {
            CVX_REQUIRE(sm->n_boxes >= 1 && sm->n_boxes <= 4, "cvx_adam_run_smoother_f32: n_boxes must be 1..4");
            for (int i = 0; i < sm->n_boxes; ++i) CVX_REQUIRE(sm->box_k[i] >= 1 && (sm->box_k[i] & 1), "cvx_adam_run_smoother_f32: box size must be odd");
        }
    }
    if (fast && !fused && sm->kind == 0 && !boxchain_fast_supported(*sm, h, w, d))
        return fail(CVX_ERR_UNSUPPORTED, "adam_mode fast: box chain outside the separable kernel's range (odd sizes <= 9, lines of at most 320 voxels)");
    const int CP = (C + 3) / 4 * 4;
    float* Fcl = cv.take<float>((size_t)CP * (V + 1));
    float* Mcl = cv.take<float>((size_t)CP * (V + 1));
    if (features_are_records) { Fcl = const_cast<float*>(F2); Mcl = const_cast<float*>(M2); }       // built by the producer (mind.hip::k_mind_finish_pool)
    else if (niter > 0) {
        int rc;
        if ((rc = launch_to_chunked(F2, C, V, Fcl, f16_features, s)) || (rc = launch_to_chunked(M2, C, V, Mcl, f16_features, s))) return rc;
    }

    // MeanBackward of lambda*mean(diff^2): lambda / N_axis in float32                       (:167-169)
    const float nH = (float)((int64_t)3 * (h - 1) * w * d), nW = (float)((int64_t)3 * h * (w - 1) * d),
                nD = (float)((int64_t)3 * h * w * (d - 1));
    const float cH = lambda_weight / nH, cW = lambda_weight / nW, cD = lambda_weight / nD;
    const float gsc = ((1.0f / (float)V) * cost_scale) / (float)C;     // MeanBackward, MulBackward, MeanBackward
    int snap = 0;
    for (int it = 0; it < niter; ++it) {
        int rc;
        const int step = step0 + it + 1;
        const double beta1 = 0.9, beta2 = 0.999;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        const AdamConsts ac = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)(-(1.0 / bc1)), adam_sqrt_table()};
        if (fast == 2 && fused) { if ((rc = launch_box3_fast(P, U, h, w, d, nullptr, nullptr, nullptr, 1.0, 1.0, nullptr, s))) return rc; }   // "fast_all": separable forward boxes too
        else if (fast == 2 && sm->kind == 0) { if ((rc = launch_boxchain_fast(P, U, h, w, d, *sm, false, s))) return rc; }             // ... for a box chain of the sweep
        else if (fused) { if ((rc = launch_box3x3(P, U, h, w, d, false, nullptr, nullptr, nullptr, ac, nullptr, s))) return rc; }
        else if ((rc = launch_smoother(P, U, t1, 3, h, w, d, *sm, false, s))) return rc;
        profile_mark_kernel("adam.forward_boxes", s);
        const bool last = it == niter - 1;
        if (!(last && !keep_state && !grad_out)) {
        float* gsave = (grad_out && it == niter - 1) ? grad_out : nullptr;
        if (fast) {
            if ((rc = launch_warp_grad_fast(Fcl, Mcl, C, h, w, d, U, base_h, base_w, base_d, gsc, cH, cW, cD, gU, f16_features, s))) return rc;
            profile_mark_kernel("adam.warp_gradient", s);
            if (fused) { if ((rc = launch_box3_fast(gU, nullptr, h, w, d, P, m, v, bc1, bc2, gsave, s))) return rc; }
            else {
                // sweep smoothers: a box chain through the separable passes (adjoint = reversed box order), a Gaussian through its exact
                // 1-D convolutions; the update as an element-wise kernel
                if (sm->kind == 0) { if ((rc = launch_boxchain_fast(gU, t2, h, w, d, *sm, true, s))) return rc; }
                else if ((rc = launch_smoother(gU, t2, t1, 3, h, w, d, *sm, true, s))) return rc;
                if ((rc = launch_adam_update_fast(t2, P, m, v, 3 * V, bc1, bc2, s))) return rc;
                if (gsave) (void)hipMemcpyAsync(gsave, t2, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice, s);
            }
        } else {
        if ((rc = launch_warp_grad(Fcl, Mcl, C, h, w, d, U, base_h, base_w, base_d, gsc, cH, cW, cD, gU, f16_features, s))) return rc;
        profile_mark_kernel("adam.warp_gradient", s);
        if (fused) { if ((rc = launch_box3x3(gU, nullptr, h, w, d, true, P, m, v, ac, gsave, s))) return rc; }
        else {
            if ((rc = launch_smoother(gU, t2, t1, 3, h, w, d, *sm, true, s))) return rc;
            hipLaunchKernelGGL(k_adam_update, dim3((unsigned)cdiv64((int64_t)(3 * V), 256)), dim3(256), 0, s, t2, P, m, v, 3 * V, ac);
            if (gsave) (void)hipMemcpyAsync(gsave, t2, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice, s);
        }
        }
        }
        if (!(last && !keep_state && !grad_out)) profile_mark_kernel("adam.adjoint_update", s);
        while (snap < n_snap && snapshot_iters_host[snap] == it + 1) {
            (void)hipMemcpyAsync(snapshots + (size_t)snap * 3 * V, U, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice, s);
            ++snap;
        }
    }
    return check_last("adam_run");
}
