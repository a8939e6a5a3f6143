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

__global__ __launch_bounds__(256) void k_warp_grad(const float* __restrict__ F2, const float* __restrict__ M2, int C, int h,
                                                   int w, int d, const float* __restrict__ U, const float* __restrict__ bh,
                                                   const float* __restrict__ bw, const float* __restrict__ bd, float gsc,
                                                   float cH, float cW, float cD, float* __restrict__ gU) {
    const size_t V = (size_t)h * w * d;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= V) return;
    const int x = (int)(p % d), y = (int)((p / d) % w), z = (int)(p / ((size_t)d * w));
    const float sc0 = (float)((h - 1) / 2.0), sc1 = (float)((w - 1) / 2.0), sc2 = (float)((d - 1) / 2.0);   // (:171)
    const float uH = U[p], uW = U[V + p], uD = U[2 * V + p];
    Tri t;
    tri_setup(t, bd[x] + fdiv(uD, sc2), bw[y] + fdiv(uW, sc1), bh[z] + fdiv(uH, sc0), h, w, d);
    const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float fx0 = (float)x0, fy0 = (float)y0, fz0 = (float)z0, fx1 = (float)x1, fy1 = (float)y1, fz1 = (float)z1;
    const bool b000 = inb3(z0, y0, x0, h, w, d), b001 = inb3(z0, y0, x1, h, w, d), b010 = inb3(z0, y1, x0, h, w, d),
               b011 = inb3(z0, y1, x1, h, w, d), b100 = inb3(z1, y0, x0, h, w, d), b101 = inb3(z1, y0, x1, h, w, d),
               b110 = inb3(z1, y1, x0, h, w, d), b111 = inb3(z1, y1, x1, h, w, d);
    const size_t i000 = ((size_t)z0 * w + y0) * d + x0, i001 = i000 + 1, i010 = i000 + d, i011 = i010 + 1,
                 i100 = i000 + (size_t)w * d, i101 = i100 + 1, i110 = i100 + d, i111 = i110 + 1;
    float gix = 0.f, giy = 0.f, giz = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* mv = M2 + (size_t)c * V;
        const float v000 = b000 ? mv[i000] : 0.f, v001 = b001 ? mv[i001] : 0.f, v010 = b010 ? mv[i010] : 0.f,
                    v011 = b011 ? mv[i011] : 0.f, v100 = b100 ? mv[i100] : 0.f, v101 = b101 ? mv[i101] : 0.f,
                    v110 = b110 ? mv[i110] : 0.f, v111 = b111 ? mv[i111] : 0.f;
        // forward sample, corners accumulated in ATen order (skipped corners add nothing)
        float wv = 0.0f;
        if (b000) wv += v000 * t.tnw;
        if (b001) wv += v001 * t.tne;
        if (b010) wv += v010 * t.tsw;
        if (b011) wv += v011 * t.tse;
        if (b100) wv += v100 * t.bnw;
        if (b101) wv += v101 * t.bne;
        if (b110) wv += v110 * t.bsw;
        if (b111) wv += v111 * t.bse;
        const float df = wv - F2[(size_t)c * V + p];
        const float gOut = gsc * (2.0f * df);                      // PowBackward0: grad * (2 * self)
        if (b000) { gix -= v000 * (fy1 - t.iy) * (fz1 - t.iz) * gOut; giy -= v000 * (fx1 - t.ix) * (fz1 - t.iz) * gOut; giz -= v000 * (fx1 - t.ix) * (fy1 - t.iy) * gOut; }
        if (b001) { gix += v001 * (fy1 - t.iy) * (fz1 - t.iz) * gOut; giy -= v001 * (t.ix - fx0) * (fz1 - t.iz) * gOut; giz -= v001 * (t.ix - fx0) * (fy1 - t.iy) * gOut; }
        if (b010) { gix -= v010 * (t.iy - fy0) * (fz1 - t.iz) * gOut; giy += v010 * (fx1 - t.ix) * (fz1 - t.iz) * gOut; giz -= v010 * (fx1 - t.ix) * (t.iy - fy0) * gOut; }
        if (b011) { gix += v011 * (t.iy - fy0) * (fz1 - t.iz) * gOut; giy += v011 * (t.ix - fx0) * (fz1 - t.iz) * gOut; giz -= v011 * (t.ix - fx0) * (t.iy - fy0) * gOut; }
        if (b100) { gix -= v100 * (fy1 - t.iy) * (t.iz - fz0) * gOut; giy -= v100 * (fx1 - t.ix) * (t.iz - fz0) * gOut; giz += v100 * (fx1 - t.ix) * (fy1 - t.iy) * gOut; }
        if (b101) { gix += v101 * (fy1 - t.iy) * (t.iz - fz0) * gOut; giy -= v101 * (t.ix - fx0) * (t.iz - fz0) * gOut; giz += v101 * (t.ix - fx0) * (fy1 - t.iy) * gOut; }
        if (b110) { gix -= v110 * (t.iy - fy0) * (t.iz - fz0) * gOut; giy += v110 * (fx1 - t.ix) * (t.iz - fz0) * gOut; giz += v110 * (fx1 - t.ix) * (t.iy - fy0) * gOut; }
        if (b111) { gix += v111 * (t.iy - fy0) * (t.iz - fz0) * gOut; giy += v111 * (t.ix - fx0) * (t.iz - fz0) * gOut; giz += v111 * (t.ix - fx0) * (t.iy - fy0) * gOut; }
    }
    // grad wrt the normalised grid (x,y,z) = (size/2)*gi ; flip ; / scale -> grad wrt U (H,W,D)
    float g[3];
    g[0] = fdiv(((float)h / 2.0f) * giz, sc0);
    g[1] = fdiv(((float)w / 2.0f) * giy, sc1);
    g[2] = fdiv(((float)d / 2.0f) * gix, sc2);
    const size_t sH = (size_t)w * d;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* Ua = U + (size_t)a * V;
        const float uc = Ua[p];
        float acc = g[a];
        if (x < d - 1) acc += -(cD * (2.0f * (Ua[p + 1] - uc)));
        if (x > 0)     acc +=  (cD * (2.0f * (uc - Ua[p - 1])));
        if (z < h - 1) acc += -(cH * (2.0f * (Ua[p + sH] - uc)));
        if (z > 0)     acc +=  (cH * (2.0f * (uc - Ua[p - sH])));
        if (y < w - 1) acc += -(cW * (2.0f * (Ua[p + d] - uc)));
        if (y > 0)     acc +=  (cW * (2.0f * (uc - Ua[p - d])));
        gU[(size_t)a * V + p] = acc;
    }
}

__global__ __launch_bounds__(256) void k_adam_update(const float* __restrict__ G, float* __restrict__ P, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, float w1, float b2, float omb2,
                                                     float bc2s, float neg_step) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = G[i];
    const float mo = m[i];
    const float mm = __builtin_fmaf(w1, g - mo, mo);          // exp_avg.lerp_(grad, 1-beta1)
    float vv = v[i] * b2;                                      // exp_avg_sq.mul_(beta2)
    vv = __builtin_fmaf(omb2 * g, g, vv);                      // .addcmul_(grad, grad, value=1-beta2)
    const float den = fdiv(fsqrt(vv), bc2s) + 1e-8f;           // (sqrt / bias_correction2_sqrt).add_(eps)
    P[i] = P[i] + fdiv(neg_step * mm, den);                    // addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mm;
    v[i] = vv;
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_adam_workspace_bytes(int C, int h, int w, int d) {
    (void)C;
    return 3 * (256 + sizeof(float) * 3 * (size_t)h * w * d) + 256;
}

extern "C" int cvx_adam_run_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v,
                                float lambda_weight, int niter, int step0, float cost_scale, const float* base_h,
                                const float* base_w, const float* base_d, float* U, float* grad_out,
                                const int* snapshot_iters_host, int n_snap, float* snapshots, void* workspace,
                                size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(F2 && M2 && P && m && v && U && base_h && base_w && base_d, "cvx_adam_run_f32: null pointer");
    CVX_REQUIRE(C > 0 && h > 1 && w > 1 && d > 1, "cvx_adam_run_f32: bad extent C=%d %dx%dx%d", C, h, w, d);
    CVX_REQUIRE(niter >= 0 && step0 >= 0, "cvx_adam_run_f32: negative iteration count");
    CVX_REQUIRE(n_snap == 0 || (snapshot_iters_host && snapshots), "cvx_adam_run_f32: snapshot buffers missing");
    if (!workspace || workspace_bytes < cvx_adam_workspace_bytes(C, h, w, d))
        return fail(CVX_ERR_WORKSPACE, "cvx_adam_run_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const size_t V = (size_t)h * w * d;
    Carver cv(workspace, workspace_bytes);
    float* t1 = cv.take<float>(3 * V);
    float* t2 = cv.take<float>(3 * V);
    float* gU = cv.take<float>(3 * V);

    // MeanBackward of lambda*mean(diff^2): lambda / N_axis in float32                       (:167-169)
    const float nH = (float)((int64_t)3 * (h - 1) * w * d), nW = (float)((int64_t)3 * h * (w - 1) * d),
                nD = (float)((int64_t)3 * h * w * (d - 1));
    const float cH = lambda_weight / nH, cW = lambda_weight / nW, cD = lambda_weight / nD;
    const float gsc = ((1.0f / (float)V) * cost_scale) / (float)C;     // MeanBackward, MulBackward, MeanBackward
    const dim3 gv((unsigned)cdiv64((int64_t)V, 256)), g3((unsigned)cdiv64((int64_t)(3 * V), 256));
    int snap = 0;
    for (int it = 0; it < niter; ++it) {
        int rc;
        if ((rc = launch_box_zero(P, t1, 3, h, w, d, 3, false, s))) return rc;
        if ((rc = launch_box_zero(t1, t2, 3, h, w, d, 3, false, s))) return rc;
        if ((rc = launch_box_zero(t2, U, 3, h, w, d, 3, false, s))) return rc;
        hipLaunchKernelGGL(k_warp_grad, gv, dim3(256), 0, s, F2, M2, C, h, w, d, U, base_h, base_w, base_d, gsc, cH, cW, cD, gU);
        if ((rc = launch_box_zero(gU, t1, 3, h, w, d, 3, true, s))) return rc;
        if ((rc = launch_box_zero(t1, t2, 3, h, w, d, 3, true, s))) return rc;
        if ((rc = launch_box_zero(t2, t1, 3, h, w, d, 3, true, s))) return rc;
        const int step = step0 + it + 1;
        const double beta1 = 0.9, beta2 = 0.999;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        hipLaunchKernelGGL(k_adam_update, g3, dim3(256), 0, s, t1, P, m, v, 3 * V, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)sqrt(bc2), (float)(-(1.0 / bc1)));
        while (snap < n_snap && snapshot_iters_host[snap] == it + 1) {
            (void)hipMemcpyAsync(snapshots + (size_t)snap * 3 * V, U, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice, s);
            ++snap;
        }
        if (grad_out && it == niter - 1)
            (void)hipMemcpyAsync(grad_out, t1, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice, s);
    }
    return check_last("adam_run");
}
