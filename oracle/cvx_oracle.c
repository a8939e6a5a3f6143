/*
 * cvx_oracle.c -- CPU restatement of the convexAdam hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in convexadam_amd/csrc.  It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (package convexadam_amd) never imports, links or executes anything under oracle/.
 *
 * It restates, operator by operator, what the upstream PyTorch reference computes on its CPU
 * float32 path, in the SAME floating-point evaluation order as the ATen CPU kernels the reference
 * calls, so that results can be compared bit-for-bit where the reference itself is deterministic.
 * Each function cites the reference lines (relative to /root/reference) it follows.
 *
 * Pinning: the oracle is checked (tests/test_oracle_vs_golden.py) against golden vectors produced
 * by importing and running the reference in the build container (tests/golden/make_golden*.py), and
 * live against the reference itself where it is mounted (tests/test_oracle_vs_reference_live.py).
 * By default three sites are not the reference's own bits (a SLEEF-style expf, the IEEE sqrt, an exactly
 * rounded global mean: <= 1 ulp each); with the optional tables / thread count of orc_set_exp_table,
 * orc_set_sqrt_table and orc_set_mean_threads (tests/mkl_tables.py) the oracle equals the reference
 * BIT FOR BIT end to end, including the full-size benchmark pair after 80 Adam iterations.
 *
 * Plain C99, scalar IEEE-754 binary32 arithmetic, compiled with -ffp-contract=off so that no
 * multiply-add is fused unless written as fmaf().  OpenMP is used only over independent outputs
 * (never inside a floating-point reduction), so results do not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static float outer_sum_rows(const float* vals, int64_t size, int ilp);
ORC_API float orc_torch_sum(const float* x, int64_t n, int threads, int vec);
/* 0 (default): the global mean of MINDSSC is the exactly rounded one; T > 0: torch's own sum with T threads (reference-bits mode) */
static int g_mean_threads = 0, g_mean_vec = 8;
ORC_API void orc_set_mean_threads(int threads, int vec) { g_mean_threads = threads; g_mean_vec = vec > 0 ? vec : 8; }

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * exp(): torch's CPU float exp is Sleef's expf (1.0-ULP variant, FMA build).  Restated from the
 * published SLEEF algorithm (sleefsimdsp.c, xexpf): Cody-Waite reduction by ln2 in two pieces, a
 * degree-6 polynomial evaluated with fused multiply-adds, ldexp in two halves.  Used by MINDSSC
 * (src/convexAdam/convex_adam_utils.py:63).
 * ---------------------------------------------------------------------------------------------- */
static inline float orc_ldexp2kf(float d, int e) {
    union { int32_t i; float f; } a, b;
    a.i = ((e >> 1) + 127) << 23;
    b.i = ((e - (e >> 1)) + 127) << 23;
    return d * a.f * b.f;
}
ORC_API float orc_expf(float d) {
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    float qf = rintf(d * R_LN2f);
    int q = (int)qf;
    float s = fmaf(qf, -L2Uf, d);
    s = fmaf(qf, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = orc_ldexp2kf(u, q);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}
ORC_API void orc_expf_array(const float* in, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = orc_expf(in[i]);
}
/* Optional restatement of the reference build's exp at MINDSSC (torch CPU -> MKL vsExp): orc_expf corrected by the tabulated
 * difference to the host's torch.exp.  Two bits per argument x <= 0, keyed by the bit pattern of |x| minus `first` (0 = equal,
 * 1 = one ulp above orc_expf, 2 = one ulp below); arguments outside [first, first + count) keep orc_expf.  The table is built at run
 * time from torch.exp itself (tests/mkl_tables.py) -- it is the exhaustive list of the 1-ulp differences, not an algorithm.
 * NULL (default) = orc_expf, which is what the HIP kernels use unless they are given the same table. */
static const uint8_t* g_exp_tbl = NULL;
static uint32_t g_exp_first = 0, g_exp_count = 0;
ORC_API void orc_set_exp_table(const uint8_t* tbl, uint32_t first, uint32_t count) { g_exp_tbl = tbl; g_exp_first = first; g_exp_count = count; }
static inline float orc_mind_exp(float negx) {       /* negx = -x >= 0 (or -0.0) */
    float r = orc_expf(-negx);
    if (g_exp_tbl) {
        uint32_t b; memcpy(&b, &negx, 4);
        b &= 0x7fffffffu;
        const uint32_t k = b - g_exp_first;
        if (b >= g_exp_first && k < g_exp_count) {
            const uint32_t code = (g_exp_tbl[k >> 2] >> ((k & 3) * 2)) & 3u;
            if (code) { uint32_t rb; memcpy(&rb, &r, 4); rb += (code == 1) ? 1u : 0xffffffffu; memcpy(&r, &rb, 4); }
        }
    }
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * torch.linspace(-1, 1, S) (float32, CPU): step = 2/(S-1) in float; first half start + step*i,
 * second half end - step*(S-1-i), each evaluated with a single rounding (checked against torch for
 * S = 2..399 in tests/test_host_logic.py).  affine_grid(align_corners=False) scales it by (S-1)/S
 * in two float ops; the search mesh of convex_adam_MIND.py:127 (align_corners=True) multiplies by
 * disp_hw -- which is NOT always an exact integer (hw=6: -1.9999999).
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_linspace_pm1(int S, float* out) {
    if (S == 1) { out[0] = -1.0f; return; }
    const float step = (1.0f - (-1.0f)) / (float)(S - 1);
    const int half = S / 2;
    for (int i = 0; i < S; ++i)
        out[i] = (i < half) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(S - 1 - i), 1.0f);
}
ORC_API void orc_affine_base(int S, float* out) { /* align_corners=False identity coordinate */
    orc_linspace_pm1(S, out);
    for (int i = 0; i < S; ++i) out[i] = (out[i] * (float)(S - 1)) / (float)S;
}
ORC_API void orc_disp_mesh(int hw, float* out /* [3][n^3] */) {
    /* convex_adam_MIND.py:127 ; flat k = (dD+hw)*n*n + (dW+hw)*n + (dH+hw), channel c = axis c */
    const int n = 2 * hw + 1;
    float lin[257];
    if (n == 1) lin[0] = 0.0f; /* affine_grid with one step yields 0 */
    else orc_linspace_pm1(n, lin);
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) for (int c = 0; c < n; ++c) {
        const int k = (a * n + b) * n + c;
        out[0 * n * n * n + k] = lin[c] * (float)hw;
        out[1 * n * n * n + k] = lin[b] * (float)hw;
        out[2 * n * n * n + k] = lin[a] * (float)hw;
    }
}

/* ------------------------------------------------------------------------------------------------
 * avg_pool3d(k, stride=1, padding=k/2, count_include_pad=True): raster-order sum of the in-range
 * taps (first spatial dim slowest), starting from 0, then ONE division by k^3.
 * (ATen AveragePool3d; used at convex_adam_utils.py:85,96,107 and convex_adam_MIND.py:166,191.)
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_box_zero(const float* in, float* out, int C, int H, int W, int D, int k) {
    const int p = k / 2;
    const float div = (float)(k * k * k);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int h = 0; h < H; ++h) {
            const float* ic = in + (size_t)c * H * W * D;
            float* oc = out + (size_t)c * H * W * D;
            for (int w = 0; w < W; ++w)
                for (int d = 0; d < D; ++d) {
                    const int h0 = h - p < 0 ? 0 : h - p, h1 = h + p >= H ? H - 1 : h + p;
                    const int w0 = w - p < 0 ? 0 : w - p, w1 = w + p >= W ? W - 1 : w + p;
                    const int d0 = d - p < 0 ? 0 : d - p, d1 = d + p >= D ? D - 1 : d + p;
                    float s = 0.0f;
                    for (int z = h0; z <= h1; ++z)
                        for (int y = w0; y <= w1; ++y)
                            for (int x = d0; x <= d1; ++x) s += ic[((size_t)z * W + y) * D + x];
                    oc[((size_t)h * W + w) * D + d] = s / div;
                }
        }
}

/* The same operator with an EVEN kernel (the reference's even `selected_smooth`, convex_adam_MIND.py:184-191: the announced "+1" is
 * overwritten at :189): padding k/2 on both sides of a window of k taps makes every axis ONE voxel longer -- out is (H+1, W+1, D+1),
 * output o covers inputs o-k/2 .. o+k/2-1; raster-order sum of the in-range taps from 0, one division by k^3 (count_include_pad). */
ORC_API void orc_box_grow(const float* in, float* out, int C, int H, int W, int D, int k) {
    const int p = k / 2, Ho = H + 1, Wo = W + 1, Do = D + 1;
    const float div = (float)(k * k * k);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int h = 0; h < Ho; ++h) {
            const float* ic = in + (size_t)c * H * W * D;
            float* oc = out + (size_t)c * Ho * Wo * Do;
            for (int w = 0; w < Wo; ++w)
                for (int d = 0; d < Do; ++d) {
                    const int h0 = h - p < 0 ? 0 : h - p, h1 = h - p + k > H ? H : h - p + k;
                    const int w0 = w - p < 0 ? 0 : w - p, w1 = w - p + k > W ? W : w - p + k;
                    const int d0 = d - p < 0 ? 0 : d - p, d1 = d - p + k > D ? D : d - p + k;
                    float s = 0.0f;
                    for (int z = h0; z < h1; ++z)
                        for (int y = w0; y < w1; ++y)
                            for (int x = d0; x < d1; ++x) s += ic[((size_t)z * W + y) * D + x];
                    oc[((size_t)h * Wo + w) * Do + d] = s / div;
                }
        }
}

/* adjoint of orc_box_zero as ATen's avg_pool3d_backward evaluates it: every output position o (in
 * raster order) adds gradOut[o]/k^3 to each input position of its window; gathered here per input
 * position in that same order of arrival. */
ORC_API void orc_box_zero_backward(const float* gout, float* gin, int C, int H, int W, int D, int k) {
    const int p = k / 2;
    const float div = (float)(k * k * k);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int h = 0; h < H; ++h) {
            const float* gc = gout + (size_t)c * H * W * D;
            float* ic = gin + (size_t)c * H * W * D;
            for (int w = 0; w < W; ++w)
                for (int d = 0; d < D; ++d) {
                    const int h0 = h - p < 0 ? 0 : h - p, h1 = h + p >= H ? H - 1 : h + p;
                    const int w0 = w - p < 0 ? 0 : w - p, w1 = w + p >= W ? W - 1 : w + p;
                    const int d0 = d - p < 0 ? 0 : d - p, d1 = d + p >= D ? D - 1 : d + p;
                    float s = 0.0f;
                    for (int z = h0; z <= h1; ++z)
                        for (int y = w0; y <= w1; ++y)
                            for (int x = d0; x <= d1; ++x) s += gc[((size_t)z * W + y) * D + x] / div;
                    ic[((size_t)h * W + w) * D + d] = s;
                }
        }
}

/* nn.Sequential(ReplicationPad3d(1), AvgPool3d(3, stride=1)) of convex_adam_MIND.py:40 : raster sum of the 27 taps
 * of the replicate-padded volume (all taps present, clamped coordinates), one division by 27. */
ORC_API void orc_box3_replicate(const float* in, float* out, int H, int W, int D) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
            for (int d = 0; d < D; ++d) {
                float s = 0.0f;
                for (int a = -1; a <= 1; ++a)
                    for (int b = -1; b <= 1; ++b)
                        for (int c = -1; c <= 1; ++c)
                            s += in[((size_t)clampi(h + a, 0, H - 1) * W + clampi(w + b, 0, W - 1)) * D + clampi(d + c, 0, D - 1)];
                out[((size_t)h * W + w) * D + d] = s / 27.0f;
            }
}

/* avg_pool3d(g, stride=g): floor output extent, raster sum of g^3 taps, one division.
 * (convex_adam_MIND.py:118-119,149-150) */
ORC_API void orc_avgpool_stride(const float* in, float* out, int C, int H, int W, int D, int g) {
    const int Ho = H / g, Wo = W / g, Do = D / g;
    const float div = (float)(g * g * g);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int h = 0; h < Ho; ++h)
            for (int w = 0; w < Wo; ++w)
                for (int d = 0; d < Do; ++d) {
                    const float* ic = in + (size_t)c * H * W * D;
                    float s = 0.0f;
                    for (int z = 0; z < g; ++z)
                        for (int y = 0; y < g; ++y)
                            for (int x = 0; x < g; ++x)
                                s += ic[((size_t)(h * g + z) * W + (w * g + y)) * D + (d * g + x)];
                    out[(((size_t)c * Ho + h) * Wo + w) * Do + d] = s / div;
                }
}

/* ------------------------------------------------------------------------------------------------
 * MIND-SSC, convex_adam_utils.py:24-68.
 *   shift tables in the reference's PRE-permutation channel order (derived by executing :31-47);
 *   the final permutation of :66 is applied when storing.
 * ---------------------------------------------------------------------------------------------- */
static const int MIND_O1[12][3] = {{0,0,-1},{0,-1,0},{0,-1,0},{0,0,1},{0,0,1},{1,0,0},
                                   {1,0,0},{1,0,0},{0,1,0},{0,1,0},{0,1,0},{0,1,0}};
static const int MIND_O2[12][3] = {{-1,0,0},{-1,0,0},{0,0,-1},{-1,0,0},{0,-1,0},{0,0,-1},
                                   {0,-1,0},{0,0,1},{-1,0,0},{0,0,-1},{0,0,1},{1,0,0}};
static const int MIND_PERM[12] = {6, 8, 1, 11, 2, 10, 0, 7, 9, 4, 5, 3}; /* out[j] = pre[PERM[j]] */

ORC_API void orc_mind_tables(int* o1, int* o2, int* perm) {
    memcpy(o1, MIND_O1, sizeof(MIND_O1)); memcpy(o2, MIND_O2, sizeof(MIND_O2));
    memcpy(perm, MIND_PERM, sizeof(MIND_PERM));
}

/* Order-independent (exact) accumulation of non-negative floats: each value is split against three
 * power-of-two grids so that every partial sum is exactly representable in a double; the three
 * exact sums are then combined.  The reference's own global mean (`mind_var.mean()`, :61) is a
 * multi-threaded cascade sum whose rounding depends on the thread split; it only sets the clamp
 * bounds, so the restatement uses the (near) correctly rounded value.  The HIP kernel uses the
 * identical splitting, which makes its result independent of the GPU reduction order. */
typedef struct { double m1, m2, m3; } orc_split_t;
static orc_split_t orc_split_make(double bound, double count) {
    /* u1 = ulp of m1 must satisfy count*bound/u1 < 2^52 */
    orc_split_t s;
    if (!(bound > 0.0)) bound = 1e-300;
    int e; (void)frexp(bound * count, &e);           /* bound*count < 2^e */
    double top = ldexp(1.0, e + 1);                  /* margin factor 2 */
    s.m1 = 1.5 * top;                                /* ulp(m1) = top * 2^-52 */
    s.m2 = s.m1 * 0x1p-30;
    s.m3 = s.m2 * 0x1p-30;
    return s;
}
static inline void orc_split_add(const orc_split_t* s, double v, double* a1, double* a2, double* a3) {
    double q1 = (v + s->m1) - s->m1; double r1 = v - q1;
    double q2 = (r1 + s->m2) - s->m2; double r2 = r1 - q2;
    double q3 = (r2 + s->m3) - s->m3;
    *a1 += q1; *a2 += q2; *a3 += q3;
}

/* ssd_c(x) = box_{(2r+1)^3}[ (I(clampP + o1*d) - I(clampP + o2*d))^2 ] with replicate borders for
 * both the shift (rpad1) and the box (rpad2), reference :52-56.  Writes pre-permutation order. */
static void mind_patch_ssd(const float* img, int H, int W, int D, int r, int dil, float* ssd12) {
    const size_t V = (size_t)H * W * D;
    const int k = 2 * r + 1;
    const float div = (float)(k * k * k);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < 12; ++c)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int d = 0; d < D; ++d) {
                    float s = 0.0f;
                    for (int tz = -r; tz <= r; ++tz) {
                        const int ph = clampi(h + tz, 0, H - 1);
                        for (int ty = -r; ty <= r; ++ty) {
                            const int pw = clampi(w + ty, 0, W - 1);
                            for (int tx = -r; tx <= r; ++tx) {
                                const int pd = clampi(d + tx, 0, D - 1);
                                const float a = img[((size_t)clampi(ph + MIND_O1[c][0] * dil, 0, H - 1) * W +
                                                     clampi(pw + MIND_O1[c][1] * dil, 0, W - 1)) * D +
                                                    clampi(pd + MIND_O1[c][2] * dil, 0, D - 1)];
                                const float b = img[((size_t)clampi(ph + MIND_O2[c][0] * dil, 0, H - 1) * W +
                                                     clampi(pw + MIND_O2[c][1] * dil, 0, W - 1)) * D +
                                                    clampi(pd + MIND_O2[c][2] * dil, 0, D - 1)];
                                const float df = a - b;
                                s += df * df;
                            }
                        }
                    }
                    ssd12[(size_t)c * V + ((size_t)h * W + w) * D + d] = s / div;
                }
}

/* out: [12][H][W][D] in the reference's final (permuted) channel order.
 * mean_out (optional): the global mean used for the clamp bounds. */
ORC_API void orc_mindssc(const float* img, int H, int W, int D, int radius, int dilation, float* out,
                         float* mean_out) {
    const size_t V = (size_t)H * W * D;
    float* ssd = (float*)malloc(sizeof(float) * 12 * V);
    float* var = (float*)malloc(sizeof(float) * V);
    mind_patch_ssd(img, H, W, D, radius, dilation, ssd);
    /* :59-60  mind = ssd - min_c ssd ; mind_var = mean_c(mind) = (sequential sum c=0..11) / 12 */
    float imin = img[0], imax = img[0];
    for (size_t i = 1; i < V; ++i) { if (img[i] < imin) imin = img[i]; if (img[i] > imax) imax = img[i]; }
#pragma omp parallel for schedule(static)
    for (size_t x = 0; x < V; ++x) {
        float mn = ssd[x];
        for (int c = 1; c < 12; ++c) { const float v = ssd[(size_t)c * V + x]; if (v < mn) mn = v; }
        float mc[12];
        for (int c = 0; c < 12; ++c) { const float m = ssd[(size_t)c * V + x] - mn; ssd[(size_t)c * V + x] = m; mc[c] = m; }
        var[x] = outer_sum_rows(mc, 12, x >= (V / 32) * 32) / 12.0f;
    }
    /* :61 global mean (exact accumulation, see orc_split_*) */
    const double range = (double)imax - (double)imin;
    const orc_split_t sp = orc_split_make(range * range, (double)V);
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (size_t x = 0; x < V; ++x) orc_split_add(&sp, (double)var[x], &a1, &a2, &a3);
    float gmean = (float)((a1 + (a2 + a3)) / (double)V);
    if (g_mean_threads > 0) gmean = orc_torch_sum(var, (int64_t)V, g_mean_threads, g_mean_vec) / (float)V;   /* sum_out(..).div_(numel) */
    if (mean_out) *mean_out = gmean;
    const float lo = (float)((double)gmean * 0.001), hi = (float)((double)gmean * 1000.0);
    /* :61-66 clamp, divide, exp(-x), permute */
#pragma omp parallel for schedule(static)
    for (size_t x = 0; x < V; ++x) {
        float v = var[x];
        v = v < lo ? lo : v;   /* torch.clamp: min(max(x, lo), hi); NaN propagates */
        v = v > hi ? hi : v;
        for (int j = 0; j < 12; ++j) {
            const float m = ssd[(size_t)MIND_PERM[j] * V + x] / v;
            out[(size_t)j * V + x] = orc_mind_exp(m);
        }
    }
    free(ssd); free(var);
}

/* ------------------------------------------------------------------------------------------------
 * torch.sum over an outer (strided) dimension of `size` rows: ATen's cascade (multi_row_sum,
 * 4 levels, level_step = 16 for size < 65536).  For size < 16 this is the plain sequential sum
 * starting from 0.  Used for the sums over channels (convex_adam_utils.py:60,84).
 * ---------------------------------------------------------------------------------------------- */
static inline int ceil_log2_i64(int64_t x) { int r = 0; int64_t v = 1; while (v < x) { v <<= 1; ++r; } return r; }
static float cascade_sum_strided(const float* vals, int64_t stride, int64_t size) {
    const int num_levels = 4;
    int level_power = ceil_log2_i64(size) / num_levels; if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power, level_mask = level_step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    for (; i + level_step <= size;) {
        for (int64_t j = 0; j < level_step; ++j, ++i) acc[0] += vals[i * stride];
        for (int j = 1; j < num_levels; ++j) {
            acc[j] += acc[j - 1]; acc[j - 1] = 0.f;
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] += vals[i * stride];
    for (int j = 1; j < num_levels; ++j) acc[0] += acc[j];
    return acc[0];
}
/* ATen evaluates an outer-dimension sum in blocks of 32 contiguous columns; the last (ncols mod 32)
 * columns go through `row_sum`, which keeps 4 interleaved partial sums (rows i = k mod 4), adds
 * the leftover rows to partial 0 and then folds ((p0+p1)+p2)+p3.  `ilp` selects that order.
 * (pinned empirically against torch 2.10 CPU, 1 and 8 threads; see tests/test_host_logic.py) */
static float outer_sum_rows(const float* vals, int64_t size, int ilp) {
    if (!ilp) return cascade_sum_strided(vals, 1, size);
    const int64_t n4 = size / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = cascade_sum_strided(vals + k, 4, n4);
    for (int64_t i = n4 * 4; i < size; ++i) p[0] += vals[i];
    for (int k = 1; k < 4; ++k) p[0] += p[k];
    return p[0];
}
ORC_API float orc_outer_sum_rows(const float* vals, int64_t size, int ilp) { return outer_sum_rows(vals, size, ilp); }

/* ------------------------------------------------------------------------------------------------
 * torch.sum of a whole contiguous float tensor as ATen's CPU kernel evaluates it with `threads` threads and `vec`-float vectors
 * (8 in the torch 2.10 build, also where the CPU capability is AVX-512: pinned empirically).  TensorIteratorReduce two_pass_reduction: at::parallel_for splits the n elements into
 * nt = min(threads, ceil(n / 32768)) chunks of ceil(n / nt); chunk t is reduced into slot t of a `threads`-long buffer, the buffer
 * is then reduced by the same kernel.  One chunk (SumKernel.cpp vectorized_inner_sum / row_sum): the vectors of the chunk are summed
 * lane-wise with 4 interleaved cascade accumulators (vector i goes to partial i mod 4; leftover vectors to partial 0; then
 * ((p0 + p1) + p2) + p3), the scalar tail (n mod vec) is summed first into the final accumulator, then the lanes are added in
 * order; fewer than `vec` elements: the same scheme on scalars.  This is the one site whose value depends on the reference's THREAD
 * COUNT -- `mind_var.mean()` (convex_adam_utils.py:61) -- and only matters where the variance is clamped to its bounds.
 * Pinned against torch.sum itself for 1..16 threads, tests/test_host_logic.py::test_torch_full_sum_restatement.
 * ---------------------------------------------------------------------------------------------- */
static float torch_row_sum_scalar(const float* x, int64_t n) {      /* row_sum<float> : 4-way ilp on scalars */
    const int64_t n4 = n / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = cascade_sum_strided(x + k, 4, n4);
    for (int64_t i = n4 * 4; i < n; ++i) p[0] += x[i];
    for (int k = 1; k < 4; ++k) p[0] += p[k];
    return p[0];
}
static float torch_inner_sum(const float* x, int64_t n, int vec) {
    if (n < vec) return torch_row_sum_scalar(x, n);
    const int64_t nv = n / vec, nv4 = nv / 4;
    float fin = 0.0f;
    for (int64_t k = nv * vec; k < n; ++k) fin += x[k];
    for (int lane = 0; lane < vec; ++lane) {
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = cascade_sum_strided(x + (int64_t)k * vec + lane, (int64_t)4 * vec, nv4);
        for (int64_t i = nv4 * 4; i < nv; ++i) p[0] += x[i * vec + lane];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        fin += p[0];
    }
    return fin;
}
ORC_API float orc_torch_sum(const float* x, int64_t n, int threads, int vec) {
    if (threads < 1) threads = 1;
    if (n < 32768 || threads == 1) return 0.0f + torch_inner_sum(x, n, vec);
    float buf[1024];
    if (threads > 1024) threads = 1024;
    for (int t = 0; t < threads; ++t) buf[t] = 0.0f;
    int64_t nt = (n + 32767) / 32768; if (nt > threads) nt = threads;
    const int64_t chunk = (n + nt - 1) / nt;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < nt; ++t) {
        const int64_t b = t * chunk, e = b + chunk < n ? b + chunk : n;
        if (b < e) buf[t] += torch_inner_sum(x + b, e - b, vec);
    }
    return 0.0f + torch_inner_sum(buf, threads, vec);
}

/* ------------------------------------------------------------------------------------------------
 * correlate(), convex_adam_utils.py:72-89.
 *   raw[k,x] = sum_c (F_c(x) - M0_c(x+delta_k))^2      (M0 = zero-padded moving features)
 *   ssd      = box3(box3(raw))   zero pad, /27 each    flat k = (dD+hw)n^2 + (dW+hw)n + (dH+hw)
 *   argmin over k, first minimum wins.
 * fix, mov: [C][h][w][d] ; ssd: [n^3][h][w][d] ; argmin: int64 [h][w][d]
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_correlate_ex(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box,
                              float* ssd, int64_t* argmin);
ORC_API void orc_correlate(const float* fix, const float* mov, int C, int h, int w, int d, int hw,
                           float* ssd, int64_t* argmin) {
    orc_correlate_ex(fix, mov, C, h, w, d, hw, 0, 2, ssd, argmin);
}
/* variants of the challenge scripts: cost 1 = `.abs().sum(0)` (l2r_2021_convexAdam_task3_docker.py:54), n_box 1 = a single
 * avg_pool3d (l2r_2021_convexAdam_task2_docker.py:60, task3:56) */
ORC_API void orc_correlate_ex(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box,
                              float* ssd, int64_t* argmin) {
    const int n = 2 * hw + 1;
    const size_t v = (size_t)h * w * d;
    const int64_t K = (int64_t)n * n * n;
#pragma omp parallel
    {
        float* raw = (float*)malloc(sizeof(float) * v);
        float* b1 = (float*)malloc(sizeof(float) * v);
        float* cv = (float*)malloc(sizeof(float) * (C > 0 ? C : 1));
#pragma omp for schedule(dynamic, 4)
        for (int64_t k = 0; k < K; ++k) {
            const int dD = (int)(k / (n * n)) - hw, dW = (int)((k / n) % n) - hw, dH = (int)(k % n) - hw;
            const int64_t jj = (int64_t)(dW + hw) * n + (dD + hw);       /* unfold channel, :76-77 */
            const int64_t ncols = (int64_t)h * n * n * w * d, tail_from = (ncols / 32) * 32;
            for (int z = 0; z < h; ++z)
                for (int y = 0; y < w; ++y)
                    for (int x = 0; x < d; ++x) {
                        const int mz = z + dH, my = y + dW, mx = x + dD;
                        const int inb = (mz >= 0 && mz < h && my >= 0 && my < w && mx >= 0 && mx < d);
                        for (int c = 0; c < C; ++c) {
                            const float f = fix[(size_t)c * v + ((size_t)z * w + y) * d + x];
                            const float m = inb ? mov[(size_t)c * v + ((size_t)mz * w + my) * d + mx] : 0.0f;
                            const float df = f - m;
                            cv[c] = cost ? fabsf(df) : df * df;
                        }
                        /* position of this element inside the reference's (C,h,n^2,w,d) difference tensor */
                        const int64_t flat = (((int64_t)z * n * n + jj) * w + y) * d + x;
                        raw[((size_t)z * w + y) * d + x] = outer_sum_rows(cv, C, flat >= tail_from);
                    }
            /* two zero-padded 3^3 box filters, raster order, /27 (inlined single-thread version) */
            for (int pass = 0; pass < n_box; ++pass) {
                const float* src = pass == 0 ? raw : b1;
                float* dst = pass == n_box - 1 ? ssd + (size_t)k * v : b1;
                for (int z = 0; z < h; ++z)
                    for (int y = 0; y < w; ++y)
                        for (int x = 0; x < d; ++x) {
                            const int z0 = z > 0 ? z - 1 : 0, z1 = z < h - 1 ? z + 1 : h - 1;
                            const int y0 = y > 0 ? y - 1 : 0, y1 = y < w - 1 ? y + 1 : w - 1;
                            const int x0 = x > 0 ? x - 1 : 0, x1 = x < d - 1 ? x + 1 : d - 1;
                            float s = 0.0f;
                            for (int a = z0; a <= z1; ++a)
                                for (int b = y0; b <= y1; ++b)
                                    for (int e = x0; e <= x1; ++e) s += src[((size_t)a * w + b) * d + e];
                            dst[((size_t)z * w + y) * d + x] = s / 27.0f;
                        }
            }
        }
        free(raw); free(b1); free(cv);
    }
    if (argmin) {
#pragma omp parallel for schedule(static)
        for (size_t x = 0; x < v; ++x) {
            float best = ssd[x]; int64_t bi = 0;
            /* torch.argmin: the first minimum, and a NaN counts as smaller than everything (the first NaN wins and ends the scan) */
            for (int64_t k = 1; k < K && best == best; ++k) { const float s = ssd[(size_t)k * v + x]; if (s < best || s != s) { best = s; bi = k; } }
            argmin[x] = bi;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * coupled_convex(), convex_adam_utils.py:93-109.
 *   u0 = box3(mesh[argmin]);  for coef in (.003,.01,.03,.1,.3,1):
 *        k*(x) = argmin_k  ssd[k,x] + coef * sum_a (mesh[a,k] - u[a,x])^2 ;  u = box3(mesh[k*])
 * mesh: [3][n^3] ; out: [3][h][w][d] (coarse-voxel units).
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_coupled_convex(const float* ssd, const int64_t* argmin, const float* mesh, int h, int w,
                                int d, int hw, float* out) {
    const int n = 2 * hw + 1;
    const int64_t K = (int64_t)n * n * n;
    const size_t v = (size_t)h * w * d;
    static const float coeffs[6] = {0.003f, 0.01f, 0.03f, 0.1f, 0.3f, 1.0f};
    float* sel = (float*)malloc(sizeof(float) * 3 * v);
    int64_t* am = (int64_t*)malloc(sizeof(int64_t) * v);
    memcpy(am, argmin, sizeof(int64_t) * v);
    for (int it = 0; it <= 6; ++it) {
        for (int a = 0; a < 3; ++a)
            for (size_t x = 0; x < v; ++x) sel[(size_t)a * v + x] = mesh[(size_t)a * K + am[x]];
        orc_box_zero(sel, out, 3, h, w, d, 3);
        if (it == 6) break;
        const float coef = coeffs[it];
#pragma omp parallel for schedule(static)
        for (size_t x = 0; x < v; ++x) {
            const float u0 = out[x], u1 = out[v + x], u2 = out[2 * v + x];
            float best = 0.f; int64_t bi = 0;
            for (int64_t k = 0; k < K; ++k) {
                const float e0 = mesh[k] - u0, e1 = mesh[K + k] - u1, e2 = mesh[2 * K + k] - u2;
                float q = 0.0f; q += e0 * e0; q += e1 * e1; q += e2 * e2;     /* .pow(2).sum(0) */
                const float cost = ssd[(size_t)k * v + x] + coef * q;           /* :104 */
                if (k == 0 || (best == best && (cost < best || cost != cost))) { best = cost; bi = k; }      /* first minimum; the first NaN wins */
            }
            am[x] = bi;
        }
    }
    free(sel); free(am);
}

/* ------------------------------------------------------------------------------------------------
 * grid_sample (3-D, bilinear, zeros padding, align_corners=False), ATen GridSampler.cpp evaluation
 * order.  Coordinates are normalised: comp 0 <-> last tensor dim (d), comp 2 <-> first (h).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float ix, iy, iz;
    int x0, y0, z0;
    float tnw, tne, tsw, tse, bnw, bne, bsw, bse;
} orc_tri_t;

static inline float unnorm(float g, int S) { return ((g + 1.0f) * (float)S - 1.0f) / 2.0f; }

static inline void tri_setup(orc_tri_t* t, float gx, float gy, float gz, int h, int w, int d) {
    t->ix = unnorm(gx, d); t->iy = unnorm(gy, w); t->iz = unnorm(gz, h);
    const float fx = floorf(t->ix), fy = floorf(t->iy), fz = floorf(t->iz);
    /* keep int conversion safe for wild coordinates */
    t->x0 = (int)fmaxf(fminf(fx, 1.0e9f), -1.0e9f); t->y0 = (int)fmaxf(fminf(fy, 1.0e9f), -1.0e9f);
    t->z0 = (int)fmaxf(fminf(fz, 1.0e9f), -1.0e9f);
    const float x1 = (float)(t->x0 + 1), y1 = (float)(t->y0 + 1), z1 = (float)(t->z0 + 1);
    const float x0f = (float)t->x0, y0f = (float)t->y0, z0f = (float)t->z0;
    t->tnw = (x1 - t->ix) * (y1 - t->iy) * (z1 - t->iz);
    t->tne = (t->ix - x0f) * (y1 - t->iy) * (z1 - t->iz);
    t->tsw = (x1 - t->ix) * (t->iy - y0f) * (z1 - t->iz);
    t->tse = (t->ix - x0f) * (t->iy - y0f) * (z1 - t->iz);
    t->bnw = (x1 - t->ix) * (y1 - t->iy) * (t->iz - z0f);
    t->bne = (t->ix - x0f) * (y1 - t->iy) * (t->iz - z0f);
    t->bsw = (x1 - t->ix) * (t->iy - y0f) * (t->iz - z0f);
    t->bse = (t->ix - x0f) * (t->iy - y0f) * (t->iz - z0f);
}
static inline int inb3(int z, int y, int x, int h, int w, int d) {
    return z >= 0 && z < h && y >= 0 && y < w && x >= 0 && x < d;
}
static inline float tri_sample(const orc_tri_t* t, const float* vol, int h, int w, int d) {
    const int x0 = t->x0, y0 = t->y0, z0 = t->z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    float o = 0.0f;
    if (inb3(z0, y0, x0, h, w, d)) o += vol[((size_t)z0 * w + y0) * d + x0] * t->tnw;
    if (inb3(z0, y0, x1, h, w, d)) o += vol[((size_t)z0 * w + y0) * d + x1] * t->tne;
    if (inb3(z0, y1, x0, h, w, d)) o += vol[((size_t)z0 * w + y1) * d + x0] * t->tsw;
    if (inb3(z0, y1, x1, h, w, d)) o += vol[((size_t)z0 * w + y1) * d + x1] * t->tse;
    if (inb3(z1, y0, x0, h, w, d)) o += vol[((size_t)z1 * w + y0) * d + x0] * t->bnw;
    if (inb3(z1, y0, x1, h, w, d)) o += vol[((size_t)z1 * w + y0) * d + x1] * t->bne;
    if (inb3(z1, y1, x0, h, w, d)) o += vol[((size_t)z1 * w + y1) * d + x0] * t->bsw;
    if (inb3(z1, y1, x1, h, w, d)) o += vol[((size_t)z1 * w + y1) * d + x1] * t->bse;
    return o;
}

/* generic grid_sample: vol [C][h][w][d], grid [ho][wo][do][3] (x,y,z) -> out [C][ho][wo][do] */
ORC_API void orc_grid_sample(const float* vol, int C, int h, int w, int d, const float* grid, int ho,
                             int wo, int dd, float* out) {
    const size_t vo = (size_t)ho * wo * dd, vi = (size_t)h * w * d;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < vo; ++p) {
        orc_tri_t t; tri_setup(&t, grid[3 * p], grid[3 * p + 1], grid[3 * p + 2], h, w, d);
        for (int c = 0; c < C; ++c) out[(size_t)c * vo + p] = tri_sample(&t, vol + (size_t)c * vi, h, w, d);
    }
}

/* ------------------------------------------------------------------------------------------------
 * inverse_consistency(), convex_adam_utils.py:114-129.  Fields are [3][h][w][d] with channel 0 the
 * NORMALISED displacement along the LAST axis (the caller flips, convex_adam_MIND.py:139).
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_inverse_consistency(const float* f1, const float* f2, int h, int w, int d, int iters,
                                     float* o1, float* o2) {
    const size_t v = (size_t)h * w * d;
    float* bh = (float*)malloc(sizeof(float) * h); float* bw = (float*)malloc(sizeof(float) * w);
    float* bd = (float*)malloc(sizeof(float) * d);
    orc_affine_base(h, bh); orc_affine_base(w, bw); orc_affine_base(d, bd);
    float* a1 = (float*)malloc(sizeof(float) * 3 * v); float* a2 = (float*)malloc(sizeof(float) * 3 * v);
    memcpy(o1, f1, sizeof(float) * 3 * v); memcpy(o2, f2, sizeof(float) * 3 * v);
    for (int it = 0; it < iters; ++it) {
        memcpy(a1, o1, sizeof(float) * 3 * v); memcpy(a2, o2, sizeof(float) * 3 * v);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < v; ++p) {
            const int z = (int)(p / ((size_t)w * d)), y = (int)((p / d) % w), x = (int)(p % d);
            orc_tri_t t;
            tri_setup(&t, bd[x] + a1[p], bw[y] + a1[v + p], bh[z] + a1[2 * v + p], h, w, d);
            for (int c = 0; c < 3; ++c) o1[(size_t)c * v + p] = 0.5f * (a1[(size_t)c * v + p] - tri_sample(&t, a2 + (size_t)c * v, h, w, d));
            tri_setup(&t, bd[x] + a2[p], bw[y] + a2[v + p], bh[z] + a2[2 * v + p], h, w, d);
            for (int c = 0; c < 3; ++c) o2[(size_t)c * v + p] = 0.5f * (a2[(size_t)c * v + p] - tri_sample(&t, a1 + (size_t)c * v, h, w, d));
        }
    }
    free(a1); free(a2); free(bh); free(bw); free(bd);
}

/* ------------------------------------------------------------------------------------------------
 * F.interpolate(mode='trilinear', align_corners=False, size=...), ATen UpSampleKernel (generic
 * N-d linear): src = max(fma(in/out, dst+0.5, -0.5), 0); i0 = min(floor(src), in-1); l1 = clamp(src-i0,
 * 0,1); l0 = 1-l1; i1 = i0 + (i0 < in-1).  Evaluation: innermost (last) dim first, per level
 * r = fma(v0, w0, v1*w1)  (how the FMA-contracted ATen kernel rounds `t0*w0 + t1*w1`; pinned
 * empirically against torch 2.10 CPU in the build container).
 * (convex_adam_MIND.py:141,153,182)
 * ---------------------------------------------------------------------------------------------- */
static void lin_table(int in, int out, int* i0, int* i1, float* l0, float* l1) {
    const float ratio = (float)in / (float)out;
    for (int o = 0; o < out; ++o) {
        float src = fmaf(ratio, (float)o + 0.5f, -0.5f);   /* contracted in the ATen build */
        if (src < 0.0f) src = 0.0f;
        int a = (int)floorf(src); if (a > in - 1) a = in - 1;
        float l = src - (float)a; l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
        i0[o] = a; i1[o] = a + ((a < in - 1) ? 1 : 0); l1[o] = l; l0[o] = 1.0f - l;
    }
}
ORC_API void orc_resize_trilinear(const float* in, int C, int h, int w, int d, float* out, int H, int W, int D) {
    int *h0 = malloc(sizeof(int) * H), *h1 = malloc(sizeof(int) * H), *w0 = malloc(sizeof(int) * W),
        *w1 = malloc(sizeof(int) * W), *d0 = malloc(sizeof(int) * D), *d1 = malloc(sizeof(int) * D);
    float *lh0 = malloc(sizeof(float) * H), *lh1 = malloc(sizeof(float) * H), *lw0 = malloc(sizeof(float) * W),
          *lw1 = malloc(sizeof(float) * W), *ld0 = malloc(sizeof(float) * D), *ld1 = malloc(sizeof(float) * D);
    lin_table(h, H, h0, h1, lh0, lh1); lin_table(w, W, w0, w1, lw0, lw1); lin_table(d, D, d0, d1, ld0, ld1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int z = 0; z < H; ++z) {
            const float* ic = in + (size_t)c * h * w * d;
            for (int y = 0; y < W; ++y)
                for (int x = 0; x < D; ++x) {
                    float lev1[2];
                    for (int a = 0; a < 2; ++a) {
                        const int zz = a ? h1[z] : h0[z];
                        float lev2[2];
                        for (int b = 0; b < 2; ++b) {
                            const int yy = b ? w1[y] : w0[y];
                            const float* row = ic + ((size_t)zz * w + yy) * d;
                            lev2[b] = fmaf(row[d0[x]], ld0[x], row[d1[x]] * ld1[x]);
                        }
                        lev1[a] = fmaf(lev2[0], lw0[y], lev2[1] * lw1[y]);
                    }
                    out[(((size_t)c * H + z) * W + y) * D + x] = fmaf(lev1[0], lh0[z], lev1[1] * lh1[z]);
                }
        }
    free(h0); free(h1); free(w0); free(w1); free(d0); free(d1);
    free(lh0); free(lh1); free(lw0); free(lw1); free(ld0); free(ld1);
}

/* ------------------------------------------------------------------------------------------------
 * Smoothers of the sweep scripts: kovesi_spline = chain of zero-padded box filters
 * (self_configuring/convexAdam_hyper_util.py:475-488); GaussianSmoothing = 5-tap replicate-padded convolution along
 * H, W, D (hyper_util.py:423-473).  Rounding order pinned against torch 2.10 CPU (oneDNN convolution):
 *   forward  acc = w0*x0 ; acc = fma(w_t, x_t, acc)  t = 1..4
 *   backward gxp[j] = w0*g[j] ; then t = 1..4 ascending: fma(w_t, g[j-t], acc) when the tensor has more than one
 *            channel (oneDNN picks another kernel for a single image: there the products are rounded, acc + w_t*g),
 *            padded index j; then replication_pad3d_backward adds the border entries in ascending order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int kind; int n_boxes; int box_k[4]; float gauss_w[5]; } orc_smoother;

static void gauss1d(const float* in, float* out, int C, int H, int W, int D, int axis, const float* w, int backward) {
    const size_t total = (size_t)C * H * W * D;
    const int n = axis == 0 ? H : (axis == 1 ? W : D);
    const size_t stride = axis == 0 ? (size_t)W * D : (axis == 1 ? (size_t)D : 1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; ++i) {
        const int a = (int)((i / stride) % n);
        const float* base = in + (i - (size_t)a * stride);
        if (!backward) {
            float acc = w[0] * base[(size_t)clampi(a - 2, 0, n - 1) * stride];
            for (int t = 1; t < 5; ++t) acc = fmaf(w[t], base[(size_t)clampi(a + t - 2, 0, n - 1) * stride], acc);
            out[i] = acc;
        } else {
            float r = 0.0f;
            const int j0 = a == 0 ? 0 : a + 2, j1 = a == n - 1 ? n + 3 : a + 2;
            for (int j = j0; j <= j1; ++j) {
                float acc = w[0] * ((j >= 0 && j < n) ? base[(size_t)j * stride] : 0.0f);
                for (int t = 1; t < 5; ++t) {
                    const int q = j - t;
                    const float gq = (q >= 0 && q < n) ? base[(size_t)q * stride] : 0.0f;
                    acc = (C > 1) ? fmaf(w[t], gq, acc) : acc + w[t] * gq;
                }
                r += acc;
            }
            out[i] = r;
        }
    }
}
ORC_API void orc_smooth(const float* in, float* out, int C, int H, int W, int D, const orc_smoother* sm, int backward) {
    const size_t n = (size_t)C * H * W * D;
    float* a = (float*)malloc(sizeof(float) * n); float* b = (float*)malloc(sizeof(float) * n);
    memcpy(a, in, sizeof(float) * n);
    const int nst = sm->kind == 1 ? 3 : sm->n_boxes;
    for (int i = 0; i < nst; ++i) {
        const int st = backward ? nst - 1 - i : i;
        if (sm->kind == 1) gauss1d(a, b, C, H, W, D, st, sm->gauss_w, backward);
        else if (backward) orc_box_zero_backward(a, b, C, H, W, D, sm->box_k[st]);
        else orc_box_zero(a, b, C, H, W, D, sm->box_k[st]);
        float* t = a; a = b; b = t;
    }
    memcpy(out, a, sizeof(float) * n);
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------------------
 * Adam instance optimisation, convex_adam_MIND.py:155-182.
 *   P    : [3][h][w][d]   parameter (control grid, grid units)   -- updated in place
 *   m, v : Adam moments (zero-initialised by the caller for a fresh run)
 *   F2,M2: [C][h][w][d]   pooled features of fixed / moving
 *   U    : [3][h][w][d]   disp_sample of the LAST forward pass (what the reference returns, :181)
 *   G    : optional, gradient dL/dP of the last iteration (debug / single-step parity)
 *   step0: number of Adam steps already taken (bias correction continues from there)
 * One iteration:
 *   U = box3(box3(box3(P)))                                                      (:166)
 *   reg = lam*[mean_W-diff^2 + mean_H-diff^2 + mean_D-diff^2]                    (:167-169)
 *   sample M2 at x + U*S/(S-1) (normalised: base + U/((S-1)/2))                  (:171-174)
 *   loss = mean_x( mean_c((Wc-Fc)^2) * 12 )                                       (:176-177)
 *   backward (autograd accumulation order restated below), Adam step             (:178-179)
 * ---------------------------------------------------------------------------------------------- */
/* Optional restatement of the reference build's sqrt (torch CPU -> MKL vsSqrt): the correctly rounded root or a neighbour of it as
 * tabulated from torch.sqrt itself -- two bits per (exponent parity, mantissa) class, 0 = IEEE root, 1 = one ulp above, 2 = one ulp
 * below; entries 0 .. 2^24-1 normal inputs (key = parity << 23 | mantissa), then 2^23 denormal inputs (key = mantissa); four per
 * byte (tests/mkl_tables.py; tests/golden/mkl_vssqrt_low.npz holds the golden host's table as bit maps of the low classes).
 * NULL (default) = IEEE sqrt, which is what the HIP kernels use unless they are given the same table. */
static const uint8_t* g_sqrt_codes = NULL;
ORC_API void orc_set_sqrt_table(const uint8_t* codes) { g_sqrt_codes = codes; }
static float orc_adam_sqrt(float x) {
    float r = sqrtf(x);
    if (g_sqrt_codes) {
        uint32_t b; memcpy(&b, &x, 4);
        const uint32_t e = b >> 23, mant = b & 0x7fffffu;
        if (b != 0 && e < 255) {
            const uint32_t key = e ? (((e & 1u) << 23) | mant) : ((1u << 24) | mant);
            const uint32_t code = (g_sqrt_codes[key >> 2] >> ((key & 3u) * 2u)) & 3u;
            if (code) { uint32_t rb; memcpy(&rb, &r, 4); rb += (code == 1u) ? 1u : 0xffffffffu; memcpy(&r, &rb, 4); }
        }
    }
    return r;
}

ORC_API void orc_adam_run_smoother(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                          float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                          float* U, float* G, float* loss_out, const orc_smoother* sm);
ORC_API void orc_adam_run(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                          float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                          float* U, float* G, float* loss_out) {
    orc_adam_run_smoother(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, U, G, loss_out, NULL);
}
ORC_API void orc_adam_run_smoother(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                          float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                          float* U, float* G, float* loss_out, const orc_smoother* sm) {
    const orc_smoother dflt = {0, 3, {3, 3, 3, 0}, {0, 0, 0, 0, 0}};       /* box3(box3(box3(.))), MIND:166 */
    if (!sm) sm = &dflt;
    const size_t V = (size_t)h * w * d;
    float* t1 = (float*)malloc(sizeof(float) * 3 * V);
    float* t2 = (float*)malloc(sizeof(float) * 3 * V);
    float* gU = (float*)malloc(sizeof(float) * 3 * V);
    float* bh = (float*)malloc(sizeof(float) * h); float* bw = (float*)malloc(sizeof(float) * w);
    float* bd = (float*)malloc(sizeof(float) * d);
    orc_affine_base(h, bh); orc_affine_base(w, bw); orc_affine_base(d, bd);
    const float sc[3] = {(float)((h - 1) / 2.0), (float)((w - 1) / 2.0), (float)((d - 1) / 2.0)}; /* :171 */
    /* reg-term constants: lam / N_axis (mean backward), :167-169.  axis 0 = H, 1 = W, 2 = D */
    const float nH = (float)((int64_t)3 * (h - 1) * w * d), nW = (float)((int64_t)3 * h * (w - 1) * d),
                nD = (float)((int64_t)3 * h * w * (d - 1));
    const float cH = lambda_weight / nH, cW = lambda_weight / nW, cD = lambda_weight / nD;
    /* data-term constant: d loss / d (Wc-Fc)^2 : ((1/V) * scale) / C  (MeanBackward, MulBackward, MeanBackward) */
    const float gsc = ((1.0f / (float)V) * cost_scale) / (float)C;
    const float gmx = (float)d / 2.0f, gmy = (float)w / 2.0f, gmz = (float)h / 2.0f;

    for (int it = 0; it < niter; ++it) {
        orc_smooth(P, U, 3, h, w, d, sm, 0);
        double lsum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : lsum)
        for (size_t p = 0; p < V; ++p) {
            const int z = (int)(p / ((size_t)w * d)), y = (int)((p / d) % w), x = (int)(p % d);
            const float uH = U[p], uW = U[V + p], uD = U[2 * V + p];
            orc_tri_t t;
            tri_setup(&t, bd[x] + uD / sc[2], bw[y] + uW / sc[1], bh[z] + uH / sc[0], h, w, d);
            const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
            const float fx0 = (float)x0, fy0 = (float)y0, fz0 = (float)z0, fx1 = (float)x1, fy1 = (float)y1, fz1 = (float)z1;
            const int b000 = inb3(z0, y0, x0, h, w, d), b001 = inb3(z0, y0, x1, h, w, d), b010 = inb3(z0, y1, x0, h, w, d),
                      b011 = inb3(z0, y1, x1, h, w, d), b100 = inb3(z1, y0, x0, h, w, d), b101 = inb3(z1, y0, x1, h, w, d),
                      b110 = inb3(z1, y1, x0, h, w, d), b111 = inb3(z1, y1, x1, h, w, d);
            float gix = 0.f, giy = 0.f, giz = 0.f;
            double lv = 0.0;
            for (int c = 0; c < C; ++c) {
                const float* mv = M2 + (size_t)c * V;
                const float wv = tri_sample(&t, mv, h, w, d);
                const float df = wv - F2[(size_t)c * V + p];
                lv += (double)df * df;
                const float gOut = gsc * (2.0f * df);           /* PowBackward: grad * (2 * self) */
                if (b000) { const float val = mv[((size_t)z0 * w + y0) * d + x0];
                    gix -= val * (fy1 - t.iy) * (fz1 - t.iz) * gOut; giy -= val * (fx1 - t.ix) * (fz1 - t.iz) * gOut; giz -= val * (fx1 - t.ix) * (fy1 - t.iy) * gOut; }
                if (b001) { const float val = mv[((size_t)z0 * w + y0) * d + x1];
                    gix += val * (fy1 - t.iy) * (fz1 - t.iz) * gOut; giy -= val * (t.ix - fx0) * (fz1 - t.iz) * gOut; giz -= val * (t.ix - fx0) * (fy1 - t.iy) * gOut; }
                if (b010) { const float val = mv[((size_t)z0 * w + y1) * d + x0];
                    gix -= val * (t.iy - fy0) * (fz1 - t.iz) * gOut; giy += val * (fx1 - t.ix) * (fz1 - t.iz) * gOut; giz -= val * (fx1 - t.ix) * (t.iy - fy0) * gOut; }
                if (b011) { const float val = mv[((size_t)z0 * w + y1) * d + x1];
                    gix += val * (t.iy - fy0) * (fz1 - t.iz) * gOut; giy += val * (t.ix - fx0) * (fz1 - t.iz) * gOut; giz -= val * (t.ix - fx0) * (t.iy - fy0) * gOut; }
                if (b100) { const float val = mv[((size_t)z1 * w + y0) * d + x0];
                    gix -= val * (fy1 - t.iy) * (t.iz - fz0) * gOut; giy -= val * (fx1 - t.ix) * (t.iz - fz0) * gOut; giz += val * (fx1 - t.ix) * (fy1 - t.iy) * gOut; }
                if (b101) { const float val = mv[((size_t)z1 * w + y0) * d + x1];
                    gix += val * (fy1 - t.iy) * (t.iz - fz0) * gOut; giy -= val * (t.ix - fx0) * (t.iz - fz0) * gOut; giz += val * (t.ix - fx0) * (fy1 - t.iy) * gOut; }
                if (b110) { const float val = mv[((size_t)z1 * w + y1) * d + x0];
                    gix -= val * (t.iy - fy0) * (t.iz - fz0) * gOut; giy += val * (fx1 - t.ix) * (t.iz - fz0) * gOut; giz += val * (fx1 - t.ix) * (t.iy - fy0) * gOut; }
                if (b111) { const float val = mv[((size_t)z1 * w + y1) * d + x1];
                    gix += val * (t.iy - fy0) * (t.iz - fz0) * gOut; giy += val * (t.ix - fx0) * (t.iz - fz0) * gOut; giz += val * (t.ix - fx0) * (t.iy - fy0) * gOut; }
            }
            lsum += lv;
            /* grad wrt normalised grid (x,y,z) -> flip -> / scale : grad wrt U (H,W,D) */
            float g[3];
            g[0] = (gmz * giz) / sc[0]; g[1] = (gmy * giy) / sc[1]; g[2] = (gmx * gix) / sc[2];
            /* regulariser, accumulated in autograd's arrival order:
             *   data, D[:-1], D[1:], H[:-1], H[1:], W[:-1], W[1:]    (see DESIGN.md "autograd order") */
            for (int a = 0; a < 3; ++a) {
                const float* Ua = U + (size_t)a * V;
                float acc = g[a];
                if (x < d - 1) acc += -(cD * (2.0f * (Ua[p + 1] - Ua[p])));
                if (x > 0)     acc +=  (cD * (2.0f * (Ua[p] - Ua[p - 1])));
                if (z < h - 1) acc += -(cH * (2.0f * (Ua[p + (size_t)w * d] - Ua[p])));
                if (z > 0)     acc +=  (cH * (2.0f * (Ua[p] - Ua[p - (size_t)w * d])));
                if (y < w - 1) acc += -(cW * (2.0f * (Ua[p + d] - Ua[p])));
                if (y > 0)     acc +=  (cW * (2.0f * (Ua[p] - Ua[p - d])));
                gU[(size_t)a * V + p] = acc;
            }
        }
        if (loss_out) loss_out[it] = (float)(lsum * (double)cost_scale / (double)C / (double)V);
        /* adjoint of the three box filters (symmetric operator, ATen backward order) */
        orc_smooth(gU, t1, 3, h, w, d, sm, 1);
        if (G) memcpy(G, t1, sizeof(float) * 3 * V);
        /* torch.optim.Adam (single-tensor path), lr=1, betas=(0.9,0.999), eps=1e-8 */
        const int step = step0 + it + 1;
        const double beta1 = 0.9, beta2 = 0.999;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        const float w1 = (float)(1.0 - beta1);           /* lerp weight */
        const float b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
        const float bc2s = (float)sqrt(bc2);
        const float neg_step = (float)(-(1.0 / bc1));
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < 3 * V; ++i) {
            const float g = t1[i];
            const float mm = fmaf(w1, g - m[i], m[i]);             /* exp_avg.lerp_(grad, 1-beta1) */
            float vv = v[i] * b2;                                   /* exp_avg_sq.mul_(beta2) */
            vv = fmaf(omb2 * g, g, vv);                             /* .addcmul_(grad, grad, value=1-beta2): the ATen
                                                                       kernel rounds (value*g) and fuses the rest */
            const float den = orc_adam_sqrt(vv) / bc2s + 1e-8f;      /* (sqrt / bc2_sqrt).add_(eps) */
            P[i] = P[i] + (neg_step * mm) / den;                    /* addcdiv_(exp_avg, denom, value=-step_size) */
            m[i] = mm; v[i] = vv;
        }
    }
    free(t1); free(t2); free(gU); free(bh); free(bw); free(bd);
}

/* ------------------------------------------------------------------------------------------------
 * FAST (throughput) mode of the Adam instance optimisation -- opt-in, graded by end-point error against the reference's field,
 * NOT a restatement of ATen's evaluation order.  Same mathematics as orc_adam_run (convex_adam_MIND.py:163-179), cheaper arithmetic:
 *   (1) the ADJOINT of the three chained zero-padded 3^3 boxes (a symmetric operator) as separable sums -- per axis (H, then W,
 *       then D) three chained 1-D stages  t[i] = (t[i-1] + t[i]) + t[i+1]  with zeros outside the volume after every stage (each
 *       avg_pool3d zero-pads its own input), then ONE multiplication by (float)(1/19683).  The FORWARD boxes keep ATen's order
 *       (orc_smooth): the diffusion regulariser differentiates U twice, so U's rounding pattern decides how long the trajectory
 *       stays next to the reference's -- measured on the full-size benchmark pair against the reference's own capture
 *       (tests/golden/fullsize.npz, mean EPE after 20 / 40 / 80 iterations): exact mode 5.5e-6 / 5.2e-5 / 1.23e-3, this mode
 *       8.5e-6 / 3.9e-5 / 1.28e-3, the same with separable FORWARD boxes 7.2e-5 / 1.8e-4 / 2.29e-3;
 *   (2) the warp / data-term gradient with the per-voxel set-up (coordinates, floor, the eight corner weights) exactly as
 *       ATen's, but per channel an FMA chain for the warped value and eight corner accumulators  A_k += df * v_k ; the three
 *       gradient components are combined from the A_k once per voxel.
 * Round 5: the regulariser gradient and the Adam update are ATen's again (autograd's arrival order; sqrt / bc2_sqrt + eps with the
 * IEEE square root): both are cheap next to (1) and (2), and on four full-size captures of the reference (tests/golden/fullsize.npz,
 * fullsize2.npz) the mode with them stays closer to the reference at 20 / 40 iterations on every capture and on average at 80
 * (DESIGN.md section 11).
 * Every operation is a correctly rounded IEEE operation in a FIXED order, which the HIP kernels of adam_mode = "fast"
 * (convexadam_amd/csrc/adamfast.hip) follow step by step: HIP-fast == oracle-fast bit for bit (tests/test_gpu_parity.py).
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_fast_box3x3(const float* in, float* out, int C, int h, int w, int d) {
    const size_t V = (size_t)h * w * d;
    const int dims[3] = {h, w, d};
    const size_t strides[3] = {(size_t)w * d, (size_t)d, 1};
    if (out != in) memcpy(out, in, sizeof(float) * (size_t)C * V);
    for (int axis = 0; axis < 3; ++axis) {
        const int n = dims[axis];
        const size_t st = strides[axis];
        const size_t nlines = (size_t)C * V / (size_t)n;
#pragma omp parallel
        {
            float* a = (float*)malloc(sizeof(float) * (size_t)(n + 2));
            float* b = (float*)malloc(sizeof(float) * (size_t)(n + 2));
#pragma omp for schedule(static)
            for (size_t l = 0; l < nlines; ++l) {
                /* line l: all index combinations of the other two axes (and the channel) */
                size_t base;
                if (axis == 0) base = (l / ((size_t)w * d)) * V + l % ((size_t)w * d);
                else if (axis == 1) base = (l / d) * ((size_t)w * d) + l % d;
                else base = l * (size_t)d;
                a[0] = a[n + 1] = b[0] = b[n + 1] = 0.0f;
                for (int i = 0; i < n; ++i) a[i + 1] = out[base + (size_t)i * st];
                for (int s = 0; s < 3; ++s) {
                    for (int i = 1; i <= n; ++i) b[i] = (a[i - 1] + a[i]) + a[i + 1];
                    float* t = a; a = b; b = t;
                }
                for (int i = 0; i < n; ++i) out[base + (size_t)i * st] = a[i + 1];
            }
            free(a); free(b);
        }
    }
    const float r = (float)(1.0 / 19683.0);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)C * V; ++i) out[i] = out[i] * r;
}

/* The separable restatement of a CHAIN of zero-padded box filters (kovesi_spline, self_configuring/convexAdam_hyper_util.py:475-488; the
 * packaged three 3^3 boxes are the chain {3, 3, 3}): per axis (H, W, D) the boxes of the chain one after the other as 1-D sums
 * out[i] = ((in[i-r] + in[i-r+1]) + ...) + in[i+r] with zeros outside the line, `reverse` = the adjoint's order of the boxes (the 1-D
 * clipped boxes of DIFFERENT sizes do not commute at the borders; different axes do), then one multiplication by (float)(1 / prod k^3). */
ORC_API void orc_fast_boxchain(const float* in, float* out, int C, int h, int w, int d, int n_boxes, const int* box_k, int reverse) {
    const size_t V = (size_t)h * w * d;
    const int dims[3] = {h, w, d};
    const size_t strides[3] = {(size_t)w * d, (size_t)d, 1};
    if (out != in) memcpy(out, in, sizeof(float) * (size_t)C * V);
    double prod = 1.0;
    for (int i = 0; i < n_boxes; ++i) prod *= (double)box_k[i] * box_k[i] * box_k[i];
    for (int axis = 0; axis < 3; ++axis) {
        const int n = dims[axis];
        const size_t st = strides[axis];
        const size_t nlines = (size_t)C * V / (size_t)n;
#pragma omp parallel
        {
            float* a = (float*)calloc((size_t)(n + 16), sizeof(float));
            float* b = (float*)calloc((size_t)(n + 16), sizeof(float));
#pragma omp for schedule(static)
            for (size_t l = 0; l < nlines; ++l) {
                size_t base;
                if (axis == 0) base = (l / ((size_t)w * d)) * V + l % ((size_t)w * d);
                else if (axis == 1) base = (l / d) * ((size_t)w * d) + l % d;
                else base = l * (size_t)d;
                for (int i = 0; i < n + 16; ++i) a[i] = b[i] = 0.0f;
                for (int i = 0; i < n; ++i) a[i + 8] = out[base + (size_t)i * st];
                for (int q = 0; q < n_boxes; ++q) {
                    const int r = box_k[reverse ? n_boxes - 1 - q : q] / 2;
                    for (int i = 0; i < n; ++i) {
                        float sacc = a[i + 8 - r];
                        for (int j = -r + 1; j <= r; ++j) sacc += a[i + 8 + j];
                        b[i + 8] = sacc;
                    }
                    float* t = a; a = b; b = t;
                    for (int i = 0; i < 8; ++i) { a[i] = 0.0f; a[n + 8 + i] = 0.0f; }
                }
                for (int i = 0; i < n; ++i) out[base + (size_t)i * st] = a[i + 8];
            }
            free(a); free(b);
        }
    }
    const float rr = (float)(1.0 / prod);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)C * V; ++i) out[i] = out[i] * rr;
}

ORC_API void orc_adam_run_fast_smoother(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                               float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                               float* U, float* G, int keep_last_step, int fast_forward, const orc_smoother* sm);
ORC_API void orc_adam_run_fast(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                               float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                               float* U, float* G, int keep_last_step, int fast_forward) {
    orc_adam_run_fast_smoother(F2, M2, C, h, w, d, P, m, v, lambda_weight, niter, step0, cost_scale, U, G, keep_last_step, fast_forward, NULL);
}
/* sm == NULL: the packaged three 3^3 boxes.  A box chain (kind 0): the adjoint -- and with fast_forward the forward pass -- through
 * orc_fast_boxchain; a Gaussian (kind 1) keeps the exact smoother both ways (it is three short 1-D convolutions already). */
ORC_API void orc_adam_run_fast_smoother(const float* F2, const float* M2, int C, int h, int w, int d, float* P,
                               float* m, float* v, float lambda_weight, int niter, int step0, float cost_scale,
                               float* U, float* G, int keep_last_step, int fast_forward, const orc_smoother* sm) {
    const size_t V = (size_t)h * w * d;
    float* t1 = (float*)malloc(sizeof(float) * 3 * V);
    float* gU = (float*)malloc(sizeof(float) * 3 * V);
    float* bh = (float*)malloc(sizeof(float) * h); float* bw = (float*)malloc(sizeof(float) * w);
    float* bd = (float*)malloc(sizeof(float) * d);
    orc_affine_base(h, bh); orc_affine_base(w, bw); orc_affine_base(d, bd);
    const float sc[3] = {(float)((h - 1) / 2.0), (float)((w - 1) / 2.0), (float)((d - 1) / 2.0)};
    const float nH = (float)((int64_t)3 * (h - 1) * w * d), nW = (float)((int64_t)3 * h * (w - 1) * d),
                nD = (float)((int64_t)3 * h * w * (d - 1));
    const float cH = lambda_weight / nH, cW = lambda_weight / nW, cD = lambda_weight / nD;
    const float m2H = -2.0f * cH, m2W = -2.0f * cW, m2D = -2.0f * cD;          /* exact doublings */
    const float gsc = ((1.0f / (float)V) * cost_scale) / (float)C;
    const float gsc2 = 2.0f * gsc;
    const float gmx = (float)d / 2.0f, gmy = (float)w / 2.0f, gmz = (float)h / 2.0f;
    const orc_smoother boxes3 = {0, 3, {3, 3, 3, 0}, {0, 0, 0, 0, 0}};
    if (!sm) sm = &boxes3;
    const int chain = sm->kind == 0;
    /* study switch (tools/experiments/fast_adam_epe.py): 1 = round 4's one-division update, 2 = round 4's FMA regulariser; 0 (default) =
     * the accepted arithmetic of round 5: ATen's update and ATen's regulariser order */
    const int xflags = getenv("ORC_FAST_R4") ? atoi(getenv("ORC_FAST_R4")) : 0;
    for (int it = 0; it < niter; ++it) {
        if (fast_forward && chain) orc_fast_boxchain(P, U, 3, h, w, d, sm->n_boxes, sm->box_k, 0);   /* adam_mode "fast_all": NOT accepted by the criteria above (2.29e-3 at 80 iterations) */
        else orc_smooth(P, U, 3, h, w, d, sm, 0);            /* forward smoother: ATen's order, as in orc_adam_run */
        if (it == niter - 1 && !keep_last_step) break;      /* the pipeline never observes the last gradient / update */
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < V; ++p) {
            const int z = (int)(p / ((size_t)w * d)), y = (int)((p / d) % w), x = (int)(p % d);
            const float uH = U[p], uW = U[V + p], uD = U[2 * V + p];
            orc_tri_t t;
            tri_setup(&t, bd[x] + uD / sc[2], bw[y] + uW / sc[1], bh[z] + uH / sc[0], h, w, d);
            const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
            const float fx0 = (float)x0, fy0 = (float)y0, fz0 = (float)z0, fx1 = (float)x1, fy1 = (float)y1, fz1 = (float)z1;
            const float wx[2] = {fx1 - t.ix, t.ix - fx0}, wy[2] = {fy1 - t.iy, t.iy - fy0}, wz[2] = {fz1 - t.iz, t.iz - fz0};
            const float wgt[8] = {t.tnw, t.tne, t.tsw, t.tse, t.bnw, t.bne, t.bsw, t.bse};
            size_t off[8]; int inb[8];
            for (int k = 0; k < 8; ++k) {
                const int zz = (k & 4) ? z1 : z0, yy = (k & 2) ? y1 : y0, xx = (k & 1) ? x1 : x0;
                inb[k] = inb3(zz, yy, xx, h, w, d);
                off[k] = inb[k] ? ((size_t)zz * w + yy) * d + xx : 0;
            }
            float A[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C; ++c) {
                const float* mv = M2 + (size_t)c * V;
                float vk[8];
                for (int k = 0; k < 8; ++k) vk[k] = inb[k] ? mv[off[k]] : 0.0f;     /* corners outside the volume read zero */
                float wv = vk[0] * wgt[0];
                for (int k = 1; k < 8; ++k) wv = fmaf(vk[k], wgt[k], wv);
                const float df = wv - F2[(size_t)c * V + p];
                for (int k = 0; k < 8; ++k) A[k] = fmaf(df, vk[k], A[k]);
            }
            /* d warp / d ix = sum_k (+-) wy wz v_k etc. (corner k: bit 0 = x1, bit 1 = y1, bit 2 = z1) */
            float gix = 0.f, giy = 0.f, giz = 0.f;
            for (int k = 0; k < 8; ++k) {
                const int kx = k & 1, ky = (k >> 1) & 1, kz = (k >> 2) & 1;
                const float cx = wy[ky] * wz[kz], cy = wx[kx] * wz[kz], cz = wx[kx] * wy[ky];
                gix = fmaf(kx ? cx : -cx, A[k], gix);
                giy = fmaf(ky ? cy : -cy, A[k], giy);
                giz = fmaf(kz ? cz : -cz, A[k], giz);
            }
            gix = gix * gsc2; giy = giy * gsc2; giz = giz * gsc2;
            float g[3];
            g[0] = (gmz * giz) / sc[0]; g[1] = (gmy * giy) / sc[1]; g[2] = (gmx * gix) / sc[2];
            /* diffusion regulariser: d/dU of lam * mean(diff^2) along the three axes, order D+, D-, H+, H-, W+, W- */
            for (int a = 0; a < 3; ++a) {
                const float* Ua = U + (size_t)a * V;
                const float uc = Ua[p];
                float acc = g[a];
                if (!(xflags & 2)) {       /* autograd's arrival order, as in orc_adam_run */
                    if (x < d - 1) acc += -(cD * (2.0f * (Ua[p + 1] - uc)));
                    if (x > 0)     acc +=  (cD * (2.0f * (uc - Ua[p - 1])));
                    if (z < h - 1) acc += -(cH * (2.0f * (Ua[p + (size_t)w * d] - uc)));
                    if (z > 0)     acc +=  (cH * (2.0f * (uc - Ua[p - (size_t)w * d])));
                    if (y < w - 1) acc += -(cW * (2.0f * (Ua[p + d] - uc)));
                    if (y > 0)     acc +=  (cW * (2.0f * (uc - Ua[p - d])));
                    gU[(size_t)a * V + p] = acc;
                    continue;
                }
                if (x < d - 1) acc = fmaf(m2D, Ua[p + 1] - uc, acc);
                if (x > 0)     acc = fmaf(m2D, Ua[p - 1] - uc, acc);
                if (z < h - 1) acc = fmaf(m2H, Ua[p + (size_t)w * d] - uc, acc);
                if (z > 0)     acc = fmaf(m2H, Ua[p - (size_t)w * d] - uc, acc);
                if (y < w - 1) acc = fmaf(m2W, Ua[p + d] - uc, acc);
                if (y > 0)     acc = fmaf(m2W, Ua[p - d] - uc, acc);
                gU[(size_t)a * V + p] = acc;
            }
        }
        if (chain) orc_fast_boxchain(gU, t1, 3, h, w, d, sm->n_boxes, sm->box_k, 1);
        else orc_smooth(gU, t1, 3, h, w, d, sm, 1);
        if (G) memcpy(G, t1, sizeof(float) * 3 * V);
        const int step = step0 + it + 1;
        const double beta1 = 0.9, beta2 = 0.999;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
        const float inv_bc2s = (float)(1.0 / sqrt(bc2)), bc2s = (float)sqrt(bc2);
        const float neg_step = (float)(-(1.0 / bc1));
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < 3 * V; ++i) {
            const float g = t1[i];
            const float mm = fmaf(w1, g - m[i], m[i]);
            float vv = v[i] * b2;
            vv = fmaf(omb2 * g, g, vv);
            const float den = (xflags & 1) ? fmaf(sqrtf(vv), inv_bc2s, 1e-8f) : sqrtf(vv) / bc2s + 1e-8f;    /* (sqrt / bc2_sqrt).add_(eps), IEEE sqrt */
            P[i] = P[i] + (neg_step * mm) / den;
            m[i] = mm; v[i] = vv;
        }
    }
    free(t1); free(gU); free(bh); free(bw); free(bd);
}

/* ------------------------------------------------------------------------------------------------
 * torch.pow(float32 tensor, python scalar) on the CPU, as the reference evaluates `(...).float().pow(.3)` in its label weights
 * (src/convexAdam/convex_adam_nnUNet.py:31).  Black-box finding (inputs fed, outputs compared; tests/test_oracle_vs_reference_live.py):
 * ATen's vectorised loop handles the leading blocks of 32 elements with Sleef's powf (1.0-ULP variant, FMA build, exponent rounded to
 * float32) and the trailing `n mod 32` elements with the scalar lambda std::pow(float, double exponent), i.e. (float)pow((double)x, y).
 * The Sleef routine is restated from its published algorithm (sleefsimdsp.c: xpowf = sp_expkf(sp_logkf(|x|) * y) in double-float arithmetic,
 * dd.h / df.h with fused multiply-adds); it agrees with torch.pow on every one of the 16 777 184 block elements of arange(1, 2^24).
 * Positive bases only (voxel counts).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } sp_f2;
static inline float sp_i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t sp_f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline sp_f2 sp_mk(float x, float y) { sp_f2 r = {x, y}; return r; }
static inline sp_f2 sp_dfadd2_f_f(float x, float y) { sp_f2 r; r.x = x + y; float v = r.x - x; r.y = (x - (r.x - v)) + (y - v); return r; }
static inline sp_f2 sp_dfadd2_f2_f(sp_f2 x, float y) { sp_f2 r; r.x = x.x + y; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y - v); r.y = r.y + x.y; return r; }
static inline sp_f2 sp_dfadd2_f2_f2(sp_f2 x, sp_f2 y) { sp_f2 r; r.x = x.x + y.x; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y.x - v); r.y = r.y + (x.y + y.y); return r; }
static inline sp_f2 sp_dfadd_f2_f2(sp_f2 x, sp_f2 y) { sp_f2 r; r.x = x.x + y.x; r.y = x.x - r.x + y.x + x.y + y.y; return r; }
static inline sp_f2 sp_dfadd_f_f2(float x, sp_f2 y) { sp_f2 r; r.x = x + y.x; r.y = x - r.x + y.x + y.y; return r; }
static inline sp_f2 sp_dfmul_f2_f(sp_f2 x, float y) { sp_f2 r; r.x = x.x * y; r.y = fmaf(x.y, y, fmaf(x.x, y, -r.x)); return r; }
static inline sp_f2 sp_dfmul_f2_f2(sp_f2 x, sp_f2 y) { sp_f2 r; r.x = x.x * y.x; r.y = fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -r.x))); return r; }
static inline sp_f2 sp_dfsqu(sp_f2 x) { sp_f2 r; r.x = x.x * x.x; r.y = fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -r.x)); return r; }
static inline sp_f2 sp_dfdiv(sp_f2 n, sp_f2 d) {
    float t = 1.0f / d.x; sp_f2 q; q.x = n.x * t;
    float u = fmaf(t, n.x, -q.x);
    q.y = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
    q.y = fmaf(q.x, q.y, fmaf(n.y, t, u));
    return q;
}
static inline sp_f2 sp_dfscale(sp_f2 d, float s) { return sp_mk(d.x * s, d.y * s); }
static inline sp_f2 sp_dfnormalize(sp_f2 t) { sp_f2 s; s.x = t.x + t.y; s.y = t.x - s.x + t.y; return s; }
static sp_f2 sp_logkf(float d) {
    int o = d < 1.17549435e-38f;
    if (o) d = d * (float)(1LL << 32) * (float)(1LL << 32);
    int e = ((sp_f2i(d * (1.0f / 0.75f)) >> 23) & 0xff) - 0x7f;
    float m = sp_i2f(sp_f2i(d) + ((-e) << 23));
    if (o) e -= 64;
    sp_f2 x = sp_dfdiv(sp_dfadd2_f_f(-1.0f, m), sp_dfadd2_f_f(1.0f, m));
    sp_f2 x2 = sp_dfsqu(x);
    float t = 0.240320354700088500976562f;
    t = fmaf(t, x2.x, 0.285112679004669189453125f);
    t = fmaf(t, x2.x, 0.400007992982864379882812f);
    sp_f2 c = sp_mk(0.66666662693023681640625f, 3.69183861259614332084311e-09f);
    sp_f2 s = sp_dfmul_f2_f(sp_mk(0.69314718246459960938f, -1.904654323148236017e-09f), (float)e);
    s = sp_dfadd_f2_f2(s, sp_dfscale(x, 2.0f));
    s = sp_dfadd_f2_f2(s, sp_dfmul_f2_f2(sp_dfmul_f2_f2(x2, x), sp_dfadd2_f2_f2(sp_dfmul_f2_f(x2, t), c)));
    return s;
}
static float sp_ldexpkf(float x, int q) {
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    m += 127; m = m < 0 ? 0 : m; m = m > 255 ? 255 : m;
    float u = sp_i2f(m << 23);
    x = x * u * u * u * u;
    u = sp_i2f((q + 0x7f) << 23);
    return x * u;
}
static float sp_expkf(sp_f2 d) {
    float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    int q = (int)rintf(u);
    sp_f2 s = sp_dfadd2_f2_f(d, (float)q * -0.693145751953125f);
    s = sp_dfadd2_f2_f(s, (float)q * -1.428606765330187045e-06f);
    s = sp_dfnormalize(s);
    float t = 0.00136324646882712841033936f;
    t = fmaf(t, s.x, 0.00836596917361021041870117f);
    t = fmaf(t, s.x, 0.0416710823774337768554688f);
    t = fmaf(t, s.x, 0.166665524244308471679688f);
    t = fmaf(t, s.x, 0.499999850988388061523438f);
    sp_f2 tt = sp_dfadd_f2_f2(s, sp_dfmul_f2_f(sp_dfsqu(s), t));
    tt = sp_dfadd_f_f2(1.0f, tt);
    u = tt.x + tt.y;
    u = sp_ldexpkf(u, q);
    if (d.x < -104.0f) u = 0.0f;
    return u;
}
static float sp_sleef_powf(float x, float y) {          /* x > 0 only */
    return sp_expkf(sp_dfmul_f2_f(sp_logkf(fabsf(x)), y));
}
/* element i of an n-element tensor; the vectorised loop covers blocks of 2 x (vector width of the host's ATen build) elements:
 * 32 on an AVX-512 host (the host that produced tests/golden), 16 on an AVX2 host -- like the MKL tables, a property of the reference HOST */
static int g_pow_block = 32;
ORC_API void orc_set_pow_block(int b) { g_pow_block = b > 0 ? b : 32; }
static float sp_torch_pow_at(float x, double y, int64_t i, int64_t n) {
    return i < (n / g_pow_block) * g_pow_block ? sp_sleef_powf(x, (float)y) : (float)pow((double)x, y);
}
ORC_API float orc_torch_pow_at(float x, double y, int64_t i, int64_t n) { return sp_torch_pow_at(x, y, i, n); }

/* ------------------------------------------------------------------------------------------------
 * nnUNet label features, convex_adam_nnUNet.py:19-38 (fp32 restatement; the reference stores fp16).
 * labels are float-valued integer maps; present = labels occurring in either image (ascending).
 * Returns C (number of channels); feat_* = [C][V] = 10 * w_c * onehot.
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_label_features(const float* lab_fix, const float* lab_mov, int64_t V, int max_label,
                               float mult, float* feat_fix, float* feat_mov, int* present_out) {
    int64_t* cf = (int64_t*)calloc(max_label + 1, sizeof(int64_t));
    int64_t* cm = (int64_t*)calloc(max_label + 1, sizeof(int64_t));
    for (int64_t i = 0; i < V; ++i) { cf[(int)lab_fix[i]]++; cm[(int)lab_mov[i]]++; }
    int C = 0; int* present = (int*)malloc(sizeof(int) * (max_label + 1));
    for (int l = 0; l <= max_label; ++l) if (cf[l] + cm[l] > 0) present[C++] = l;
    float* wt = (float*)malloc(sizeof(float) * C);
    /* weight = 1/((n_fix+n_mov)+eps).float().pow(.3) ; weight /= weight.mean() */
    for (int c = 0; c < C; ++c) wt[c] = 1.0f / sp_torch_pow_at((float)(cf[present[c]] + cm[present[c]]) + 1e-32f, 0.3, c, C);
    const float mean = orc_torch_sum(wt, C, 1, 8) / (float)C;                   /* weight.mean(): ATen's sum (one chunk), then / C */
    for (int c = 0; c < C; ++c) wt[c] = wt[c] / mean;
    if (feat_fix && feat_mov)
        for (int c = 0; c < C; ++c)
            for (int64_t i = 0; i < V; ++i) {
                feat_fix[(size_t)c * V + i] = mult * (((int)lab_fix[i] == present[c] ? 1.0f : 0.0f) * wt[c]);
                feat_mov[(size_t)c * V + i] = mult * (((int)lab_mov[i] == present[c] ? 1.0f : 0.0f) * wt[c]);
            }
    if (present_out) memcpy(present_out, present, sizeof(int) * C);
    free(cf); free(cm); free(present); free(wt);
    return C;
}

/* ------------------------------------------------------------------------------------------------
 * Euclidean feature transform (index of the nearest zero element), used by the masked feature path:
 * convex_adam_MIND.py:44,49 calls scipy.ndimage.distance_transform_edt(..., return_indices=True).
 * scipy is a third-party dependency of the reference (not vendored; 1.15.3 in this image); its exact transform is the
 * dimension-by-dimension Voronoi algorithm of Maurer, Qi & Raghavan (IEEE TPAMI 25(2), 2003) as implemented in
 * scipy/ndimage/src/ni_measure.c (_VoronoiFT / _ComputeFT).  Ties between equidistant sites are resolved by the
 * order of that construction (`<= 0` keeps the older site when a site is removed, `delta1 <= delta2` keeps the
 * lower site when scanning), which this restatement reproduces: tests/test_oracle_vs_golden.py compares it with scipy
 * itself on random and tie-heavy masks.
 *   obj  [H][W][D] : non-zero = object voxel (needs its nearest zero), zero = site
 *   feat [3][H][W][D] int32 : coordinates of the nearest site along axes 0, 1, 2
 * ---------------------------------------------------------------------------------------------- */
static void orc_voronoi_ft(int32_t* pf, int len, const int* coor, int d, int64_t stride, int64_t cstride, int (*f)[3], int* g) {
    int l = -1, maxl;
    for (int ii = 0; ii < len; ii++) for (int jj = 0; jj < 3; jj++) f[ii][jj] = pf[ii * stride + cstride * jj];
    for (int ii = 0; ii < len; ii++) {
        if (pf[ii * stride] < 0) continue;
        double fd = f[ii][d], wR = 0.0;
        for (int jj = 0; jj < 3; jj++) if (jj != d) { const double tw = f[ii][jj] - coor[jj]; wR += tw * tw; }
        while (l >= 1) {
            const int idx1 = g[l], idx2 = g[l - 1];
            const double f1 = f[idx1][d], a = f1 - f[idx2][d], b = fd - f1, c = a + b;
            double uR = 0.0, vR = 0.0;
            for (int jj = 0; jj < 3; jj++) if (jj != d) {
                const double cc = coor[jj], tu = f[idx2][jj] - cc, tv = f[idx1][jj] - cc;
                uR += tu * tu; vR += tv * tv;
            }
            if (c * vR - b * uR - a * wR - a * b * c <= 0.0) break;
            --l;
        }
        g[++l] = ii;
    }
    maxl = l;
    if (maxl < 0) return;
    l = 0;
    for (int ii = 0; ii < len; ii++) {
        double delta1 = 0.0;
        for (int jj = 0; jj < 3; jj++) { const double t = jj == d ? f[g[l]][jj] - ii : f[g[l]][jj] - coor[jj]; delta1 += t * t; }
        while (l < maxl) {
            double delta2 = 0.0;
            for (int jj = 0; jj < 3; jj++) { const double t = jj == d ? f[g[l + 1]][jj] - ii : f[g[l + 1]][jj] - coor[jj]; delta2 += t * t; }
            if (delta1 <= delta2) break;
            delta1 = delta2;
            ++l;
        }
        for (int jj = 0; jj < 3; jj++) pf[ii * stride + jj * cstride] = f[g[l]][jj];
    }
}
ORC_API void orc_feature_transform(const float* obj, int H, int W, int D, int32_t* feat) {
    const int64_t V = (int64_t)H * W * D;
    int maxlen = H > W ? H : W;
    if (D > maxlen) maxlen = D;
#pragma omp parallel
    {
        int (*f)[3] = malloc(sizeof(int[3]) * (size_t)maxlen);
        int* g = malloc(sizeof(int) * (size_t)maxlen);
        int coor[3];
#pragma omp for collapse(2) schedule(static)
        for (int y = 0; y < W; y++) for (int x = 0; x < D; x++) {                 /* axis 0 */
            coor[0] = 0; coor[1] = y; coor[2] = x;
            int32_t* pf = feat + (int64_t)y * D + x;
            for (int z = 0; z < H; z++) {
                int32_t* q = pf + (int64_t)z * W * D;
                if (obj[((int64_t)z * W + y) * D + x] != 0.0f) { q[0] = -1; q[V] = -1; q[2 * V] = -1; }
                else { q[0] = z; q[V] = y; q[2 * V] = x; }
            }
            orc_voronoi_ft(pf, H, coor, 0, (int64_t)W * D, V, f, g);
        }
#pragma omp for collapse(2) schedule(static)
        for (int z = 0; z < H; z++) for (int x = 0; x < D; x++) {                 /* axis 1 */
            coor[0] = z; coor[1] = 0; coor[2] = x;
            orc_voronoi_ft(feat + (int64_t)z * W * D + x, W, coor, 1, D, V, f, g);
        }
#pragma omp for collapse(2) schedule(static)
        for (int z = 0; z < H; z++) for (int y = 0; y < W; y++) {                 /* axis 2 */
            coor[0] = z; coor[1] = y; coor[2] = 0;
            orc_voronoi_ft(feat + ((int64_t)z * W + y) * D, D, coor, 2, 1, V, f, g);
        }
        free(f); free(g);
    }
}
