// correlate.hip -- dense SSD correlation volume (reference: convex_adam_utils.py:72-89).
//
//   raw[k,x] = sum_c (F_c(x) - M0_c(x + delta_k))^2          M0 = zero-padded moving features
//   ssd      = box3(box3(raw))                                 zero pad, raster-order 27-tap sums, /27
//   k        = (dD+hw)*n^2 + (dW+hw)*n + (dH+hw),  n = 2*hw+1
//
// Data flow (all float32, D fastest; "px" rows hold element x at index x+1, zero elsewhere):
//   k_corr_prep : F -> Fp [C][h][w][px]        M -> Mp [C][h+2hw][w+2hw][dq]  zero border of hw voxels
//   k_corr_raw  : one thread = one aligned run of 4 row indices x ALL n D-shifts of one (dH,dW) pair; per
//                 channel it loads 1 float4 of F and (2*PL+4)/4 float4 of the M row and updates 4*n
//                 accumulators in registers in channel order (the reference's `.sum(0)` order, incl. ATen's
//                 16-wide cascade for C >= 16).  Writes raw [K][h][w][px] with exact zeros on the borders.
//   k_corr_tail : re-evaluates the <= 31 trailing elements per H-shift whose channel sum ATen evaluates in
//                 its 4-way interleaved order (see oracle outer_sum_rows).
//   k_corr_box  : one workgroup per (k, y-tile) marches along z with two 4-plane LDS rings (raw, box1); one
//                 thread per (row, pair of columns).  Each 27-tap raster sum reads 9 aligned 16-byte windows
//                 [x-1 .. x+2] as two 8-byte halves: two packed adds + two scalar adds per tap row, no
//                 cross-lane traffic, no bank conflicts; box1 is stored shifted by one column so that the
//                 windows of the second box (pairs x = 2j-1, 2j) are aligned as well; division by 27 is the
//                 exact FMA form (div_exact<27>).  No halo recomputation in z (and none in y for OASIS).
// Roofline: HBM by bytes (K*v*4 written + 2*C*v*4 read); the reference's summation order costs
// 2 x (26 adds + 1 division) + 36 flops per output, which makes the stage VALU-bound.
#include <stdlib.h>

#include "cvx_common.h"

namespace cvx {

struct CorrGeom {
    int C, h, w, d, hw, n;
    int px;     // F/raw row pitch: element x at index x+1, multiple of 4, >= d+3
    int PL;     // left pad of Mp rows (multiple of 4, >= hw)
    int dq;     // Mp row pitch: element x at index x + PL + 1
    int hq, wq; // Mp plane extents (h+2hw, w+2hw)
};
static CorrGeom corr_geom(int C, int h, int w, int d, int hw) {
    CorrGeom g;
    g.C = C; g.h = h; g.w = w; g.d = d; g.hw = hw; g.n = 2 * hw + 1;
    g.px = (d + 3 + 3) / 4 * 4;
    g.PL = (hw + 3) / 4 * 4;
    g.dq = g.px + 2 * g.PL;
    g.hq = h + 2 * hw; g.wq = w + 2 * hw;
    return g;
}

__global__ __launch_bounds__(256) void k_corr_prep(const float* __restrict__ fix, const float* __restrict__ mov, CorrGeom g,
                                                   float* __restrict__ Fp, float* __restrict__ Mp) {
    const size_t nF = (size_t)g.C * g.h * g.w * g.px, nM = (size_t)g.C * g.hq * g.wq * g.dq;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nF) {
        const int x = (int)(i % g.px) - 1;
        const size_t r = i / g.px;     // (c*h + z)*w + y
        Fp[i] = (x >= 0 && x < g.d) ? fix[r * g.d + x] : 0.0f;
    }
    if (i < nM) {
        const int xq = (int)(i % g.dq), yq = (int)((i / g.dq) % g.wq), zq = (int)((i / ((size_t)g.dq * g.wq)) % g.hq);
        const int c = (int)(i / ((size_t)g.dq * g.wq * g.hq));
        const int x = xq - g.PL - 1, y = yq - g.hw, z = zq - g.hw;
        const bool in = x >= 0 && x < g.d && y >= 0 && y < g.w && z >= 0 && z < g.h;
        Mp[i] = in ? mov[(((size_t)c * g.h + z) * g.w + y) * g.d + x] : 0.0f;
    }
}

// ---- raw SSD: register tile of 4 row indices x n D-shifts ---------------------------------------------
template <int HW, bool CASCADE>
__global__ __launch_bounds__(256) void k_corr_raw(const float* __restrict__ Fp, const float* __restrict__ Mp, CorrGeom g,
                                                  float* __restrict__ raw) {
    constexpr int N = 2 * HW + 1;
    constexpr int PL = (HW + 3) / 4 * 4;
    constexpr int NCH = (2 * PL + 4) / 4;          // float4 chunks of the M row segment
    const int runs_per_row = g.px / 4;
    const int nruns = g.h * g.w * runs_per_row;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const int i0 = 4 * (r % runs_per_row), y = (r / runs_per_row) % g.w, z = r / (runs_per_row * g.w);   // row index i = x + 1
    const int iH = blockIdx.y % N, iW = blockIdx.y / N;   // dH + hw, dW + hw

    float acc[N][4];
    float acc1[CASCADE ? N : 1][4];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.0f;
    if (CASCADE) {
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[k][j] = 0.0f;
    }
    const size_t fstride = (size_t)g.h * g.w * g.px, mstride = (size_t)g.hq * g.wq * g.dq;
    const float* fp = Fp + ((size_t)z * g.w + y) * g.px + i0;
    const float* mp = Mp + ((size_t)(z + iH) * g.wq + (y + iW)) * g.dq + i0;   // Mp index of M(x+dD) = i + PL + dD

    // software pipeline over channels: the loads of channel c+1 are issued before the arithmetic of channel c
    float4 fq[2];
    float4 mq[2][NCH];
    fq[0] = *reinterpret_cast<const float4*>(fp);
#pragma unroll
    for (int q = 0; q < NCH; ++q) mq[0][q] = *reinterpret_cast<const float4*>(mp + 4 * q);
    auto consume = [&](const float4& f4, const float4 (&mv)[NCH], int c) {
        const float f[4] = {f4.x, f4.y, f4.z, f4.w};
        float m[4 * NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) { m[4 * q] = mv[q].x; m[4 * q + 1] = mv[q].y; m[4 * q + 2] = mv[q].z; m[4 * q + 3] = mv[q].w; }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float df = f[j] - m[PL + j + k - HW];
                acc[k][j] += df * df;
            }
        if (CASCADE && ((c & 15) == 15)) {           // ATen multi_row_sum: fold every 16 rows
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc1[k][j] += acc[k][j]; acc[k][j] = 0.0f; }
        }
    };
#pragma unroll 1
    for (int c = 0; c < g.C; c += 2) {
        if (c + 1 < g.C) {
            fq[1] = *reinterpret_cast<const float4*>(fp + (size_t)(c + 1) * fstride);
#pragma unroll
            for (int q = 0; q < NCH; ++q) mq[1][q] = *reinterpret_cast<const float4*>(mp + (size_t)(c + 1) * mstride + 4 * q);
        }
        consume(fq[0], mq[0], c);
        if (c + 1 < g.C) {
            if (c + 2 < g.C) {
                fq[0] = *reinterpret_cast<const float4*>(fp + (size_t)(c + 2) * fstride);
#pragma unroll
                for (int q = 0; q < NCH; ++q) mq[0][q] = *reinterpret_cast<const float4*>(mp + (size_t)(c + 2) * mstride + 4 * q);
            }
            consume(fq[1], mq[1], c + 1);
        }
    }
    const size_t v = (size_t)g.h * g.w * g.px;
    bool inx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inx[j] = (i0 + j >= 1) && (i0 + j <= g.d);     // x = i - 1 in [0, d)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = CASCADE ? acc[k][j] + acc1[k][j] : acc[k][j];
            o[j] = inx[j] ? s : 0.0f;                 // the boxes zero-pad: border columns must be exact zeros
        }
        const size_t kk = ((size_t)k * N + iW) * N + iH;
        *reinterpret_cast<float4*>(raw + kk * v + ((size_t)z * g.w + y) * g.px + i0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- ATen interleaved-order tail ---------------------------------------------------------------------
__device__ float sum_cascade_strided(const float* v, int stride, int size) {   // level step 16, two levels
    float a0 = 0.f, a1 = 0.f;
    int i = 0;
    for (; i + 16 <= size; i += 16) {
        for (int j = 0; j < 16; ++j) a0 += v[(i + j) * stride];
        a1 += a0; a0 = 0.f;
    }
    for (; i < size; ++i) a0 += v[i * stride];
    return a0 + a1;
}
__global__ void k_corr_tail(const float* __restrict__ fix, const float* __restrict__ mov, CorrGeom g, int64_t tail_from,
                            int ntail, float* __restrict__ raw) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntail * g.n) return;
    const int iH = t / ntail;
    const int64_t flat = tail_from + (t % ntail);     // index into the reference's (h, n^2, w, d) tensor
    const int x = (int)(flat % g.d), y = (int)((flat / g.d) % g.w);
    const int jj = (int)((flat / ((int64_t)g.d * g.w)) % (g.n * g.n)), z = (int)(flat / ((int64_t)g.d * g.w * g.n * g.n));
    const int iW = jj / g.n, iD = jj % g.n;
    const int mz = z + iH - g.hw, my = y + iW - g.hw, mx = x + iD - g.hw;
    const bool inb = mz >= 0 && mz < g.h && my >= 0 && my < g.w && mx >= 0 && mx < g.d;
    const size_t v = (size_t)g.h * g.w * g.d;
    float sq[256];
    for (int c = 0; c < g.C; ++c) {
        const float f = fix[(size_t)c * v + ((size_t)z * g.w + y) * g.d + x];
        const float m = inb ? mov[(size_t)c * v + ((size_t)mz * g.w + my) * g.d + mx] : 0.0f;
        const float df = f - m;
        sq[c] = df * df;
    }
    const int n4 = g.C / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = sum_cascade_strided(sq + k, 4, n4);
    for (int i = n4 * 4; i < g.C; ++i) p[0] += sq[i];
    p[0] += p[1]; p[0] += p[2]; p[0] += p[3];
    const size_t kk = ((size_t)iD * g.n + iW) * g.n + iH;
    raw[kk * ((size_t)g.h * g.w * g.px) + ((size_t)z * g.w + y) * g.px + x + 1] = p[0];
}

// ---- two box filters, marching along z ---------------------------------------------------------------------
// One workgroup = one displacement k x one y-tile, one thread = one (row, pair of columns).  The workgroup walks
// the h planes once: at step t it (1) stores raw plane t (prefetched into registers during the previous step) into
// a 4-slot LDS ring, (2) evaluates the first box for plane t-2 from ring slots t-3..t-1 into a second 4-slot ring
// (stored shifted by one column so that the second box's windows are aligned too), (3) evaluates the second box for
// plane t-4 and stores it to global memory; one barrier per step.  No halo recomputation in z, and none in y when
// the whole plane fits (w + 2 <= 1024 / pairs-per-row).
struct BoxGeom {
    int h, w, d, px;
    int nj;             // column pairs per row (covers row indices 0 .. 2*nj+1)
    int Ty, nytiles;    // output rows per tile
    int ry;             // rows handled concurrently = Ty + 2 ; workgroup = nj * ry threads
    int nthreads;
};

// raster-order partial sum over the 9 taps of ONE plane for the two outputs whose 4-wide window starts at `win`
// (plain scalar adds: on gfx950 v_pk_add_f32 issues at half the rate of v_add_f32 -- 5.0 vs 2.5 cycles per
// wave-instruction measured -- so packing buys nothing and its v_pk_mov shuffles cost extra)
__device__ __forceinline__ void box9_pair(const float* __restrict__ win, int px, float& s0, float& s1) {
    f32x2 lo[3], hi[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) { lo[b] = lds_load2(win + (b - 1) * px); hi[b] = lds_load2(win + (b - 1) * px + 2); }   // 6 loads in flight
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        s0 += lo[b].x; s1 += lo[b].y;     // taps x-1 | x
        s0 += lo[b].y; s1 += hi[b].x;     // taps x   | x+1
        s0 += hi[b].x; s1 += hi[b].y;     // taps x+1 | x+2
    }
}

// one marching step with compile-time ring slots (S = t mod 4): all LDS offsets are scalar constants x plane pitch.
// (Keeping the two older planes of each sum in registers instead of re-reading them from LDS was measured slower:
//  141 VGPRs halve the number of resident workgroups.)
template <int S>
__device__ __forceinline__ void box_step(int t, const BoxGeom& b, int pp, int px, bool stage_row, bool stager, float4& pre,
                                         const float* __restrict__& sp, size_t gplane, float* __restrict__ sdst,
                                         bool row1, bool row2, const float* __restrict__ wA, float* __restrict__ oB,
                                         const float* __restrict__ wB, float* __restrict__& dst, bool m0, bool m1, int xa, int xb) {
    constexpr int S1 = (S + 1) & 3, S2 = (S + 2) & 3, S3 = (S + 3) & 3;      // slots of planes t-3, t-2, t-1 (= t+1, t+2, t+3 mod 4)
    // (1) raw plane t -> ring slot S, prefetch plane t+1
    if (stage_row) *reinterpret_cast<float4*>(sdst + S * pp) = (stager && t < b.h) ? pre : make_float4(0.f, 0.f, 0.f, 0.f);
    if (stager && t + 1 < b.h) { sp += gplane; pre = *reinterpret_cast<const float4*>(sp); }
    // (2) first box for plane t-2 (taps in planes t-3, t-2, t-1), zero outside the volume
    {
        float s0 = 0.0f, s1 = 0.0f;
        if (row1 && t >= 2 && t - 2 < b.h) {
            box9_pair(wA + S1 * pp, px, s0, s1);
            box9_pair(wA + S2 * pp, px, s0, s1);
            box9_pair(wA + S3 * pp, px, s0, s1);
            s0 = m0 ? div_exact<27>(s0) : 0.0f;       // the second pool zero-pads box1
            s1 = m1 ? div_exact<27>(s1) : 0.0f;
        }
        lds_store2(oB + S2 * pp, f32x2{s0, s1});     // box1 plane t-2 lives in slot (t-2) mod 4
    }
    // (3) second box for plane t-4 (taps in box1 planes t-5, t-4, t-3 = slots S3, S, S1)
    if (t >= 4) {
        if (row2) {
            float s0 = 0.0f, s1 = 0.0f;
            box9_pair(wB + S3 * pp, px, s0, s1);
            box9_pair(wB + S * pp, px, s0, s1);
            box9_pair(wB + S1 * pp, px, s0, s1);
            if (xa >= 0 && xa < b.d) dst[xa] = div_exact<27>(s0);
            if (xb < b.d) dst[xb] = div_exact<27>(s1);
        }
        dst += (size_t)b.w * b.d;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_corr_box(const float* __restrict__ raw, BoxGeom b, float* __restrict__ ssd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int k = blockIdx.x, y0 = blockIdx.y * b.Ty;
    const int ty = min(b.Ty, b.w - y0);
    const int rows = b.Ty + 4;                         // LDS rows hold y0-2 .. y0+Ty+1
    const int px = b.px, pp = rows * px;               // row / plane pitch in LDS
    float* A = lds;                                    // raw ring, 4 planes (element x at index x+1)
    float* B = lds + 4 * pp;                           // box1 ring, 4 planes (element x at index x+2)
    const float* rk = raw + (size_t)k * ((size_t)b.h * b.w * px);

    for (int i = tid * 4; i < 8 * pp; i += b.nthreads * 4) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging role: thread -> (LDS row, 16-byte chunk) of one plane
    const int c4 = px / 4;
    const int srow = tid / c4, scx = tid % c4;
    const int sgy = y0 - 2 + srow;
    const bool stage_row = srow < rows;
    const bool stager = stage_row && sgy >= 0 && sgy < b.w;
    const float* sp = rk + (size_t)(stager ? sgy : 0) * px + 4 * scx;
    float* sdst = A + srow * px + 4 * scx;
    // compute role: px/2 lanes per row, so that lane l reads LDS dwords 2l .. 2l+3 relative to lane 0: the 8-byte
    // accesses of a wavefront are contiguous and bank-conflict free (19 lanes per 40-float row were 2-way conflicted:
    // 45 % of the LDS cycles in the PMC profile); the last lane(s) of a row are dummies
    const int lpr = px / 2;
    const int j = tid % lpr, ry = tid / lpr;
    const int gy1 = y0 - 1 + ry;                       // row of the first box
    const bool row1 = gy1 >= 0 && gy1 < b.w && j < b.nj && ry < b.ry;
    const bool row2 = ry < ty && j < b.nj;            // row y0 + ry of the second box
    const float* wA = A + (ry + 1) * px + 2 * j;       // window x = 2j-1 .. 2j+2 -> outputs x = 2j, 2j+1
    float* oB = B + (ry + 1) * px + 2 * j + 2;         // where those two outputs live in the shifted layout
    const float* wB = B + (ry + 2) * px + 2 * j;       // window x = 2j-2 .. 2j+1 -> outputs x = 2j-1, 2j
    float* dst = ssd + (size_t)k * ((size_t)b.h * b.w * b.d) + (size_t)(y0 + ry) * b.d;
    const bool m0 = 2 * j < b.d, m1 = 2 * j + 1 < b.d;
    const int xa = 2 * j - 1, xb = 2 * j;

    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (stager && b.h > 0) pre = *reinterpret_cast<const float4*>(sp);
    __syncthreads();
    const size_t gplane = (size_t)b.w * px;
    const int nsteps = b.h + 4;
    for (int t = 0; t < nsteps; t += 4) {
        box_step<0>(t, b, pp, px, stage_row, stager, pre, sp, gplane, sdst, row1, row2, wA, oB, wB, dst, m0, m1, xa, xb);
        if (t + 1 < nsteps) box_step<1>(t + 1, b, pp, px, stage_row, stager, pre, sp, gplane, sdst, row1, row2, wA, oB, wB, dst, m0, m1, xa, xb);
        if (t + 2 < nsteps) box_step<2>(t + 2, b, pp, px, stage_row, stager, pre, sp, gplane, sdst, row1, row2, wA, oB, wB, dst, m0, m1, xa, xb);
        if (t + 3 < nsteps) box_step<3>(t + 3, b, pp, px, stage_row, stager, pre, sp, gplane, sdst, row1, row2, wA, oB, wB, dst, m0, m1, xa, xb);
    }
}

static BoxGeom box_geom(int h, int w, int d, int px) {
    BoxGeom b;
    b.h = h; b.w = w; b.d = d; b.px = px;
    b.nj = (d + 2) / 2;                                // pairs j = 0 .. d/2 reach x = d-1 in both passes
    b.Ty = b.nytiles = b.ry = b.nthreads = 0;
    if (b.nj > 340) return b;
    static int max_threads = 0;
    if (max_threads == 0) {
        const char* e = getenv("CVX_BOX_THREADS");
        max_threads = e ? atoi(e) : 1024;
        if (max_threads < 64 || max_threads > 1024) max_threads = 1024;
    }
    int ry = max_threads / (px / 2);                   // rows per workgroup (px/2 lanes per row)
    if (ry > w + 2) ry = w + 2;
    if (ry < 3) return b;
    b.nytiles = cdiv(w, ry - 2);
    b.Ty = cdiv(w, b.nytiles);
    b.nytiles = cdiv(w, b.Ty);
    b.ry = b.Ty + 2;
    b.nthreads = (px / 2) * b.ry;
    const int stagers = (b.Ty + 4) * (px / 4);         // every LDS row needs a staging thread
    if (b.nthreads < stagers) b.nthreads = stagers;
    if (b.nthreads > 1024 || sizeof(float) * 8 * (size_t)(b.Ty + 4) * px > 160 * 1024) { b.nthreads = 0; return b; }
    return b;
}

template <int HW>
static void corr_raw_dispatch(const float* Fp, const float* Mp, const CorrGeom& g, float* raw, hipStream_t s) {
    const int nruns = g.h * g.w * (g.px / 4);
    const dim3 grid(cdiv(nruns, 256), g.n * g.n);
    if (g.C >= 16) hipLaunchKernelGGL((k_corr_raw<HW, true>), grid, dim3(256), 0, s, Fp, Mp, g, raw);
    else hipLaunchKernelGGL((k_corr_raw<HW, false>), grid, dim3(256), 0, s, Fp, Mp, g, raw);
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_correlate_workspace_bytes(int C, int h, int w, int d, int disp_hw) {
    const CorrGeom g = corr_geom(C, h, w, d, disp_hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    size_t used = 0;
    used = carve_size(used, sizeof(float) * (size_t)C * h * w * g.px);            // Fp
    used = carve_size(used, sizeof(float) * (size_t)C * g.hq * g.wq * g.dq);      // Mp
    used = carve_size(used, sizeof(float) * K * h * w * g.px);                    // raw
    used = carve_size(used, sizeof(unsigned long long) * (size_t)h * w * d);      // argmin keys
    return used + 256;
}

extern "C" int cvx_correlate_f32(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, float* ssd,
                                 int64_t* argmin, void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(fix && mov && ssd && workspace, "cvx_correlate_f32: null pointer");
    CVX_REQUIRE(C > 0 && C < 256 && h > 0 && w > 0 && d > 0, "cvx_correlate_f32: bad extent C=%d %dx%dx%d", C, h, w, d);
    CVX_REQUIRE(disp_hw >= 0, "cvx_correlate_f32: negative disp_hw");
    if (disp_hw > 8) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: disp_hw %d > 8 not built", disp_hw);
    if (workspace_bytes < cvx_correlate_workspace_bytes(C, h, w, d, disp_hw))
        return fail(CVX_ERR_WORKSPACE, "cvx_correlate_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const CorrGeom g = corr_geom(C, h, w, d, disp_hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    const BoxGeom b = box_geom(h, w, d, g.px);
    if (b.nthreads == 0 && !corr_box2_supported(h, w, d, g.px))
        return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: coarse rows of %d voxels are too long for the LDS box kernel", d);
    Carver cv(workspace, workspace_bytes);
    float* Fp = cv.take<float>((size_t)C * h * w * g.px);
    float* Mp = cv.take<float>((size_t)C * g.hq * g.wq * g.dq);
    float* raw = cv.take<float>(K * h * w * g.px);
    unsigned long long* keys = cv.take<unsigned long long>((size_t)h * w * d);

    const size_t nprep = (size_t)C * g.hq * g.wq * g.dq;   // >= nF
    hipLaunchKernelGGL(k_corr_prep, dim3((unsigned)cdiv64((int64_t)nprep, 256)), dim3(256), 0, s, fix, mov, g, Fp, Mp);
    switch (disp_hw) {
        case 0: corr_raw_dispatch<0>(Fp, Mp, g, raw, s); break;
        case 1: corr_raw_dispatch<1>(Fp, Mp, g, raw, s); break;
        case 2: corr_raw_dispatch<2>(Fp, Mp, g, raw, s); break;
        case 3: corr_raw_dispatch<3>(Fp, Mp, g, raw, s); break;
        case 4: corr_raw_dispatch<4>(Fp, Mp, g, raw, s); break;
        case 5: corr_raw_dispatch<5>(Fp, Mp, g, raw, s); break;
        case 6: corr_raw_dispatch<6>(Fp, Mp, g, raw, s); break;
        case 7: corr_raw_dispatch<7>(Fp, Mp, g, raw, s); break;
        default: corr_raw_dispatch<8>(Fp, Mp, g, raw, s); break;
    }
    // ATen's interleaved tail: last (ncols mod 32) columns of the (h, n^2, w, d) difference tensor
    const int64_t ncols = (int64_t)h * g.n * g.n * w * d, tail_from = (ncols / 32) * 32;
    const int ntail = (int)(ncols - tail_from);
    if (ntail > 0)
        hipLaunchKernelGGL(k_corr_tail, dim3(cdiv(ntail * g.n, 64)), dim3(64), 0, s, fix, mov, g, tail_from, ntail, raw);

    static const bool old_box = getenv("CVX_CORR_BOX_V1") != nullptr;
    int rc;
    if (!old_box && corr_box2_supported(h, w, d, g.px)) rc = launch_corr_box2(raw, (int)K, h, w, d, g.px, ssd, s);
    else {
        if (b.nthreads == 0) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: coarse rows of %d voxels are too long for the LDS box kernel", d);
        const size_t lds = sizeof(float) * 8 * (size_t)(b.Ty + 4) * b.px;
        static size_t granted = 0;
        ensure_dynamic_lds(&k_corr_box, lds, granted);
        hipLaunchKernelGGL(k_corr_box, dim3((unsigned)K, b.nytiles), dim3(b.nthreads), lds, s, raw, b, ssd);
        rc = check_last("correlate");
    }
    if (rc) return rc;
    if (argmin) return launch_argmin(ssd, nullptr, nullptr, 0.0f, false, (int)K, (size_t)h * w * d, keys, argmin, s);
    return CVX_OK;
}
