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
//   k_corr_box2 : (corrbox.hip) the two zero-padded box filters as a z-marching pipeline, raw -> ssd.
// Roofline: HBM by bytes (K*v*4 written + 2*C*v*4 read); the reference's summation order costs
// 2 x (26 adds + 1 division) + 36 flops per output, which makes the stage VALU-bound.

#include <algorithm>

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
// FAST (certified-fast arithmetic, certify.hip): one FMA per channel and output in plain channel order -- a chain of C roundings, no cascade
template <int HW, bool CASCADE, bool FAST = false>
__global__ __launch_bounds__(256) void k_corr_raw(const float* __restrict__ Fp, const float* __restrict__ Mp, CorrGeom g,
                                                  float* __restrict__ raw) {
    static_assert(!(CASCADE && FAST), "the fast arithmetic has no cascade");
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
                if (FAST) acc[k][j] = __builtin_fmaf(df, df, acc[k][j]);
                else acc[k][j] += df * df;
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

// the same tail values into a compact side buffer tail[iH][32] for the fused kernel (corrfused.hip)
__global__ void k_corr_tail_compact(const float* __restrict__ fix, const float* __restrict__ mov, CorrGeom g, int64_t tail_from,
                                    int ntail, int sad, float* __restrict__ tail) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntail * g.n) return;
    const int iH = t / ntail;
    const int64_t flat = tail_from + (t % ntail);
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
        sq[c] = sad ? fabsf(df) : df * df;
    }
    const int n4 = g.C / 4;
    float p[4];
    for (int k = 0; k < 4; ++k) p[k] = sum_cascade_strided(sq + k, 4, n4);       // (one cascade level below 64 channels)
    for (int i = n4 * 4; i < g.C; ++i) p[0] += sq[i];
    p[0] += p[1]; p[0] += p[2]; p[0] += p[3];
    tail[iH * 32 + (t % ntail)] = p[0];
}

void launch_corr_prep_generic(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int px, int PL, int dq, float* Fp,
                              float* Mp, hipStream_t s) {
    CorrGeom g = corr_geom(C, h, w, d, hw);
    g.px = px; g.PL = PL; g.dq = dq;
    const size_t nprep = std::max((size_t)C * g.hq * g.wq * g.dq, (size_t)C * h * w * g.px);
    hipLaunchKernelGGL(k_corr_prep, dim3((unsigned)cdiv64((int64_t)nprep, 256)), dim3(256), 0, s, fix, mov, g, Fp, Mp);
}

void launch_corr_tail_compact(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int sad, float* tail, hipStream_t s) {
    const CorrGeom g = corr_geom(C, h, w, d, hw);
    const int64_t ncols = (int64_t)h * g.n * g.n * w * d, tail_from = (ncols / 32) * 32;
    const int ntail = (int)(ncols - tail_from);
    if (ntail > 0) hipLaunchKernelGGL(k_corr_tail_compact, dim3(cdiv(ntail * g.n, 64)), dim3(64), 0, s, fix, mov, g, tail_from, ntail, sad, tail);
}

template <int HW>
static void corr_raw_dispatch(const float* Fp, const float* Mp, const CorrGeom& g, float* raw, hipStream_t s, bool fast = false) {
    const int nruns = g.h * g.w * (g.px / 4);
    const dim3 grid(cdiv(nruns, 256), g.n * g.n);
    if (fast) hipLaunchKernelGGL((k_corr_raw<HW, false, true>), grid, dim3(256), 0, s, Fp, Mp, g, raw);
    else if (g.C >= 16) hipLaunchKernelGGL((k_corr_raw<HW, true>), grid, dim3(256), 0, s, Fp, Mp, g, raw);
    else hipLaunchKernelGGL((k_corr_raw<HW, false>), grid, dim3(256), 0, s, Fp, Mp, g, raw);
}

// ---- the certified-fast volume: which kernel produces it (option corr_cert: 1 = the role kernel of corrfused.hip in its fast arithmetic
// without the final scaling -- the faster one as measured --, 2 = the staged kernel of corrcert.hip; 0 in the pipeline = exact volumes) -------
// ---- C >= 16 with few work items or many channels: the ROUND-1 PAIR OF KERNELS in the certified-fast arithmetic -- k_corr_raw with one FMA per
// channel (every wavefront on the channel sums, where the role kernel has a third of them) through the raw intermediate, then the two boxes as
// separable running sums without divisions (corrbox.hip, FAST).  Same class of arithmetic as the role kernel's (C + 13 roundings, non-negative
// terms only): the certification constants cover it.  Taken when the work is large enough to pay for the certified passes (K v C >= 1e9:
// BASELINE configs[3], 32 label channels on 40 x 48 x 40 with 729 displacements).
static bool certfast_unfused_ok(int C, int h, int w, int d, int hw) {
    if (C < 16 || C > 128 || hw > 8 || options().corr_cert == 2 || options().cert_unfused == 0) return false;
    const CorrGeom g = corr_geom(C, h, w, d, hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    return corr_box2_supported(h, w, d, g.px) && K * h * w * g.px * sizeof(float) <= ((size_t)2 << 30) && ((double)K * h * w * d * C >= 1e9 || options().cert_unfused == 2);
}
static bool certfast_use_unfused(int C, int h, int w, int d, int hw) {
    if (!certfast_unfused_ok(C, h, w, d, hw)) return false;
    const bool role_good = corr_fused_supported(C, h, w, d, hw) && C <= 32 && corr_fused_items(C, h, w, d, hw) >= 384 && !corr_fused_tiled(C, h, w, d, hw);
    return !role_good;
}
static size_t certfast_unfused_workspace(int C, int h, int w, int d, int hw) {
    const CorrGeom g = corr_geom(C, h, w, d, hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    size_t used = 0;
    used = carve_size(used, sizeof(float) * (size_t)C * h * w * g.px);
    used = carve_size(used, sizeof(float) * (size_t)C * g.hq * g.wq * g.dq);
    used = carve_size(used, sizeof(float) * K * h * w * g.px);
    return used + 256;
}
static int launch_corr_certfast_unfused(const float* fix, const float* mov, int C, int h, int w, int d, int hw, float* ssdu, void* workspace, size_t workspace_bytes,
                                        hipStream_t s) {
    if (workspace_bytes < certfast_unfused_workspace(C, h, w, d, hw)) return fail(CVX_ERR_WORKSPACE, "correlate (certified-fast, two kernels): workspace too small");
    const CorrGeom g = corr_geom(C, h, w, d, hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    Carver cv(workspace, workspace_bytes);
    float* Fp = cv.take<float>((size_t)C * h * w * g.px);
    float* Mp = cv.take<float>((size_t)C * g.hq * g.wq * g.dq);
    float* raw = cv.take<float>(K * h * w * g.px);
    const size_t nprep = (size_t)C * g.hq * g.wq * g.dq;
    hipLaunchKernelGGL(k_corr_prep, dim3((unsigned)cdiv64((int64_t)nprep, 256)), dim3(256), 0, s, fix, mov, g, Fp, Mp);
    corr_call_prep_hook(s);
    switch (hw) {
        case 0: corr_raw_dispatch<0>(Fp, Mp, g, raw, s, true); break;
        case 1: corr_raw_dispatch<1>(Fp, Mp, g, raw, s, true); break;
        case 2: corr_raw_dispatch<2>(Fp, Mp, g, raw, s, true); break;
        case 3: corr_raw_dispatch<3>(Fp, Mp, g, raw, s, true); break;
        case 4: corr_raw_dispatch<4>(Fp, Mp, g, raw, s, true); break;
        case 5: corr_raw_dispatch<5>(Fp, Mp, g, raw, s, true); break;
        case 6: corr_raw_dispatch<6>(Fp, Mp, g, raw, s, true); break;
        case 7: corr_raw_dispatch<7>(Fp, Mp, g, raw, s, true); break;
        default: corr_raw_dispatch<8>(Fp, Mp, g, raw, s, true); break;
    }
    return launch_corr_box2(raw, (int)K, h, w, d, g.px, ssdu, s, true);
}

static bool certfast_use_staged(int C, int h, int w, int d, int hw) {
    const bool staged_ok = corr_cert_supported(C, h, w, d, hw), fused_ok = corr_fused_supported(C, h, w, d, hw) && C <= 128;
    if (options().corr_cert == 2) return staged_ok;
    return staged_ok && !fused_ok;
}
bool corr_certfast_supported(int C, int h, int w, int d, int hw) {
    return corr_cert_supported(C, h, w, d, hw) || (corr_fused_supported(C, h, w, d, hw) && C <= 128) || certfast_unfused_ok(C, h, w, d, hw);
}
// does the certified path pay for this geometry in the whole-pair pipeline?  (below 16 channels always; from 16 on where a fast kernel beats the exact pair)
bool corr_certfast_pays(int C, int h, int w, int d, int hw) {
    if (C < 16) return true;
    if (certfast_use_unfused(C, h, w, d, hw)) return true;
    return C <= 32 && corr_fused_supported(C, h, w, d, hw) && corr_fused_items(C, h, w, d, hw) >= 384 && !corr_fused_tiled(C, h, w, d, hw);
}
size_t corr_certfast_workspace_bytes(int C, int h, int w, int d, int hw) {
    const size_t a = corr_cert_supported(C, h, w, d, hw) ? corr_cert_workspace_bytes(C, h, w, d, hw) : 0;
    const size_t b = corr_fused_supported(C, h, w, d, hw) ? corr_fused_workspace_bytes(C, h, w, d, hw) : 0;
    const size_t c = certfast_unfused_ok(C, h, w, d, hw) ? certfast_unfused_workspace(C, h, w, d, hw) : 0;
    const size_t m = a > b ? a : b;
    return align_up(m > c ? m : c, 256);
}
int launch_corr_certfast(const float* fix, const float* mov, int C, int h, int w, int d, int hw, float* ssdu, void* workspace, size_t workspace_bytes,
                         hipStream_t s) {
    if (certfast_use_unfused(C, h, w, d, hw)) return launch_corr_certfast_unfused(fix, mov, C, h, w, d, hw, ssdu, workspace, workspace_bytes, s);
    if (certfast_use_staged(C, h, w, d, hw)) return launch_corr_cert(fix, mov, C, h, w, d, hw, ssdu, workspace, workspace_bytes, s);
    return launch_corr_fused(fix, mov, C, h, w, d, hw, 0, 2, /*fast=*/2, 0, ssdu, workspace, workspace_bytes, s);
}

}  // namespace cvx

using namespace cvx;

// Which path the packaged operator takes.  The fused kernel covers every shape (C up to 255 through the cascade sum, tall planes through
// y tiles); for C >= 16 its raw stage -- one third of the wavefronts -- carries most of the work and the round-1 kernels (all wavefronts
// on the raw SSD, then the box pipeline) are faster when their raw intermediate is affordable (tools/time_corr.py: C = 32 at 26x32x37,
// hw 6: 0.40 vs 0.63 ms), so they stay the default there; option corr_fused_all = 1 selects the fused kernel for every C.
bool cvx::corr_use_unfused(int C, int h, int w, int d, int hw, bool variant) {
    if (!corr_fused_supported(C, h, w, d, hw)) return true;
    if (variant || options().corr_fused_all != 0 || C < 16 || hw > 8) return false;      // (the round-1 raw kernel is instantiated for hw <= 8)
    const CorrGeom g = corr_geom(C, h, w, d, hw);
    return corr_box2_supported(h, w, d, g.px) && (size_t)g.n * g.n * g.n * h * w * g.px * sizeof(float) <= ((size_t)2 << 30);
}

static size_t correlate_workspace_exact(int C, int h, int w, int d, int disp_hw);
extern "C" size_t cvx_correlate_workspace_bytes(int C, int h, int w, int d, int disp_hw) {
    // (the certified-fast variant, cvx_corr_opts.fast = 2, stages its own padded copies)
    const size_t exact = correlate_workspace_exact(C, h, w, d, disp_hw);
    const size_t cert = corr_certfast_supported(C, h, w, d, disp_hw) ? corr_certfast_workspace_bytes(C, h, w, d, disp_hw) + corr_certify_workspace_bytes(C, h, w, d, disp_hw, true) + 512 : 0;
    return exact > cert ? exact : cert;
}
static size_t correlate_workspace_exact(int C, int h, int w, int d, int disp_hw) {
    const size_t fused = corr_fused_supported(C, h, w, d, disp_hw)
                             ? carve_size(corr_fused_workspace_bytes(C, h, w, d, disp_hw), sizeof(unsigned long long) * (size_t)h * w * d) + 256 : 0;
    if (fused && !corr_use_unfused(C, h, w, d, disp_hw, false)) return fused;         // fused kernel: no raw intermediate
    const CorrGeom g = corr_geom(C, h, w, d, disp_hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    size_t used = 0;
    used = carve_size(used, sizeof(float) * (size_t)C * h * w * g.px);            // Fp
    used = carve_size(used, sizeof(float) * (size_t)C * g.hq * g.wq * g.dq);      // Mp
    used = carve_size(used, sizeof(float) * K * h * w * g.px);                    // raw
    used = carve_size(used, sizeof(unsigned long long) * (size_t)h * w * d);      // argmin keys
    return (used + 256 > fused ? used + 256 : fused);                             // (the variants of cvx_correlate_ex_f32 take the fused kernel)
}

extern "C" int cvx_correlate_f32(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, float* ssd,
                                 int64_t* argmin, void* workspace, size_t workspace_bytes, void* stream) {
    return cvx_correlate_ex_f32(fix, mov, C, h, w, d, disp_hw, nullptr, ssd, argmin, workspace, workspace_bytes, stream);
}

extern "C" int cvx_correlate_ex_f32(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, const cvx_corr_opts* opts,
                                    float* ssd, int64_t* argmin, void* workspace, size_t workspace_bytes, void* stream) {
    const int cost = opts ? opts->cost : 0, n_box = opts ? opts->n_box : 2, fast = opts ? opts->fast : 0, f16 = opts ? opts->f16 : 0;
    CVX_REQUIRE((cost == 0 || cost == 1) && (n_box == 1 || n_box == 2) && (fast == 0 || fast == 1 || fast == 2) && f16 >= 0 && f16 <= 2, "cvx_correlate_ex_f32: bad options");
    CVX_REQUIRE(!(f16 && (cost != 0 || n_box != 2)), "cvx_correlate_ex_f32: fp16 storage exists for the SSD cost with two boxes only");
    CVX_REQUIRE(!(fast && (cost != 0 || n_box != 2)), "cvx_correlate_ex_f32: the fast mode exists for the SSD cost with two boxes only");
    CVX_REQUIRE(fix && mov && ssd && workspace, "cvx_correlate_f32: null pointer");
    CVX_REQUIRE(C > 0 && C < 256 && h > 0 && w > 0 && d > 0, "cvx_correlate_f32: bad extent C=%d %dx%dx%d", C, h, w, d);
    CVX_REQUIRE(disp_hw >= 0, "cvx_correlate_f32: negative disp_hw");
    if (disp_hw > CVX_MAX_DISP_HW) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: disp_hw %d > %d not built", disp_hw, CVX_MAX_DISP_HW);
    if (workspace_bytes < cvx_correlate_workspace_bytes(C, h, w, d, disp_hw))
        return fail(CVX_ERR_WORKSPACE, "cvx_correlate_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const CorrGeom g = corr_geom(C, h, w, d, disp_hw);
    const size_t K = (size_t)g.n * g.n * g.n;
    if (fast == 2) {
        // certified-fast arithmetic: `ssd` receives the UNSCALED sums (729 x the mean, to within 2^-16 relative); `argmin` -- if asked for --
        // is the reference's argmin (first minimum of the EXACT volume), certified from the fast one and resolved exactly where it cannot be
        if (f16) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_ex_f32: the certified-fast volume is float32");
        if (!corr_certfast_supported(C, h, w, d, disp_hw)) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_ex_f32: certified-fast correlation not built for this geometry");
        const size_t cws = corr_certfast_workspace_bytes(C, h, w, d, disp_hw);
        int rc = launch_corr_certfast(fix, mov, C, h, w, d, disp_hw, ssd, workspace, cws, s);
        if (rc || !argmin) return rc;
        return corr_certified_argmin(ssd, fix, mov, C, h, w, d, disp_hw, argmin, static_cast<char*>(workspace) + align_up(cws, 256),
                                     corr_certify_workspace_bytes(C, h, w, d, disp_hw, true), s);
    }
    const bool variant = cost != 0 || n_box != 2 || fast || f16;
    if (!corr_use_unfused(C, h, w, d, disp_hw, variant)) {
        const size_t fws = corr_fused_workspace_bytes(C, h, w, d, disp_hw);
        int rc = launch_corr_fused(fix, mov, C, h, w, d, disp_hw, cost, n_box, fast, f16, ssd, workspace, fws, s);
        if (rc) return rc;
        if (argmin) {
            unsigned long long* keys = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + align_up(fws, 256));
            return launch_argmin(ssd, f16 == 2, nullptr, nullptr, 0.0f, false, (int)K, (size_t)h * w * d, keys, argmin, s);
        }
        return CVX_OK;
    }
    if (variant)
        return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_ex_f32: cost / n_box / fast / fp16 variants need the fused kernel (option corr_unfused is set, or the grid is outside its range: the fused kernel covers rows of d <= ~1270 voxels when the plane has w <= 320 / ceil((d + 6) / 4) rows, else (y tiles) d <= ~250)");
    if (disp_hw > 8) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: disp_hw %d > 8 needs the fused kernel (option corr_unfused is set, or the grid is outside its range: the fused kernel covers rows of d <= ~1270 voxels when the plane has w <= 320 / ceil((d + 6) / 4) rows, else (y tiles) d <= ~250)", disp_hw);
    if (!corr_box2_supported(h, w, d, g.px))
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

    int rc = launch_corr_box2(raw, (int)K, h, w, d, g.px, ssd, s);
    if (rc) return rc;
    if (argmin) return launch_argmin(ssd, false, nullptr, nullptr, 0.0f, false, (int)K, (size_t)h * w * d, keys, argmin, s);
    return CVX_OK;
}
