// correlate.hip -- dense SSD correlation volume (reference: convex_adam_utils.py:72-89).
//
//   raw[k,x] = sum_c (F_c(x) - M0_c(x + delta_k))^2          M0 = zero-padded moving features
//   ssd      = box3(box3(raw))                                 zero pad, raster-order 27-tap sums, /27
//   k        = (dD+hw)*n^2 + (dW+hw)*n + (dH+hw),  n = 2*hw+1
//
// Data flow (all float32, D fastest):
//   k_corr_prep : F -> Fp [C][h][w][dp]  (rows padded to a multiple of 4 floats)
//                 M -> Mp [C][h+2hw][w+2hw][dq] zero border of hw voxels (no bounds checks later)
//   k_corr_raw  : one thread = one run of 4 voxels along D x ALL n D-shifts of one (dH,dW) pair;
//                 per channel it loads 1 float4 of F and (2*PL+4)/4 float4 of the M row and updates
//                 4*n accumulators in registers in channel order (the reference's `.sum(0)` order,
//                 incl. ATen's 16-wide cascade for C >= 16).  Writes raw [K][h][w][dp].
//   k_corr_tail : re-evaluates the <= 31 trailing elements per H-shift whose channel sum ATen
//                 evaluates in its 4-way interleaved order (see oracle outer_sum_rows).
//   k_corr_box  : one workgroup per (k, z-slab): slab (+2 halo planes each side) staged in LDS with
//                 zero borders, box -> registers -> LDS in place -> box -> global ssd [K][h][w][d].
// Roofline: HBM; algorithmic bytes = K*v*4 written + 2*C*v*4 read; the raw
// intermediate adds 2*K*v*4 of traffic that stays largely in the 256 MiB Infinity Cache.
#include "cvx_common.h"

namespace cvx {

struct CorrGeom {
    int C, h, w, d, hw, n;
    int dp;     // padded F/raw row length (multiple of 4)
    int PL;     // left pad of Mp rows (multiple of 4, >= hw)
    int dq;     // Mp row length
    int hq, wq; // Mp plane extents (h+2hw, w+2hw)
};
static CorrGeom corr_geom(int C, int h, int w, int d, int hw) {
    CorrGeom g;
    g.C = C; g.h = h; g.w = w; g.d = d; g.hw = hw; g.n = 2 * hw + 1;
    g.dp = (d + 3) / 4 * 4;
    g.PL = (hw + 3) / 4 * 4;
    g.dq = g.dp + 2 * g.PL;
    g.hq = h + 2 * hw; g.wq = w + 2 * hw;
    return g;
}

__global__ __launch_bounds__(256) void k_corr_prep(const float* __restrict__ fix, const float* __restrict__ mov, CorrGeom g,
                                                   float* __restrict__ Fp, float* __restrict__ Mp) {
    const size_t nF = (size_t)g.C * g.h * g.w * g.dp, nM = (size_t)g.C * g.hq * g.wq * g.dq;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nF) {
        const int x = (int)(i % g.dp);
        const size_t r = i / g.dp;     // (c*h + z)*w + y
        Fp[i] = x < g.d ? fix[r * g.d + x] : 0.0f;
    }
    if (i < nM) {
        const int xq = (int)(i % g.dq), yq = (int)((i / g.dq) % g.wq), zq = (int)((i / ((size_t)g.dq * g.wq)) % g.hq);
        const int c = (int)(i / ((size_t)g.dq * g.wq * g.hq));
        const int x = xq - g.PL, y = yq - g.hw, z = zq - g.hw;
        const bool in = x >= 0 && x < g.d && y >= 0 && y < g.w && z >= 0 && z < g.h;
        Mp[i] = in ? mov[(((size_t)c * g.h + z) * g.w + y) * g.d + x] : 0.0f;
    }
}

// ---- raw SSD: register tile of 4 voxels x n D-shifts -----------------------------------------------
template <int HW, bool CASCADE>
__global__ __launch_bounds__(256) void k_corr_raw(const float* __restrict__ Fp, const float* __restrict__ Mp, CorrGeom g,
                                                  float* __restrict__ raw) {
    constexpr int N = 2 * HW + 1;
    constexpr int PL = (HW + 3) / 4 * 4;
    constexpr int NCH = (2 * PL + 4) / 4;          // float4 chunks of the M row segment
    const int runs_per_row = g.dp / 4;
    const int nruns = g.h * g.w * runs_per_row;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const int x0 = 4 * (r % runs_per_row), y = (r / runs_per_row) % g.w, z = r / (runs_per_row * g.w);
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
    const size_t fstride = (size_t)g.h * g.w * g.dp, mstride = (size_t)g.hq * g.wq * g.dq;
    const float* fp = Fp + ((size_t)z * g.w + y) * g.dp + x0;
    const float* mp = Mp + ((size_t)(z + iH) * g.wq + (y + iW)) * g.dq + x0;   // covers x0-PL .. x0+PL+3

#pragma unroll 1
    for (int c = 0; c < g.C; ++c) {
        const float4 f4 = *reinterpret_cast<const float4*>(fp + (size_t)c * fstride);
        const float f[4] = {f4.x, f4.y, f4.z, f4.w};
        float m[4 * NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(mp + (size_t)c * mstride + 4 * q);
            m[4 * q] = v.x; m[4 * q + 1] = v.y; m[4 * q + 2] = v.z; m[4 * q + 3] = v.w;
        }
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
    }
    const size_t v = (size_t)g.h * g.w * g.dp;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = CASCADE ? acc[k][j] + acc1[k][j] : acc[k][j];
        const size_t kk = ((size_t)k * N + iW) * N + iH;
        *reinterpret_cast<float4*>(raw + kk * v + ((size_t)z * g.w + y) * g.dp + x0) = make_float4(o[0], o[1], o[2], o[3]);
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
    raw[kk * ((size_t)g.h * g.w * g.dp) + ((size_t)z * g.w + y) * g.dp + x] = p[0];
}

// ---- two box filters per (k, z-slab) in LDS ------------------------------------------------------------
// A wavefront handles RPW rows x rp aligned runs of 4 columns (rp = dp/4 <= 64, RPW = 64/rp, the remaining lanes
// idle): one conflict-free ds_read_b128 per tap row per lane, the two neighbouring columns come from the
// adjacent lanes' registers (DPP wave shift); at the two ends of a row they are the zero border.
constexpr int BOX_NT = 512, BOX_NW = BOX_NT / 64, BOX_MAXG = 8;

struct BoxGeom {
    int h, w, d, dp, rp, rpw;
    int Tz, nslabs;
    int wy;     // w + 2
    int dx;     // dp + 8 : element x lives at index x + 4
};

__device__ __forceinline__ void box_run(const float* __restrict__ lds, const BoxGeom& b, int p, int y, int xr, float (&s)[4]) {
    // 27-tap raster order around (plane slot p, row y, columns 4*xr .. 4*xr+3); taps outside the volume are zeros
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = 0.0f;
    const bool first = xr == 0, last = xr == b.rp - 1;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int bb = -1; bb <= 1; ++bb) {
            const f32x4 q = lds_load4(lds + ((size_t)(p + a) * b.wy + (y + 1 + bb)) * b.dx + 4 * xr + 4);
            float lft = lane_prev(q.w), rgt = lane_next(q.x);
            lft = first ? 0.0f : lft;
            rgt = last ? 0.0f : rgt;
            s[0] += lft; s[0] += q.x; s[0] += q.y;
            s[1] += q.x; s[1] += q.y; s[1] += q.z;
            s[2] += q.y; s[2] += q.z; s[2] += q.w;
            s[3] += q.z; s[3] += q.w; s[3] += rgt;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = fdiv(s[j], 27.0f);
}

__global__ __launch_bounds__(BOX_NT) void k_corr_box(const float* __restrict__ raw, BoxGeom b, float* __restrict__ ssd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x, slab = blockIdx.y;
    const int z0 = slab * b.Tz;
    const int tz = min(b.Tz, b.h - z0);               // output planes of this slab
    const int nplanes = b.Tz + 4;                     // slots: plane z lives in slot z - (z0 - 2)
    const size_t plane_lds = (size_t)b.wy * b.dx;
    const size_t vraw = (size_t)b.h * b.w * b.dp;
    const float* rk = raw + (size_t)k * vraw;

    // 1. zero the whole buffer, then copy the in-volume planes (masking the padded columns x >= d)
    for (int i = tid * 4; i < (int)(nplanes * plane_lds); i += BOX_NT * 4)
        *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int zlo = max(z0 - 2, 0), zhi = min(z0 + tz + 2, b.h);     // [zlo, zhi)
    const int rp = b.rp;
    const int ncopy = (zhi - zlo) * b.w * rp;
    for (int i = tid; i < ncopy; i += BOX_NT) {
        const int xr = i % rp, y = (i / rp) % b.w, z = zlo + i / (rp * b.w);
        float4 v = *reinterpret_cast<const float4*>(rk + ((size_t)z * b.w + y) * b.dp + 4 * xr);
        const int x = 4 * xr;
        if (x + 3 >= b.d) {
            if (x + 0 >= b.d) v.x = 0.f;
            if (x + 1 >= b.d) v.y = 0.f;
            if (x + 2 >= b.d) v.z = 0.f;
            v.w = 0.f;
        }
        *reinterpret_cast<float4*>(lds + ((size_t)(z - (z0 - 2)) * b.wy + (y + 1)) * b.dx + x + 4) = v;
    }
    __syncthreads();

    const int xr = lane % rp, rsub = lane / rp;
    const bool lane_ok = rsub < b.rpw;
    // 2. first box on planes [z0-1, z0+tz+1) ∩ volume -> registers
    const int b1lo = max(z0 - 1, 0), b1hi = min(z0 + tz + 1, b.h);
    const int rows1 = (b1hi - b1lo) * b.w;
    float keep[BOX_MAXG][4];
#pragma unroll
    for (int i = 0; i < BOX_MAXG; ++i) {
        const int row = (wave + i * BOX_NW) * b.rpw + rsub;
        if ((wave + i * BOX_NW) * b.rpw < rows1) {                       // wave-uniform: all lanes take part in the shifts
            const int rr = (lane_ok && row < rows1) ? row : 0;
            box_run(lds, b, b1lo + rr / b.w - (z0 - 2), rr % b.w, xr, keep[i]);
        }
    }
    __syncthreads();
    // 3. in place: the second pool sees box1 only inside the volume (zeros elsewhere)
#pragma unroll
    for (int i = 0; i < BOX_MAXG; ++i) {
        const int row = (wave + i * BOX_NW) * b.rpw + rsub;
        if (lane_ok && row < rows1) {
            const int y = row % b.w, z = b1lo + row / b.w;
            const int x = 4 * xr;
            f32x4 v = {keep[i][0], keep[i][1], keep[i][2], keep[i][3]};
            if (x + 0 >= b.d) v.x = 0.f;
            if (x + 1 >= b.d) v.y = 0.f;
            if (x + 2 >= b.d) v.z = 0.f;
            if (x + 3 >= b.d) v.w = 0.f;
            lds_store4(lds + ((size_t)(z - (z0 - 2)) * b.wy + (y + 1)) * b.dx + x + 4, v);
        }
    }
    // slots of planes z0-2 and z0+tz+1 still hold raw values, but the second box only reads
    // [z0-1, z0+tz]: in-volume planes there now hold box1, out-of-volume slots are still zero
    __syncthreads();

    // 4. second box on planes [z0, z0+tz) -> global
    const int rows2 = tz * b.w;
    float* ok = ssd + (size_t)k * ((size_t)b.h * b.w * b.d);
#pragma unroll
    for (int i = 0; i < BOX_MAXG; ++i) {
        const int row = (wave + i * BOX_NW) * b.rpw + rsub;
        if ((wave + i * BOX_NW) * b.rpw < rows2) {
            const bool act = lane_ok && row < rows2;
            const int rr = act ? row : 0;
            const int y = rr % b.w, z = z0 + rr / b.w;
            float s[4];
            box_run(lds, b, z - (z0 - 2), y, xr, s);
            if (act) {
                float* dst = ok + ((size_t)z * b.w + y) * b.d + 4 * xr;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * xr + j < b.d) dst[j] = s[j];
            }
        }
    }
}

static BoxGeom box_geom(int h, int w, int d) {
    BoxGeom b;
    b.h = h; b.w = w; b.d = d; b.dp = (d + 3) / 4 * 4;
    b.rp = b.dp / 4;
    b.wy = w + 2; b.dx = b.dp + 8;
    b.Tz = 0; b.nslabs = 0;
    if (b.rp > 64) return b;                                           // rows longer than a wavefront: not built
    b.rpw = 64 / b.rp;
    const size_t plane_bytes = sizeof(float) * (size_t)b.wy * b.dx;
    int tz_lds = (int)((72 * 1024) / plane_bytes) - 4;                 // two workgroups per CU
    if (tz_lds < 1) tz_lds = (int)((156 * 1024) / plane_bytes) - 4;    // large planes: one workgroup per CU
    const int max_rows = BOX_NW * BOX_MAXG * b.rpw;                    // rows of the first box one workgroup can hold
    int tz_reg = max_rows / w - 2;
    int tzmax = tz_lds < tz_reg ? tz_lds : tz_reg;
    if (tzmax > h) tzmax = h;
    if (tzmax < 1) return b;                                           // plane too large for this kernel
    b.nslabs = cdiv(h, tzmax);
    b.Tz = cdiv(h, b.nslabs);
    b.nslabs = cdiv(h, b.Tz);
    return b;
}

template <int HW>
static void corr_raw_dispatch(const float* Fp, const float* Mp, const CorrGeom& g, float* raw, hipStream_t s) {
    const int nruns = g.h * g.w * (g.dp / 4);
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
    used = carve_size(used, sizeof(float) * (size_t)C * h * w * g.dp);            // Fp
    used = carve_size(used, sizeof(float) * (size_t)C * g.hq * g.wq * g.dq);      // Mp
    used = carve_size(used, sizeof(float) * K * h * w * g.dp);                    // raw
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
    const BoxGeom b = box_geom(h, w, d);
    if (b.nslabs == 0) return fail(CVX_ERR_UNSUPPORTED, "cvx_correlate_f32: coarse plane %dx%d too large for the LDS box kernel", w, d);
    Carver cv(workspace, workspace_bytes);
    float* Fp = cv.take<float>((size_t)C * h * w * g.dp);
    float* Mp = cv.take<float>((size_t)C * g.hq * g.wq * g.dq);
    float* raw = cv.take<float>(K * h * w * g.dp);
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

    const size_t lds = sizeof(float) * (size_t)(b.Tz + 4) * b.wy * b.dx;
    static size_t granted = 0;
    ensure_dynamic_lds(&k_corr_box, lds, granted);
    hipLaunchKernelGGL(k_corr_box, dim3((unsigned)K, b.nslabs), dim3(BOX_NT), lds, s, raw, b, ssd);
    int rc = check_last("correlate");
    if (rc) return rc;
    if (argmin) return launch_argmin(ssd, nullptr, nullptr, 0.0f, false, (int)K, (size_t)h * w * d, keys, argmin, s);
    return CVX_OK;
}
