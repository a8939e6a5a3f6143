// corrbox.hip -- the two zero-padded 3^3 box filters of the SSD volume as a z-marching pipeline
// (reference: convex_adam_utils.py:84-86, two F.avg_pool3d(.,3,stride=1,padding=1) over the displacement planes).
//
// One workgroup = one displacement k x one y tile; it walks the h planes once.  Two groups of specialised wavefronts
// run as a skewed pipeline with one barrier per step: at step t the loader publishes raw plane t (fetched into
// registers during the previous step), the box-1 waves turn raw planes t-3..t-1 into box-1 plane t-2 and the box-2
// waves turn box-1 planes t-5..t-3 into output plane t-4.  LDS holds only the newest plane of each stage (double
// buffered); a thread keeps the 3 rows x 6 columns windows of the two older planes of its 4 output columns in
// registers, so each value is read from LDS 3 times instead of 27.  Both 27-tap sums run in ATen's raster order from
// registers: 108 adds + 4 exact FMA divisions per 4 outputs.
// Row layout: raw rows (pitch px, element x at index x+1) land in LDS at index + 4; box 1 evaluates columns
// 4q .. 4q+3 and stores them at index 4q+4, box 2 evaluates columns 4q-3 .. 4q: every window is one aligned
// ds_read_b128 + ds_read_b64, every stage write one aligned ds_write_b128.  Values outside the volume are exact zeros
// in every stage (each avg_pool3d zero-pads its own input).
#include <stdlib.h>

#include "cvx_common.h"

namespace cvx {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 4-byte aligned 16-byte access (global memory only)

struct CB2Ctx {
    const float* rk;          // raw volume of this displacement   [h][w][px]
    float* ok;                // output volume of this displacement [h][w][d]
    float *S0, *S1;
    int h, w, d, px, lpr, RS, slot;
    int y0, ty;               // first output row and number of output rows of this tile
    int nsteps;
};

struct CB2Loader {
    bool ldr;
    unsigned goff;            // gy*px + 4j
    float* lds0;
    float4 reg;               // plane t+1, fetched one step ahead of its publication
};

__device__ __forceinline__ void cb2_load_step(const CB2Ctx& c, CB2Loader& L, int t) {
    if (L.ldr && t <= c.h) {
        *reinterpret_cast<float4*>(L.lds0 + (t & 1) * c.slot) = L.reg;        // plane t (zeros for t = h)
        L.reg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t + 1 < c.h) L.reg = *reinterpret_cast<const float4*>(c.rk + (size_t)(t + 1) * c.w * c.px + L.goff);
    }
}

// ROLE 1: box 1 (raw -> S1), ROLE 2: box 2 (S1 -> global).  `id` = index of the thread inside its role.
// FAST (certified-fast arithmetic, certify.hip): a plane's 3 x 3 sum is evaluated separably (column sums over the three rows, then three neighbours) and the
// three planes of a window meet in two running values per column (A = plane n-1, B = planes n-2 + n-1); no divisions: box 1 hands 27 x its mean to
// box 2, the volume is 729 x the mean.  28 additions per 4 outputs instead of 108 + 4 divisions; additions of non-negative terms only.
template <int ROLE, bool FAST>
__device__ __forceinline__ void cb2_run(const CB2Ctx& c, CB2Loader& L, int id) {
    // rows of this role: box 1 needs rows y0-1 .. y0+ty clipped to the volume (outside it is zero padding), box 2 rows y0 ..
    const int glo = ROLE == 1 ? max(0, c.y0 - 1) : c.y0;
    const int ghi = ROLE == 1 ? min(c.w - 1, c.y0 + c.ty) : c.y0 + c.ty - 1;
    const int nrow = ghi - glo + 1;
    const bool active = id < nrow * c.lpr;
    const int idc = active ? id : nrow * c.lpr - 1;
    const int q = idc % c.lpr, gy = glo + idc / c.lpr;
    const int lr = gy - (c.y0 - 2);                                        // LDS row of the thread's output row
    const int widx = ROLE == 1 ? 4 * q + 4 : 4 * q;                       // window start index
    const float* src = (ROLE == 1 ? c.S0 : c.S1) + (lr - 1) * c.RS + widx;
    float* dst = c.S1 + lr * c.RS + 4 * q + 4;
    const int c0 = ROLE == 1 ? 4 * q : 4 * q - 3;
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ok[j] = active && c0 + j >= 0 && c0 + j < c.d;
    const bool allok = ok[0] && ok[1] && ok[2] && ok[3];
    float* orow = c.ok + (size_t)gy * c.d + c0;
    const size_t oplane = (size_t)c.w * c.d;
    const int RS = c.RS;
    // steps: ROLE 1 loads at t = 1, computes plane t-2 for t = 2 .. h+2 (plane h = zeros);
    //        ROLE 2 loads at t = 3, computes plane t-4 for t = 4 .. h+3
    constexpr int T0 = ROLE == 1 ? 1 : 3;
    const int tlast = ROLE == 1 ? c.h + 2 : c.h + 3;

    // Register state: the windows of the two newest planes (win[n % 2]) and, instead of the window of the oldest
    // plane, its finished raster-order prefix: the 27-tap sum of plane z starts with the 9 taps of plane z-1 added to
    // +0.0, which depends on plane z-1 alone and is evaluated when that plane arrives (pre[n % 2], two steps ahead).
    float win[2][3][6], pre[2][4];
    float fa[4] = {0.f, 0.f, 0.f, 0.f}, fb[4] = {0.f, 0.f, 0.f, 0.f};       // FAST: A, B
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) win[a][i][j] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[a][j] = 0.0f;                    // plane -1 is zero padding: prefix +0.0
    }
    // one step with newest plane n (n % 2 = PAR): FIRST = plane 0 arrives (nothing to emit yet)
    auto step = [&](auto par, auto first, int t) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first)::value;
        cb2_load_step(c, L, t);
        const float* sp = src + ((t - 1) & 1) * c.slot;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 a = lds_load4(sp + i * RS);
            const f32x2 b = lds_load2(sp + i * RS + 4);
            win[PAR][i][0] = a.x; win[PAR][i][1] = a.y; win[PAR][i][2] = a.z; win[PAR][i][3] = a.w;
            win[PAR][i][4] = b.x; win[PAR][i][5] = b.y;
        }
        float s[4];
        if (FAST) {
            float cs[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) cs[j] = (win[PAR][0][j] + win[PAR][1][j]) + win[PAR][2][j];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = (cs[j] + cs[j + 1]) + cs[j + 2];
                s[j] = fb[j] + p;
                fb[j] = fa[j] + p;
                fa[j] = p;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!FAST) s[j] = pre[PAR][j];                  // prefix of plane n-2
        if (!FIRST && !FAST) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)                               // planes n-1, n
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s[j] += win[(PAR + 1 + pl) % 2][i][j];
                        s[j] += win[(PAR + 1 + pl) % 2][i][j + 1];
                        s[j] += win[(PAR + 1 + pl) % 2][i][j + 2];
                    }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                    // prefix of plane n, used two steps later
            if (FAST) break;
            float p = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { p += win[PAR][i][j]; p += win[PAR][i][j + 1]; p += win[PAR][i][j + 2]; }
            pre[PAR][j] = p;
        }
        if (!FIRST) {
            if (ROLE == 1) {
                const bool planeok = t - 2 < c.h;
                f32x4 o;
                o.x = (planeok && ok[0]) ? (FAST ? s[0] : div_exact<27>(s[0])) : 0.0f;
                o.y = (planeok && ok[1]) ? (FAST ? s[1] : div_exact<27>(s[1])) : 0.0f;
                o.z = (planeok && ok[2]) ? (FAST ? s[2] : div_exact<27>(s[2])) : 0.0f;
                o.w = (planeok && ok[3]) ? (FAST ? s[3] : div_exact<27>(s[3])) : 0.0f;
                if (active) lds_store4(dst + (t & 1) * c.slot, o);
            } else {
                float* op = orow + (size_t)(t - 4) * oplane;
                if (allok) {                                 // one 16-byte store (rows of d floats are only 4-byte aligned)
                    f32x4u o = {FAST ? s[0] : div_exact<27>(s[0]), FAST ? s[1] : div_exact<27>(s[1]), FAST ? s[2] : div_exact<27>(s[2]), FAST ? s[3] : div_exact<27>(s[3])};
                    *reinterpret_cast<f32x4u*>(op) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ok[j]) op[j] = FAST ? s[j] : div_exact<27>(s[j]);
                }
            }
        }
        cvx_barrier();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int t = 0;
    for (; t < T0; ++t) {                                     // T0 is odd: t = 0 [, 1, 2]
        cb2_load_step(c, L, t);
        cvx_barrier();
    }
    step(P0{}, Yes{}, t); ++t;                                // plane 0 arrives (t = T0, T0 - 1 is even for both roles)
    for (; t + 1 <= tlast; t += 2) { step(P1{}, No{}, t); step(P0{}, No{}, t + 1); }
    if (t <= tlast) { step(P1{}, No{}, t); ++t; }
    for (; t < c.nsteps; ++t) {
        cb2_load_step(c, L, t);
        cvx_barrier();
    }
}

struct CB2Geom {
    int h, w, d, px, lpr, RS, Ty, nyt, rows, nw1, nw2, nthreads;
    size_t lds_bytes;
};

template <bool FAST>
__global__ __launch_bounds__(1024) void k_corr_box2(const float* __restrict__ raw, CB2Geom b, float* __restrict__ ssd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    CB2Ctx c;
    const int k = blockIdx.x;
    c.h = b.h; c.w = b.w; c.d = b.d; c.px = b.px; c.lpr = b.lpr; c.RS = b.RS;
    c.slot = b.rows * b.RS;
    c.y0 = blockIdx.y * b.Ty;
    c.ty = min(b.Ty, b.w - c.y0);
    c.nsteps = b.h + 4;
    c.rk = raw + (size_t)k * ((size_t)b.h * b.w * b.px);
    c.ok = ssd + (size_t)k * ((size_t)b.h * b.w * b.d);
    c.S0 = lds;
    c.S1 = lds + 2 * c.slot;
    for (int i = tid * 4; i < 4 * c.slot; i += b.nthreads * 4) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    // loader: raw rows y0-2 .. y0+ty+1 clipped to the volume, lpr 16-byte chunks each
    CB2Loader L;
    const int glo = max(0, c.y0 - 2), ghi = min(b.w - 1, c.y0 + c.ty + 1);
    L.ldr = tid < (ghi - glo + 1) * b.lpr;
    const int lrow = glo + tid / b.lpr, lj = tid % b.lpr;
    L.goff = (unsigned)(lrow * b.px + 4 * lj);
    L.lds0 = c.S0 + (lrow - (c.y0 - 2)) * b.RS + 4 * lj + 4;
    L.reg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.ldr && b.h > 0) L.reg = *reinterpret_cast<const float4*>(c.rk + L.goff);
    cvx_barrier();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < b.nw1) cb2_run<1, FAST>(c, L, tid);
    else cb2_run<2, FAST>(c, L, tid - 64 * b.nw1);
}

static CB2Geom cb2_geom(int h, int w, int d, int px) {
    CB2Geom b;
    b.h = h; b.w = w; b.d = d; b.px = px;
    b.lpr = px / 4;
    // row stride: >= px + 8 floats, and (RS/4) = lpr (mod 8) so that the 16-byte window reads of consecutive lanes stay
    // on distinct bank groups across a row boundary (lane -> (row, quad) is row-major with lpr quads per row)
    int rs4 = b.lpr + 2;
    while ((rs4 & 7) != (b.lpr & 7)) ++rs4;
    b.RS = 4 * rs4;
    b.nthreads = 0;
    // largest y tile whose two roles (and the loader rows) fit 1024 threads
    int Ty = w;
    for (;; --Ty) {
        if (Ty < 1) return b;
        const int nw1 = cdiv((Ty + 2) * b.lpr, 64), nw2 = cdiv(Ty * b.lpr, 64);
        if (64 * (nw1 + nw2) <= 1024 && (Ty + 4) * b.lpr <= 64 * (nw1 + nw2)) break;
    }
    b.nyt = cdiv(w, Ty);
    b.Ty = cdiv(w, b.nyt);
    b.nyt = cdiv(w, b.Ty);
    // rows of box 1 actually inside the volume: at most Ty + 2, exactly w for a single tile
    const int r1 = b.nyt == 1 ? w : b.Ty + 2;
    b.nw1 = cdiv(r1 * b.lpr, 64);
    b.nw2 = cdiv(b.Ty * b.lpr, 64);
    b.rows = b.Ty + 4;
    b.nthreads = 64 * (b.nw1 + b.nw2);
    const int loaders = (b.nyt == 1 ? w : b.Ty + 4) * b.lpr;
    if (b.nthreads < loaders) b.nthreads = cdiv(loaders, 64) * 64;       // (extra waves join role 2 as inactive lanes)
    b.lds_bytes = sizeof(float) * 4 * (size_t)b.rows * b.RS;
    if (b.nthreads > 1024 || b.lds_bytes > 160 * 1024) b.nthreads = 0;
    return b;
}

bool corr_box2_supported(int h, int w, int d, int px) { return cb2_geom(h, w, d, px).nthreads != 0; }

int launch_corr_box2(const float* raw, int K, int h, int w, int d, int px, float* ssd, hipStream_t s, bool fast) {
    const CB2Geom b = cb2_geom(h, w, d, px);
    if (b.nthreads == 0) return fail(CVX_ERR_UNSUPPORTED, "correlate: rows of %d voxels are too long for the LDS box kernel", d);
    static size_t granted = 0, granted_fast = 0;
    if (fast) {
        ensure_dynamic_lds(&k_corr_box2<true>, b.lds_bytes, granted_fast);
        hipLaunchKernelGGL(k_corr_box2<true>, dim3((unsigned)K, b.nyt), dim3(b.nthreads), b.lds_bytes, s, raw, b, ssd);
    } else {
        ensure_dynamic_lds(&k_corr_box2<false>, b.lds_bytes, granted);
        hipLaunchKernelGGL(k_corr_box2<false>, dim3((unsigned)K, b.nyt), dim3(b.nthreads), b.lds_bytes, s, raw, b, ssd);
    }
    return check_last("corr_box2");
}

}  // namespace cvx
