// corrfused.hip -- the SSD correlation volume in ONE kernel: raw SSD -> box filter -> box filter -> ssd, nothing but the
// cost volume itself goes to HBM (reference: correlate, convex_adam_utils.py:72-89).
//
//   raw[k,x] = sum_c (F_c(x) - M0_c(x + delta_k))^2   (channel order of `.sum(0)`, M0 = zero-padded moving features)
//   ssd      = avg_pool3d(avg_pool3d(raw, 3, 1, 1), 3, 1, 1): two zero-padded 27-tap means in ATen's raster order
//
// Work item = one workgroup = a group of G <= 5 displacements that share (dH, dW) and are adjacent in dD; it marches along z
// over whole (w x d) planes.  The 2*hw+1 D-shifts are split into groups of 4 with a last group of 3..5 (13 -> 4+4+5), so a
// thread of the raw stage loads one 16-byte piece of the fixed row and two of the moving row per channel for 4 voxels x G
// shifts (20 outputs, 60 flops per 48 bytes from L1/L2; F and M together are 3 MB and never leave the caches), and the
// BASELINE workload gives 3 x 169 = 507 items: one round on the 2 x 256 workgroup slots of the chip.
//
// Three groups of specialised wavefronts run concurrently, one plane per step (step s: raw plane s, box-1 plane s-2, output
// plane s-4):
//   raw    thread = (row y, 4 columns): accumulates the G x 4 squared differences over the channels in registers
//   box 1  thread = (row y, 4 columns): newest raw plane -> box-1 plane
//   box 2  thread = (row y, 4 columns): newest box-1 plane -> 16-byte store of the cost volume
// A box thread reads ONLY the newest plane of its input (3 rows x 6 columns from LDS) and carries, per displacement and column,
// two running sums in registers: when plane m arrives it finishes output plane m-1 as (prefix(m-2) + taps(m-1)) + taps(m) --
// ATen's raster order: z slowest -- and starts the next two sums; 26 adds + 1 exact division per output, every tap read once.
// LDS holds rings of G+2 planes per stage; a step is cut into sub-intervals of two displacements (one barrier each): the box-1
// plane of displacement g overwrites the plane box 2 consumed two displacements -- one sub-interval -- earlier, so no stage
// needs a second copy of its planes and no thread parks finished results in registers (64 VGPRs, two workgroups = 30 waves
// per CU).  Rows are stored back to back (pitch 4*lpr, the zero column x = -1 of a row is the zero tail of the previous row):
// lane -> 16-byte slot is the identity, so every ds_read_b128 / ds_write_b128 is conflict-free.
// The <= 31 trailing elements whose channel sum ATen evaluates in its interleaved order come from k_corr_tail (correlate.hip)
// through a small side buffer.  C >= 16 (cascade sum) and planes of more than 320 quads keep the unfused path.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "cvx_common.h"

namespace cvx {

constexpr int CF_GMAX = 5;

struct CFGeom {
    int C, h, w, d, hw, n;
    int lpr, RS;            // quads per row; row pitch of LDS planes and of Fp (floats) = 4 * lpr
    int dq, hq, wq;         // Mp row pitch, plane extents (h + 2hw, w + 2hw); element x at index x + 1 + hw
    int ng, gs;             // D-shift groups per (dH, dW); shifts per group (the last group takes the rest, <= 5)
    int tiled, T, nyt;      // planes taller than one role can hold (5 wavefronts x 64 quads) are cut into y tiles of T output rows; the
                            // raw stage then evaluates rows y0-2 .. y0+T+1 and the first box y0-1 .. y0+T (halo rows are recomputed)
    int wpr;                // wavefronts per role
    int PF;                 // floats per LDS plane: 4 + (w + 2) * RS + 4
    int64_t tail_from;      // first flat index (h, n^2, w, d order) of ATen's interleaved-order tail; ntail = ncols - tail_from
    int ntail;
    int colocate;           // option cf_map = 1 (see k_corr_fused)
    int prio;               // issue priorities (option cf_prio): base-4 digits first-round raw / box, second-round raw / box
    unsigned long long* dbg;   // optional residency census (CVX_CF_CENSUS): per workgroup {start, end, HW_ID, XCC_ID}
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte global access at 4-byte alignment

__device__ __forceinline__ int cf_group_size(int n, int ng, int grp, int gs) { return grp < ng - 1 ? gs : n - gs * (ng - 1); }

// box stage of one displacement: `src` = window origin of the thread in the newest input plane, `rs` = its row pitch (a zero
// block with pitch 0 stands in for an all-zero plane); mid / pre = the two running sums per column.  Returns the four finished sums
// (not yet divided).  Rows 1 and 2 run as a rolled loop: one row of the window (6 values) is live at a time, which is what keeps
// 5 displacements x 8 running sums + a window inside 64 VGPRs; the LDS latency per row is covered by the other wavefronts.
__device__ __forceinline__ void cf_box_item(const float* src, int rs, float (&mid)[4], float (&pre)[4], float (&fin)[4]) {
    float f[4], m[4], p[4];
    {
        const f32x4 a = lds_load4(src);
        const f32x2 b = lds_load2(src + 4);
        src += rs;
        const float w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = mid[j] + w[j]; m[j] = pre[j] + w[j]; p[j] = w[j];             // 0 + first tap = first tap
            f[j] += w[j + 1]; m[j] += w[j + 1]; p[j] += w[j + 1];
            f[j] += w[j + 2]; m[j] += w[j + 2]; p[j] += w[j + 2];
        }
    }
#pragma unroll 1
    for (int i = 1; i < 3; ++i) {
        const f32x4 a = lds_load4(src);
        const f32x2 b = lds_load2(src + 4);
        src += rs;
        const float w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] += w[j]; m[j] += w[j]; p[j] += w[j];
            f[j] += w[j + 1]; m[j] += w[j + 1]; p[j] += w[j + 1];
            f[j] += w[j + 2]; m[j] += w[j + 2]; p[j] += w[j + 2];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { fin[j] = f[j]; mid[j] = m[j]; pre[j] = p[j]; }
}

// FAST variant of the box stage (opt-in, not bit-compatible): the 3 x 3 plane sum is evaluated separably (column sums over the three
// rows, then three neighbours) and the three planes are combined with two running values per column: A = p(m-1), B = p(m-2) + p(m-1).
// 28 adds per 4 outputs instead of 104; the divisions by 27 are folded into one multiplication at the end of the chain.
__device__ __forceinline__ void cf_box_item_fast(const float* src, int rs, float (&B)[4], float (&A)[4], float (&out)[4]) {
    float c[6];
    {
        const f32x4 a = lds_load4(src);
        const f32x2 b = lds_load2(src + 4);
        src += rs;
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y;
    }
#pragma unroll 1
    for (int i = 1; i < 3; ++i) {
        const f32x4 a = lds_load4(src);
        const f32x2 b = lds_load2(src + 4);
        src += rs;
        c[0] += a.x; c[1] += a.y; c[2] += a.z; c[3] += a.w; c[4] += b.x; c[5] += b.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float p = (c[j] + c[j + 1]) + c[j + 2];
        out[j] = B[j] + p;
        B[j] = A[j] + p;
        A[j] = p;
    }
}

// what every role needs to know about its work item
struct CFItem {
    int iH, iW, grp, y, q;  // y = row index of the thread inside its role's row range
    int y0;                 // first output row of the y tile (0 without tiling)
    bool active;
};

__device__ __forceinline__ float4 cf_ld16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uni_off) { return buffer_load16(rsrc, lane_off, uni_off); }

// ---- raw stage: G x 4 channel sums per thread and step ---------------------------------------------------------------------
// CT = compile-time channel count (12: fully unrolled software pipeline, every register static) or 0 (run-time count, rolled loop)
// CASC: ATen's cascade order of `.sum(0)` for C >= 16 (blocks of 16 channels are folded into a second accumulator, cvx_common.h cascade_seq)
template <int G, int CT, bool FAST, bool SAD, bool CASC, bool TILED>
__device__ __forceinline__ void cf_raw(const float* __restrict__ Fp, const float* __restrict__ Mp, const float* __restrict__ tail,
                                       const CFGeom& g, const CFItem& it, float* S0, int nsteps) {
    constexpr int R = G + 2, NSUB = (G + 1) / 2;
    const int n = g.n, nn = n * n, q = it.q, RS = g.RS, PF = g.PF;
    const int C = CT ? CT : g.C;
    static_assert(!CASC || CT == 0, "the cascade sum uses the rolled channel loop");
    // tiled: local row it.y stands for the global row y0 - 2 + it.y, rows outside the volume are written as zeros (the boxes zero-pad)
    const int y = TILED ? it.y0 - 2 + it.y : it.y;
    const bool rowok = !TILED || (y >= 0 && y < g.w);
    const int yc = !TILED ? y : (y < 0 ? 0 : (y >= g.w ? g.w - 1 : y));
    // per-thread byte offsets; everything else of an address is wave-uniform and travels in the scalar offset
    const unsigned foff = 4u * (unsigned)(yc * RS + 4 * q);
    const unsigned moff = 4u * (unsigned)((yc + it.iW) * g.dq + 4 * q + g.gs * it.grp);
    const unsigned doff = (unsigned)((TILED ? it.y : it.y + 1) * RS + 4 * q);
    const unsigned fstride = 4u * (unsigned)(g.h * g.w * RS), mstride = 4u * (unsigned)(g.hq * g.wq * g.dq);     // bytes per channel
    const unsigned fplane = 4u * (unsigned)(g.w * RS), mplane = 4u * (unsigned)(g.wq * g.dq);                    // bytes per plane
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fp), 0, (int)(fstride * (unsigned)g.C), 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Mp), 0, (int)(mstride * (unsigned)g.C + 32u), 0x00020000);
    // does this item hold elements of ATen's interleaved-order tail (the last < 32 elements of the (h, n^2, w, d) tensor)?
    const int ylast = TILED ? min(g.w - 1, it.y0 + g.T + 1) : g.w - 1;              // last row this workgroup evaluates
    const int64_t item_last = (((int64_t)(g.h - 1) * nn + it.iW * n + g.gs * it.grp + G - 1) * g.w + ylast) * g.d + g.d - 1;
    const bool tail_item = !FAST && g.ntail > 0 && item_last >= g.tail_from;
    int base = 0;                                          // (s * G) mod R
    for (int s = 0; s < nsteps; ++s) {
        const bool live = s < g.h;
        float acc[G][4];
        float acc1[CASC ? G : 1][4];
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[k][j] = 0.0f; if (CASC) acc1[k][j] = 0.0f; }
        const unsigned fz = (unsigned)s * fplane, mz = (unsigned)(s + it.iH) * mplane;     // uniform
        auto consume = [&](const float4& f4, const float4& m0, const float4& m1) {
            const float f[4] = {f4.x, f4.y, f4.z, f4.w};
            const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int k = 0; k < G; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float df = f[j] - m[j + k];
                    if (SAD) acc[k][j] += __builtin_fabsf(df);                       // .abs().sum(0)   l2r_2021 task 3 :54
                    else if (FAST) acc[k][j] = __builtin_fmaf(df, df, acc[k][j]);
                    else acc[k][j] += df * df;                                       // .pow(2).sum(0)
                }
        };
        // channels per sub-interval follow the box stages' load (two displacements, two, one)
        auto part_end = [&](int part) { return part == NSUB - 1 ? C : (C * (2 * (part + 1) < G ? 2 * (part + 1) : G) + (G >> 1)) / G; };
        if (CT) {
            float4 fa, ma0, ma1, fb, mb0, mb1;             // two register sets, channel c in set (c & 1)
            if (live) { fa = cf_ld16(fr, foff, fz); ma0 = cf_ld16(mr, moff, mz); ma1 = cf_ld16(mr, moff + 16, mz); }
#pragma unroll
            for (int part = 0; part < NSUB; ++part) {
                if (live) {
#pragma unroll
                    for (int c = part == 0 ? 0 : (CT * (2 * part < G ? 2 * part : G) + (G >> 1)) / G;
                         c < (part == NSUB - 1 ? CT : (CT * (2 * (part + 1) < G ? 2 * (part + 1) : G) + (G >> 1)) / G); ++c) {
                        if (c + 1 < CT) {
                            const unsigned fo = fz + (unsigned)(c + 1) * fstride, mo = mz + (unsigned)(c + 1) * mstride;
                            if (c & 1) { fa = cf_ld16(fr, foff, fo); ma0 = cf_ld16(mr, moff, mo); ma1 = cf_ld16(mr, moff + 16, mo); }
                            else { fb = cf_ld16(fr, foff, fo); mb0 = cf_ld16(mr, moff, mo); mb1 = cf_ld16(mr, moff + 16, mo); }
                        }
                        if (c & 1) consume(fb, mb0, mb1); else consume(fa, ma0, ma1);
                    }
                }
                if (part < NSUB - 1) cvx_barrier();
            }
        } else {
            // run-time channel count: rolled loop, software-pipelined by hand -- channel c + 1 is requested before channel c is consumed
            // (the loop is unrolled by two so that the two register sets alternate without copies)
            int c = 0;
            float4 fn, mn0, mn1;
            if (live) { fn = cf_ld16(fr, foff, fz); mn0 = cf_ld16(mr, moff, mz); mn1 = cf_ld16(mr, moff + 16, mz); }
#pragma unroll
            for (int part = 0; part < NSUB; ++part) {
                const int cend = part_end(part);
                if (live) {
#pragma unroll 2
                    for (; c < cend; ++c) {
                        const float4 f4 = fn, m0 = mn0, m1 = mn1;
                        if (c + 1 < C) {
                            const unsigned fo = fz + (unsigned)(c + 1) * fstride, mo = mz + (unsigned)(c + 1) * mstride;
                            fn = cf_ld16(fr, foff, fo); mn0 = cf_ld16(mr, moff, mo); mn1 = cf_ld16(mr, moff + 16, mo);
                        }
                        consume(f4, m0, m1);
                        if (CASC && (c & 15) == 15) {                      // a block of 16 channels is complete
#pragma unroll
                            for (int k = 0; k < G; ++k)
#pragma unroll
                                for (int j = 0; j < 4; ++j) { acc1[k][j] += acc[k][j]; acc[k][j] = 0.0f; }
                        }
                    }
                }
                if (part < NSUB - 1) cvx_barrier();
            }
        }
        if (CASC) {
#pragma unroll
            for (int k = 0; k < G; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k][j] += acc1[k][j];       // partial block + folded blocks (0 + a1 when C % 16 == 0)
        }
        if (live && it.active) {
            // all planes of the step go to the ring in the last sub-interval (their slots were consumed earlier in the step)
            // rare: replace the channel sums of the tail elements (the last < 32 elements of the (h, n^2, w, d) tensor; on tiny volumes
            // they span more than one plane)
            if (tail_item && rowok && (int64_t)(s + 1) * nn * g.w * g.d > g.tail_from) {
#pragma unroll 1
                for (int k = 0; k < G; ++k)
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {
                        const int x = 4 * q + j - 1;
                        const int64_t flat = (((int64_t)s * nn + it.iW * n + g.gs * it.grp + k) * g.w + y) * g.d + x;
                        if (x >= 0 && x < g.d && flat >= g.tail_from) {
                            const float t = tail[it.iH * 32 + (int)(flat - g.tail_from)];
#pragma unroll
                            for (int kk = 0; kk < G; ++kk)
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
                                    if (kk == k && jj == j) acc[kk][jj] = t;
                        }
                    }
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                int slot = base + k; slot = slot >= R ? slot - R : slot;
                float* o = S0 + doff + slot * PF;
                if (rowok) lds_store4(o, f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]});
                else lds_store4(o, f32x4{0.f, 0.f, 0.f, 0.f});   // (tiled) row outside the volume
                if (q == 0) o[0] = 0.0f;                         // x = -1
                if (4 * q + 3 > g.d)                             // x >= d: the boxes zero-pad
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j > g.d) o[j] = 0.0f;
            }
        }
        cvx_barrier();
        base += G; base = base >= R ? base - R : base; base = base >= R ? base - R : base;
    }
}

// ---- box stages: FIRST = reads the raw planes (else the box-1 planes), LAST = writes the cost volume (else the box-1 planes);
// FIRST && LAST is the single-box variant of the challenge scripts (l2r_2021_convexAdam_task2_docker.py:60) ------------------------
typedef _Float16 h16x4u __attribute__((ext_vector_type(4), aligned(2)));   // four half-precision values at 2-byte alignment

// OT = element type of the cost volume: float, or __half (fp16 STORAGE: half the bytes written here and read by the argmin passes)
template <int G, bool FIRST, bool LAST, bool FAST, bool ONEBOX, bool F16, typename OT, bool TILED, bool UNSCALED = false>
__device__ __forceinline__ void cf_box(const CFGeom& g, const CFItem& it, const float* Sin, float* S1, const float* zero, OT* __restrict__ ssd) {
    constexpr int R = G + 2, NSUB = (G + 1) / 2;
    const int n = g.n, nn = n * n, q = it.q, RS = g.RS, PF = g.PF;
    // local row it.y of this role: y = global output row, rin = first of the three input rows in a ring plane, rout = output row in a
    // box-1 plane.  Untiled planes carry a zero row above and below (ring row = y + 1); tiled planes hold exactly the rows of the tile.
    const int y = !TILED ? it.y : (FIRST && !LAST ? it.y0 - 1 + it.y : it.y0 + it.y);
    const int rin = TILED && FIRST && LAST ? it.y + 1 : it.y;
    const int rout = TILED ? it.y : it.y + 1;
    const bool rowok = !TILED || (y >= 0 && y < g.w);
    float mid[G][4], pre[G][4];
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { mid[k][j] = 0.0f; pre[k][j] = 0.0f; }
    // window origin in a ring plane: rows y-1 .. y+1 = ring rows y .. y+2; a FIRST stage reads raw columns 4q-1 .. 4q+4 (index x + 1)
    // and produces columns 4q .. 4q+3, the second stage reads box-1 columns 4q-4 .. 4q+1 (index x) and produces 4q-3 .. 4q
    const float* srcbase = Sin + rin * RS + 4 * q - (FIRST ? 0 : 4);
    const unsigned doff1 = (unsigned)(rout * RS + 4 * q);
    const int c0 = FIRST ? 4 * q : 4 * q - 3;                                          // first column of the four outputs
    const unsigned vol = (unsigned)(g.h * g.w * g.d), plane = (unsigned)(g.w * g.d);
    const unsigned ooff = (unsigned)sizeof(OT) * (unsigned)(y * g.d + 4 * q);          // bytes, relative to (plane base + c0 - 4q)
    OT* ssd_item = ssd + ((size_t)((g.gs * it.grp) * n + it.iW) * n + it.iH) * vol - (FIRST ? 0 : 3);   // uniform
    const size_t kstride = (size_t)nn * vol;                                           // next D-shift
    const bool full = c0 >= 0 && c0 + 3 < g.d;
    const int jlo = c0 < 0 ? -c0 : 0, jhi = min(4, g.d - c0);                          // valid columns of a partial quad
    const int lag = FIRST ? 1 : 3;                                                     // newest input plane of step s = s - lag
    const int nsteps = g.h + (ONEBOX ? 2 : 4);
    constexpr float SCALE = ONEBOX ? 1.0f / 27.0f : 1.0f / 729.0f;                     // fast mode: one multiplication for all divisions
    int base_prev = 0, base = 0;                               // ((s-1) * G) mod R, (s * G) mod R
    for (int s = 0; s < nsteps; ++s) {
        const int m = s - lag;
        const bool have = m >= 0 && m <= g.h;                 // m == h: the zero plane that closes the last output
        const bool readable = m >= 0 && m < g.h;
        const bool emit = m >= 1 && m <= g.h;
        if (!have) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) cvx_barrier();
        } else {
            const float* sb = readable ? srcbase : zero;
            const int rs = readable ? RS : 0, pf = readable ? PF : 0;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    constexpr int KMAX = G - 1;
                    const int k = 2 * sub + kk < KMAX ? 2 * sub + kk : KMAX;   // compile-time after unrolling
                    if (2 * sub + kk < G) {
                        int sp = base_prev + k; sp = sp >= R ? sp - R : sp;
                        float fin[4], o[4];
                        if (FAST) {
                            cf_box_item_fast(sb + sp * pf, rs, mid[k], pre[k], fin);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (LAST && !UNSCALED) ? fin[j] * SCALE : fin[j];     // UNSCALED: the certified pipeline's volume (no multiplication that could underflow)
                        } else {
                            cf_box_item(sb + sp * pf, rs, mid[k], pre[k], fin);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = div_exact<27>(fin[j]);
                        }
                        if (F16 && LAST) {                   // cost volume kept at half precision (values)
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = __half2float(__float2half_rn(o[j]));
                        }
                        if (emit && it.active && (rowok || !LAST)) {
                            if (!LAST) {
                                int sn = base + k; sn = sn >= R ? sn - R : sn;
                                float* ol = S1 + doff1 + sn * PF;
                                if (rowok) lds_store4(ol, f32x4{o[0], o[1], o[2], o[3]});
                                else lds_store4(ol, f32x4{0.f, 0.f, 0.f, 0.f});      // (tiled) the second box zero-pads outside the volume
                                if (4 * q + 3 >= g.d)
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if (4 * q + j >= g.d) ol[j] = 0.0f;
                            } else {
                                char* ob = reinterpret_cast<char*>(ssd_item + (size_t)k * kstride + (size_t)(m - 1) * plane);   // uniform
                                if (sizeof(OT) == 2) {
                                    if (full) *reinterpret_cast<h16x4u*>(ob + ooff) = h16x4u{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                                    else {
                                        _Float16* oe = reinterpret_cast<_Float16*>(ob + ooff);
                                        if (jlo <= 0 && jhi > 0) oe[0] = (_Float16)o[0];
                                        if (jlo <= 1 && jhi > 1) oe[1] = (_Float16)o[1];
                                        if (jlo <= 2 && jhi > 2) oe[2] = (_Float16)o[2];
                                        if (jlo <= 3 && jhi > 3) oe[3] = (_Float16)o[3];
                                    }
                                } else if (full) *reinterpret_cast<f32x4u*>(ob + ooff) = f32x4u{o[0], o[1], o[2], o[3]};   // (a non-temporal store changed nothing: 393 vs 385 MB of traffic)
                                else {
                                    float* oe = reinterpret_cast<float*>(ob + ooff);
                                    if (jlo <= 0 && jhi > 0) oe[0] = o[0];
                                    if (jlo <= 1 && jhi > 1) oe[1] = o[1];
                                    if (jlo <= 2 && jhi > 2) oe[2] = o[2];
                                    if (jlo <= 3 && jhi > 3) oe[3] = o[3];
                                }
                            }
                        }
                    }
                }
                cvx_barrier();
            }
        }
        base_prev = base;
        base += G; base = base >= R ? base - R : base; base = base >= R ? base - R : base;
    }
}

// MODE bits: 1 = FAST (FMA + separable sums, not bit-compatible), 2 = SAD cost, 4 = single box, 8 = cost volume rounded to fp16 (values in a
// float32 buffer), 16 = cost volume STORED as fp16 (the buffer holds __half: same values as 8, half the bytes), 32 = FAST without the
// final multiplication by 1 / 729 (the unscaled volume certify.hip takes its decisions on)
template <int G, int MODE, bool CASC, bool TILED>
__device__ __forceinline__ void cf_roles(int role, const float* Fp, const float* Mp, const float* tail, const CFGeom& g, const CFItem& it,
                                         float* lds, float* S0, float* S1, void* ssd_any) {
    constexpr bool FAST = (MODE & 1) != 0, SAD = (MODE & 2) != 0, ONEBOX = (MODE & 4) != 0, F16 = (MODE & 8) != 0, HALF = (MODE & 16) != 0, UNS = (MODE & 32) != 0;
    using OT = typename std::conditional<HALF, __half, float>::type;
    OT* ssd = static_cast<OT*>(ssd_any);
    if (role == 0) {
        if (CASC) cf_raw<G, 0, FAST, SAD, true, TILED>(Fp, Mp, tail, g, it, S0, ONEBOX ? g.h + 2 : g.h + 4);
        else if (g.C == 12) cf_raw<G, 12, FAST, SAD, false, TILED>(Fp, Mp, tail, g, it, S0, ONEBOX ? g.h + 2 : g.h + 4);
        else cf_raw<G, 0, FAST, SAD, false, TILED>(Fp, Mp, tail, g, it, S0, ONEBOX ? g.h + 2 : g.h + 4);
    } else if (ONEBOX) cf_box<G, true, true, FAST, true, F16, OT, TILED>(g, it, S0, S1, lds, ssd);
    else if (role == 1) cf_box<G, true, false, FAST, false, F16, OT, TILED>(g, it, S0, S1, lds, ssd);
    else cf_box<G, false, true, FAST, false, F16, OT, TILED, UNS>(g, it, S1, S1, lds, ssd);
}

// A launch may carry a SECOND problem of the same geometry (the reverse direction of a pair, pipeline.hip): blocks items1 .. 2 items1 - 1
// work on (Fp2, Mp2, tail2, ssd2).  One launch of 2 x 507 items instead of two of 507: a CU whose first-dispatched workgroup has finished
// (at ~76 % of a single launch) takes an item of the other direction instead of idling until the boundary.  Measured
// (option corr_dual, round 5): 0.360 ms for both directions against 0.374, but 540 MB of fresh cost volume no longer fit the 256 MB
// Infinity Cache and the plain argmin passes that follow pay 0.05 ms more than the correlation saved -- off by default.
struct CFSecond { const float* Fp; const float* Mp; const float* tail; void* ssd; int items1; };

template <int GMAX, int MODE, bool CASC, bool TILED>
__global__ __launch_bounds__(1024, (CASC ? 4 : 8)) void k_corr_fused(const float* __restrict__ Fp1, const float* __restrict__ Mp1,
                                                        const float* __restrict__ tail1, CFGeom g, void* __restrict__ ssd1, CFSecond two) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int n = g.n, nn = n * n;
    const bool second_problem = two.items1 > 0 && (int)blockIdx.x >= two.items1;           // (uniform)
    const float* __restrict__ Fp = second_problem ? two.Fp : Fp1;
    const float* __restrict__ Mp = second_problem ? two.Mp : Mp1;
    const float* __restrict__ tail = second_problem ? two.tail : tail1;
    void* __restrict__ ssd = second_problem ? two.ssd : ssd1;
    const int bx = second_problem ? (int)blockIdx.x - two.items1 : (int)blockIdx.x;
    CFItem it;
    // items: the large last groups first (they take longest), then the groups of four; y tiles innermost
    int pair;
    const int bi = TILED ? bx / g.nyt : bx;
    it.y0 = TILED ? (bx - bi * g.nyt) * g.T : 0;
    if (g.colocate) {
        // co-location experiment (option cf_map = 1; one round of <= 512 workgroups, three groups): workgroups b and b + 256 share a CU (census), so
        // slot s < nn holds (pair s, last group) then (pair s, the group next to it) -- the same moving rows, 16 bytes apart, through one L1 --
        // and the first groups of all pairs fill the remaining slots two by two
        const int rnd = bx >> 8, sl = bx & 255;
        if (sl < nn) { pair = sl; it.grp = rnd == 0 ? g.ng - 1 : g.ng - 2; }
        else { pair = (sl - nn) + rnd * (256 - nn); it.grp = 0; }
        if (pair >= nn) return;
    } else if (bi < nn) { it.grp = g.ng - 1; pair = bi; }
    else { const int b = bi - nn; it.grp = b / nn; pair = b - it.grp * nn; }
    it.iH = pair % n; it.iW = pair / n;
    const int G = cf_group_size(n, g.ng, it.grp, g.gs);
    if (g.dbg && tid == 0) {
        g.dbg[4 * blockIdx.x] = __builtin_amdgcn_s_memtime();
        g.dbg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        g.dbg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    }
    float* S0 = lds + 16 + 4;                                 // (16 zeros first) plane p of stage 0: S0 + p * PF, row r = y + 1, index i = x + 1
    float* S1 = S0 + (size_t)(GMAX + 2) * g.PF;               // stage 1: index = x
    const int nlds = 16 + ((MODE & 4) ? 1 : 2) * (GMAX + 2) * g.PF;
    for (int i = tid * 4; i < nlds; i += blockDim.x * 4) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    cvx_barrier();

    const int role = __builtin_amdgcn_readfirstlane(tid / (64 * g.wpr));
    const int tr = tid - role * 64 * g.wpr;
    // rows of this role: whole planes, or (tiled) T + 4 raw rows, T + 2 rows of the first box (T for a single box), T output rows
    const bool onebox = (MODE & 4) != 0;
    int rows = !TILED ? g.w : (role == 0 ? g.T + 4 : (role == 1 && !onebox ? g.T + 2 : g.T));
    if (TILED && role > 0 && (role == 2 || onebox)) rows = min(rows, g.w - it.y0);        // last tile: output rows inside the volume only
    it.active = tr < rows * g.lpr;
    {
        // issue priority of this wavefront: the raw stage is the longest instruction stream of every sub-interval and runs above the
        // boxes; with two workgroups on a CU (one launch round of 257..512 items) the one dispatched second may be given its own pair
        const bool second = two.items1 == 0 && gridDim.x > 256 && gridDim.x <= 512 && blockIdx.x >= 256;
        const int pr = (g.prio >> ((second ? 0 : 4) + (role == 0 ? 2 : 0))) & 3;
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
    const int trc = it.active ? tr : rows * g.lpr - 1;
    it.y = trc / g.lpr; it.q = trc - it.y * g.lpr;
    // the group size is a compile-time constant inside the roles (ring arithmetic, register arrays, no idle accumulators)
    switch (G) {
        case 5: cf_roles<5, MODE, CASC, TILED>(role, Fp, Mp, tail, g, it, lds, S0, S1, ssd); break;
        case 4: cf_roles<4, MODE, CASC, TILED>(role, Fp, Mp, tail, g, it, lds, S0, S1, ssd); break;
        case 3: cf_roles<3, MODE, CASC, TILED>(role, Fp, Mp, tail, g, it, lds, S0, S1, ssd); break;
        case 2: cf_roles<2, MODE, CASC, TILED>(role, Fp, Mp, tail, g, it, lds, S0, S1, ssd); break;
        default: cf_roles<1, MODE, CASC, TILED>(role, Fp, Mp, tail, g, it, lds, S0, S1, ssd); break;
    }
    if (g.dbg && tid == 0) g.dbg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static CFGeom cf_geom(int C, int h, int w, int d, int hw) {
    CFGeom g{};
    g.C = C; g.h = h; g.w = w; g.d = d; g.hw = hw; g.n = 2 * hw + 1;
    g.lpr = (d + 3 + 3) / 4;
    g.RS = 4 * g.lpr;
    const int n = g.n;
    g.gs = 4;
    g.ng = (n >= 4 && n % 4 <= 1) ? n / 4 : (n + 3) / 4;
    g.dq = g.RS + 4 * g.ng + 4;
    g.hq = h + 2 * hw; g.wq = w + 2 * hw;
    // a role has at most 5 wavefronts (3 roles = 15 of the 16 a workgroup may hold): planes of up to 320 quads go through whole, taller
    // ones in y tiles whose halo rows (2 + 2 for the raw stage, 1 + 1 for the first box) are recomputed by the neighbouring tile
    const int max_rows = 320 / g.lpr;
    if (w <= max_rows) { g.tiled = 0; g.T = w; g.nyt = 1; g.wpr = cdiv(w * g.lpr, 64); g.PF = (w + 2) * g.RS + 8; }
    else {
        g.tiled = 1;
        g.T = max_rows - 4 > 0 ? max_rows - 4 : 0;
        g.nyt = g.T > 0 ? cdiv(w, g.T) : 0;
        if (g.T > 0) g.T = cdiv(w, g.nyt);                                          // balanced tiles
        g.wpr = cdiv((g.T + 4) * g.lpr, 64);
        g.PF = (g.T + 4) * g.RS + 8;
    }
    const int64_t ncols = (int64_t)h * n * n * w * d;
    g.tail_from = (ncols / 32) * 32;
    g.ntail = (int)(ncols - g.tail_from);
    return g;
}
static size_t cf_lds_bytes(const CFGeom& g) { return sizeof(float) * (16 + 2 * (size_t)(CF_GMAX + 2) * g.PF); }

bool corr_fused_supported(int C, int h, int w, int d, int hw) {
    const bool off = options().corr_unfused != 0;
    if (off || C < 1 || C > 255 || hw < 0 || hw > CVX_MAX_DISP_HW || h < 1 || w < 1 || d < 1) return false;
    const CFGeom g = cf_geom(C, h, w, d, hw);
    if (g.T < 1 || g.wpr < 1 || 3 * g.wpr > 16 || cf_lds_bytes(g) > 160 * 1024) return false;
    // 32-bit byte offsets inside a feature copy and inside one displacement plane of the cost volume
    return (size_t)C * g.hq * g.wq * g.dq * 4 + 64 < ((size_t)1 << 31) && (size_t)h * w * d * 4 < ((size_t)1 << 31);
}

// work items of one launch (0: geometry not supported) -- the callers' measure of whether the kernel fills the 2 x 256 workgroup slots
int corr_fused_items(int C, int h, int w, int d, int hw) {
    if (!corr_fused_supported(C, h, w, d, hw)) return 0;
    const CFGeom g = cf_geom(C, h, w, d, hw);
    return g.n * g.n * g.ng * (g.nyt > 0 ? g.nyt : 1);
}

bool corr_fused_tiled(int C, int h, int w, int d, int hw) { return cf_geom(C, h, w, d, hw).tiled != 0; }

size_t corr_fused_workspace_bytes(int C, int h, int w, int d, int hw) {
    const CFGeom g = cf_geom(C, h, w, d, hw);
    size_t used = 0;
    used = carve_size(used, sizeof(float) * (size_t)C * h * w * g.RS);            // Fp
    used = carve_size(used, sizeof(float) * ((size_t)C * g.hq * g.wq * g.dq + 8));  // Mp
    used = carve_size(used, sizeof(float) * 32 * g.n);                             // tail values
    used = carve_size(used, 32 * ((size_t)g.n * g.n * g.ng * (g.nyt > 0 ? g.nyt : 1) + 8));  // residency census (CVX_CF_CENSUS; + 8: the 512 slots of option cf_map)
    return used + 256;
}

// prep / tail kernels of correlate.hip
void launch_corr_prep_generic(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int px, int PL, int dq, float* Fp,
                              float* Mp, hipStream_t s);
void launch_corr_tail_compact(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int sad, float* tail, hipStream_t s);

template <int MODE, bool CASC, bool TILED>
static void cf_launch_c(const CFGeom& gl, const float* Fp, const float* Mp, const float* tail, void* ssd, const CFSecond* second, hipStream_t s) {
    constexpr bool ONEBOX = (MODE & 4) != 0;
    const size_t lds = sizeof(float) * (16 + (ONEBOX ? 1 : 2) * (size_t)(CF_GMAX + 2) * gl.PF);
    static size_t granted = 0;
    ensure_dynamic_lds(&k_corr_fused<CF_GMAX, MODE, CASC, TILED>, lds, granted);
    const int items = gl.n * gl.n * gl.ng * gl.nyt;
    CFSecond two = {nullptr, nullptr, nullptr, nullptr, 0};
    if (second) { two = *second; two.items1 = items; }
    hipLaunchKernelGGL((k_corr_fused<CF_GMAX, MODE, CASC, TILED>), dim3(second ? 2 * items : (gl.colocate ? 512 : items)), dim3((ONEBOX ? 2 : 3) * 64 * gl.wpr), lds, s, Fp, Mp, tail, gl, ssd, two);
}
template <int MODE>
static void cf_launch(const CFGeom& gl, const float* Fp, const float* Mp, const float* tail, void* ssd, hipStream_t s, const CFSecond* second = nullptr) {
    // C >= 16: ATen's cascade channel sum; tiled: planes taller than one role holds (both are separate instantiations so that the
    // packaged configuration keeps its 64-register budget)
    // (the fast arithmetics sum the channels as ONE FMA chain -- no cascade, the 64-register instantiation and two workgroups per CU also for C >= 16:
    //  the certification bound of certify.hip counts C + 18 roundings for it, C <= 128)
    if (gl.C >= 16 && !(MODE & 1)) { if (gl.tiled) cf_launch_c<MODE, true, true>(gl, Fp, Mp, tail, ssd, second, s); else cf_launch_c<MODE, true, false>(gl, Fp, Mp, tail, ssd, second, s); }
    else if (gl.tiled) cf_launch_c<MODE, false, true>(gl, Fp, Mp, tail, ssd, second, s);
    else cf_launch_c<MODE, false, false>(gl, Fp, Mp, tail, ssd, second, s);
}

// opts: cost 0 = SSD / 1 = SAD, n_box 2 / 1, fast 0 / 1 (fast: SSD with two boxes only)
// f16: 0 float32 cost volume; 1 values rounded to half precision, float32 buffer; 2 the buffer holds __half (fp16 storage)
int launch_corr_fused(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box, int fast, int f16,
                      void* ssd, void* workspace, size_t workspace_bytes, hipStream_t s) {
    return launch_corr_fused_dual(fix, mov, C, h, w, d, hw, cost, n_box, fast, f16, ssd, nullptr, workspace, workspace_bytes, nullptr, s);
}

// profiling aid of the whole-pair pipeline: called on the stream between the feature copies (k_corr_prep, tail values) and the kernel, so that a
// stage interval can be the KERNEL's own duration (bench.py's `roofline`); nullptr (default) = nothing.  Per calling thread.
static thread_local void (*t_after_prep)(hipStream_t) = nullptr;
void corr_fused_set_prep_hook(void (*hook)(hipStream_t)) { t_after_prep = hook; }
void corr_call_prep_hook(hipStream_t s) { if (t_after_prep) t_after_prep(s); }

// ssd_rev != nullptr: BOTH directions of a pair in one launch -- ssd = correlate(fix, mov), ssd_rev = correlate(mov, fix); workspace_rev
// = a second workspace of corr_fused_workspace_bytes (the reverse direction's padded feature copies)
int launch_corr_fused_dual(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box, int fast, int f16,
                           void* ssd, void* ssd_rev, void* workspace, size_t workspace_bytes, void* workspace_rev, hipStream_t s) {
    const CFGeom g = cf_geom(C, h, w, d, hw);
    // every argument check comes before the first launch: a refused call leaves nothing on the stream
    if (workspace_bytes < corr_fused_workspace_bytes(C, h, w, d, hw)) return fail(CVX_ERR_WORKSPACE, "correlate (fused): workspace too small");
    if (fast && (cost != 0 || n_box != 2)) return fail(CVX_ERR_UNSUPPORTED, "correlate: the fast mode exists for the SSD cost with two boxes only");
    if (fast == 2 && f16) return fail(CVX_ERR_UNSUPPORTED, "correlate: the unscaled fast volume is float32");
    if (f16 && (cost != 0 || n_box != 2)) return fail(CVX_ERR_UNSUPPORTED, "correlate: fp16 storage exists for the SSD cost with two boxes only");
    if (f16 < 0 || f16 > 2) return fail(CVX_ERR_INVALID_ARG, "correlate: f16 must be 0, 1 or 2");
    Carver cv(workspace, workspace_bytes);
    float* Fp = cv.take<float>((size_t)C * h * w * g.RS);
    float* Mp = cv.take<float>((size_t)C * g.hq * g.wq * g.dq + 8);
    float* tail = cv.take<float>((size_t)32 * g.n);
    unsigned long long* census_buf = cv.take<unsigned long long>((size_t)4 * (g.n * g.n * g.ng * g.nyt + 8));
    launch_corr_prep_generic(fix, mov, C, h, w, d, hw, g.RS, hw, g.dq, Fp, Mp, s);
    if (g.ntail > 0 && !fast) launch_corr_tail_compact(fix, mov, C, h, w, d, hw, cost, tail, s);
    CFSecond two = {nullptr, nullptr, nullptr, nullptr, 0};
    if (ssd_rev) {
        if (!workspace_rev) return fail(CVX_ERR_WORKSPACE, "correlate (fused, both directions): second workspace missing");
        Carver cr(workspace_rev, workspace_bytes);
        float* Fp2 = cr.take<float>((size_t)C * h * w * g.RS);
        float* Mp2 = cr.take<float>((size_t)C * g.hq * g.wq * g.dq + 8);
        float* tail2 = cr.take<float>((size_t)32 * g.n);
        launch_corr_prep_generic(mov, fix, C, h, w, d, hw, g.RS, hw, g.dq, Fp2, Mp2, s);
        if (g.ntail > 0 && !fast) launch_corr_tail_compact(mov, fix, C, h, w, d, hw, cost, tail2, s);
        two.Fp = Fp2; two.Mp = Mp2; two.tail = tail2; two.ssd = ssd_rev;
    }
    const CFSecond* sec = ssd_rev ? &two : nullptr;
    if (t_after_prep) t_after_prep(s);
    CFGeom gl = g;
    gl.prio = (int)options().cf_prio;
    // (pairs of a slot: 2 x (256 - nn) >= nn first groups)
    gl.colocate = (options().cf_map == 1 && !ssd_rev && !g.tiled && g.ng == 3 && g.gs == 4 && g.n * g.n <= 256 && 2 * (256 - g.n * g.n) >= g.n * g.n) ? 1 : 0;
    gl.dbg = (options().cf_census && !ssd_rev) ? census_buf : nullptr;      // debugging aid: per-workgroup start / end / placement in the workspace
    if (fast == 2) cf_launch<1 + 32>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (fast && f16 == 2) cf_launch<1 + 8 + 16>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (fast && f16) cf_launch<9>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (fast) cf_launch<1>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (f16 == 2) cf_launch<8 + 16>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (f16) cf_launch<8>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (cost == 0 && n_box == 2) cf_launch<0>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (cost == 0) cf_launch<4>(gl, Fp, Mp, tail, ssd, s, sec);
    else if (n_box == 2) cf_launch<2>(gl, Fp, Mp, tail, ssd, s, sec);
    else cf_launch<6>(gl, Fp, Mp, tail, ssd, s, sec);
    return check_last("corr_fused");
}

}  // namespace cvx
