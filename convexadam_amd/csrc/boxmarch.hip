// boxmarch.hip -- three chained 3^3 box filters of the Adam control grid as ONE z-marching kernel
// (reference: the three F.avg_pool3d(.,3,stride=1,padding=1) of convex_adam_MIND.py:166 and their autograd adjoint).
//
// A workgroup owns one channel, 8 rows (y) x the full row length (x) x a chunk of planes (z) and marches along z.
// The three passes run CONCURRENTLY as a skewed pipeline on specialised wavefronts: at step t the loader stages input
// plane zlo+t, the pass-1 waves produce plane zlo+t-2 of stage 1, the pass-2 waves plane zlo+t-4 of stage 2 and the
// pass-3 waves the output plane zlo+t-6; one barrier per step.  LDS holds only the NEWEST plane of every stage
// (double buffered, 39 KB for 126-voxel rows): a thread keeps the 3 rows x 6 columns windows of the two previous
// planes of its 4 output columns in registers, so every stage value is read from LDS 3 times instead of 27 and the
// 27-tap raster-order sums run from registers (108 adds + 4 exact divisions per 4 outputs, ~135 instructions).
// Rows are stored with a per-stage shift (stage 0: column c at index c+7, stage 1: c+6, stage 2: c+5) so that
// every window read is one aligned ds_read_b128 + ds_read_b64 at the same index 4q+4 and every stage write one
// aligned ds_write_b128; pass k evaluates columns 4q-3+k .. 4q+k, the last pass lands on 16-byte aligned rows of
// the output (and of P, m, v for the fused Adam update).  Rows of up to 62 voxels: no halo along x (the row pads are the zero padding
// of avg_pool3d); longer rows are cut into x tiles of <= 56 columns plus one aligned quad of halo per side (two half-size
// workgroups per CU that run out of step); 3 rows of halo along y, 3 planes (+ pipeline fill) along z.
//   forward  (ATen avg_pool3d):           out = (raster sum of 27 taps) / 27            at every pass
//   backward (ATen avg_pool3d_backward):  out = raster sum of (tap / 27): taps are divided when they are staged
//                                          (IEEE division, the dividend may be -0.0), the last pass stores the sum
#include "cvx_common.h"

namespace cvx {

// QPR = quads (16-byte slots) per LDS row, YT = output rows per tile, CPT = output columns per thread (4, or 2: twice the wavefronts
// with half the per-step instruction stream each -- a step is bound by the serial issue of ONE wavefront)
// DPP (QPR = 16, CPT = 4 only): every stage keeps column c at index c; a thread reads ONE aligned 16-byte piece per window row and takes
// the two halo columns from its neighbour lanes (v_mov_b32_dpp row_shr / row_shl: a row of 16 lanes IS a tile row; the lanes at the row
// ends receive 0, which only reaches the tile's discarded halo columns or stands for the zero padding of the volume): 3 LDS reads
// instead of 6 per step, no misaligned loader stores, no bank conflicts.
template <int QPR, int YT, int CPT = 4, bool DPP = false>
struct BMGeomT {
    static constexpr int TPR = QPR * 4 / CPT;                              // threads per row
    static constexpr int RPW = 64 / TPR;                                   // rows per wavefront
    static_assert(CPT == 4 || CPT == 2, "4 or 2 columns per thread");
    static_assert(TPR <= 64 && 64 % TPR == 0, "a row must not straddle wavefronts");
    static constexpr int ROWS0 = YT + 6, ROWS1 = YT + 4, ROWS2 = YT + 2, ROWS3 = YT;
    static constexpr int NW1 = (ROWS1 + RPW - 1) / RPW, NW2 = (ROWS2 + RPW - 1) / RPW, NW3 = (ROWS3 + RPW - 1) / RPW;
    static constexpr int NT = 64 * (NW1 + NW2 + NW3);
    static constexpr int NWA = 2;                                          // wavefronts of the Adam role (variant AROLE of the adjoint kernel)
    static constexpr int NTA = NT + 64 * NWA;
    static constexpr int RS = DPP ? 4 * QPR : 4 * QPR + 8;                 // LDS row stride in floats
    static_assert(!DPP || (QPR == 16 && CPT == 4), "the DPP halo needs rows of exactly 16 lanes");
};

// Optional explicit work list (kernel argument, 2 KB): entry = z0 | zn << 12 | column << 20 for workgroup blockIdx.x, 0xffffffff =
// no work.  Used for UNEVEN z chunks: with two workgroups per CU the one dispatched second shares an already busy CU and
// advances ~25 % slower per step than the first one (tools/adam_census.py: 13.3 vs 16.9 us for equal chunks), so the
// workgroups of the second dispatch round get shorter chunks and both rounds finish together.
struct BMTable { int n; unsigned v[512]; };

// per-workgroup constants shared by the three roles
struct BMCtx {
    const float* ic;                       // input channel
    float *oc, *Pc, *mc, *vc, *gs;          // output channel / Adam state / saved gradient (may be null)
    float *S0, *S1, *S2, *S3;               // S3: plain adjoint sums of the newest output plane (Adam variant only)
    int h, w, d, z0, y0, zn, nsteps;
    int xl0;                                // global column of local column 0 (0, or x tile start - 4: an aligned halo of one quad)
    int ox0, ox1;                           // global columns this workgroup writes: [ox0, ox1)
    size_t wd;                              // plane stride w*d
    unsigned e_off[2];                      // Adam variant: offset (y0+row)*d + col inside a plane of the <= 2 elements this thread updates
    unsigned e_lds[2];                      //               and their index in an S3 slot; 0xffffffff = none
    unsigned long long* phases;             // (experiment build only) per-wave phase clocks
    int prio_par;                           // -1: no priority play; 0 / 1: this workgroup issues at raised priority on even / odd steps
    bool vec;
    AdamConsts ac;
};

// loader state of one thread: columns 4lq .. 4lq+3 of input row lgy, one plane per step, register staged
struct BMLoader {
    bool ldr, lrow;
    int lq;
    unsigned loff;                          // lgy*d + 4lq
    float* lds0;
    float4 reg;
};

// x / 27 correctly rounded for every float including -0.0 (IEEE: -0.0 / 27 = -0.0)
__device__ __forceinline__ float div27_signed(float x) { return x == 0.0f ? x : div_exact<27>(x); }

template <bool BACKWARD, bool VEC>
__device__ __forceinline__ void bm_issue(const BMCtx& c, BMLoader& L, int gz) {
    L.reg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.lrow && gz >= 0 && gz < c.h) {
        const float* rowp = c.ic + (size_t)gz * c.wd + L.loff;
        if (VEC) L.reg = *reinterpret_cast<const float4*>(rowp);
        else {
            L.reg.x = rowp[0];
            if (4 * L.lq + 1 < c.d) L.reg.y = rowp[1];
            if (4 * L.lq + 2 < c.d) L.reg.z = rowp[2];
            if (4 * L.lq + 3 < c.d) L.reg.w = rowp[3];
        }
    }
}

// loader part of step t: publish the plane fetched during the previous step, start fetching the next one
template <int SLOT0, bool BACKWARD, bool VEC, bool DPP>
__device__ __forceinline__ void bm_load_step(const BMCtx& c, BMLoader& L, int t) {
    if (DPP) {
        if (L.ldr && t <= c.zn + 5) {
            float mx = L.reg.x, my = L.reg.y, mz = L.reg.z, mw = L.reg.w;
            if (BACKWARD) { mx = div27_signed(mx); my = div27_signed(my); mz = div27_signed(mz); mw = div27_signed(mw); }
            lds_store4(L.lds0 + (t & 1) * SLOT0, f32x4{mx, my, mz, mw});      // aligned: index 4lq
            if (t + 1 <= c.zn + 5) bm_issue<BACKWARD, VEC>(c, L, c.z0 - 3 + t + 1);
        }
        return;
    }
    if (L.ldr && t <= c.zn + 5) {
        float* p = L.lds0 + (t & 1) * SLOT0;                         // indices 4lq+7 .. 4lq+10: b32 + b64 + b32
        // the 8-byte store needs an even-aligned register pair, the middle of a 16-byte load is an odd one: without the barrier below
        // the compiler carries the PAIR through the loop and copies into it right after the load is issued, i.e. it waits for the
        // prefetch in the step that requested it.  New values defined here keep the copies (and the wait) on this side of the step.
        float mx = L.reg.x, my = L.reg.y, mz = L.reg.z, mw = L.reg.w;
        asm volatile("" : "+v"(my), "+v"(mz));
        // backward: every tap is gradOut / 27 (divided here, when the plane is published -- not when it is requested, which would wait
        // for the load at once); the dividend may be -0.0, which div_exact maps to +0.0 -> sign fix
        if (BACKWARD) { mx = div27_signed(mx); my = div27_signed(my); mz = div27_signed(mz); mw = div27_signed(mw); }
        p[0] = mx;
        const f32x2 mid = {my, mz};
        lds_store2(p + 1, mid);
        p[3] = mw;
        if (t + 1 <= c.zn + 5) bm_issue<BACKWARD, VEC>(c, L, c.z0 - 3 + t + 1);
    }
}

// Adam variant, every thread: update the elements (e_row, e_col) of the plane that the pass-3 waves published in S3
// during the previous step (plane index z0 + t - 10).  Spreading the update over all wavefronts keeps the
// per-step work of the roles balanced; P, m, v of a plane are fetched one step ahead so that their latency is
// not on the critical path of the step barrier.
struct BMAdamPre { float p[2], m[2], v[2]; };

template <int QPR, int YT, int CPT, bool DPP>
__device__ __forceinline__ void bm_adam_step(const BMCtx& c, BMAdamPre& pre, int t) {
    using G = BMGeomT<QPR, YT, CPT, DPP>;
    constexpr int SLOT3 = G::ROWS3 * G::RS;
    if (t >= 10 && t <= c.zn + 9) {
        const size_t po = (size_t)(c.z0 + t - 10) * c.wd;             // uniform plane offset
        float *Pz = c.Pc + po, *mz = c.mc + po, *vz = c.vc + po;
        const float* sp = c.S3 + ((t - 1) & 1) * SLOT3;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (c.e_lds[k] != 0xffffffffu) {
                const float g = sp[c.e_lds[k]];
                float pp = pre.p[k], mm = pre.m[k], vv = pre.v[k];
                adam_update(g, pp, mm, vv, c.ac);
                Pz[c.e_off[k]] = pp; mz[c.e_off[k]] = mm; vz[c.e_off[k]] = vv;
                if (c.gs) (c.gs + po)[c.e_off[k]] = g;
            }
    }
    if (t + 1 >= 10 && t + 1 <= c.zn + 9) {
        const size_t po = (size_t)(c.z0 + t + 1 - 10) * c.wd;
        const float *Pz = c.Pc + po, *mz = c.mc + po, *vz = c.vc + po;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (c.e_lds[k] != 0xffffffffu) { pre.p[k] = Pz[c.e_off[k]]; pre.m[k] = mz[c.e_off[k]]; pre.v[k] = vz[c.e_off[k]]; }
    }
}

// AROLE variant: the Adam update as a FOURTH role (two extra wavefronts) instead of a share of every wavefront's step.  The per-phase
// clocks of the spread version (tools/box_phases.py) show ~1 000 of the ~2 700 clocks of an adjoint step inside the update (two IEEE
// divisions, a square root, the wait for P, m, v and four stores) on EVERY wavefront, while the pass-3 wavefronts idle 1 300 clocks at
// the barrier; as its own role the update runs beside the three passes and the step shrinks to the passes' own length.
// EPT elements per thread: the tile's YT x (ox1 - ox0) outputs over 128 threads.
template <int QPR, int YT, int CPT, bool DPP, int EPT>
__device__ __forceinline__ void bm_adam_role(const BMCtx& c, int atid) {
    using G = BMGeomT<QPR, YT, CPT, DPP>;
    constexpr int SLOT3 = G::ROWS3 * G::RS, NA = 64 * G::NWA;
    unsigned e_off[EPT], e_lds[EPT];
    const int ow = c.ox1 - c.ox0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = atid + k * NA;
        const int row = e / ow, col = e - row * ow;
        const bool have = row < YT && c.y0 + row < c.w;
        e_off[k] = (unsigned)((c.y0 + (have ? row : 0)) * c.d + c.ox0 + (have ? col : 0));
        e_lds[k] = have ? (unsigned)(row * G::RS + (c.ox0 - c.xl0) + col + (DPP ? 0 : 4)) : 0xffffffffu;
    }
    float pp[EPT], pm[EPT], pv[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) { pp[k] = 0.f; pm[k] = 0.f; pv[k] = 0.f; }
    for (int t = 0; t < c.nsteps; ++t) {
        if (t >= 10 && t <= c.zn + 9) {
            const size_t po = (size_t)(c.z0 + t - 10) * c.wd;             // uniform plane offset
            float *Pz = c.Pc + po, *mz = c.mc + po, *vz = c.vc + po;
            const float* sp = c.S3 + ((t - 1) & 1) * SLOT3;
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (e_lds[k] != 0xffffffffu) {
                    const float g = sp[e_lds[k]];
                    float p = pp[k], m = pm[k], v = pv[k];
                    adam_update(g, p, m, v, c.ac);
                    Pz[e_off[k]] = p; mz[e_off[k]] = m; vz[e_off[k]] = v;
                    if (c.gs) (c.gs + po)[e_off[k]] = g;
                }
        }
        if (t + 1 >= 10 && t + 1 <= c.zn + 9) {                          // P, m, v of the next plane: requested a step ahead
            const size_t po = (size_t)(c.z0 + t + 1 - 10) * c.wd;
            const float *Pz = c.Pc + po, *mz = c.mc + po, *vz = c.vc + po;
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (e_lds[k] != 0xffffffffu) { pp[k] = Pz[e_off[k]]; pm[k] = mz[e_off[k]]; pv[k] = vz[e_off[k]]; }
        }
        cvx_barrier();
    }
}

// The whole march of one role.  Every role executes exactly nsteps barriers.  Lanes beyond the role's last row
// compute on a clamped row and only their stores are masked, so that the window registers never pass through a
// divergent merge (no register copies).
template <int K, int QPR, int YT, int CPT, bool BACKWARD, bool ADAM, bool VEC, bool PK, bool DPP, bool AROLE>
__device__ __forceinline__ void bm_run(const BMCtx& c, BMLoader& L, int wk, int lane) {
    using G = BMGeomT<QPR, YT, CPT, DPP>;
    constexpr int SLOT0 = G::ROWS0 * G::RS, SLOT1 = G::ROWS1 * G::RS, SLOT2 = G::ROWS2 * G::RS;
    constexpr int ROWS = YT + 6 - 2 * K;
    constexpr int SRC_SLOT = K == 1 ? SLOT0 : (K == 2 ? SLOT1 : SLOT2), DST_SLOT = K == 1 ? SLOT1 : SLOT2;
    constexpr int WIN = CPT + 2;                                         // window columns per row
    const int r_raw = wk * G::RPW + lane / G::TPR, q = lane % G::TPR;
    const bool active = r_raw < ROWS;
    const int r = active ? r_raw : ROWS - 1;
    const int c0 = DPP ? CPT * q : CPT * q - 3 + K;                      // first output column (local)
    const int gy = c.y0 - 3 + K + r;
    const bool rowok = active && gy >= 0 && gy < c.w;
    bool ok[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) ok[j] = rowok && c.xl0 + c0 + j >= 0 && c.xl0 + c0 + j < c.d;
    // window of pass K = columns c0-1 .. c0+CPT of stage K-1 = indices CPT*q+4 .. (stage shifts 7, 6, 5); outputs at CPT*q+4 .. of stage K
    const float* src = (K == 1 ? c.S0 : (K == 2 ? c.S1 : c.S2)) + r * G::RS + CPT * q + (DPP ? 0 : 4);
    float* dst = (K == 1 ? c.S1 : c.S2) + r * G::RS + CPT * q + (DPP ? 0 : 4);
    const int gx = c.xl0 + CPT * q;                                      // pass 3: first global column of this thread
    const int ncol = gx >= c.ox0 ? c.ox1 - gx : 0;                       //         and how many of its columns this workgroup owns
    const unsigned rowbase = (unsigned)((gy < 0 ? 0 : gy) * c.d + (gx < 0 ? 0 : gx));
    const int tlast = c.zn + 5 + K;

    BMAdamPre apre = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    // Register state: two running sums per output column.  The 27-tap raster-order sum of output plane z is
    // ((0 + taps(z-1)) + taps(z)) + taps(z+1); when input plane n arrives the thread finishes plane n-1 from `mid` (= prefix of
    // planes n-2, n-1), advances `mid` from `pre` (= taps of plane n-1 added to +0.0) and restarts `pre`: every tap is read once
    // and only the newest plane's 3 x WIN window is live.
    // (mid, pre) travel as ONE register pair per column: the two running sums receive the same taps, so after the first tap of a
    // plane (which also restarts `pre` from +0.0) every tap costs one scalar add for the finishing sum and one v_pk_add_f32 with a
    // broadcast operand for the pair -- 19 instead of 27 issue slots per column and plane, the same additions in the same order
    // (packed fp32 adds run at half rate, so the VALU time is unchanged; what shrinks is the serial issue of one wavefront).
    f32x2 mp[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) mp[j] = f32x2{0.f, 0.f};
    // one step; n = t - (3K-2) counts the input planes of this role; EMIT: n >= 2, an output plane is due
#ifdef CVX_BM_PHASES        // experiment build (tools/box_phases.sh): where does a step's time go?  shader clocks per phase, summed over the emit steps
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define CVX_PH(i)                                                                  \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
        if (EMIT && tprev) ph[i] += now_ - tprev;                                  \
        tprev = now_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)
#else
#define CVX_PH(i) do {} while (0)
#endif
    auto step = [&](auto emit, int t) {
        constexpr bool EMIT = decltype(emit)::value;
        CVX_PH(4);                                     // (time since the previous stamp = the barrier wait)
        if (c.prio_par >= 0) {                         // (wave-uniform) the two workgroups of a CU take turns at the issue arbitration
            if ((t + c.prio_par) & 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        bm_load_step<SLOT0, BACKWARD, VEC, DPP>(c, L, t);
        const float* sp = src + ((t - 1) & 1) * SRC_SLOT;
        float win[3][WIN];
#pragma unroll
        for (int i = 0; i < 3; ++i) {                 // all three rows in flight: one LDS round trip per step
            if constexpr (DPP) {
                const f32x4 a = lds_load4(sp + i * G::RS);
                win[i][1] = a.x; win[i][2] = a.y; win[i][3] = a.z; win[i][4] = a.w;
                // column 4q-1 = the last value of the left neighbour, column 4q+4 = the first value of the right one; 0 at the row ends
                win[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.w), 0x111, 0xf, 0xf, true));       // row_shr:1
                win[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), 0x101, 0xf, 0xf, true));       // row_shl:1
            } else if (CPT == 4) {
                const f32x4 a = lds_load4(sp + i * G::RS);
                const f32x2 b = lds_load2(sp + i * G::RS + 4);
                win[i][0] = a.x; win[i][1] = a.y; win[i][2] = a.z; win[i][3] = a.w; win[i][WIN - 2] = b.x; win[i][WIN - 1] = b.y;
            } else {
                const f32x2 a = lds_load2(sp + i * G::RS);
                const f32x2 b = lds_load2(sp + i * G::RS + 2);
                win[i][0] = a.x; win[i][1] = a.y; win[i][WIN - 2] = b.x; win[i][WIN - 1] = b.y;
            }
        }
#ifdef CVX_BM_PHASES
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        CVX_PH(0);                                     // loader part + LDS window reads
        float f[CPT];
        if (PK) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                // first tap: f = mid + w, mid' = pre + w, pre' = +0.0 + w (a -0.0 tap must not survive, like ATen's sum)
                const float w0 = win[0][j];
                f[j] = mp[j].x + w0;
                f32x2 n;
                n.x = mp[j].y + w0;
                n.y = 0.0f + w0;
                f[j] += win[0][j + 1]; n += win[0][j + 1];
                f[j] += win[0][j + 2]; n += win[0][j + 2];
#pragma unroll
                for (int i = 1; i < 3; ++i) {
                    f[j] += win[i][j]; n += win[i][j];
                    f[j] += win[i][j + 1]; n += win[i][j + 1];
                    f[j] += win[i][j + 2]; n += win[i][j + 2];
                }
                mp[j] = n;
            }
        } else {                                      // three scalar adds per tap
            float m[CPT], p[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) { f[j] = mp[j].x; m[j] = mp[j].y; p[j] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float (&w)[WIN] = win[i];
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    f[j] += w[j]; m[j] += w[j]; p[j] += w[j];
                    f[j] += w[j + 1]; m[j] += w[j + 1]; p[j] += w[j + 1];
                    f[j] += w[j + 2]; m[j] += w[j + 2]; p[j] += w[j + 2];
                }
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) mp[j] = f32x2{m[j], p[j]};
        }
#ifdef CVX_BM_PHASES
#pragma unroll
        for (int j = 0; j < CPT; ++j) asm volatile("" ::"v"(f[j]), "v"(mp[j].x), "v"(mp[j].y));
#endif
        CVX_PH(1);                                     // the 27-tap sums
        const float (&s)[CPT] = f;
        if (EMIT) {
            const int gz = c.z0 - (2 * K + 3) + t;
            const bool planeok = gz >= 0 && gz < c.h;
            if (K < 3) {
                float o[CPT];
#pragma unroll
                for (int j = 0; j < CPT; ++j) o[j] = (planeok && ok[j]) ? div_exact<27>(s[j]) : 0.0f;
                if (active) {
                    if (CPT == 4) lds_store4(dst + (t & 1) * DST_SLOT, f32x4{o[0], o[1], o[CPT - 2], o[CPT - 1]});
                    else lds_store2(dst + (t & 1) * DST_SLOT, f32x2{o[0], o[1]});
                }
            } else if (ADAM) {
                // plain adjoint sums of this plane -> S3 (index = column + 4); consumed by bm_adam_step of the next step
                float* o3 = c.S3 + (t & 1) * (G::ROWS3 * G::RS) + r * G::RS + CPT * q + (DPP ? 0 : 4);
                if (active) {
                    if (CPT == 4) lds_store4(o3, f32x4{s[0], s[1], s[CPT - 2], s[CPT - 1]});
                    else lds_store2(o3, f32x2{s[0], s[1]});
                }
            } else if (planeok && rowok && ncol > 0) {
                float* oz = c.oc + (size_t)gz * c.wd;
                float g[CPT];
#pragma unroll
                for (int j = 0; j < CPT; ++j) g[j] = BACKWARD ? s[j] : div_exact<27>(s[j]);
                if (VEC) {
                    if (CPT == 4) *reinterpret_cast<float4*>(oz + rowbase) = make_float4(g[0], g[1], g[CPT - 2], g[CPT - 1]);
                    else *reinterpret_cast<float2*>(oz + rowbase) = make_float2(g[0], g[1]);
                } else {
#pragma unroll
                    for (int j = 0; j < CPT; ++j) if (j < ncol) oz[rowbase + j] = g[j];
                }
            }
        }
        CVX_PH(2);                                     // division + stage store (LDS or global)
        if (ADAM && !AROLE) bm_adam_step<QPR, YT, CPT, DPP>(c, apre, t);
        CVX_PH(3);                                     // Adam update of this thread's elements
        cvx_barrier();
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int t = 0;
    for (; t < 3 * K - 2; ++t) { bm_load_step<SLOT0, BACKWARD, VEC, DPP>(c, L, t); cvx_barrier(); }                    // (Adam starts at t = 10)
    step(No{}, t); ++t;                                 // t = 3K-2: input plane 0
    step(No{}, t); ++t;                                 // t = 3K-1: input plane 1
#pragma unroll 1
    for (; t <= tlast; ++t) step(Yes{}, t);             // t = 3K ..: output planes
    for (; t < c.nsteps; ++t) {
        bm_load_step<SLOT0, BACKWARD, VEC, DPP>(c, L, t);
        if (ADAM && !AROLE) bm_adam_step<QPR, YT, CPT, DPP>(c, apre, t);
        cvx_barrier();
    }
#ifdef CVX_BM_PHASES
    if (c.phases && lane == 0) {                       // slot = (workgroup, role K, wave of the role): 8 values
        unsigned long long* o = c.phases + ((size_t)blockIdx.x * 16 + (K - 1) * 4 + wk) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ph[i];
        o[5] = (unsigned long long)(tlast - 3 * K + 1);           // emit steps
    }
#endif
#undef CVX_PH
}

template <int QPR, int YT, int CPT, bool BACKWARD, bool ADAM, bool VEC, bool PK, bool DPP, bool AROLE>
__global__ __launch_bounds__((AROLE ? BMGeomT<QPR, YT, CPT, DPP>::NTA : BMGeomT<QPR, YT, CPT, DPP>::NT)) void k_box3_march(const float* __restrict__ in, float* __restrict__ out, int h,
                                                                int w, int d, int zc, int nzc, int nyt, float* __restrict__ P,
                                                                float* __restrict__ m, float* __restrict__ v, AdamConsts ac,
                                                                float* __restrict__ gsave, int vec_ok, int nxt, int tw,
                                                                unsigned long long* __restrict__ census, int prio_mode, BMTable tbl) {
    using G = BMGeomT<QPR, YT, CPT, DPP>;
    constexpr int NTK = AROLE ? G::NTA : G::NT;                       // threads of this launch
    static_assert(!AROLE || ADAM, "the Adam role exists in the adjoint + Adam kernel only");
    if (census && threadIdx.x == 0) {
        census[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        census[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
    constexpr int SLOT0 = G::ROWS0 * G::RS, SLOT1 = G::ROWS1 * G::RS, SLOT2 = G::ROWS2 * G::RS;
    __shared__ __attribute__((aligned(16))) float S0[2 * SLOT0];
    __shared__ __attribute__((aligned(16))) float S1[2 * SLOT1];
    __shared__ __attribute__((aligned(16))) float S2[2 * SLOT2];
    __shared__ __attribute__((aligned(16))) float S3[ADAM ? 2 * G::ROWS3 * G::RS : 4];
    int xi, yi, ch, z0w, znw;
    if (tbl.n > 0) {                                   // explicit work list (uneven z chunks)
        const unsigned e = tbl.v[blockIdx.x];
        if (e == 0xffffffffu) return;
        const int col = (int)(e >> 20);
        z0w = (int)(e & 0xfffu); znw = (int)((e >> 12) & 0xffu);
        xi = col % nxt; yi = (col / nxt) % nyt; ch = col / (nxt * nyt);
    } else {
        // XCD-aware order: XCD q (workgroups q, q+8, ..) takes the q-th contiguous run of (channel, y tile, x tile, z chunk) tuples
        const int nblk = 3 * nyt * nxt * nzc;
        const int b = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        if (b >= nblk) return;
        const int zi = b % nzc;
        xi = (b / nzc) % nxt; yi = (b / (nzc * nxt)) % nyt; ch = b / (nzc * nxt * nyt);
        z0w = zi * zc; znw = min(zc, h - z0w);
    }
    const size_t V = (size_t)h * w * d;
    BMCtx c;
    c.ic = in + (size_t)ch * V;
    c.oc = out ? out + (size_t)ch * V : nullptr;
    c.Pc = P ? P + (size_t)ch * V : nullptr;
    c.mc = m ? m + (size_t)ch * V : nullptr;
    c.vc = v ? v + (size_t)ch * V : nullptr;
    c.gs = gsave ? gsave + (size_t)ch * V : nullptr;
    c.S0 = S0; c.S1 = S1; c.S2 = S2; c.S3 = S3;
    c.wd = (size_t)w * d;
    c.h = h; c.w = w; c.d = d; c.z0 = z0w; c.y0 = yi * YT;
    // x tiles (nxt > 1): tw columns each plus one quad of halo on either side, so that every 16-byte access stays aligned; the three
    // passes lose one column per side each, the tile's own columns are local 4 .. 4 + tw - 1
    c.xl0 = nxt > 1 ? xi * tw - 4 : 0;
    c.ox0 = nxt > 1 ? xi * tw : 0;
    c.ox1 = nxt > 1 ? min(d, (xi + 1) * tw) : d;
    c.zn = znw;
    c.nsteps = c.zn + (ADAM ? 10 : 9);
    c.vec = vec_ok != 0;
    c.ac = ac;
    // (experiment build) behind the census slots of all three kernels: forward kernel first, adjoint kernel 1024 x 16 x 8 entries later
    c.phases = census ? (BACKWARD ? census - 4 * 1024 : census) + 4 * (8192 + 4096) + (BACKWARD ? 1024 * 16 * 8 : 0) : nullptr;
    // prio_mode 1: alternate by step, phase from the workgroup's slot id on its CU (HW_ID.TG_ID); 2: the same from the block index
    c.prio_par = prio_mode == 1 ? (int)((__builtin_amdgcn_s_getreg((4 << 0) | (16 << 6) | (3 << 11))) & 1u)
               : prio_mode == 2 ? (int)((blockIdx.x >> 8) & 1u) : -1;
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * SLOT0; i += NTK) S0[i] = 0.0f;
    for (int i = tid; i < 2 * SLOT1; i += NTK) S1[i] = 0.0f;
    for (int i = tid; i < 2 * SLOT2; i += NTK) S2[i] = 0.0f;
    if (ADAM) for (int i = tid; i < 2 * G::ROWS3 * G::RS; i += NTK) S3[i] = 0.0f;

    if (ADAM && !AROLE) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * G::NT;                           // 8 rows x (<= 126) columns <= 1008 elements <= 2 per thread
            const int ow = c.ox1 - c.ox0;
            const int row = e / ow, col = e - row * ow;
            const bool have = row < YT && c.y0 + row < w;
            c.e_off[k] = (unsigned)((c.y0 + row) * d + c.ox0 + col);
            c.e_lds[k] = have ? (unsigned)(row * G::RS + (c.ox0 - c.xl0) + col + (DPP ? 0 : 4)) : 0xffffffffu;
        }
    }
    BMLoader L;
    L.ldr = tid < G::ROWS0 * QPR;
    const int lr = tid / QPR;
    L.lq = tid % QPR;
    const int lgy = c.y0 - 3 + lr;
    const int lgx = c.xl0 + 4 * L.lq;
    L.lrow = L.ldr && lgy >= 0 && lgy < w && lgx >= 0 && lgx < d;
    L.loff = (unsigned)((lgy < 0 ? 0 : lgy) * d + (lgx < 0 ? 0 : lgx));
    L.lds0 = S0 + lr * G::RS + 4 * L.lq + (DPP ? 0 : 7);
    bm_issue<BACKWARD, VEC>(c, L, c.z0 - 3);
    cvx_barrier();
    if (census) {                                        // arrival of the first input plane
        const float4 probe = L.reg;
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(probe.x));
        if (threadIdx.x == 0) census[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }

    // role of this wavefront (wave-uniform, kept in a scalar register)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (wave < G::NW1) bm_run<1, QPR, YT, CPT, BACKWARD, ADAM, VEC, PK, DPP, AROLE>(c, L, wave, lane);
    else if (wave < G::NW1 + G::NW2) bm_run<2, QPR, YT, CPT, BACKWARD, ADAM, VEC, PK, DPP, AROLE>(c, L, wave - G::NW1, lane);
    else if (!AROLE || wave < G::NW1 + G::NW2 + G::NW3) bm_run<3, QPR, YT, CPT, BACKWARD, ADAM, VEC, PK, DPP, AROLE>(c, L, wave - G::NW1 - G::NW2, lane);
    else bm_adam_role<QPR, YT, CPT, DPP, (YT * (4 * QPR - 2) + 127) / 128>(c, tid - G::NT);          // rows of at most 4 * QPR - 2 outputs
    if (census && threadIdx.x == 0) census[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
}

bool box3_march_supported(int d) { return d <= 126; }

// Work list with uneven z chunks for a chip of 8 XCDs x 32 CUs (blockIdx.x % 8 = XCD, workgroups of an XCD are dealt to its CUs in
// index order): `ncol` columns (channel x y tile x x tile) of h planes, k chunks each; XCD q owns a contiguous run of columns; its
// first 32 workgroups (first dispatch round: alone on a CU, or the older of two) get the long chunks, the rest the short ones;
// ratio_pct = long : short in percent.  Returns false when the shape does not give 33..64 workgroups per XCD (uniform chunks then).
static bool bm_uneven_table(BMTable& T, unsigned& grid, int h, int ncol, long long ratio_pct) {
    if (ratio_pct <= 100 || ncol < 8 || ncol > 256) return false;
    const int k = 512 / ncol;
    if (k < 2 || k > 64) return false;
    int colq0[9];
    for (int q = 0; q <= 8; ++q) colq0[q] = (int)((long long)q * ncol / 8);
    int maxn = 0;
    for (int q = 0; q < 8; ++q) {
        const int nq = (colq0[q + 1] - colq0[q]) * k;
        if (nq <= 32 || nq > 64) return false;
        maxn = nq > maxn ? nq : maxn;
    }
    const double rho = (double)ratio_pct / 100.0;
    for (int i = 0; i < 512; ++i) T.v[i] = 0xffffffffu;
    for (int q = 0; q < 8; ++q) {
        const int cols = colq0[q + 1] - colq0[q];
        const int base = 32 / cols, rem = 32 % cols;                 // long chunks per column: 32 in this XCD
        int jl = 0, js = 32;                                         // next long / short slot of this XCD
        for (int ci = 0; ci < cols; ++ci) {
            const int nL = base + (ci < rem ? 1 : 0), nS = k - nL;
            if (nL > k) return false;
            const double S = (double)h / (rho * nL + nS), Lg = rho * S;
            if (S < 4.0 || Lg > 200.0) return false;
            // alternate long / short as far as the counts allow, starting with the more numerous kind; boundaries by rounding the running sum
            double acc = 0.0;
            int z0 = 0, l = nL, sh = nS;
            for (int c = 0; c < k; ++c) {
                const bool lng = (l > sh) || (l == sh && (c & 1) == 0) ? l > 0 : !(sh > 0);
                acc += lng ? Lg : S;
                if (lng) --l; else --sh;
                const int z1 = c == k - 1 ? h : (int)(acc + 0.5);
                const int zn = z1 - z0;
                if (zn < 1 || zn > 255) return false;
                const int j = lng ? jl++ : js++;
                T.v[j * 8 + q] = (unsigned)z0 | ((unsigned)zn << 12) | ((unsigned)(colq0[q] + ci) << 20);
                z0 = z1;
            }
        }
    }
    T.n = 8 * maxn;
    grid = (unsigned)T.n;
    return true;
}

template <int QPR, int YT, int CPT = 4, bool DPP = false>
static int launch_qpr(const float* in, float* out, int h, int w, int d, int nxt, int tw, bool backward, float* P, float* m, float* v,
                      AdamConsts ac, float* gsave, hipStream_t s) {
    using G = BMGeomT<QPR, YT, CPT, DPP>;
    const int nyt = cdiv(w, YT);
    long long wg_target = options().box_wg_target;                            // workgroups to aim for (z chunks follow from it);
    if (wg_target <= 0) wg_target = nxt > 1 ? 512 : 256;                      // 0 = automatic: two x-tile workgroups share a CU
    const int nz_target = wg_target / (3 * nyt * nxt) > 0 ? (int)(wg_target / (3 * nyt * nxt)) : 1;
    int zc = cdiv(h, nz_target);
    if (zc < 4) zc = 4;
    const int nzc = cdiv(h, zc);
    unsigned grid = (unsigned)((3 * nyt * nxt * nzc + 7) / 8 * 8);
    static thread_local BMTable tbl;                                          // (2 KB, passed by value)
    tbl.n = 0;
    // the uneven work list encodes a dispatch model of 8 XCDs x 32 CUs: used only on a device of that shape (results do not depend on it)
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); (void)hipGetLastError(); return n; }();
    if (cus == 256 && options().box_wg_target <= 0 && nxt > 1 && h <= 4095) (void)bm_uneven_table(tbl, grid, h, 3 * nyt * nxt, options().box_uneven);
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int vec = (d % 4 == 0) && al(in) && al(out) && al(P) && al(m) && al(v) && al(gsave);
    const bool arole = options().box_adam_role != 0 && G::NTA <= 1024;          // Adam update as a role of its own (two extra wavefronts)
    const bool pk = options().box_pk != 0;                                     // packed (mid, pre) pair: v_pk_add_f32 with a broadcast tap
    // debugging aid (option census_ptr): forward kernel -> slots [0, 4096), adjoint kernel -> [4096, 8192)
    unsigned long long* census = reinterpret_cast<unsigned long long*>(options().census_ptr);
    if (census && backward) census += 4 * 1024;
#define CVX_BM_LAUNCH(B, A)                                                                                                                        \
    do {                                                                                                                                           \
        if (vec && A && arole) hipLaunchKernelGGL((k_box3_march<QPR, YT, CPT, B, A, true, false, DPP, A>), dim3(grid), dim3(G::NTA), 0, s, in, out, h, w, d, zc, nzc, nyt, P, m, v, ac, gsave, vec, nxt, tw, census, (int)options().box_prio, tbl); \
        else if (vec && pk) hipLaunchKernelGGL((k_box3_march<QPR, YT, CPT, B, A, true, true, DPP, false>), dim3(grid), dim3(G::NT), 0, s, in, out, h, w, d, zc, nzc, nyt, P, m, v, ac, gsave, vec, nxt, tw, census, (int)options().box_prio, tbl); \
        else if (vec) hipLaunchKernelGGL((k_box3_march<QPR, YT, CPT, B, A, true, false, DPP, false>), dim3(grid), dim3(G::NT), 0, s, in, out, h, w, d, zc, nzc, nyt, P, m, v, ac, gsave, vec, nxt, tw, census, (int)options().box_prio, tbl); \
        else hipLaunchKernelGGL((k_box3_march<QPR, YT, CPT, B, A, false, false, false, false>), dim3(grid), dim3(G::NT), 0, s, in, out, h, w, d, zc, nzc, nyt, P, m, v, ac, gsave, vec, nxt, tw, census, (int)options().box_prio, tbl); \
    } while (0)
    if (!backward) CVX_BM_LAUNCH(false, false);
    else if (!P) CVX_BM_LAUNCH(true, false);
    else CVX_BM_LAUNCH(true, true);
#undef CVX_BM_LAUNCH
    return check_last("box3_march");
}

int launch_box3_march(const float* in, float* out, int h, int w, int d, bool backward, float* P, float* m, float* v,
                      AdamConsts ac, float* gsave, hipStream_t s) {
    // Long rows are cut into x tiles of <= 56 columns (+ 2 x 4 of halo = 16 quads): twice as many workgroups of half the size, two of
    // which share a CU and run out of step, so that the LDS phase of one overlaps the VALU phase of the other.
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (d % 4 == 0) && al(in) && al(out) && al(P) && al(m) && al(v) && al(gsave);
    const long long xs = options().box_xsplit;                                 // -1 automatic, 0 off, n >= 2: that many x tiles
    int nxt = 1, tw = d;
    if (vec && d > 62 && xs != 0) {
        nxt = xs > 1 ? (int)xs : cdiv(d, 56);
        tw = (cdiv(d, nxt) + 3) / 4 * 4;
        if (tw > 56 || tw < 8 || (nxt - 1) * tw >= d) { nxt = 1; tw = d; }      // (a requested split that would leave a tile empty is ignored)
    }
    const bool cpt2 = options().box_cpt == 2;                                    // two columns per thread (x tiles / short rows only)
    if (nxt > 1) {
        if (options().box_yt == 4) return launch_qpr<16, 4>(in, out, h, w, d, nxt, tw, backward, P, m, v, ac, gsave, s);
        if (cpt2) return launch_qpr<16, 8, 2>(in, out, h, w, d, nxt, tw, backward, P, m, v, ac, gsave, s);
        if (options().box_yt == 16) return launch_qpr<16, 16>(in, out, h, w, d, nxt, tw, backward, P, m, v, ac, gsave, s);
        if (options().box_dpp != 0) return launch_qpr<16, 8, 4, true>(in, out, h, w, d, nxt, tw, backward, P, m, v, ac, gsave, s);     // (nxt > 1 implies 16-byte aligned rows)
        return launch_qpr<16, 8>(in, out, h, w, d, nxt, tw, backward, P, m, v, ac, gsave, s);
    }
    if (options().box_yt == 4) {              // 4-row tiles: 9-wave workgroups, three per CU
        if (d <= 30) return launch_qpr<8, 4>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
        if (d <= 62) return launch_qpr<16, 4>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
        return launch_qpr<32, 4>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
    }
    if (d <= 30) return cpt2 ? launch_qpr<8, 8, 2>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s) : launch_qpr<8, 8>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
    if (d <= 62) return cpt2 ? launch_qpr<16, 8, 2>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s) : launch_qpr<16, 8>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
    return launch_qpr<32, 8>(in, out, h, w, d, 1, d, backward, P, m, v, ac, gsave, s);
}

}  // namespace cvx
