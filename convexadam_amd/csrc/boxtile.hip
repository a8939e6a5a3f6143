// boxtile.hip -- the three chained 3^3 FORWARD boxes of the Adam control grid in ATen's order, without a marching pipeline
// (reference: the three F.avg_pool3d(., 3, stride=1, padding=1) of convex_adam_MIND.py:166; same results, bit for bit, as
// k_box3_march<.., forward> of boxmarch.hip).
//
// Why a second kernel: the z-marching pipeline meets at one barrier per plane (21 dependent steps per workgroup, LDS write ->
// barrier -> LDS read -> 27 dependent adds -> division -> LDS write) and holds every CU at half its VALU rate for 18 us on the
// benchmark grid.  Here a workgroup owns one channel x one TZ x TY x 56 output tile and runs the three passes one
// after the other with TWO barriers in total:
//   pass 1  stage 0 (P, straight from global memory / L1: three 16-byte buffer loads per plane and thread, the two halo columns of a
//           quad from the neighbour lanes by DPP row shifts)                     -> stage 1 in LDS, (TZ+4) x (TY+4) x 60
//   pass 2  stage 1 (one aligned ds_read_b128 + ds_read_b64 per window row)      -> stage 2 in LDS, (TZ+2) x (TY+2) x 58
//   pass 3  stage 2                                                              -> U, TZ x TY x 56 (16-byte stores)
// Every stage is stored one column further left than its input (stage 1 at index x - x0 + 2, stage 2 at x - x0 + 1, the output at
// x - x0), so the 6-column window of an output quad always starts on a 16-byte boundary.
// A work item = 16 lanes = one row of quads x one SEGMENT of consecutive planes: the thread walks its segment along z and keeps, per
// output column, the two running raster sums of the marching kernels (`mid` = taps of planes n-2, n-1; `pre` = taps of plane n-1),
// so every tap is read once per item and an output costs exactly its 27 additions + one exact division whatever the segment
// length -- a segment's first and last plane contribute to one and two sums only, i.e. cutting z into segments costs window loads
// (L + 2 planes for L outputs), not additions.  Items are independent: no barrier inside a pass, the wavefronts of a CU overlap each
// other's LDS latency freely.
// Zero padding: every avg_pool3d pads its own input, so a stage value outside the VOLUME is stored as 0.
#include "cvx_common.h"

namespace cvx {

namespace {

constexpr unsigned BT_OOB = 0x80000000u;            // buffer offset beyond num_records: the load returns 0

// one plane of one item: 3 window rows x 6 columns
struct BTWin { float w[3][6]; };

// LIVE sums of a plane: F = finish the output of the previous plane (mid + taps), M = advance mid (pre + taps), P = restart pre
// (measured and dropped: (mid, pre) as one register pair fed by v_pk_add_f32 with a broadcast tap -- 18 instead of 27 issue slots per
// column and plane, the same additions: 17.2 us against 15.6-16.4 for the forward tiles under rocprofv3)
template <bool F, bool M, bool P>
__device__ __forceinline__ void bt_accum(const BTWin& t, float (&mid)[4], float (&pre)[4], float (&fin)[4]) {
    float f[4], m[4], p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = mid[j]; m[j] = pre[j]; p[j] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (F) { f[j] += t.w[i][j]; f[j] += t.w[i][j + 1]; f[j] += t.w[i][j + 2]; }
            if (M) { m[j] += t.w[i][j]; m[j] += t.w[i][j + 1]; m[j] += t.w[i][j + 2]; }
            if (P) { p[j] += t.w[i][j]; p[j] += t.w[i][j + 1]; p[j] += t.w[i][j + 2]; }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) { fin[j] = f[j]; mid[j] = m[j]; pre[j] = p[j]; }
}

// Walks one segment: planes n = 0 .. L+1 of the item's input (L >= 2 outputs); load(n, win) fetches plane n, emit(k, fin) receives the
// raster sums of output k = 0 .. L-1.  The window of plane n + 1 is requested before plane n is summed.
// prep(win) completes a fetched window right before it is summed (pass 1: the halo columns by DPP -- touching the registers of a
// global load any earlier would wait for it in the step that requested it).
template <typename Load, typename Prep, typename Emit>
__device__ __forceinline__ void bt_walk(int L, Load load, Prep prep, Emit emit) {
    float mid[4] = {0.f, 0.f, 0.f, 0.f}, pre[4] = {0.f, 0.f, 0.f, 0.f}, fin[4];
    BTWin A, B;                                            // two window buffers used in turn: no register copies between planes
    load(0, A);
    load(1, B);
    prep(A);
    bt_accum<false, false, true>(A, mid, pre, fin);
    load(2, A);
    prep(B);
    bt_accum<false, true, true>(B, mid, pre, fin);
    int n = 2;                                             // A holds plane n
#pragma unroll 1
    for (; n + 1 < L; n += 2) {
        load(n + 1, B);
        prep(A);
        bt_accum<true, true, true>(A, mid, pre, fin);
        emit(n - 2, fin);
        load(n + 2, A);
        prep(B);
        bt_accum<true, true, true>(B, mid, pre, fin);
        emit(n - 1, fin);
    }
    if (n < L) {                                           // one full plane left (n = L - 1, in A)
        load(L, B);
        prep(A);
        bt_accum<true, true, true>(A, mid, pre, fin);
        emit(L - 3, fin);
        load(L + 1, A);
        prep(B);
        bt_accum<true, true, false>(B, mid, pre, fin);
        emit(L - 2, fin);
        prep(A);
        bt_accum<true, false, false>(A, mid, pre, fin);
        emit(L - 1, fin);
    } else {                                               // A holds plane L
        load(L + 1, B);
        prep(A);
        bt_accum<true, true, false>(A, mid, pre, fin);
        emit(L - 2, fin);
        prep(B);
        bt_accum<true, false, false>(B, mid, pre, fin);
        emit(L - 1, fin);
    }
}

}  // namespace

// TZ x TY x (4 TXQ) output tile, NW wavefronts, NS1 / NS2 / NS3 z segments per row in passes 1 / 2 / 3 (run-time: they only cut the
// planes of a row into work items)
// BACKWARD: ATen's avg_pool3d_backward order of the three adjoint boxes -- every tap is gradOut / 27 (IEEE division, the dividend may be
// -0.0), the sums are plain: the global taps are divided when they are consumed, stages 1 and 2 are stored divided, the last pass keeps
// the sum; ADAM: the last pass applies torch.optim.Adam's update to P, m, v in place (gsave optionally receives G) instead of storing G.
template <int TZ, int TY, int TXQ, int NW, int WPS, bool BACKWARD, bool ADAM>
__global__ __launch_bounds__(64 * NW, WPS) void k_box3_tile(const float* __restrict__ in, float* __restrict__ out, int h, int w, int d,
                                                           int ntz, int nty, int ntx, int ntiles, int NS1, int NS2, int NS3, FastDiv dvz, FastDiv dvx, FastDiv dvy,
                                                           FastDiv dn1, FastDiv dn2, FastDiv dn3, float* __restrict__ P, float* __restrict__ m, float* __restrict__ v, AdamConsts ac,
                                                           float* __restrict__ gsave) {
    constexpr int TX = 4 * TXQ, RS = TX + 4;                          // LDS row stride (floats): stage 1 holds TX + 4 columns
    constexpr int Z1 = TZ + 4, Y1 = TY + 4, Z2 = TZ + 2, Y2 = TY + 2;
    constexpr int SLOTS = NW * 4;                                      // 16-lane items per round
    static_assert(TXQ + 2 <= 16, "a row of quads must fit 16 lanes");
    __shared__ __attribute__((aligned(16))) float S[Z1 * Y1 * RS + Z2 * Y2 * RS + 8];
    float* S1 = S;
    float* S2 = S + Z1 * Y1 * RS;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int bid = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int b = (bid & 7) * per_xcd + (bid >> 3);                             // XCD q takes the q-th contiguous run of tiles
    if (b >= ntiles) return;
    // (divisions by launch constants on the scalar unit, cvx_common.h FastDiv)
    const int b1 = fastdiv(b, dvz), b2 = fastdiv(b1, dvx), c = fastdiv(b2, dvy);
    const int tz = b - b1 * ntz, tx = b1 - b2 * ntx, ty = b2 - c * nty;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const size_t V = (size_t)h * w * d;
    const float* ic = in + (size_t)c * V;
    float* oc = out ? out + (size_t)c * V : nullptr;
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ic), 0, (int)(V * sizeof(float)), 0x00020000);
    const int tid = threadIdx.x, q = tid & 15, slot = tid >> 4;
    const int wd = w * d;

    // ---- pass 1: stage 0 (global) -> S1.  Lane q: columns x0 - 4 + 4q .. + 3 (local index i = 4q .. 4q + 3; kept: i = 2 .. TX + 5)
    for (int item = slot; item < Y1 * NS1; item += SLOTS) {
        const int r = item % Y1, sg = item / Y1;
        const int p0 = fastdiv(sg * Z1, dn1), L = fastdiv((sg + 1) * Z1, dn1) - p0;   // stage-1 planes p0 .. p0 + L - 1  <->  z = z0 - 2 + p
        const int gy = y0 - 2 + r, gx = x0 - 4 + 4 * q;
        const bool qin = gx >= 0 && gx < d && q < TXQ + 2;             // (d % 4 == 0: a quad is inside or outside as a whole)
        unsigned ro[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int yy = gy - 1 + i;
            ro[i] = (qin && yy >= 0 && yy < w) ? (unsigned)(yy * d + gx) * 4u : BT_OOB;
        }
        const int zin0 = z0 - 3 + p0;                                  // first input plane of the segment
        auto load = [&](int n, BTWin& t) {
            const int z = zin0 + n;
            const bool zok = z >= 0 && z < h;
            const unsigned zo = (unsigned)(zok ? z : 0) * (unsigned)wd * 4u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4 a = buffer_load16(ir, (zok && ro[i] != BT_OOB) ? ro[i] + zo : BT_OOB, 0);
                t.w[i][1] = a.x; t.w[i][2] = a.y; t.w[i][3] = a.z; t.w[i][4] = a.w;
            }
        };
        auto prep = [&](BTWin& t) {
            if (BACKWARD) {                       // taps of the adjoint: gradOut / 27, sign of zero kept
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 1; j <= 4; ++j) t.w[i][j] = t.w[i][j] == 0.0f ? t.w[i][j] : div_exact<27>(t.w[i][j]);
            }
            // column 4q - 1 = the last value of the left neighbour's quad, 4q + 4 = the first of the right one's; the row ends
            // receive 0: they only feed the discarded columns i = 0 and i = 63
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t.w[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t.w[i][4]), 0x111, 0xf, 0xf, true));       // row_shr:1
                t.w[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t.w[i][1]), 0x101, 0xf, 0xf, true));       // row_shl:1
            }
        };
        const bool rowin = qin && gy >= 0 && gy < w;
        float* dst = S1 + (p0 * Y1 + r) * RS + 4 * q - 2;              // stage-1 index j = i - 2
        auto emit = [&](int k, const float (&fin)[4]) {
            const int z = z0 - 2 + p0 + k;
            const bool ok = rowin && z >= 0 && z < h;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ok ? div_exact<27>(fin[j]) : 0.0f;
            float* p = dst + k * (Y1 * RS);
            if (q > 0 && q < TXQ + 2) lds_store2(p, f32x2{o[0], o[1]});
            if (q < TXQ + 1) lds_store2(p + 2, f32x2{o[2], o[3]});
        };
        bt_walk(L, load, prep, emit);
    }
    cvx_barrier();

    // ---- pass 2: S1 -> S2.  Lane q < TXQ + 1: stage-2 index j2 = 4q .. 4q + 3  <->  x = x0 - 1 + j2; window = S1 index 4q .. 4q + 5
    for (int item = slot; item < Y2 * NS2; item += SLOTS) {
        const int r = item % Y2, sg = item / Y2;
        const int p0 = fastdiv(sg * Z2, dn2), L = fastdiv((sg + 1) * Z2, dn2) - p0;   // stage-2 planes p0 ..  <->  z = z0 - 1 + p
        if (q < TXQ + 1) {
            const float* src = S1 + (p0 * Y1 + r) * RS + 4 * q;
            auto load = [&](int n, BTWin& t) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float* rp = src + (n * Y1 + i) * RS;
                    const f32x4 a = lds_load4(rp);
                    const f32x2 e = lds_load2(rp + 4);
                    t.w[i][0] = a.x; t.w[i][1] = a.y; t.w[i][2] = a.z; t.w[i][3] = a.w; t.w[i][4] = e.x; t.w[i][5] = e.y;
                }
            };
            const int gy = y0 - 1 + r, gx = x0 - 1 + 4 * q;
            const bool rowin = gy >= 0 && gy < w;
            bool xin[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xin[j] = rowin && gx + j >= 0 && gx + j < d;
            float* dst = S2 + (p0 * Y2 + r) * RS + 4 * q;
            auto emit = [&](int k, const float (&fin)[4]) {
                const int z = z0 - 1 + p0 + k;
                const bool zok = z >= 0 && z < h;
                f32x4 o;
                o.x = (zok && xin[0]) ? div_exact<27>(fin[0]) : 0.0f;
                o.y = (zok && xin[1]) ? div_exact<27>(fin[1]) : 0.0f;
                o.z = (zok && xin[2]) ? div_exact<27>(fin[2]) : 0.0f;
                o.w = (zok && xin[3]) ? div_exact<27>(fin[3]) : 0.0f;
                lds_store4(dst + k * (Y2 * RS), o);
            };
            bt_walk(L, load, [](BTWin&) {}, emit);
        }
    }
    cvx_barrier();

    // ---- pass 3: S2 -> U.  Lane q < TXQ: columns x0 + 4q .. + 3; window = S2 index 4q .. 4q + 5
    for (int item = slot; item < TY * NS3; item += SLOTS) {
        const int r = item % TY, sg = item / TY;
        const int p0 = fastdiv(sg * TZ, dn3), L = fastdiv((sg + 1) * TZ, dn3) - p0;
        const int gy = y0 + r, gx = x0 + 4 * q;
        if (q < TXQ && gy < w && gx < d) {
            const float* src = S2 + (p0 * Y2 + r) * RS + 4 * q;
            auto load = [&](int n, BTWin& t) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float* rp = src + (n * Y2 + i) * RS;
                    const f32x4 a = lds_load4(rp);
                    const f32x2 e = lds_load2(rp + 4);
                    t.w[i][0] = a.x; t.w[i][1] = a.y; t.w[i][2] = a.z; t.w[i][3] = a.w; t.w[i][4] = e.x; t.w[i][5] = e.y;
                }
            };
            const size_t o0 = ((size_t)(z0 + p0) * w + gy) * d + gx;
            auto emit = [&](int k, const float (&fin)[4]) {
                if (z0 + p0 + k >= h) return;
                const size_t o = o0 + (size_t)k * wd;
                if (!BACKWARD) {
                    *reinterpret_cast<float4*>(oc + o) = make_float4(div_exact<27>(fin[0]), div_exact<27>(fin[1]), div_exact<27>(fin[2]), div_exact<27>(fin[3]));
                } else if (!ADAM) {
                    *reinterpret_cast<float4*>(oc + o) = make_float4(fin[0], fin[1], fin[2], fin[3]);
                } else {
                    float* Pc = P + (size_t)c * V + o; float* mc = m + (size_t)c * V + o; float* vc = v + (size_t)c * V + o;
                    const float4 p4 = *reinterpret_cast<const float4*>(Pc), m4 = *reinterpret_cast<const float4*>(mc), v4 = *reinterpret_cast<const float4*>(vc);
                    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) adam_update(fin[j], pp[j], mm[j], vv[j], ac);
                    *reinterpret_cast<float4*>(Pc) = make_float4(pp[0], pp[1], pp[2], pp[3]);
                    *reinterpret_cast<float4*>(mc) = make_float4(mm[0], mm[1], mm[2], mm[3]);
                    *reinterpret_cast<float4*>(vc) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    if (gsave) *reinterpret_cast<float4*>(gsave + (size_t)c * V + o) = make_float4(fin[0], fin[1], fin[2], fin[3]);
                }
            };
            bt_walk(L, load, [](BTWin&) {}, emit);
        }
    }
}

// rows of whole 16-byte quads, 16-byte aligned volumes; 3 channels
bool box3_tile_supported(const float* in, const float* out, int h, int w, int d, const float* P, const float* m, const float* v, const float* gsave) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (d % 4 == 0) && al(in) && al(out) && al(P) && al(m) && al(v) && al(gsave) && in != out && (size_t)h * w * d * 4 < ((size_t)1 << 31);
}
bool box3_tile_fwd_supported(const float* in, const float* out, int h, int w, int d) { return out && box3_tile_supported(in, out, h, w, d, nullptr, nullptr, nullptr, nullptr); }

template <int TZ, int TY, int TXQ, int NW, int WPS>
static int launch_tile_t(const float* in, float* out, int h, int w, int d, int ns1, int ns2, int ns3, bool backward, float* P, float* m, float* v,
                         AdamConsts ac, float* gsave, hipStream_t s) {
    const int ntz = cdiv(h, TZ), nty = cdiv(w, TY), ntx = cdiv(d, 4 * TXQ);
    const int ntiles = 3 * ntz * nty * ntx;
    const unsigned nb = (unsigned)((ntiles + 7) / 8 * 8);
    // segments of at least two planes (bt_walk), at most one per two planes
    auto clampns = [](int ns, int planes) { return ns < 1 ? 1 : (ns > planes / 2 ? planes / 2 : ns); };
    const int a1 = clampns(ns1, TZ + 4), a2 = clampns(ns2, TZ + 2), a3 = clampns(ns3, TZ);
#define CVX_BT_LAUNCH(B, A) hipLaunchKernelGGL((k_box3_tile<TZ, TY, TXQ, NW, WPS, B, A>), dim3(nb), dim3(64 * NW), 0, s, in, out, h, w, d, ntz, nty, ntx, ntiles, a1, a2, a3, fastdiv_make(ntz), fastdiv_make(ntx), fastdiv_make(nty), fastdiv_make(a1), fastdiv_make(a2), fastdiv_make(a3), P, m, v, ac, gsave)
    if (!backward) CVX_BT_LAUNCH(false, false);
    else if (!P) CVX_BT_LAUNCH(true, false);
    else CVX_BT_LAUNCH(true, true);
#undef CVX_BT_LAUNCH
    return check_last("box3_tile");
}

// variant (option box_fwd_tile / box_bwd_tile) = kind * 1000 + NS1 * 100 + NS2 * 10 + NS3 (z segments per row in the three passes; 0 = the kind's default):
// kind 1 = 12 x 8 x 56 tiles, 8 wavefronts, two workgroups per CU (79.7 KB of LDS each); kind 2 = 12 x 16 x 56, 16 wavefronts, one per CU
// (137 KB).  Measured on the benchmark grid 80 x 96 x 112 under rocprofv3 (profiles/r05_boxtile_sweep.txt): marching kernel 19.8-20.0 us,
// kind 2 with 4 / 3 / 4 segments 15.6-16.6 us, kind 1 18.7-19.6 us; both kinds issue the marching kernel's 8.1 M VALU wave-instructions
// (the 27 additions per output and stage are ATen's) and run at the ~3.3 clocks per instruction of four wavefronts per SIMD -- the LDS
// footprint of the stage tiles leaves no room for more (a 64-register, 8-wavefront build of kind 1 spills and takes 25 us).
int launch_box3_tile(const float* in, float* out, int h, int w, int d, int variant, bool backward, float* P, float* m, float* v, AdamConsts ac,
                     float* gsave, hipStream_t s) {
    const int kind = variant / 1000, ns = variant % 1000;
    int ns1 = ns / 100, ns2 = (ns / 10) % 10, ns3 = ns % 10;
    if (kind == 1) {
        if (!ns) { ns1 = 5; ns2 = 3; ns3 = 4; }
        return launch_tile_t<12, 8, 14, 8, 4>(in, out, h, w, d, ns1, ns2, ns3, backward, P, m, v, ac, gsave, s);
    }
    if (!ns) { ns1 = 4; ns2 = 3; ns3 = 4; }
    return launch_tile_t<12, 16, 14, 16, 4>(in, out, h, w, d, ns1, ns2, ns3, backward, P, m, v, ac, gsave, s);
}
int launch_box3_tile_fwd(const float* in, float* out, int h, int w, int d, int variant, hipStream_t s) {
    return launch_box3_tile(in, out, h, w, d, variant, false, nullptr, nullptr, nullptr, AdamConsts{}, nullptr, s);
}

// automatic choice (option box_fwd_tile / box_bwd_tile = -1): the large tiles when they fill the chip
int box3_tile_fwd_auto(int h, int w, int d) {
    const int n2 = 3 * cdiv(h, 12) * cdiv(w, 16) * cdiv(d, 56);
    return n2 >= 192 ? 2000 : 0;
}

}  // namespace cvx
