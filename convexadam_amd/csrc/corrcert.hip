// corrcert.hip -- the SSD correlation volume in the CERTIFIED-FAST arithmetic (reference: correlate, convex_adam_utils.py:72-89).
//
// The pipeline consumes the cost volume only through argmin decisions (the plain argmin :87 and the six coupled passes :98-107), so
// it does not need ATen's evaluation order -- it needs values with a PROVEN distance to ATen's, and an exact evaluator for the
// decisions that distance cannot settle (certify.hip).  This kernel produces
//
//     ssdu[k,x] = sum over the two nested 3^3 windows of raw'[k,.]           (UNSCALED: 729 x the mean, no division)
//     raw'[k,x] = fma chain over the channels of (F_c(x) - M0_c(x + delta_k))^2
//
// with the boxes evaluated separably per axis as S.R.S (S = zero-extended 3-tap sum, R = restriction to the volume: the second
// avg_pool3d pads the FIRST one's output with zeros, convex_adam_utils.py:84).  All terms are non-negative, so every partial sum
// carries a relative error of at most (number of roundings on its path) x 2^-24: |ssdu / 729 - ssd| <= E_REL x ssd with E_REL = 2^-16
// covering ATen's own ~70 roundings, this kernel's ~30 and the two corrected edge terms (factor 2) with a margin of 2.5 (DESIGN 12.1;
// measured: 5.8e-7).  ssdu == 0 iff all 125 channel sums are 0 iff the exact entry is 0: there is no multiplication that could underflow.
//
// Work decomposition.  A workgroup marches along z over whole (w x d) planes for ONE (dH, dW) row of the search window and up to nine
// adjacent D-shifts: two thread sets of wps wavefronts, each set owning <= 5 shifts (thread = 4 voxels x B shifts).  2197 shifts over
// 256 CUs is 8.6 per CU: rows are cut 4 + 5 | 4 and the leftover fours of two adjacent dW rows share a workgroup (their M tiles differ
// by one row), which gives 169 + 78 + 13 workgroups of 9 / 8 / 4 shifts for n = 13 -- one round, the short ones last.
//   stage  (LDS-DMA)   F plane and the M tile of the step, CH channels at a time, double buffered: every staged byte is read from the
//                      L2 once per workgroup-step (620 MB per launch; the role kernel of corrfused.hip moves 2 GB through the L1).  The
//                      staging copies are laid out so that a tile of one channel is ONE contiguous run: lane address = base + 16 tid
//   raw    (phase 1)   lane = (quad column, row y): conflict-free ds_read_b128 of 1 F + 2 M quads per channel for 4 x B outputs;
//                      then S.R.S along y with v_fmac_f32_dpp wave_shr / wave_shl (row masks as multipliers), columns >= d zeroed
//   exch               the y-filtered quads go through LDS once (the transposition to row-major threads)
//   box    (phase 2)   thread = (row y, quad) row-major: x halo from LDS, S.R.S along x in registers (zero-extended sums + two edge
//                      corrections), S.R.S along z as two chained running sums, coalesced 16-byte stores of plane q - 2
// No MFMA: the work is a stencil over a gather (north_star); the bound is VALU issue (39 operations per output).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cvx_common.h"

namespace cvx {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int CC_CH = 4;          // channels per staged chunk
// Staging helpers (the wavefronts of a set of <= 4 shifts stage 64 quads each of their tile's run besides the loaders): built and measured --
// the helpers then wait for their own loads at every chunk barrier (1.45 us under the load of 254 workgroups: the L2s deliver ~10 TB/s of
// the 0.97 GB a launch stages), 141 -> 169 us for the 9-shift workgroups -- and left off.
constexpr bool CC_HELPERS = false;
constexpr int CC_MAXT = 4;        // tile types: pair workgroups per row (n <= 31: 3) + the single set

struct CCGeom {                   // (host side; the kernel receives the CCKern subset)
    int C, h, w, d, hw, n;
    int lpr, T;                   // quads per row (>= 2 zero columns behind the row), quads per plane
    int cpw, wps;                 // quad columns per wavefront in phase 1; wavefronts per set
    int FQ, MQ;                   // quads per row of the staged tiles (odd: conflict-free 16-byte reads at row stride)
    int HQ, WQ;                   // padded moving copies: planes, rows
    int np;                       // pair workgroups per row: shifts [pstart[i], pstart[i] + 4 + pB2[i])
    int pstart[CC_MAXT], pB2[CC_MAXT];
    int sB, sstart;               // leftover single set of a row (sB = 0: none)
    int ntype;                    // tile types = np + (sB > 0): type t copies the moving features from shift index tstart[t] on
    int tstart[CC_MAXT];
    int nch, exq;                 // chunks per step; quads per exchange plane: T + 2
    unsigned off_F, off_M[CC_MAXT];   // byte offsets inside the staging buffer
    unsigned chanF, chanM;        // bytes per channel
    unsigned stage_bytes;
    bool ok;                      // the cut of a row fits the tile types
};

static CCGeom cc_geom(int C, int h, int w, int d, int hw) {
    CCGeom g{};
    g.C = C; g.h = h; g.w = w; g.d = d; g.hw = hw; g.n = 2 * hw + 1;
    g.lpr = (d + 2 + 3) / 4;
    g.T = w * g.lpr;
    g.cpw = w <= 64 ? 64 / w : 0;
    const int w1 = g.cpw ? cdiv(g.lpr, g.cpw) : 0, w2 = cdiv(g.T, 64);
    g.wps = w1 > w2 ? w1 : w2;
    g.FQ = g.lpr | 1;
    g.MQ = (g.lpr + 2) | 1;
    g.HQ = h + 2 * hw; g.WQ = w + 2 * hw + 1;
    // cut of a row's n shifts: pairs (4, B2) while at least 6 remain, then one single set of at most 5
    int rem = g.n, start = 0;
    g.np = 0; g.ntype = 0;
    while (rem >= 6 && g.np < CC_MAXT - 1) {
        const int b2 = rem - 4 >= 5 ? 5 : rem - 4;
        g.pstart[g.np] = start; g.pB2[g.np] = b2; g.tstart[g.ntype++] = start;
        ++g.np; start += 4 + b2; rem -= 4 + b2;
    }
    g.ok = rem <= 5;
    g.sB = rem; g.sstart = start;
    if (rem) g.tstart[g.ntype++] = start;
    g.nch = cdiv(C, CC_CH);
    g.exq = g.T + 2;
    g.chanF = 16u * (unsigned)(h * w * g.FQ);
    g.chanM = 16u * (unsigned)(g.HQ * g.WQ * g.MQ);
    unsigned off = 0;
    g.off_F = off; off += (unsigned)align_up((size_t)g.chanF * C, 256);
    for (int i = 0; i < g.ntype; ++i) { g.off_M[i] = off; off += (unsigned)align_up((size_t)g.chanM * C, 256); }
    g.stage_bytes = off;
    return g;
}
static int cc_exch_planes(const CCGeom& g) {
    int m = g.sB ? 2 * g.sB : 0;
    for (int i = 0; i < g.np; ++i) m = m > 4 + g.pB2[i] ? m : 4 + g.pB2[i];
    return m;
}
// largest staging layout of a launch: four channels of (F plane + one M tile of up to w + 1 rows), or -- the workgroups that hold the
// single sets of two dH -- two channels of (F plane + two tiles of w rows)
static size_t cc_chunk_quads(const CCGeom& g) {
    const size_t a = (size_t)CC_CH * ((size_t)g.w * g.FQ + (size_t)(g.w + 1) * g.MQ), b = 2 * ((size_t)g.w * g.FQ + 2 * (size_t)g.w * g.MQ);
    return a > b ? a : b;
}
static int cc_nwg(const CCGeom& g) { return g.n * (g.n * g.np + (g.sB ? g.n / 2 : 0)) + (g.sB ? (g.n + 1) / 2 : 0); }
static size_t cc_lds_bytes(const CCGeom& g) { return 16 * (2 * cc_chunk_quads(g) + (size_t)cc_exch_planes(g) * g.exq); }

bool corr_cert_supported(int C, int h, int w, int d, int hw) {
    // C <= 128: the certification constants (2^-16, certify.hip) cover (C + 18) roundings of the fast chain + (1 + 15 + C/16 + 54) of ATen's
    // cascade = 224 u at C = 128; beyond that the a-priori distance passes 256 u = 2^-16 and the exact path is taken
    if (C < 1 || C > 128 || hw < 0 || hw > CVX_MAX_DISP_HW || h < 1 || w < 1 || w > 64 || d < 1) return false;
    const CCGeom g = cc_geom(C, h, w, d, hw);
    if (!g.ok || g.wps < 1 || 2 * g.wps + 2 > 16) return false;
    if (cdiv(w * g.FQ, 128) + cdiv((w + 1) * g.MQ, 128) > 8 || cdiv(w * g.FQ, 128) + 2 * cdiv(w * g.MQ, 128) > 12) return false;      // pieces per channel the loaders hold
    if (cc_lds_bytes(g) > 160 * 1024) return false;
    const size_t total = (size_t)g.off_M[g.ntype - 1] + (size_t)g.chanM * C;
    return total < ((size_t)1 << 31) && (size_t)h * w * d * 4 < ((size_t)1 << 31);
}
size_t corr_cert_workspace_bytes(int C, int h, int w, int d, int hw) {
    const CCGeom g = cc_geom(C, h, w, d, hw);
    return (size_t)g.stage_bytes + 512 + 32 * 1024;          // (+ the residency census of option cc_debug)
}

// ---- staging copies: Fp [C][h][w][4 FQ] (zeros behind d); Mp_t [C][HQ][WQ][4 MQ] per tile type t with element i = the moving
// feature at shift index tstart[t] + i, i.e. at column tstart[t] + i - hw (zero outside the volume); rows / planes shifted by hw ------
__global__ __launch_bounds__(256) void k_cc_prep(const float* __restrict__ fix, const float* __restrict__ mov, CCGeom g, char* __restrict__ stage) {
    const size_t nF = (size_t)g.C * g.h * g.w * g.FQ, nM = (size_t)g.C * g.HQ * g.WQ * g.MQ;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nF) {
        const int q = (int)(i % g.FQ);
        const size_t r = i / g.FQ;                      // (c*h + z)*w + y
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (4 * q + j < g.d) ? fix[r * g.d + 4 * q + j] : 0.0f;
        reinterpret_cast<float4*>(stage + g.off_F)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (i < nM) {
        const int q = (int)(i % g.MQ), yy = (int)((i / g.MQ) % g.WQ), zz = (int)((i / ((size_t)g.MQ * g.WQ)) % g.HQ);
        const int c = (int)(i / ((size_t)g.MQ * g.WQ * g.HQ));
        const int y = yy - g.hw, z = zz - g.hw;
        const bool rowin = y >= 0 && y < g.w && z >= 0 && z < g.h;
        const float* row = mov + (((size_t)c * g.h + (rowin ? z : 0)) * g.w + (rowin ? y : 0)) * g.d;
        for (int a = 0; a < g.ntype; ++a) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = g.tstart[a] + 4 * q + j - g.hw;
                o[j] = (rowin && x >= 0 && x < g.d) ? row[x] : 0.0f;
            }
            reinterpret_cast<float4*>(stage + g.off_M[a])[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- what the kernel needs to know ---------------------------------------------------------------------------------------------
struct CCKern {
    int C, h, w, d, n;
    int lpr, T, cpw, wps, FQ, MQ, HQ, WQ;
    int np, pstart[CC_MAXT - 1], pB2[CC_MAXT - 1];
    int sB, sstart, exq;
    int nwg;
    int chunk_floats;             // floats per staging buffer (the largest layout of the launch)
    unsigned off_F, off_M[CC_MAXT], chanF, chanM, stage_bytes;
    int dbg;                      // option cc_debug bits: 1 census, 2 compute wavefronts idle, 4 loaders do not commit, 8 loaders do not fetch (timing experiments)
    unsigned long long* census;   // option cc_debug: per workgroup {start, end, class / XCC_ID, shifts} (s_memrealtime ticks of 10 ns)
};
// A work item: one or two sets (shifts [start, start + B) of row (iH, iW)); the sets read one shared M tile, or -- the leftover single
// sets of two dH, which share nothing but the F plane -- one tile each, and then stage two channels per chunk instead of four.
struct CCItem {
    int nsets, ntile, chn;
    int B[2], iH[2], iW[2], start[2], rowoff[2], gq[2];
    int type, row0, RM;
    int hstart[2];                // first quad of the tile run that the set's own wavefronts stage (sets of <= 4 shifts help the loaders; -1: none)
    int lstart[2];                // per tile: quads the helpers cover = where the loaders' share of the run begins
};
// Launch order.  The hardware deals workgroups to the 8 XCDs round-robin (block b -> XCD b % 8), and every XCD has its own 4 MB L2:
// the staging copies (10 MB) only stay L2-resident if an XCD's workgroups walk the SAME planes at the same time.  So the work list is
// sorted by dH (at step z every row of one dH reads M plane z + dH and the same F plane) and each XCD takes one contiguous eighth
// of it.  Per dH: n * np pair workgroups (class 0), n / 2 workgroups with the single sets of two adjacent rows (class 1); the single
// set of the last row shares a workgroup with the one of dH + 1 (class 2, listed under the even dH; two M tiles), the very last
// one stays alone (class 3).  2 sets of at most 5 shifts per workgroup would otherwise leave n lone single sets -- 260 workgroups
// for n = 13, four more than the chip has CUs, i.e. a second launch round (measured: 315 instead of 180 us).
__device__ __forceinline__ void cc_order(const CCKern& g, int b, int& cls, int& iH, int& idx) {
    const int n = g.n, nwg = g.nwg;
    const int base = n * g.np + (g.sB ? n / 2 : 0);        // entries of every dH; even dH carry one more (the cross / lone entry)
    const int extra = g.sB ? 1 : 0;
    const int per2 = 2 * base + extra;
    const int x = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int lo = x * q + (x < r ? x : r);                // this XCD's share [lo, lo + cnt) of the dH-major list, taken in list order
    const int L = lo + slot;
    const int pi = L / per2, rem = L - pi * per2;
    int j;
    if (rem < base + extra) { iH = 2 * pi; j = rem; }
    else { iH = 2 * pi + 1; j = rem - base - extra; }
    const int n0 = n * g.np;
    if (j < n0) { cls = 0; idx = j; }
    else if (j < base) { cls = 1; idx = j - n0; }
    else { cls = iH + 1 < n ? 2 : 3; idx = 0; }
}
__device__ __forceinline__ CCItem cc_decode(const CCKern& g, int b) {
    CCItem it;
    const int n = g.n;
    int cls, iH0, idx;
    cc_order(g, b, cls, iH0, idx);
    it.iH[0] = iH0; it.iH[1] = iH0;
    it.ntile = 1; it.chn = CC_CH;
    if (cls == 0) {
        const int iW = idx / g.np, pi = idx - iW * g.np;
        it.nsets = 2; it.type = pi; it.row0 = iW; it.RM = g.w;
        it.B[0] = 4; it.iW[0] = iW; it.start[0] = g.pstart[pi]; it.rowoff[0] = 0; it.gq[0] = 0;
        it.B[1] = g.pB2[pi]; it.iW[1] = iW; it.start[1] = g.pstart[pi] + 4; it.rowoff[1] = 0; it.gq[1] = 1;
    } else {
        it.type = g.np;
        it.B[0] = g.sB; it.start[0] = g.sstart; it.rowoff[0] = 0; it.gq[0] = 0;
        it.B[1] = g.sB; it.start[1] = g.sstart; it.rowoff[1] = 0; it.gq[1] = 0;
        if (cls == 1) { it.nsets = 2; it.row0 = 2 * idx; it.RM = g.w + 1; it.iW[0] = 2 * idx; it.iW[1] = 2 * idx + 1; it.rowoff[1] = 1; }
        else {
            it.row0 = n - 1; it.RM = g.w; it.iW[0] = n - 1; it.iW[1] = n - 1;
            if (cls == 2) { it.nsets = 2; it.ntile = 2; it.chn = 2; it.iH[1] = iH0 + 1; }
            else { it.nsets = 1; it.B[1] = 0; }
        }
    }
    // staging helpers: the wavefronts of a set of at most four shifts have registers to spare and stage wps * 64 quads of their tile's run
    {
        const int H = g.wps * 64, nM = it.RM * g.MQ;
        it.hstart[0] = it.hstart[1] = -1; it.lstart[0] = it.lstart[1] = 0;
        for (int q = 0; q < it.nsets; ++q) {
            const int t = it.ntile == 2 ? q : 0;
            if (CC_HELPERS && it.B[q] <= 4 && it.lstart[t] < nM) { it.hstart[q] = it.lstart[t]; it.lstart[t] = it.lstart[t] + H < nM ? it.lstart[t] + H : nM; }
        }
    }
    return it;
}

// y-direction taps through DPP: acc[j] += a[j](lane -/+ 1) * m.  The sources were written at least two instructions earlier (s_nop 1
// covers the VALU-write -> DPP-read hazard at the head of a block; inside a block no DPP source is written).
#define CC_DPP4(ctrl)                                                                                                               \
    asm("s_nop 1\n\t"                                                                                                              \
        "v_fmac_f32_dpp %0, %4, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                           \
        "v_fmac_f32_dpp %1, %5, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                           \
        "v_fmac_f32_dpp %2, %6, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                           \
        "v_fmac_f32_dpp %3, %7, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0"                                               \
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                                    \
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(m))
#define CC_DPP4M(ctrl)                                                                                                              \
    asm("s_nop 1\n\t"                                                                                                              \
        "v_fmac_f32_dpp %0, %4, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                           \
        "v_fmac_f32_dpp %1, %5, %9 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                           \
        "v_fmac_f32_dpp %2, %6, %10 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                                          \
        "v_fmac_f32_dpp %3, %7, %11 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0"                                              \
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                                    \
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]))
__device__ __forceinline__ void cc_tap_up4(float (&acc)[4], const float (&a)[4], const float (&m)[4]) { CC_DPP4M("wave_shr:1"); }
__device__ __forceinline__ void cc_tap_dn4(float (&acc)[4], const float (&a)[4], const float (&m)[4]) { CC_DPP4M("wave_shl:1"); }
__device__ __forceinline__ void cc_tap_up(float (&acc)[4], const float (&a)[4], float m) { CC_DPP4("wave_shr:1"); }
__device__ __forceinline__ void cc_tap_dn(float (&acc)[4], const float (&a)[4], float m) { CC_DPP4("wave_shl:1"); }

// workgroup barrier of this kernel: the waits are explicit (a __syncthreads() fence would drain the LDS-DMA queue again)
#ifdef CVX_RACE_JITTER
__device__ __forceinline__ void cc_barrier() { cvx_jitter(); __builtin_amdgcn_s_barrier(); cvx_jitter(); }
#else
__device__ __forceinline__ void cc_barrier() { __builtin_amdgcn_s_barrier(); }
#endif
// hides a value from loop-invariant code motion: what is derived from it is recomputed where it is used instead of living in
// registers across the whole march (the kernel runs at 168 registers per thread)
__device__ __forceinline__ int cc_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

struct CCCtx {
    float* lds;                    // [2][chunk] then the exchange planes
    int chunk_floats;              // floats per staging buffer
    float* exch;                   // exchange planes of THIS set (plane k at k * exq * 4)
    int nF, nM;                    // quads per channel of the F plane / of one M tile
    int chn, nch;                  // channels per chunk, chunks per plane
    __amdgpu_buffer_rsrc_t rsrc;   // the staging copies
    unsigned uM_step, chanM;
};

// ---- staging: two LOADER wavefronts per workgroup ---------------------------------------------------------------------------------
// (The LDS-DMA path moves ~25 GB/s per CU, a fifth of what this kernel needs -- measured: +146 us per launch; staging from the compute
// wavefronts' own registers leaves half a chunk of latency cover and costs them 16-32 registers -- measured: +39 us of stalls.)  The loaders
// own nothing else: each of their 128 threads keeps one chunk in flight in registers (pieces of 16 bytes: per channel one contiguous
// run of the F copy and one of each M tile, lane address = base + 16 * piece), requested while the compute wavefronts consume chunk g
// and committed to the other buffer (ds_write_b128) behind the barrier of chunk g + 1: a full chunk of latency cover.  A workgroup's
// wavefronts are dealt to the SIMDs cyclically, so wavefronts 10 and 11 land on the two SIMDs that hold only two compute wavefronts;
// they run at raised issue priority (a short instruction stream that everything else waits for).
constexpr int CC_LOADER_THREADS = 128;
__device__ __forceinline__ f32x4 cc_ld16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uni_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)uni_off, 0));
}
struct CCLoad {
    __amdgpu_buffer_rsrc_t rsrc;
    int lt;                        // loader thread 0 .. 127
    int rF, rM, ntile;             // pieces per channel of this thread's share: ceil(nF / 128), ceil(max share of a tile / 128)
    int lstart[2], lcount[2];      // the loaders' share of a tile's run: quads [lstart, lstart + lcount)
    unsigned uF0, uM0[2], uF_step, uM_step, chanF, chanM;
};
// piece r of a channel: r < RF the F run, then RM pieces per tile.  RF / RM / NTILE are compile-time (0 = take the run-time counts: the
// generic, slower instantiation); everything but the lane offset (16 * lt) and the tail mask of a run's last piece is wave-uniform.
template <int CHN, int RF, int RM, int NTILE, int NPC, bool FETCH>
__device__ __forceinline__ void cc_loader_pieces(const CCLoad& L, const CCCtx& c, int p, int ci, int buf, f32x4 (&R)[CHN][NPC]) {
    const int rF = RF ? RF : L.rF, rM = RM ? RM : L.rM, ntile = NTILE ? NTILE : L.ntile;
    const unsigned lane_off = 16u * (unsigned)L.lt;
    const bool tailF = L.lt < c.nF - (rF - 1) * CC_LOADER_THREADS;
    float* dst = c.lds + buf * c.chunk_floats + 4 * L.lt;
#pragma unroll
    for (int cl = 0; cl < CHN; ++cl) {
        const int ch = ci * CHN + cl;
        const unsigned uF = L.uF0 + (unsigned)p * L.uF_step + (unsigned)ch * L.chanF;
#pragma unroll
        for (int r = 0; r < NPC; ++r) {
            if (r < rF) {
                if (r < rF - 1 || tailF) {
                    if (FETCH) R[cl][r] = cc_ld16(L.rsrc, lane_off, uF + 2048u * (unsigned)r);
                    else lds_store4(dst + (cl * c.nF + r * CC_LOADER_THREADS) * 4, R[cl][r]);
                }
            } else if (r < rF + ntile * rM) {
                const int t = (r - rF) >= rM ? 1 : 0, rr = r - rF - t * rM;
                const int ls = t ? L.lstart[1] : L.lstart[0], lc = t ? L.lcount[1] : L.lcount[0];
                const unsigned um = t ? L.uM0[1] : L.uM0[0];
                if (rr * CC_LOADER_THREADS + L.lt < lc) {
                    if (FETCH) R[cl][r] = cc_ld16(L.rsrc, lane_off, um + (unsigned)p * L.uM_step + (unsigned)ch * L.chanM + 16u * (unsigned)ls + 2048u * (unsigned)rr);
                    else lds_store4(dst + (CHN * c.nF + (t * CHN + cl) * c.nM + ls + rr * CC_LOADER_THREADS) * 4, R[cl][r]);
                }
            }
        }
    }
}
template <int CHN, int RF, int RM, int NTILE, int NPC>
__device__ __forceinline__ void cc_loader_run(const CCKern& g, const CCLoad& L, const CCCtx& c) {
    const int h = g.h, nch = c.nch;
    f32x4 R[CHN][NPC];
    __builtin_amdgcn_s_setprio(3);
    // chunk 0 goes in synchronously; chunk 1 is requested before the march starts
    cc_loader_pieces<CHN, RF, RM, NTILE, NPC, true>(L, c, 0, 0, 0, R);
    cc_loader_pieces<CHN, RF, RM, NTILE, NPC, false>(L, c, 0, 0, 0, R);
    int np = nch > 1 ? 0 : 1, nci = nch > 1 ? 1 : 0;             // the chunk held in R
    bool have = np < h;
    if (have) cc_loader_pieces<CHN, RF, RM, NTILE, NPC, true>(L, c, np, nci, 0, R);
    int gc = 0;
    for (int p = 0; p < h; ++p)
        for (int ci = 0; ci < nch; ++ci) {
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0) only: own LDS writes are out (the loads in flight are NOT waited for)
            cc_barrier();                                        // chunk gc is visible; buffer (gc + 1) & 1 is free
            if (have) {
                if (!(g.dbg & 4)) cc_loader_pieces<CHN, RF, RM, NTILE, NPC, false>(L, c, np, nci, (gc + 1) & 1, R);
                if (++nci == nch) { nci = 0; ++np; }
                have = np < h;
                if (have && !(g.dbg & 8)) cc_loader_pieces<CHN, RF, RM, NTILE, NPC, true>(L, c, np, nci, 0, R);
            }
            ++gc;
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    cc_barrier();                                                // (the compute wavefronts' barrier of step h)
}

struct CCSetSel { int iH, iW, start, rowoff, gq, tile, hstart; unsigned uM0; };
template <int B, bool CT, int CHN>
__device__ __forceinline__ void cc_run(const CCKern& g, const CCCtx& c, const CCItem& it, const CCSetSel st, float* __restrict__ ssd, int lw, int lane) {
    const int h = g.h, w = g.w, d = g.d, lpr = g.lpr;
    // ---- phase-1 identity: lane = cq * w + y
    const int cq = lane / w, y1 = lane - cq * w;
    const int q1 = lw * g.cpw + cq;
    const bool act1 = cq < g.cpw && q1 < lpr;
    const int q1c = q1 < lpr ? q1 : lpr - 1;
    const int foff = (y1 * g.FQ + q1c) * 4;
    const int moff = (CHN * c.nF + st.tile * CHN * c.nM) * 4 + ((y1 + st.rowoff) * g.MQ + q1c + st.gq) * 4;
    const int fstep = c.nF * 4, mstep = c.nM * 4;
    const int xw = (1 + y1 * lpr + q1c) * 4;                           // exchange quad of (y1, q1)
    // ---- phase-2 identity: row-major quads
    // (rows of the FULL quads first, the partial / empty last quads of all rows behind them: only the last wavefront runs the element-wise stores)
    const int t2 = lw * 64 + lane;
    const bool act2 = t2 < g.T;
    const int t2c = t2 < g.T ? t2 : g.T - 1;
    const int qfull = d / 4, nfull = w * qfull;
    int y2, q2;
    if (t2c < nfull) { y2 = t2c / qfull; q2 = t2c - y2 * qfull; }
    else { const int tt = t2c - nfull, npart = lpr - qfull; y2 = tt / npart; q2 = qfull + tt - y2 * npart; }
    const int xr = (1 + y2 * lpr + q2) * 4;
    const int c0 = 4 * q2;
    const bool full = c0 + 3 < d;
    const int nn = g.n * g.n;
    const size_t vol = (size_t)h * w * d;
    // stores through a buffer descriptor: one 32-bit lane offset, everything else of an address is wave-uniform (scalar offset)
    const unsigned out_lane = 4u * (unsigned)(y2 * d + c0);
    const size_t out_item = ((size_t)st.start * nn + (size_t)st.iW * g.n + st.iH) * vol;           // floats
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(ssd + out_item, 0, (int)(unsigned)std::min<size_t>(0xfffffff0u, 4 * ((size_t)(B - 1) * nn * vol + vol)), 0x00020000);
    const unsigned kstride_b = 4u * (unsigned)((size_t)nn * vol), plane_b = 4u * (unsigned)(w * d);

    // z state per (shift, column): X = raw(q-1), B1 = raw(q-2) + raw(q-1), Y / B2 the same for the first z sums.  The newest value of a
    // plane is written into the array the plane before last used (Xa / Xb alternate: no register copies), hence the loop in steps of two.
    float Xa[B][4], Xb[B][4], Ya[B][4], Yb[B][4], B1[B][4], B2[B][4];
#pragma unroll
    for (int k = 0; k < B; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { Xa[k][j] = Xb[k][j] = Ya[k][j] = Yb[k][j] = B1[k][j] = B2[k][j] = 0.0f; }

    int gc = 0;                                       // chunk counter: buffer = gc & 1
    const int nch = c.nch;
    // ---- staging helper (sets of at most four shifts): this wavefront's 64 quads of the tile run, one chunk in flight like the loaders'
    constexpr bool HELP = CC_HELPERS && B <= 4;
    const int hq = st.hstart + lw * 64 + lane;        // quad of the run
    const bool hon = HELP && st.hstart >= 0 && hq < c.nM && lw * 64 + lane < g.wps * 64;
    f32x4 HR[HELP ? CHN : 1];
    int hp = nch > 1 ? 0 : 1, hci = nch > 1 ? 1 : 0;  // the chunk held in HR
    auto help_fetch = [&](const int p, const int ci) __attribute__((always_inline)) {
#pragma unroll
        for (int cl = 0; cl < CHN; ++cl)
            if (hon) HR[HELP ? cl : 0] = cc_ld16(c.rsrc, 16u * (unsigned)hq, st.uM0 + (unsigned)p * c.uM_step + (unsigned)(ci * CHN + cl) * c.chanM);
    };
    auto help_commit = [&](const int buf) __attribute__((always_inline)) {
        float* dst = c.lds + buf * c.chunk_floats + (CHN * c.nF + st.tile * CHN * c.nM + hq) * 4;
#pragma unroll
        for (int cl = 0; cl < CHN; ++cl)
            if (hon) lds_store4(dst + cl * c.nM * 4, HR[HELP ? cl : 0]);
    };
    // behind the barrier that opens chunk gc: commit the chunk in flight (gc + 1) to the other buffer, request chunk gc + 2
    auto help_tick = [&]() __attribute__((always_inline)) {
        if (HELP && st.hstart >= 0 && hp < h) {
            help_commit((gc + 1) & 1);
            if (++hci == nch) { hci = 0; ++hp; }
            if (hp < h) help_fetch(hp, hci);
        }
    };
    if (HELP && st.hstart >= 0) {
        help_fetch(0, 0); help_commit(0);             // chunk 0 synchronously (the first barrier of the march orders it)
        if (hp < h) help_fetch(hp, hci);
    }
    // x / z passes of plane q (the plane the previous step's y pass left in the exchange planes), stores of plane q - 2
    auto phase2 = [&](const int q, float (&Xn)[B][4], const float (&Xo)[B][4], float (&Yn)[B][4], const float (&Yo)[B][4]) __attribute__((always_inline)) {
        const bool feed = q >= 1 && q <= h;          // the first z sum of plane q - 1 lies inside the volume
        const bool emit = q >= 2 && act2;
        const int qo = cc_opaque(c0), xro = cc_opaque(xr);
        float ec[4];                                 // -(number of row ends the column is: first and / or last)
#pragma unroll
        for (int j = 0; j < 4; ++j) ec[j] = -((qo + j == 0 ? 1.0f : 0.0f) + (qo + j == d - 1 ? 1.0f : 0.0f));
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (q < h) {
                const float* e = c.exch + k * g.exq * 4 + xro;
                const f32x4 a4 = lds_load4(e);
                const f32x2 l2 = lds_load2(e - 2), r2 = lds_load2(e + 4);
                const float a[4] = {a4.x, a4.y, a4.z, a4.w};
                // zero-extended S.S over the row, then the two edge corrections (S.R.S = S.S - v at the first and last column)
                const float p1 = l2.y + a[0], p2 = a[1] + a[2], p3 = a[3] + r2.x;
                const float tm = l2.x + p1, t0 = p1 + a[1], t1 = a[0] + p2, t2_ = p2 + a[3], t3 = a[2] + p3, t4 = p3 + r2.y;
                const float s01 = t0 + t1, s23 = t2_ + t3;
                Xn[k][0] = __builtin_fmaf(ec[0], a[0], tm + s01);
                Xn[k][1] = __builtin_fmaf(ec[1], a[1], s01 + t2_);
                Xn[k][2] = __builtin_fmaf(ec[2], a[2], t1 + s23);
                Xn[k][3] = __builtin_fmaf(ec[3], a[3], s23 + t4);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) Xn[k][j] = 0.0f;
            }
            float o[4];
            if (feed) {                                          // (wave-uniform: no per-value selects)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Yn[k][j] = B1[k][j] + Xn[k][j];              // raw(q-2) + raw(q-1) + raw(q): first z sum of plane q - 1
                    B1[k][j] = Xo[k][j] + Xn[k][j];
                    o[j] = B2[k][j] + Yn[k][j];                  // second z sum of plane q - 2
                    B2[k][j] = Yo[k][j] + Yn[k][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    B1[k][j] = Xo[k][j] + Xn[k][j];
                    Yn[k][j] = 0.0f;
                    o[j] = B2[k][j];
                    B2[k][j] = Yo[k][j];
                }
            }
            if (emit) {
                const unsigned so = (unsigned)k * kstride_b + (unsigned)(q - 2) * plane_b;
                if (full) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{o[0], o[1], o[2], o[3]}), orsrc, (int)out_lane, (int)so, 2);          // (aux 2 = nt: the cost volume streams past the L2 that holds the staging copies)
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < d) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[j]), orsrc, (int)out_lane + 4 * j, (int)so, 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // one shift at a time (register pressure)
        }
    };
    // One step = the channel sums and the y pass of plane p; the x / z passes of plane p - 1 run behind the first barrier of the step (its
    // stores then have a whole chunk's arithmetic to drain, and the exchange planes need no barrier of their own: the step's first
    // barrier orders them).
    auto step = [&](const int p, float (&Xn)[B][4], const float (&Xo)[B][4], float (&Yn)[B][4], const float (&Yo)[B][4]) __attribute__((always_inline)) {
        float acc[B][4];
        // channel sums of one staged chunk (buffer gc & 1) into acc
        auto chunk = [&](const int ci) __attribute__((always_inline)) {
            const float* buf = c.lds + (gc & 1) * c.chunk_floats;
            const int cn = CT ? CHN : (g.C - ci * CHN < CHN ? g.C - ci * CHN : CHN);
            // one channel ahead: the window of channel cl + 1 is requested before channel cl is consumed, into the OTHER of two register sets
            // (no copies); running pointers: the per-channel addresses are not loop invariants the compiler could park in registers
            const float* pf = buf + cc_opaque(foff);
            const float* pm = buf + cc_opaque(moff);
            f32x4 fq[2], m0q[2], m1q[2];
            fq[0] = lds_load4(pf); m0q[0] = lds_load4(pm); m1q[0] = lds_load4(pm + 4);
#pragma unroll
            for (int cl = 0; cl < CHN; ++cl) {
                if (CT || cl < cn) {
                    if (cl + 1 < CHN && (CT || cl + 1 < cn)) {
                        pf += fstep; pm += mstep;
                        fq[(cl + 1) & 1] = lds_load4(pf);
                        m0q[(cl + 1) & 1] = lds_load4(pm);
                        m1q[(cl + 1) & 1] = lds_load4(pm + 4);
                    }
                    const f32x4 f4 = fq[cl & 1], m0 = m0q[cl & 1], m1 = m1q[cl & 1];
                    const float f[4] = {f4.x, f4.y, f4.z, f4.w};
                    const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                    for (int k = 0; k < B; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float df = f[j] - m[j + k];
                            acc[k][j] = __builtin_fmaf(df, df, acc[k][j]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (p <= h) {             // p < h: chunk 0 of plane p is staged;  p == h: the last plane's y pass is in the exchange planes
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wavefront's LDS writes (exchange planes, staged pieces) are out
            cc_barrier();
            if (p < h) help_tick();
        }
        // (p = 0: plane -1 = the zeroed exchange planes, nothing fed, nothing stored -- one call site, executed by every step, so that the
        // arrays it overwrites are dead across the loop edge)
        phase2(p - 1, Xn, Xo, Yn, Yo);
        if (p < h) {
#pragma unroll
            for (int k = 0; k < B; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k][j] = 0.0f;
            chunk(0);
            ++gc;
            for (int ci = 1; ci < nch; ++ci) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                cc_barrier();
                help_tick();
                chunk(ci);
                ++gc;
            }
            // S.R.S along y; columns >= d become exact zeros (the x pass zero-extends its input)
            const int yo = cc_opaque(y1), qo = cc_opaque(q1), xwo = cc_opaque(xw);
            const float mu = yo > 0 ? 1.0f : 0.0f, md = yo < w - 1 ? 1.0f : 0.0f;
            // column masks: folded into the second sum's taps where the registers allow it (sets of at most four shifts), else one more product
            constexpr bool FOLD = B <= 4;
            float mc[4], muc[FOLD ? 4 : 1], mdc[FOLD ? 4 : 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { mc[j] = (4 * qo + j < d) ? 1.0f : 0.0f; if (FOLD) { muc[j] = mu * mc[j]; mdc[j] = md * mc[j]; } }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                float t[4], u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = acc[k][j];
                cc_tap_up(t, acc[k], mu);
                cc_tap_dn(t, acc[k], md);
                if (FOLD) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = t[j] * mc[j];
                    cc_tap_up4(u, t, reinterpret_cast<const float (&)[4]>(muc));
                    cc_tap_dn4(u, t, reinterpret_cast<const float (&)[4]>(mdc));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = t[j];
                    cc_tap_up(u, t, mu);
                    cc_tap_dn(u, t, md);
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] *= mc[j];
                }
                if (act1) lds_store4(c.exch + k * g.exq * 4 + xwo, f32x4{u[0], u[1], u[2], u[3]});
                __builtin_amdgcn_sched_barrier(0);          // one shift at a time: the scheduler must not interleave all B (register pressure)
            }
        }
    };
    // (plane q's x / z passes write the arrays of parity q: step p handles plane p - 1)
    for (int p = 0; p < h + 3; p += 2) {
        step(p, Xb, Xa, Yb, Ya);
        if (p + 1 < h + 3) step(p + 1, Xa, Xb, Ya, Yb);
    }
}

template <int CT, int MAXT>
__global__ __launch_bounds__(MAXT) void k_corr_cert(CCKern g, char* __restrict__ stage, float* __restrict__ ssd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const CCItem it = cc_decode(g, (int)blockIdx.x);
    const int si = __builtin_amdgcn_readfirstlane(wave / g.wps), lw = wave - si * g.wps;
    if (g.census && tid == 0) {
        g.census[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        g.census[4 * blockIdx.x + 2] = ((unsigned long long)it.ntile << 32) | __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        g.census[4 * blockIdx.x + 3] = (unsigned long long)(it.B[0] + it.B[1]);
    }
    // wavefronts of a set the item does not have leave at once (a lone single set): the barriers count the others only
    if (si < 2 && si >= it.nsets) return;
    CCCtx c;
    c.lds = lds;
    c.nF = g.w * g.FQ;
    c.nM = it.RM * g.MQ;
    c.chn = it.chn;
    c.nch = (g.C + it.chn - 1) / it.chn;
    c.chunk_floats = g.chunk_floats;
    float* exch0 = lds + 2 * c.chunk_floats;
    c.exch = exch0 + (si == 1 ? it.B[0] : 0) * g.exq * 4;
    const unsigned uM_step = 16u * (unsigned)(g.WQ * g.MQ);
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc(stage, 0, (int)g.stage_bytes, 0x00020000);
    c.uM_step = uM_step; c.chanM = g.chanM;
    if (si >= 2) {
        CCLoad L;
        L.rsrc = c.rsrc;
        L.lt = tid - 2 * g.wps * 64;
        L.rF = (c.nF + CC_LOADER_THREADS - 1) / CC_LOADER_THREADS;
        L.ntile = it.ntile;
        L.lstart[0] = it.lstart[0]; L.lstart[1] = it.lstart[1];
        L.lcount[0] = c.nM - it.lstart[0]; L.lcount[1] = it.ntile == 2 ? c.nM - it.lstart[1] : 0;
        L.rM = ((L.lcount[0] > L.lcount[1] ? L.lcount[0] : L.lcount[1]) + CC_LOADER_THREADS - 1) / CC_LOADER_THREADS;
        L.uF_step = 16u * (unsigned)(g.w * g.FQ);
        L.uM_step = uM_step;
        L.uF0 = g.off_F;
        L.uM0[0] = g.off_M[it.type] + (unsigned)it.iH[0] * uM_step + 16u * (unsigned)(it.row0 * g.MQ);
        L.uM0[1] = g.off_M[it.type] + (unsigned)it.iH[1] * uM_step + 16u * (unsigned)(it.row0 * g.MQ);
        L.chanF = g.chanF; L.chanM = g.chanM;
        // (the benchmark geometry -- F planes of 3 x 128 pieces, at most one piece left of a tile's run behind the helpers -- has straight-line
        // instantiations)
        if (it.chn == 2) {
            if (L.rF == 3 && L.rM == 4) cc_loader_run<2, 3, 4, 2, 11>(g, L, c);
            else if (L.rF == 3 && L.rM == 1) cc_loader_run<2, 3, 1, 2, 5>(g, L, c);
            else cc_loader_run<2, 0, 0, 0, 12>(g, L, c);
        } else if (L.rF == 3 && L.rM == 4) cc_loader_run<4, 3, 4, 1, 7>(g, L, c);
        else if (L.rF == 3 && L.rM == 1) cc_loader_run<4, 3, 1, 1, 4>(g, L, c);
        else cc_loader_run<4, 0, 0, 0, 8>(g, L, c);
        return;
    }
    // zero this set's exchange planes once: the quads before the first and behind the last row are never written
    {
        const int nex = (si ? it.B[1] : it.B[0]) * g.exq;
        for (int i = lw * 64 + lane; i < nex; i += g.wps * 64) lds_store4(c.exch + i * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    const int Bs = __builtin_amdgcn_readfirstlane(si ? it.B[1] : it.B[0]);
    const unsigned uMt0 = g.off_M[it.type] + 16u * (unsigned)(it.row0 * g.MQ);
    const CCSetSel st = si ? CCSetSel{it.iH[1], it.iW[1], it.start[1], it.rowoff[1], it.gq[1], it.ntile - 1, it.hstart[1], uMt0 + (unsigned)it.iH[it.ntile - 1] * uM_step}
                           : CCSetSel{it.iH[0], it.iW[0], it.start[0], it.rowoff[0], it.gq[0], 0, it.hstart[0], uMt0 + (unsigned)it.iH[0] * uM_step};
    // (the shifts per set and the channels per chunk are compile-time inside the march; CT: every chunk is full)
#define CC_RUN(BB)                                                                                                    \
    do {                                                                                                             \
        if (it.chn == 2) cc_run<BB, CT != 0, 2>(g, c, it, st, ssd, lw, lane); else cc_run<BB, CT != 0, CC_CH>(g, c, it, st, ssd, lw, lane); \
    } while (0)
    switch (Bs) {
        case 5: CC_RUN(5); break;
        case 4: CC_RUN(4); break;
        case 3: CC_RUN(3); break;
        case 2: CC_RUN(2); break;
        default: CC_RUN(1); break;
    }
#undef CC_RUN
    if (g.census && tid == 0) g.census[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

// fix, mov [C][h][w][d] -> ssdu [n^3][h][w][d]: UNSCALED certified-fast cost volume (729 x the reference's value to within E_REL)
int launch_corr_cert(const float* fix, const float* mov, int C, int h, int w, int d, int hw, float* ssdu, void* workspace, size_t workspace_bytes,
                     hipStream_t s) {
    if (!corr_cert_supported(C, h, w, d, hw)) return fail(CVX_ERR_UNSUPPORTED, "correlate (certified fast): geometry outside the kernel's range");
    if (workspace_bytes < corr_cert_workspace_bytes(C, h, w, d, hw)) return fail(CVX_ERR_WORKSPACE, "correlate (certified fast): workspace too small");
    const CCGeom g = cc_geom(C, h, w, d, hw);
    char* stage = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
    const size_t nprep = std::max((size_t)C * g.HQ * g.WQ * g.MQ, (size_t)C * h * w * g.FQ);
    hipLaunchKernelGGL(k_cc_prep, dim3((unsigned)cdiv64((int64_t)nprep, 256)), dim3(256), 0, s, fix, mov, g, stage);
    CCKern k{};
    k.C = C; k.h = h; k.w = w; k.d = d; k.n = g.n;
    k.lpr = g.lpr; k.T = g.T; k.cpw = g.cpw; k.wps = g.wps; k.FQ = g.FQ; k.MQ = g.MQ; k.HQ = g.HQ; k.WQ = g.WQ;
    k.np = g.np;
    for (int i = 0; i < CC_MAXT - 1; ++i) { k.pstart[i] = g.pstart[i]; k.pB2[i] = g.pB2[i]; }
    k.sB = g.sB; k.sstart = g.sstart; k.exq = g.exq;
    k.nwg = cc_nwg(g);
    k.chunk_floats = (int)cc_chunk_quads(g) * 4;
    k.off_F = g.off_F;
    for (int i = 0; i < CC_MAXT; ++i) k.off_M[i] = g.off_M[i];
    k.chanF = g.chanF; k.chanM = g.chanM; k.stage_bytes = g.stage_bytes;
    k.dbg = (int)options().cc_debug;
    k.census = options().cc_debug ? reinterpret_cast<unsigned long long*>(stage + align_up((size_t)g.stage_bytes, 256)) : nullptr;
    const size_t lds = cc_lds_bytes(g);
    const dim3 block(2 * g.wps * 64 + CC_LOADER_THREADS);
#define CC_LAUNCH(CT, MAXT)                                                                         \
    do {                                                                                           \
        static size_t granted = 0;                                                                 \
        ensure_dynamic_lds(&k_corr_cert<CT, MAXT>, lds, granted);                                  \
        hipLaunchKernelGGL((k_corr_cert<CT, MAXT>), dim3(k.nwg), block, lds, s, k, stage, ssdu);   \
    } while (0)
    // (768 threads = 3 wavefronts per SIMD: 168 registers per thread)
    if (C % CC_CH == 0) { if (block.x <= 768) CC_LAUNCH(1, 768); else CC_LAUNCH(1, 1024); }
    else { if (block.x <= 768) CC_LAUNCH(0, 768); else CC_LAUNCH(0, 1024); }
#undef CC_LAUNCH
    return check_last("corr_cert");
}

}  // namespace cvx
