// certify.hip -- the reference's argmin decisions (torch.argmin(ssd, 0) convex_adam_utils.py:87; the six coupled passes :98-107) taken
// on the CERTIFIED-FAST cost volume of corrcert.hip, bit for bit the decisions the exact volume gives.
//
// ssdu[k,x] is the unscaled fast volume: the exact entry ssd[k,x] lies in the interval
//     [ max(ssdu * CLO - TINY, 0),  ssdu * CHI + min(ssdu, TINY) ]        CLO / CHI = (1 -/+ 2^-16) / 729
// (relative part: DESIGN 12.1; TINY covers the denormal range, where ATen's divisions may round to zero; ssdu == 0 <=> ssd == 0).
// Rounding is monotonic, so the exact cost fl(ssd + pen) of a displacement lies in [lo, hi] = [fl(lower + pen), fl(upper + pen)].
// A decision is CERTAIN when the second smallest lo exceeds the smallest hi (then every other displacement costs strictly more than
// the one with the smallest lo); otherwise the candidates {k : lo_k <= min hi} are evaluated EXACTLY -- one wavefront restates
// ATen's arithmetic for that entry: 125 channel sums in `.sum(0)`'s order (cascade for C >= 16, the interleaved order of the tensor's
// last < 32 elements), 27 raster-order box sums / 27, one more -- and the first minimum of the exact costs wins.  Zero backgrounds
// stay certain: their entries are exact zeros on both sides, the penalties decide.  On the benchmark pair 0..3 voxels of 30 784 need the
// evaluator per pass.
//
// Pruning is the branch and bound of convex.hip with the bounds widened to what is known: the lower bound of every cost is
// fl(lower(min_k ssdu) + pen_k), the upper bound of the minimum is hi of any displacement (the previous winner, or the lattice point
// nearest to u).  Same launches per pass as the exact path (a voxel kernel, a wavefront kernel), both directions of a pair in each.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "cvx_common.h"

namespace cvx {

// interval of the exact entry (see the header): ATen's path rounds at most 67 times between an input and an output (1 product, 12 channel
// additions, 2 x (26 additions + 1 division)), the fast kernels at most 25-30 (with the factor 2 of the two subtracted edge terms): together
// below 7.6e-6 = 2^-17 relative; the constants use 2^-16 (measured distance: 5.8e-7)
#define CERT_CLO ((float)((1.0 - 1.0 / 65536.0) / 729.0))
#define CERT_CHI ((float)((1.0 + 1.0 / 65536.0) / 729.0))
#define CERT_TINY 1.0e-40f
__device__ __forceinline__ float cert_lower(float s) { return fmaxf(__builtin_fmaf(s, CERT_CLO, -CERT_TINY), 0.0f); }
__device__ __forceinline__ float cert_upper(float s) { return __builtin_fmaf(s, CERT_CHI, fminf(s, CERT_TINY)); }

__device__ __forceinline__ unsigned ordered_bits(float v) {              // the value part of pack_min_key
    unsigned b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    if (v != v) b = 0u;
    return b;
}
__device__ __forceinline__ float from_ordered_bits(unsigned b) { return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b); }

// a voxel the voxel kernel leaves to the wavefront kernel, with everything it already knows (the wavefront kernel then starts its scan one
// memory round trip after its launch instead of four)
struct CertRec { unsigned x, box; int kp; float bound, slo, uc, ub, ua; };

struct CertProb {
    const float* ssdu; const float* fix; const float* mov; const float* tail;
    unsigned long long* key; unsigned* sec;       // plain pass: (min, index) and the runner-up value
    int* idx0; int* idxA; int* idxB;              // winners: plain pass, coupled passes (ping-pong)
    float* smin;                                  // min_k ssdu[k,x] (NaN: the column holds a NaN)
    unsigned* list; int* counts;                  // flagged voxels of the plain pass; one counter per pass: [0] plain, [1..6] coupled, [7] statistics, [8] work items
    unsigned long long* work; unsigned work_cap;  // plain pass: (voxel << 32 | displacement) entries that need the exact evaluator
    struct CertRec* rec;                          // work records of the coupled passes (the voxel kernel hands its box to the wavefront kernel)
    float* u;                                     // [3][v] running smoothed field = the result
    int64_t* argmin_out;                          // optional int64 copy of the plain winners
};
struct CertGeo { int C, h, w, d, hw, n, K, ntail; long long tail_from; };
struct CertArgs { CertProb p[2]; CertGeo g; const float* mesh; float coef; int pass; };

// ---- exact evaluation of ONE entry by one wavefront (sm = 160 floats of LDS that belong to the calling wavefront) ----------------
// WAVE: the workgroup holds several wavefronts -- LDS accesses of ONE wavefront complete in order, so a wavefront-level fence replaces the
// workgroup barrier
template <bool WAVE>
__device__ __forceinline__ void cert_sync() {
    if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __syncthreads();
}
template <bool WAVE>
__device__ float cert_exact_entry_t(const CertGeo& G, const CertProb& P, int k, int xlin, float* sm, int lane) {
    const int n = G.n, nn = n * n, h = G.h, w = G.w, d = G.d, hw = G.hw;
    const int iH = k % n, iW = (k / n) % n, iD = k / nn;
    const int x = xlin % d, y = (xlin / d) % w, z = xlin / (d * w);
    const size_t v = (size_t)h * w * d;
    for (int t = lane; t < 125; t += 64) {
        const int a = t / 25 - 2, b = (t / 5) % 5 - 2, c = t % 5 - 2;
        const int pz = z + a, py = y + b, px = x + c;
        float val = 0.0f;
        if (pz >= 0 && pz < h && py >= 0 && py < w && px >= 0 && px < d) {
            const long long flat = (((long long)pz * nn + iW * n + iD) * w + py) * d + px;   // position in the reference's (h, n^2, w, d) tensor
            if (G.ntail > 0 && flat >= G.tail_from) val = P.tail[iH * 32 + (int)(flat - G.tail_from)];
            else {
                const int mz = pz + iH - hw, my = py + iW - hw, mx = px + iD - hw;
                const bool inb = mz >= 0 && mz < h && my >= 0 && my < w && mx >= 0 && mx < d;
                const float* fp = P.fix + ((size_t)pz * w + py) * d + px;
                const float* mp = P.mov + ((size_t)(inb ? mz : 0) * w + (inb ? my : 0)) * d + (inb ? mx : 0);
                float a0 = 0.0f, a1 = 0.0f;
                // (16 channels at a time: all their loads are in flight together, the sum then runs in ATen's order)
                for (int c0 = 0; c0 < G.C; c0 += 16) {
                    float fv[16], mv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool on = c0 + i < G.C;
                        fv[i] = on ? fp[(size_t)(c0 + i) * v] : 0.0f;
                        mv[i] = (on && inb) ? mp[(size_t)(c0 + i) * v] : 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (c0 + i < G.C) {
                            const float df = fv[i] - mv[i];
                            a0 += df * df;                                    // .pow(2).sum(0)
                        }
                    }
                    if (c0 + 16 <= G.C) { a1 += a0; a0 = 0.0f; }              // ATen's cascade (two levels below 256 channels): a full block of 16 is folded
                }
                val = G.C >= 16 ? a0 + a1 : a0;
            }
        }
        sm[t] = val;
    }
    cert_sync<WAVE>();
    if (lane < 27) {
        const int a = lane / 9 - 1, b = (lane / 3) % 3 - 1, c = lane % 3 - 1;
        const int pz = z + a, py = y + b, px = x + c;
        float s = 0.0f;
        const bool in = pz >= 0 && pz < h && py >= 0 && py < w && px >= 0 && px < d;
        if (in) {
            for (int aa = -1; aa <= 1; ++aa)
                for (int bb = -1; bb <= 1; ++bb)
                    for (int cc = -1; cc <= 1; ++cc) {
                        const int qz = pz + aa, qy = py + bb, qx = px + cc;
                        if (qz >= 0 && qz < h && qy >= 0 && qy < w && qx >= 0 && qx < d) s += sm[((a + aa + 2) * 5 + (b + bb + 2)) * 5 + (c + cc + 2)];
                    }
            s = div_exact<27>(s);
        }
        sm[128 + lane] = in ? s : -1.0f;           // (-1 marks a tap outside the volume; box sums are never negative)
    }
    cert_sync<WAVE>();
    float s = 0.0f;
    for (int t = 0; t < 27; ++t) {
        const float b = sm[128 + t];
        if (b >= 0.0f || b != b) s += b;
    }
    cert_sync<WAVE>();                                // sm is reused by the next entry
    return div_exact<27>(s);
}

__device__ float cert_exact_entry(const CertGeo& G, const CertProb& P, int k, int xlin, float* sm, int lane) { return cert_exact_entry_t<false>(G, P, k, xlin, sm, lane); }
__device__ float cert_exact_entry_wave(const CertGeo& G, const CertProb& P, int k, int xlin, float* sm, int lane) { return cert_exact_entry_t<true>(G, P, k, xlin, sm, lane); }

// ---- plain argmin: one streaming pass, (min, index) through a 64-bit atomicMin, the runner-up through what that atomic displaces --
__global__ __launch_bounds__(256) void k_cert_plain_stream(CertArgs A, int kslice, int vec4) {
    const CertProb& P = A.p[blockIdx.z];
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    const int K = A.g.K;
    const size_t x0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int k0 = blockIdx.y * kslice, k1 = min(k0 + kslice, K);
    if (x0 >= v) return;
    const int nv = (int)min((size_t)4, v - x0);
    const float INF = __uint_as_float(0x7f800000u);
    // branch-free scan: (best, index) with torch.argmin's comparison, the runner-up as a plain minimum (+inf = none), and a flag for
    // entries in the denormal zone (ATen's divisions may have rounded those to zero)
    float best[4], second[4];
    int bi[4];
    bool tiny[4] = {false, false, false, false};
    const float* p = P.ssdu + (size_t)k0 * v + x0;
    auto load4 = [&](const float* q, float (&c4)[4]) {
        if (vec4) { const float4 t = *reinterpret_cast<const float4*>(q); c4[0] = t.x; c4[1] = t.y; c4[2] = t.z; c4[3] = t.w; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[j] = j < nv ? q[j] : 0.0f;
        }
    };
    {
        float c4[4];
        load4(p, c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { best[j] = c4[j]; bi[j] = k0; second[j] = INF; tiny[j] = (c4[j] > 0.0f) & (c4[j] < 1.0e-36f); }
        p += v;
    }
#pragma unroll 8
    for (int k = k0 + 1; k < k1; ++k, p += v) {
        float c4[4];
        load4(p, c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float c = c4[j];
            tiny[j] = tiny[j] | ((c > 0.0f) & (c < 1.0e-36f));
            const bool lt = argmin_better(c, best[j]);
            second[j] = fminf(second[j], lt ? best[j] : c);       // (fminf drops a NaN: a column with a NaN is decided by its first NaN anyway)
            best[j] = lt ? c : best[j];
            bi[j] = lt ? k : bi[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= nv) continue;
        const unsigned long long mine = pack_min_key(best[j], (unsigned)bi[j]);
        const unsigned long long old = atomicMin(&P.key[x0 + j], mine);
        unsigned cand = second[j] == INF ? 0xffffffffu : ordered_bits(second[j]);
        if (old != ~0ull) {
            const unsigned loser = (unsigned)((old < mine ? mine : old) >> 32);
            cand = loser < cand ? loser : cand;
        }
        if (tiny[j]) cand = 0u;                     // sends the voxel to the exact evaluator
        if (cand != 0xffffffffu) atomicMin(&P.sec[x0 + j], cand);
    }
}

// certain winners -> idx0 / smin; the rest -> list
__global__ __launch_bounds__(256) void k_cert_plain_finalize(CertArgs A) {
    const CertProb& P = A.p[blockIdx.y];
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= v) return;
    const unsigned long long key = P.key[x];
    const unsigned b1 = (unsigned)(key >> 32), b2 = P.sec[x];
    const int k1 = (int)(unsigned)(key & 0xffffffffull);
    const float s1 = b1 == 0u ? __uint_as_float(0x7fc00000u) : from_ordered_bits(b1);
    P.smin[x] = s1;
    P.idx0[x] = k1;
    if (P.argmin_out) P.argmin_out[x] = k1;
    // first NaN / single displacement / exact zero (the first of the zeros; no entry of the column in the denormal zone) / a clear runner-up
    bool certain = b1 == 0u || b2 == 0xffffffffu;
    if (!certain && b2 != 0u) certain = s1 == 0.0f || cert_lower(from_ordered_bits(b2)) > cert_upper(s1);
    if (!certain) P.list[atomicAdd(&P.counts[0], 1)] = (unsigned)x;
}

// flagged voxels of the plain pass.  The smallest upper bound is the one of the minimum (the key); every entry whose lower bound does not exceed it
// must be evaluated exactly, and the first minimum of the exact values wins.  Three launches:
//   k_cert_plain_resolve  one wavefront per flagged voxel scans the column (1 024 entries at a time, 16 loads per lane in flight) and APPENDS its
//                         candidates to a work list; exact zeros need no evaluation (the first of them is a candidate at cost 0)
//   k_cert_plain_eval     one wavefront per work item: the exact entry, a 64-bit atomicMin on (value, index) of its voxel
//   k_cert_plain_commit   winners of the flagged voxels
// (round 6, first version: the scanning wavefront evaluated its candidates itself, one after the other -- a boundary voxel of a masked image whose
// column holds 720 IDENTICAL tiny non-zero sums (every displacement into the flat background of the other image) took 1.8 ms, the stage 4.3 ms.)
__global__ __launch_bounds__(64) void k_cert_plain_resolve(CertArgs A) {
    __shared__ float sm[160];
    const CertProb& P = A.p[blockIdx.y];
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    const int K = A.g.K, lane = threadIdx.x;
    const int cnt = P.counts[0];
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const unsigned x = P.list[e];
        const float* col = P.ssdu + x;
        const float U = cert_upper(P.smin[x]);
        unsigned long long bestkey = ~0ull;
        for (int kb = 0; kb < K; kb += 1024) {
            float s[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int k = kb + 64 * i + lane; s[i] = k < K ? col[(size_t)k * v] : __uint_as_float(0x7f800000u); }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // exact zeros: only the first of them can win (they all cost 0) -- no loop over a zero background's thousands of entries
                const unsigned long long mz = __ballot(s[i] == 0.0f);
                if (mz) {
                    const unsigned long long key = pack_min_key(0.0f, (unsigned)(kb + 64 * i + __ffsll((long long)mz) - 1));
                    bestkey = key < bestkey ? key : bestkey;
                }
                const bool mine = s[i] != 0.0f && cert_lower(s[i]) <= U;
                const unsigned long long m = __ballot(mine);
                if (m) {
                    const int nm = __popcll(m);
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&P.counts[8], nm);
                    base = __shfl(base, 0);
                    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (mine && (unsigned)slot < P.work_cap) P.work[slot] = ((unsigned long long)x << 32) | (unsigned)(kb + 64 * i + lane);
                    // a full list (cannot happen with the capacity of the carve unless thousands of voxels tie): the rest is evaluated here
                    unsigned long long rest = __ballot(mine && (unsigned)slot >= P.work_cap);
                    while (rest) {
                        const int l = __ffsll((long long)rest) - 1;
                        rest &= rest - 1;
                        const int kk = kb + 64 * i + l;
                        const float ex = cert_exact_entry(A.g, P, kk, (int)x, sm, lane);
                        const unsigned long long key = pack_min_key(ex, (unsigned)kk);
                        bestkey = key < bestkey ? key : bestkey;
                    }
                }
            }
        }
        if (lane == 0) P.key[x] = bestkey;             // (the plain pass's key is spent: its value and index live in smin / idx0)
    }
}
__global__ __launch_bounds__(256) void k_cert_plain_eval(CertArgs A) {
    __shared__ float smem[4][160];
    const CertProb& P = A.p[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = min(P.counts[8], (int)P.work_cap);
    for (int e = blockIdx.x * 4 + wv; e < n; e += gridDim.x * 4) {
        const unsigned long long it = P.work[e];
        const unsigned x = (unsigned)(it >> 32);
        const int kk = (int)(unsigned)(it & 0xffffffffull);
        const float ex = cert_exact_entry_wave(A.g, P, kk, (int)x, smem[wv], lane);
        if (lane == 0) atomicMin(&P.key[x], pack_min_key(ex, (unsigned)kk));
    }
}
__global__ __launch_bounds__(256) void k_cert_plain_commit(CertArgs A) {
    const CertProb& P = A.p[blockIdx.y];
    const int cnt = P.counts[0];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += gridDim.x * blockDim.x) {
        const unsigned x = P.list[e];
        const int kw = (int)(unsigned)(P.key[x] & 0xffffffffull);
        P.idx0[x] = kw;
        if (P.argmin_out) P.argmin_out[x] = kw;
    }
}

// ---- coupled passes --------------------------------------------------------------------------------------------------------------
// The pipeline's search mesh is a product grid (cvx_disp_mesh_f32: component a of displacement k depends on the a-th digit of
// k = (ia * n + ib) * n + ic only), so the kernels of the coupled passes keep its 3 x n axis values in LDS and never wait for a mesh load: one
// dependent memory round trip less in every stage of these latency-bound kernels.  Same values, same operations -- same bits.
struct CertMesh {
    const float* ax;        // LDS [3][32]: ax[a * 32 + i] = mesh[a][i * n^a]
    int n; float rn;
    __device__ __forceinline__ void split(int k, int& ic, int& ib, int& ia) const {
        // k < 2^15, n < 2^10: (k + 0.5) / n stays 0.5 / n away from every integer -- far more than the rounding of the float product
        const int q = (int)(((float)k + 0.5f) * rn);
        ic = k - q * n;
        ia = (int)(((float)q + 0.5f) * rn);
        ib = q - ia * n;
    }
};
__device__ __forceinline__ CertMesh cert_mesh_load(float* lds96, const float* __restrict__ mesh, int K, int n) {
    const int t = threadIdx.x;
    if (t < n) { lds96[t] = mesh[t]; lds96[32 + t] = mesh[K + t * n]; lds96[64 + t] = mesh[2 * K + t * n * n]; }
    __syncthreads();
    return CertMesh{lds96, n, 1.0f / (float)n};
}

// u_a(x) = avg_pool3d(mesh[a, idx], 3, padding=1)(x): raster-order sum of the in-range taps, / 27  (convex_adam_utils.py:96,107)
__device__ __forceinline__ void cert_smooth_winner(const int* __restrict__ idx, const CertMesh& M, int h, int w, int d, size_t i,
                                                   float& o0, float& o1, float& o2) {
    const int x = (int)(i % d), y = (int)((i / d) % w), z = (int)(i / ((size_t)d * w));
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int a = -1; a <= 1; ++a) {
        const int za = z + a;
        const bool zok = za >= 0 && za < h;
        const int zc = zok ? za : z;
        int kk[9];
        bool ok[9];
#pragma unroll
        for (int b = -1; b <= 1; ++b)
#pragma unroll
            for (int c = -1; c <= 1; ++c) {
                const int yb = y + b, xc = x + c;
                const bool in = zok && yb >= 0 && yb < w && xc >= 0 && xc < d;
                const int t = (b + 1) * 3 + (c + 1);
                ok[t] = in;
                kk[t] = idx[((size_t)zc * w + (in ? yb : y)) * d + (in ? xc : x)];
            }
        float m0[9], m1[9], m2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int ic, ib, ia;
            M.split(kk[t], ic, ib, ia);
            m0[t] = M.ax[ic]; m1[t] = M.ax[32 + ib]; m2[t] = M.ax[64 + ia];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            s0 += ok[t] ? m0[t] : 0.0f;
            s1 += ok[t] ? m1[t] : 0.0f;
            s2 += ok[t] ? m2[t] : 0.0f;
        }
    }
    o0 = fdiv(s0, 27.0f);
    o1 = fdiv(s1, 27.0f);
    o2 = fdiv(s2, 27.0f);
}

__device__ __forceinline__ float cert_pen(const CertMesh& M, int k, float uc, float ub, float ua, float coef) {
    int ic, ib, ia;
    M.split(k, ic, ib, ia);
    const float e0 = M.ax[ic] - uc, e1 = M.ax[32 + ib] - ub, e2 = M.ax[64 + ia] - ua;
    float q = e0 * e0;          // (..).pow(2).sum(0): sequential over the 3 components
    q += e1 * e1;
    q += e2 * e2;
    return coef * q;            // coeffs[j]*(...)                                                      (:104)
}

struct CertBox {
    float uc, ub, ua, slo, bound;
    int c_lo, c_hi, b_lo, b_hi, a_lo, a_hi;
    long long vol;
    bool degenerate;
};
// the admissible box: every displacement whose lower cost bound fl(slo + pen) does not exceed `bound` (an upper bound of the minimum)
__device__ __forceinline__ CertBox cert_box(const CertMesh& M, float uc, float ub, float ua, float coef, int K, int n, int kp,
                                            float ssdu_kp, float smin_u, const float* __restrict__ col, size_t v, int refine_above) {
    CertBox c;
    c.uc = uc; c.ub = ub; c.ua = ua;
    c.slo = cert_lower(smin_u);
    c.bound = cert_upper(ssdu_kp) + cert_pen(M, kp, uc, ub, ua, coef);
    const float hwf = (float)((n - 1) / 2);
    auto close_box = [&]() {
        const float qmax = fdiv((c.bound - c.slo) + fabsf(c.bound) * 2.384185791015625e-07f, coef) * 1.00001f;
        const float R = fsqrt(fmaxf(qmax, 0.0f)) * 1.00001f + 1.0e-4f;
        c.c_lo = max((int)ceilf(c.uc - R + hwf), 0); c.c_hi = min((int)floorf(c.uc + R + hwf), n - 1);
        c.b_lo = max((int)ceilf(c.ub - R + hwf), 0); c.b_hi = min((int)floorf(c.ub + R + hwf), n - 1);
        c.a_lo = max((int)ceilf(c.ua - R + hwf), 0); c.a_hi = min((int)floorf(c.ua + R + hwf), n - 1);
        c.vol = (long long)max(c.c_hi - c.c_lo + 1, 0) * max(c.b_hi - c.b_lo + 1, 0) * max(c.a_hi - c.a_lo + 1, 0);
        c.degenerate = !(coef > 0.0f) || !(R == R) || c.vol <= 0;
    };
    close_box();
    if (!c.degenerate && c.vol > refine_above && uc == uc && ub == ub && ua == ua) {
        const int kn = (min(max((int)rintf(ua + hwf), 0), n - 1) * n + min(max((int)rintf(ub + hwf), 0), n - 1)) * n + min(max((int)rintf(uc + hwf), 0), n - 1);
        if (kn != kp) {
            const float near_hi = cert_upper(col[(size_t)kn * v]) + cert_pen(M, kn, uc, ub, ua, coef);
            if (near_hi < c.bound) { c.bound = near_hi; close_box(); }
        }
    }
    if (c.degenerate) { c.c_lo = c.b_lo = c.a_lo = 0; c.c_hi = c.b_hi = c.a_hi = n - 1; c.vol = (long long)n * n * n; }
    return c;
}

// one thread per voxel: smoothing of the previous winners, the admissible box, boxes of at most 8 displacements decided here
__global__ __launch_bounds__(64) void k_cert_voxel(CertArgs A, int refine) {
    const CertProb& P = A.p[blockIdx.y];
    const int h = A.g.h, w = A.g.w, d = A.g.d, K = A.g.K, n = A.g.n;
    const size_t v = (size_t)h * w * d;
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float axes[96];
    const CertMesh M = cert_mesh_load(axes, A.mesh, K, n);
    if (x >= v) return;
    const int* prev = A.pass == 1 ? P.idx0 : ((A.pass & 1) ? P.idxB : P.idxA);
    int* next = (A.pass & 1) ? P.idxA : P.idxB;
    const int kp = prev[x];
    const float s_kp = P.ssdu[(size_t)kp * v + x];
    const float sm_x = P.smin[x];
    float uc, ub, ua;
    cert_smooth_winner(prev, M, h, w, d, x, uc, ub, ua);
    P.u[x] = uc; P.u[v + x] = ub; P.u[2 * v + x] = ua;
    // a column with a NaN: torch.argmin returns the first NaN in every pass (the penalty is finite) = the plain pass's winner
    if (sm_x != sm_x) { next[x] = P.idx0[x]; return; }
    const CertBox c = cert_box(M, uc, ub, ua, A.coef, K, n, kp, s_kp, sm_x, P.ssdu + x, v, refine);
    constexpr int NB = 8;
    const unsigned boxpack = (unsigned)c.c_lo | ((unsigned)c.c_hi << 5) | ((unsigned)c.b_lo << 10) | ((unsigned)c.b_hi << 15) | ((unsigned)c.a_lo << 20) | ((unsigned)c.a_hi << 25) | (c.degenerate ? 1u << 30 : 0u);
    const CertRec rec = {(unsigned)x, boxpack, kp, c.bound, c.slo, c.uc, c.ub, c.ua};
    if (c.vol > NB) { P.rec[atomicAdd(&P.counts[A.pass], 1)] = rec; return; }
    int kk[NB];
    float pen[NB], val[NB];
    bool need[NB];
    {
        int ic = c.c_lo, ib = c.b_lo, ia = c.a_lo;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool in = j < (int)c.vol;
            kk[j] = (ia * n + ib) * n + ic;
            pen[j] = cert_pen(M, kk[j], c.uc, c.ub, c.ua, A.coef);
            need[j] = in && !(c.slo + pen[j] > c.bound);
            if (j + 1 < (int)c.vol) {
                if (++ic > c.c_hi) { ic = c.c_lo; if (++ib > c.b_hi) { ib = c.b_lo; ++ia; } }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) val[j] = need[j] ? P.ssdu[(size_t)kk[j] * v + x] : 0.0f;
    float lo1 = 0.f, lo2 = 0.f, U = __uint_as_float(0x7f800000u);
    int k1 = -1, cnt = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (!need[j]) continue;
        const float lo = cert_lower(val[j]) + pen[j], hi = cert_upper(val[j]) + pen[j];
        U = fminf(U, hi);
        if (cnt == 0) { lo1 = lo; k1 = kk[j]; }
        else if (lo < lo1) { lo2 = lo1; lo1 = lo; k1 = kk[j]; }
        else if (cnt == 1 || lo < lo2) lo2 = lo;
        ++cnt;
    }
    if (cnt == 0) { next[x] = kp; return; }         // cannot happen (kp passes its own test); keeps the output defined
    if (cnt == 1 || lo2 > U) { next[x] = k1; return; }
    P.rec[atomicAdd(&P.counts[A.pass], 1)] = rec;
}

// one wavefront per listed voxel: scan of the admissible box (512 displacements per round, 8 loads per lane in flight), exact
// evaluation of what the intervals leave open
__global__ __launch_bounds__(256) void k_cert_wave(CertArgs A) {
    __shared__ float smem[4][160];
    __shared__ float axes[96];
    const CertProb& P = A.p[blockIdx.y];
    const int h = A.g.h, w = A.g.w, d = A.g.d, K = A.g.K, n = A.g.n;
    const size_t v = (size_t)h * w * d;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* sm = smem[wv];
    const CertMesh M = cert_mesh_load(axes, A.mesh, K, n);
    int* next = (A.pass & 1) ? P.idxA : P.idxB;
    const int cnt = P.counts[A.pass];
    const float INF = __uint_as_float(0x7f800000u);
    for (int e = blockIdx.x * 4 + wv; e < cnt; e += gridDim.x * 4) {
        const CertRec r = P.rec[e];
        const size_t x = r.x;
        const int kp = r.kp;
        const float* col = P.ssdu + x;
        CertBox c;
        c.uc = r.uc; c.ub = r.ub; c.ua = r.ua; c.slo = r.slo; c.bound = r.bound;
        c.c_lo = r.box & 31; c.c_hi = (r.box >> 5) & 31; c.b_lo = (r.box >> 10) & 31; c.b_hi = (r.box >> 15) & 31; c.a_lo = (r.box >> 20) & 31; c.a_hi = (r.box >> 25) & 31;
        c.degenerate = ((r.box >> 30) & 1u) != 0;
        c.vol = (long long)(c.c_hi - c.c_lo + 1) * (c.b_hi - c.b_lo + 1) * (c.a_hi - c.a_lo + 1);
        const int nc = c.c_hi - c.c_lo + 1, nb = c.b_hi - c.b_lo + 1, ncb = nc * nb;
        const float rc = 1.0f / (float)nc, rab = 1.0f / (float)ncb;
        // round 1 over the box: smallest lo (first index among equals), runner-up lo, smallest hi.  All loads of a round
        // -- of 768 per round -- are in flight together (slots beyond the box are skipped wave-uniformly)
        unsigned long long key = ~0ull;
        float lo2 = INF, U = INF;
        constexpr int NL = 12;
        for (long long i0 = 0; i0 < c.vol; i0 += 64 * NL) {
            int kk[NL];
            float pen[NL], val[NL];
            bool need[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                need[j] = false; kk[j] = 0; pen[j] = 0.0f;
                if (i0 + 64 * j < c.vol) {                                     // (uniform)
                    const int i = (int)i0 + lane + 64 * j;
                    need[j] = i < (int)c.vol;
                    const int ii = need[j] ? i : 0;
                    // box position of flat index ii without integer division: ii < 2^15 and the divisors are below 2^10, so (ii + 0.5) / divisor
                    // stays 0.5 / divisor away from every integer -- far more than the rounding of the float product
                    const int qa = (int)(((float)ii + 0.5f) * rab), ra = ii - qa * ncb;
                    const int qb = (int)(((float)ra + 0.5f) * rc);
                    const int ic = c.c_lo + ra - qb * nc, ib = c.b_lo + qb, ia = c.a_lo + qa;
                    kk[j] = (ia * n + ib) * n + ic;
                    pen[j] = cert_pen(M, kk[j], c.uc, c.ub, c.ua, A.coef);
                    need[j] = need[j] && (c.degenerate || !(c.slo + pen[j] > c.bound));
                }
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) val[j] = need[j] ? col[(size_t)kk[j] * v] : 0.0f;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                if (!need[j]) continue;
                const float lo = cert_lower(val[j]) + pen[j], hi = cert_upper(val[j]) + pen[j];
                U = fminf(U, hi);
                const unsigned long long kx = pack_min_key(lo, (unsigned)kk[j]);
                if (kx < key) { if (key != ~0ull) lo2 = fminf(lo2, from_ordered_bits((unsigned)(key >> 32))); key = kx; }
                else lo2 = fminf(lo2, lo);
            }
        }
        unsigned long long gkey = key;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(gkey, o); gkey = t < gkey ? t : gkey; }
        float mine2 = key == gkey ? lo2 : fminf(lo2, key != ~0ull ? from_ordered_bits((unsigned)(key >> 32)) : INF);
        for (int o = 32; o > 0; o >>= 1) { mine2 = fminf(mine2, __shfl_xor(mine2, o)); U = fminf(U, __shfl_xor(U, o)); }
        if (gkey == ~0ull) { if (lane == 0) next[x] = kp; continue; }             // (cannot happen)
        if (mine2 > U) { if (lane == 0) next[x] = (int)(unsigned)(gkey & 0xffffffffull); continue; }
        // round 2: the candidates lo <= U, exactly
        if (lane == 0) atomicAdd(&P.counts[7], 1);                               // (statistics: decisions that needed the exact evaluator)
        unsigned long long bestkey = ~0ull;
        for (long long i0 = 0; i0 < c.vol; i0 += 64 * NL) {
            int kk[NL];
            float pen[NL], val[NL];
            bool need[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                need[j] = false; kk[j] = 0; pen[j] = 0.0f;
                if (i0 + 64 * j < c.vol) {
                    const int i = (int)i0 + lane + 64 * j;
                    need[j] = i < (int)c.vol;
                    const int ii = need[j] ? i : 0;
                    const int qa = (int)(((float)ii + 0.5f) * rab), ra = ii - qa * ncb;
                    const int qb = (int)(((float)ra + 0.5f) * rc);
                    kk[j] = ((c.a_lo + qa) * n + c.b_lo + qb) * n + c.c_lo + ra - qb * nc;
                    pen[j] = cert_pen(M, kk[j], c.uc, c.ub, c.ua, A.coef);
                    need[j] = need[j] && (c.degenerate || !(c.slo + pen[j] > c.bound));
                }
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) val[j] = need[j] ? col[(size_t)kk[j] * v] : 0.0f;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                if (!(i0 + 64 * j < c.vol)) continue;
                unsigned long long m = __ballot(need[j] && cert_lower(val[j]) + pen[j] <= U);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int k2 = __shfl(kk[j], l);
                    const float sk = __shfl(val[j], l), pk = __shfl(pen[j], l);
                    const float ex = sk == 0.0f ? 0.0f : cert_exact_entry_wave(A.g, P, k2, (int)x, sm, lane);
                    const unsigned long long kx = pack_min_key(ex + pk, (unsigned)k2);      // ssd + coeffs[j]*(...)
                    bestkey = kx < bestkey ? kx : bestkey;
                }
            }
        }
        if (lane == 0) next[x] = (int)(unsigned)(bestkey & 0xffffffffull);
    }
}

__global__ __launch_bounds__(256) void k_cert_gather(CertArgs A) {
    const CertProb& P = A.p[blockIdx.y];
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float axes[96];
    const CertMesh M = cert_mesh_load(axes, A.mesh, A.g.K, A.g.n);
    if (i >= v) return;
    const int* prev = A.pass == 1 ? P.idx0 : ((A.pass & 1) ? P.idxB : P.idxA);
    float o0, o1, o2;
    cert_smooth_winner(prev, M, A.g.h, A.g.w, A.g.d, i, o0, o1, o2);
    P.u[i] = o0; P.u[v + i] = o1; P.u[2 * v + i] = o2;
}

__global__ __launch_bounds__(256) void k_cert_arm(CertArgs A, int nprob) {
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int q = 0; q < nprob; ++q) {
        if (i < v) { A.p[q].key[i] = ~0ull; A.p[q].sec[i] = 0xffffffffu; }
        if (i < 16) A.p[q].counts[i] = 0;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
void launch_corr_tail_compact(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int sad, float* tail, hipStream_t s);

// plain = the plain argmin only (cvx_correlate_ex_f32 with fast = 2): keys, runner-up, winners, minimum, list -- 24 bytes per voxel;
// the coupled passes add the two ping-pong winner arrays and the work records
// capacity of the plain pass's work list: four candidates per voxel on average, at least 64 K (a column holds n^3 entries)
static size_t cert_work_cap(size_t v) { return 4 * v + 65536; }
size_t corr_certify_workspace_bytes(int C, int h, int w, int d, int hw, bool plain) {
    (void)C;
    const size_t v = (size_t)h * w * d;
    const int n = 2 * hw + 1;
    size_t used = 0;
    used = carve_size(used, sizeof(unsigned long long) * v);      // key
    used = carve_size(used, sizeof(unsigned) * v);                // sec
    used = carve_size(used, sizeof(int) * v);                     // idx0
    used = carve_size(used, sizeof(float) * v);                   // smin
    used = carve_size(used, sizeof(unsigned) * v);                // list
    used = carve_size(used, sizeof(int) * 16);                    // counts
    used = carve_size(used, sizeof(unsigned long long) * cert_work_cap(v));   // work list of the plain pass
    used = carve_size(used, sizeof(float) * 32 * n);              // tail
    if (!plain) {
        for (int i = 0; i < 2; ++i) used = carve_size(used, sizeof(int) * v);   // idxA, idxB
        used = carve_size(used, sizeof(CertRec) * v);             // records
    }
    return used + 256;
}

struct CertCarve { unsigned long long* key; unsigned* sec; int* idx0; int* idxA; int* idxB; float* smin; unsigned* list; CertRec* rec; int* counts; float* tail; unsigned long long* work; unsigned work_cap; };
static CertCarve cert_carve(void* workspace, size_t workspace_bytes, int h, int w, int d, int hw, bool plain = false) {
    const size_t v = (size_t)h * w * d;
    const int n = 2 * hw + 1;
    Carver cv(workspace, workspace_bytes);
    CertCarve c{};
    c.key = cv.take<unsigned long long>(v);
    c.sec = cv.take<unsigned>(v);
    c.idx0 = cv.take<int>(v);
    c.smin = cv.take<float>(v);
    c.list = cv.take<unsigned>(v);
    c.counts = cv.take<int>(16);
    c.work_cap = (unsigned)cert_work_cap(v);
    c.work = cv.take<unsigned long long>(c.work_cap);
    c.tail = cv.take<float>((size_t)32 * n);
    if (!plain) {
        c.idxA = cv.take<int>(v); c.idxB = cv.take<int>(v);
        c.rec = cv.take<CertRec>(v);
    }
    return c;
}
static CertGeo cert_geo(int C, int h, int w, int d, int hw) {
    CertGeo g;
    g.C = C; g.h = h; g.w = w; g.d = d; g.hw = hw; g.n = 2 * hw + 1; g.K = g.n * g.n * g.n;
    const long long ncols = (long long)h * g.n * g.n * w * d;
    g.tail_from = (ncols / 32) * 32;
    g.ntail = (int)(ncols - g.tail_from);
    return g;
}
static CertProb cert_prob(const float* ssdu, const float* fix, const float* mov, const CertCarve& c, float* u, int64_t* argmin_out) {
    return CertProb{ssdu, fix, mov, c.tail, c.key, c.sec, c.idx0, c.idxA, c.idxB, c.smin, c.list, c.counts, c.work, c.work_cap, c.rec, u, argmin_out};
}

// the plain argmin in three parts so that a caller can stream each volume while it is still in the Infinity Cache (right behind its
// correlation kernel): arm (keys, counters, the interleaved-order tail values), stream (one problem), finish (certify + resolve)
static void cert_arm(CertArgs& A, int nprob, hipStream_t s) {
    const CertGeo& g = A.g;
    const size_t v = (size_t)g.h * g.w * g.d;
    hipLaunchKernelGGL(k_cert_arm, dim3((unsigned)cdiv64((int64_t)(v > 16 ? v : 16), 256)), dim3(256), 0, s, A, nprob);
    for (int q = 0; q < nprob; ++q)
        if (g.ntail > 0) launch_corr_tail_compact(A.p[q].fix, A.p[q].mov, g.C, g.h, g.w, g.d, g.hw, 0, const_cast<float*>(A.p[q].tail), s);
}
// problems [q0, q0 + nq)
static void cert_stream(CertArgs A, int q0, int nq, hipStream_t s) {
    const CertGeo& g = A.g;
    const size_t v = (size_t)g.h * g.w * g.d;
    const int K = g.K;
    if (q0) { A.p[0] = A.p[q0]; }
    bool vec4 = v % 4 == 0;
    for (int q = 0; q < nq; ++q) vec4 = vec4 && (reinterpret_cast<uintptr_t>(A.p[q].ssdu) & 15) == 0;
    const int xb = (int)cdiv64((int64_t)cdiv64((int64_t)v, 4), 256);
    int nslices = cdiv(512, xb);
    if (nslices > K) nslices = K;
    if (nslices < 1) nslices = 1;
    const int kslice = cdiv(K, nslices);
    nslices = cdiv(K, kslice);
    hipLaunchKernelGGL(k_cert_plain_stream, dim3(xb, nslices, nq), dim3(256), 0, s, A, kslice, vec4 ? 1 : 0);
}
static int cert_finish(CertArgs& A, int nprob, hipStream_t s) {
    const size_t v = (size_t)A.g.h * A.g.w * A.g.d;
    hipLaunchKernelGGL(k_cert_plain_finalize, dim3((unsigned)cdiv64((int64_t)v, 256), nprob), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_cert_plain_resolve, dim3(256, nprob), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_cert_plain_eval, dim3(256, nprob), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_cert_plain_commit, dim3(8, nprob), dim3(256), 0, s, A);
    return check_last("certified argmin");
}
static int cert_plain(CertArgs& A, int nprob, bool arm, hipStream_t s) {
    if (arm) cert_arm(A, nprob, s);
    cert_stream(A, 0, nprob, s);
    return cert_finish(A, nprob, s);
}

int corr_certified_argmin(const float* ssdu, const float* fix, const float* mov, int C, int h, int w, int d, int hw, int64_t* argmin,
                          void* workspace, size_t workspace_bytes, hipStream_t s) {
    if (workspace_bytes < corr_certify_workspace_bytes(C, h, w, d, hw, true)) return fail(CVX_ERR_WORKSPACE, "certified argmin: workspace too small");
    const CertCarve c = cert_carve(workspace, workspace_bytes, h, w, d, hw, true);
    CertArgs A{};
    A.g = cert_geo(C, h, w, d, hw);
    A.p[0] = cert_prob(ssdu, fix, mov, c, nullptr, argmin);
    A.p[1] = A.p[0];
    return cert_plain(A, 1, true, s);
}

// both directions (ssduB == nullptr: one) from the certified-fast volumes to the smoothed fields: plain argmin + six coupled passes
// stage: 0 = everything; or in sequence 1 = arm, 2 = stream the first volume, 3 = stream the second, 4 = certify + resolve the plain argmin,
// 5 = the coupled passes (the pipeline interleaves 2 / 3 with the correlation kernels: each volume is streamed from the Infinity Cache)
int coupled_convex_cert_impl(const float* ssduA, const float* fixA, const float* movA, float* outA, void* wsA, const float* ssduB, const float* fixB,
                             const float* movB, float* outB, void* wsB, const float* mesh, int C, int h, int w, int d, int hw, size_t workspace_bytes,
                             hipStream_t s, int stage) {
    if (workspace_bytes < corr_certify_workspace_bytes(C, h, w, d, hw, false)) return fail(CVX_ERR_WORKSPACE, "certified coupled convex: workspace too small");
    const int nprob = ssduB ? 2 : 1;
    CertArgs A{};
    A.g = cert_geo(C, h, w, d, hw);
    A.mesh = mesh;
    const CertCarve ca = cert_carve(wsA, workspace_bytes, h, w, d, hw);
    A.p[0] = cert_prob(ssduA, fixA, movA, ca, outA, nullptr);
    A.p[1] = A.p[0];
    if (ssduB) { const CertCarve cb = cert_carve(wsB, workspace_bytes, h, w, d, hw); A.p[1] = cert_prob(ssduB, fixB, movB, cb, outB, nullptr); }
    if (stage == 1) { cert_arm(A, nprob, s); return check_last("certified argmin"); }
    if (stage == 2) { cert_stream(A, 0, 1, s); return check_last("certified argmin"); }
    if (stage == 3) { if (nprob > 1) cert_stream(A, 1, 1, s); return check_last("certified argmin"); }
    if (stage == 4) return cert_finish(A, nprob, s);
    if (stage == 0) { const int rc = cert_plain(A, nprob, true, s); if (rc) return rc; }
    const size_t v = (size_t)h * w * d;
    const int refine = options().prune_refine != 0 ? 8 : 0x7fffffff;
    static const float coeffs[6] = {0.003f, 0.01f, 0.03f, 0.1f, 0.3f, 1.0f};      // torch.tensor([...]) float32 (:98)
    for (int it = 0; it < 6; ++it) {
        A.coef = coeffs[it]; A.pass = it + 1;
        hipLaunchKernelGGL(k_cert_voxel, dim3((unsigned)cdiv64((int64_t)v, 64), nprob), dim3(64), 0, s, A, refine);
        hipLaunchKernelGGL(k_cert_wave, dim3(512, nprob), dim3(256), 0, s, A);
    }
    if (getenv("CVX_CERT_TRACE")) {                       // debugging aid (synchronises): how many voxels each pass sent to the wavefront kernel
        for (int q = 0; q < nprob; ++q) {
            int cnt[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(cnt, A.p[q].counts, sizeof(cnt), hipMemcpyDeviceToHost);
            fprintf(stderr, "cert trace: problem %d: plain flagged %d; listed per coupled pass %d %d %d %d %d %d, %d of them evaluated exactly (of %zu voxels)\n", q, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7], v);
        }
    }
    A.pass = 7;
    hipLaunchKernelGGL(k_cert_gather, dim3((unsigned)cdiv64((int64_t)v, 256), nprob), dim3(256), 0, s, A);
    return check_last("certified coupled convex");
}

}  // namespace cvx
