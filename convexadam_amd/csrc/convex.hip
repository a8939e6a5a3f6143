// convex.hip -- argmin over the search window, coupled-convex regularisation and inverse consistency.
//
// References: torch.argmin(ssd, 0) convex_adam_utils.py:87; coupled_convex :93-109;
// inverse_consistency :114-129.
//
// k_argmin streams the cost volume once ([K][v] float32, v contiguous): the grid is
// (v/256) x (K-slices); each thread walks its K-slice for one voxel (coalesced across the wave),
// keeps the first minimum, and merges slices with one 64-bit atomicMin on an order-preserving
// (cost bits << 32 | k) key -- ties resolve to the lowest k, like ATen's CPU argmin.
// The coupled variant adds coef * sum_a (mesh[a,k] - u[a,x])^2 in the reference's evaluation order.
// HBM-bound: K*v*4 bytes per pass, 1 + 6 passes per direction.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "cvx_common.h"

namespace cvx {

// Element type of the cost volume: float32, or half precision (fp16 STORAGE of the reference's GPU default, convex_adam_MIND.py:79,
// SURVEY 8(f).4): values are widened to float32 on load -- exact -- and every comparison / sum runs in float32 as before.
template <typename ST> struct SsdIO;
template <> struct SsdIO<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
};
template <> struct SsdIO<__half> {
    static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ float4 ld4(const __half* p) {                      // one 8-byte load
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        const h16x4 r = *reinterpret_cast<const h16x4*>(p);
        return make_float4((float)r.x, (float)r.y, (float)r.z, (float)r.w);
    }
};

template <bool COUPLED, typename ST>
__global__ __launch_bounds__(256) void k_argmin(const ST* __restrict__ ssd, const float* __restrict__ mesh,
                                                const float* __restrict__ u, float coef, int K, size_t v, int kslice,
                                                unsigned long long* __restrict__ keys) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = blockIdx.y * kslice, k1 = min(k0 + kslice, K);
    if (x >= v) return;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f;
    if (COUPLED) { u0 = u[x]; u1 = u[v + x]; u2 = u[2 * v + x]; }
    float best = 0.f;
    int bi = -1;
    const ST* p = ssd + (size_t)k0 * v + x;
#pragma unroll 4
    for (int k = k0; k < k1; ++k, p += v) {
        float cost = SsdIO<ST>::ld(p);
        if (COUPLED) {
            const float e0 = mesh[k] - u0, e1 = mesh[K + k] - u1, e2 = mesh[2 * K + k] - u2;
            float q = e0 * e0;          // (..).pow(2).sum(0): sequential over the 3 components
            q += e1 * e1;
            q += e2 * e2;
            cost = cost + coef * q;     // ssd + coeffs[j]*(...)                       (:104)
        }
        if ((bi < 0) | argmin_better(cost, best)) { best = cost; bi = k; }
    }
    if (bi >= 0) atomicMin(&keys[x], pack_min_key(best, (unsigned)bi));
}

// Same pass with four consecutive voxels per thread (one 16-byte load per displacement plane, four loads in flight):
// 4 KB per wavefront in flight instead of 1 KB -- the 4-byte version is latency-bound at ~4 TB/s.  Needs v % 4 == 0.
template <bool COUPLED, typename ST>
__global__ __launch_bounds__(256) void k_argmin4(const ST* __restrict__ ssd, const float* __restrict__ mesh,
                                                 const float* __restrict__ u, float coef, int K, size_t v, int kslice,
                                                 unsigned long long* __restrict__ keys) {
    const size_t x = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int k0 = blockIdx.y * kslice, k1 = min(k0 + kslice, K);
    if (x >= v) return;
    float u0[4] = {0.f, 0.f, 0.f, 0.f}, u1[4] = {0.f, 0.f, 0.f, 0.f}, u2[4] = {0.f, 0.f, 0.f, 0.f};
    if (COUPLED) {
        const float4 a = *reinterpret_cast<const float4*>(u + x), b = *reinterpret_cast<const float4*>(u + v + x),
                     c = *reinterpret_cast<const float4*>(u + 2 * v + x);
        u0[0] = a.x; u0[1] = a.y; u0[2] = a.z; u0[3] = a.w;
        u1[0] = b.x; u1[1] = b.y; u1[2] = b.z; u1[3] = b.w;
        u2[0] = c.x; u2[1] = c.y; u2[2] = c.z; u2[3] = c.w;
    }
    float best[4] = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {-1, -1, -1, -1};
    const ST* p = ssd + (size_t)k0 * v + x;
#pragma unroll 4
    for (int k = k0; k < k1; ++k, p += v) {
        const float4 q4 = SsdIO<ST>::ld4(p);
        const float cst[4] = {q4.x, q4.y, q4.z, q4.w};
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        if (COUPLED) { m0 = mesh[k]; m1 = mesh[K + k]; m2 = mesh[2 * K + k]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cost = cst[j];
            if (COUPLED) {
                const float e0 = m0 - u0[j], e1 = m1 - u1[j], e2 = m2 - u2[j];
                float q = e0 * e0;      // (..).pow(2).sum(0): sequential over the 3 components
                q += e1 * e1;
                q += e2 * e2;
                cost = cost + coef * q; // ssd + coeffs[j]*(...)                       (:104)
            }
            if ((bi[j] < 0) | argmin_better(cost, best[j])) { best[j] = cost; bi[j] = k; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (bi[j] >= 0) atomicMin(&keys[x + j], pack_min_key(best[j], (unsigned)bi[j]));
}

// ---- coupled pass with exact pruning (branch and bound) ----------------------------------------------------------------------
// For voxel x let smin = min_k ssd[k,x] (known from the plain argmin) and B = the reference cost of ANY displacement under this
// pass's (coef, u) -- here the previous pass's winner.  Rounding is monotonic and ssd[k,x] >= smin, so
// cost_k = fl(ssd[k,x] + p_k) >= fl(smin + p_k) with p_k = fl(coef * q_k): a displacement with fl(smin + p_k) > B costs strictly
// more than B >= the minimum and can neither win nor tie -- its cost-volume entry is never read.  p_k needs no memory access.
// Only displacements inside the axis-aligned box around the admissible ball
//     |delta - u|^2 <= ((B - smin) + 2^-22 |B|) / coef * (1 + 1e-5)
// are visited (the margins make the box a superset of everything that passes the exact test; inside the box the exact test and
// the reference's cost expression decide).  After the first box filter u sits on the previous winner for most voxels: on the
// benchmark pair the median box holds ONE displacement, the mean 13 (coef 0.003) .. 1.1 (coef 1); a few voxels in flat cost
// regions keep the whole window.  Two kernels: k_argmin_voxel -- one thread per voxel for boxes of at most `limit`
// displacements, larger ones are appended to a list; k_argmin_wave -- one wavefront per listed voxel, the lanes scan the box
// in parallel and merge with a 64-bit min on the (cost, index) key (lowest index among equal costs, like ATen's argmin).
// The result is the reference's argmin bit for bit; the worst case (every voxel listed) degrades to a scan of the whole volume.
// mesh[0][k] belongs to the fastest index of k = (a*n + b)*n + c, mesh[2][k] to the slowest; delta_i ~= i - hw.
// u_a(x) = avg_pool3d(mesh[a, idx], 3, padding=1)(x): raster-order sum of the in-range taps of the 3^3 window, / 27
// (convex_adam_utils.py:96,107).  idx holds displacement indices (int32) or (cost, index) keys (low word = index).
template <typename IndexT>
__device__ __forceinline__ void smooth_winner(const IndexT* __restrict__ idx, const float* __restrict__ mesh, int K, int h, int w,
                                              int d, size_t i, float& o0, float& o1, float& o2) {
    const int x = (int)(i % d), y = (int)((i / d) % w), z = (int)(i / ((size_t)d * w));
    // Fully unrolled with clamped (always valid) addresses: the 9 winners of a plane are fetched together, then their 27 mesh
    // entries; taps outside the volume contribute an exact +0.0 (a partial sum that starts at +0.0 is never -0.0, so adding
    // +0.0 changes nothing) -- two memory round trips per plane instead of two per tap.
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
                kk[t] = (int)(unsigned)idx[((size_t)zc * w + (in ? yb : y)) * d + (in ? xc : x)];
            }
        float m0[9], m1[9], m2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { m0[t] = mesh[kk[t]]; m1[t] = mesh[K + kk[t]]; m2[t] = mesh[2 * K + kk[t]]; }
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

struct CandBox {
    float uc, ub, ua, sm, bound;
    int kp, c_lo, c_hi, b_lo, b_hi, a_lo, a_hi;
    long long vol;
    bool degenerate;
};
// kp = previous winner of the voxel (low 32 bits of its key), ssd_kp = ssd[kp, x], sm_x = smin[x]: fetched by the caller, who can
// issue these loads before the smoothing step instead of after it (one memory round trip less on the critical path).
// col = ssd + x (the voxel's column, stride v).  A box of more than `refine_above` displacements is tried again with the cost of the
// lattice point NEAREST to u as the bound (any displacement's cost is a valid bound): where many displacements tie at the minimum --
// zero background: every in-volume displacement of a background voxel costs 0 -- the previous winner is the FIRST of them and may lie
// far from u, while the nearest one costs the same and closes the ball to a few displacements.  One more load, only for boxes that
// would otherwise go to the list; both kernels evaluate the same function, so they see the same box.
template <typename ST>
__device__ __forceinline__ CandBox cand_box(const float* __restrict__ mesh, float uc, float ub, float ua, float coef, int K, int n,
                                            int kp, float ssd_kp, float sm_x, const ST* __restrict__ col, size_t v, int refine_above) {
    CandBox c;
    c.uc = uc; c.ub = ub; c.ua = ua; c.sm = sm_x;
    c.kp = kp;
    const float e0 = mesh[c.kp] - c.uc, e1 = mesh[K + c.kp] - c.ub, e2 = mesh[2 * K + c.kp] - c.ua;
    float q = e0 * e0;
    q += e1 * e1;
    q += e2 * e2;
    c.bound = ssd_kp + coef * q;                                        // the reference cost of the previous winner
    const float hwf = (float)((n - 1) / 2);
    auto close_box = [&]() {
        const float qmax = fdiv((c.bound - c.sm) + fabsf(c.bound) * 2.384185791015625e-07f, coef) * 1.00001f;
        const float R = fsqrt(fmaxf(qmax, 0.0f)) * 1.00001f + 1.0e-4f;
        c.c_lo = max((int)ceilf(c.uc - R + hwf), 0); c.c_hi = min((int)floorf(c.uc + R + hwf), n - 1);
        c.b_lo = max((int)ceilf(c.ub - R + hwf), 0); c.b_hi = min((int)floorf(c.ub + R + hwf), n - 1);
        c.a_lo = max((int)ceilf(c.ua - R + hwf), 0); c.a_hi = min((int)floorf(c.ua + R + hwf), n - 1);
        c.vol = (long long)max(c.c_hi - c.c_lo + 1, 0) * max(c.b_hi - c.b_lo + 1, 0) * max(c.a_hi - c.a_lo + 1, 0);
        c.degenerate = !(coef > 0.0f) || !(R == R) || c.vol <= 0;       // no usable bound: scan the whole window
    };
    close_box();
    if (!c.degenerate && c.vol > refine_above && uc == uc && ub == ub && ua == ua) {
        const int kn = (min(max((int)rintf(ua + hwf), 0), n - 1) * n + min(max((int)rintf(ub + hwf), 0), n - 1)) * n + min(max((int)rintf(uc + hwf), 0), n - 1);
        if (kn != kp) {
            const float f0 = mesh[kn] - c.uc, f1 = mesh[K + kn] - c.ub, f2 = mesh[2 * K + kn] - c.ua;
            float qn = f0 * f0;
            qn += f1 * f1;
            qn += f2 * f2;
            const float near_cost = SsdIO<ST>::ld(col + (size_t)kn * v) + coef * qn;          // the reference cost of the nearest displacement
            if (near_cost < c.bound) { c.bound = near_cost; c.kp = kn; close_box(); }
        }
    }
    if (c.degenerate) { c.c_lo = c.b_lo = c.a_lo = 0; c.c_hi = c.b_hi = c.a_hi = n - 1; c.vol = (long long)n * n * n; }
    return c;
}

// Two independent problems in one launch (the forward and the reverse direction of a pair): gridDim.y = 2, and the second
// problem's buffers are the first one's displaced by these byte offsets (all workspace-carved buffers share `ws`).
struct Prob2 { ptrdiff_t ssd, argmin, out, ws; };
template <typename T>
__device__ __forceinline__ T* shifted(T* p, ptrdiff_t bytes) {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes);
}

// (the kernel first evaluates u = box3(mesh[previous winners]) for its voxel -- the reference's smoothing step between two
// passes -- and stores it for the wavefront kernel and as the running result)
template <typename PrevT, typename ST>
__global__ __launch_bounds__(64) void k_argmin_voxel(const ST* __restrict__ ssd, const float* __restrict__ mesh,
                                                     float* __restrict__ u, float coef, int K, int n, int h, int w, int d,
                                                     const float* __restrict__ smin, const PrevT* __restrict__ kprev, int limit,
                                                     unsigned long long* __restrict__ list, int* __restrict__ list_count,
                                                     int* __restrict__ next_count, unsigned long long* __restrict__ keys,
                                                     const unsigned long long* __restrict__ minkeys, int refine, Prob2 o) {
    if (blockIdx.y) {
        ssd = shifted(ssd, o.ssd); u = shifted(u, o.out); smin = shifted(smin, o.ws); kprev = shifted(kprev, o.ws);
        list = shifted(list, o.ws); list_count = shifted(list_count, o.ws); next_count = shifted(next_count, o.ws);
        keys = shifted(keys, o.ws);
        if (minkeys) minkeys = shifted(minkeys, o.ws);
    }
    const size_t v = (size_t)h * w * d;
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x == 0) *next_count = 0;                            // list length of the NEXT pass (the two counters alternate)
    if (x >= v) return;
    // the voxel's own previous winner, its cost and the per-voxel minimum do not depend on the smoothing: loads issued first
    const int kp = (int)(unsigned)kprev[x];                 // low 32 bits of a key = displacement index
    const float ssd_kp = SsdIO<ST>::ld(ssd + (size_t)kp * v + x);
    const float sm_x = smin[x];
    float uc, ub, ua;
    smooth_winner(kprev, mesh, K, h, w, d, x, uc, ub, ua);
    u[x] = uc; u[v + x] = ub; u[2 * v + x] = ua;
    // a voxel whose cost column holds a NaN: torch.argmin returns the first NaN in every pass (the penalty is finite), i.e. the plain
    // argmin's winner, and stays it.  minkeys = the (cost, index) keys of the library's own minimum pass: on the public entry points the
    // caller's `argmin` only seeds the first smoothing step and need not be the first NaN
    if (sm_x != sm_x) { keys[x] = pack_min_key(sm_x, minkeys ? (unsigned)(minkeys[x] & 0xffffffffull) : (unsigned)kp); return; }
    const CandBox c = cand_box(mesh, uc, ub, ua, coef, K, n, kp, ssd_kp, sm_x, ssd + x, v, refine);
    if (c.vol > min(limit, 8)) {        // (8 = NB below) hand the box over in chunks of 256 displacements: (voxel << 8 | chunk) work items
        const int nchunks = (int)((c.vol + 255) >> 8);
        const int at = atomicAdd(list_count, nchunks);
        for (int i = 0; i < nchunks; ++i) list[at + i] = ((unsigned long long)x << 8) | (unsigned)i;
        keys[x] = ~0ull;                // merged by the wavefronts with atomicMin
        return;
    }
    // The previous winner (or the nearest displacement, see cand_box) lies inside the box and is simply visited again in index order,
    // which keeps the reference's first-minimum rule.  The box holds at most `limit` <= 8 displacements: their penalties are evaluated
    // first and ALL admissible cost-volume entries are requested together -- one memory round trip for the box instead of one per visited
    // displacement (the serial scan skipped an entry once a better one was known; those few extra loads are cheaper than the waiting).
    constexpr int NB = 8;
    int kk[NB];
    float pen[NB], val[NB];
    bool need[NB];
    {
        int ic = c.c_lo, ib = c.b_lo, ia = c.a_lo;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool in = j < (int)c.vol;
            kk[j] = (ia * n + ib) * n + ic;                                // j-th displacement of the box in index order
            const float e0 = mesh[kk[j]] - c.uc, e1 = mesh[K + kk[j]] - c.ub, e2 = mesh[2 * K + kk[j]] - c.ua;
            float q = e0 * e0;          // (..).pow(2).sum(0): sequential over the 3 components
            q += e1 * e1;
            q += e2 * e2;
            pen[j] = coef * q;                                             // coeffs[j]*(...)                     (:104)
            need[j] = in && !(c.sm + pen[j] > c.bound);                    // c.sm + pen <= cost_k
            if (j + 1 < (int)c.vol) {                                      // advance (c fastest); stays on a valid displacement past the end
                if (++ic > c.c_hi) { ic = c.c_lo; if (++ib > c.b_hi) { ib = c.b_lo; ++ia; } }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) val[j] = need[j] ? SsdIO<ST>::ld(ssd + (size_t)kk[j] * v + x) : 0.0f;
    float best = 0.0f;
    int bi = -1;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float cost = val[j] + pen[j];                                // ssd + coeffs[j]*(...)
        if (need[j] && ((bi < 0) | argmin_better(cost, best))) { best = cost; bi = kk[j]; }
    }
    if (bi < 0) { best = c.bound; bi = c.kp; }   // cannot happen (kp passes its own test); keeps the output defined
    keys[x] = pack_min_key(best, (unsigned)bi);
}

template <typename ST>
__device__ void argmin4_stream(const ST* __restrict__ ssd, const float* __restrict__ mesh, const float* __restrict__ u, float coef, int K,
                               size_t v, unsigned long long* __restrict__ keys, bool vec);

template <typename PrevT, typename ST>
__global__ __launch_bounds__(256) void k_argmin_wave(const ST* __restrict__ ssd, const float* __restrict__ mesh,
                                                     const float* __restrict__ u, float coef, int K, int n, size_t v,
                                                     const float* __restrict__ smin, const PrevT* __restrict__ kprev,
                                                     const unsigned long long* __restrict__ list, const int* __restrict__ list_count,
                                                     unsigned long long* __restrict__ keys, int stream_above, int vec, int refine, Prob2 o) {
    if (blockIdx.y) {
        ssd = shifted(ssd, o.ssd); u = shifted(u, o.out); smin = shifted(smin, o.ws); kprev = shifted(kprev, o.ws);
        list = shifted(list, o.ws); list_count = shifted(list_count, o.ws); keys = shifted(keys, o.ws);
    }
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int cnt = *list_count;
    if (cnt > stream_above) { argmin4_stream(ssd, mesh, u, coef, K, v, keys, vec != 0); return; }   // too many large boxes: one coalesced scan
    for (int e = wave; e < cnt; e += nwaves) {
        const unsigned long long item = list[e];
        const size_t x = (size_t)(item >> 8);
        const long long first = (long long)(item & 255) << 8;              // 256 displacements of the box: 4 per lane
        const int kp = (int)(unsigned)kprev[x];
        const CandBox c = cand_box(mesh, u[x], u[v + x], u[2 * v + x], coef, K, n, kp, SsdIO<ST>::ld(ssd + (size_t)kp * v + x), smin[x], ssd + x, v, refine);
        const int nc = c.c_hi - c.c_lo + 1, nb = c.b_hi - c.b_lo + 1;
        int kk[4];
        float pen[4];
        bool need[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                       // all four loads of a lane are in flight together
            const long long i = first + lane + 64 * j;
            need[j] = i < c.vol;
            const long long ii = need[j] ? i : 0;
            const int ic = c.c_lo + (int)(ii % nc), ib = c.b_lo + (int)((ii / nc) % nb), ia = c.a_lo + (int)(ii / ((long long)nc * nb));
            kk[j] = (ia * n + ib) * n + ic;
            const float e0 = mesh[kk[j]] - c.uc, e1 = mesh[K + kk[j]] - c.ub, e2 = mesh[2 * K + kk[j]] - c.ua;
            float q = e0 * e0;
            q += e1 * e1;
            q += e2 * e2;
            pen[j] = coef * q;
            need[j] = need[j] && (c.degenerate || !(c.sm + pen[j] > c.bound));
        }
        float val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = need[j] ? SsdIO<ST>::ld(ssd + (size_t)kk[j] * v + x) : 0.0f;
        unsigned long long key = ~0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long cand = need[j] ? pack_min_key(val[j] + pen[j], (unsigned)kk[j]) : ~0ull;
            key = cand < key ? cand : key;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_down(key, o);
            key = other < key ? other : key;
        }
        if (lane == 0 && key != ~0ull) atomicMin(&keys[x], key);
    }
}

// Bounded worst case of a pruned pass: when the large boxes add up to more than `stream_above` chunks (flat cost regions -- zero
// background, masked-out tissue -- keep their whole window), gathering them displacement by displacement would read far more
// sectors than one coalesced scan of the volume, so the wavefront kernel then streams the whole pass like k_argmin4<true> (argmin4_stream below) instead of working through its list.  Voxels already settled by k_argmin_voxel receive the
// same winner again through atomicMin.
template <typename ST>
__device__ void argmin4_stream(const ST* __restrict__ ssd, const float* __restrict__ mesh, const float* __restrict__ u, float coef, int K,
                               size_t v, unsigned long long* __restrict__ keys, bool vec) {
    const int xb = (int)((((v + 3) >> 2) + 255) >> 8);                   // blocks of 256 threads x 4 voxels
    int nslices = (int)gridDim.x * 2 / xb;                                // about two tiles per workgroup
    nslices = nslices < 1 ? 1 : (nslices > K ? K : nslices);
    const int kslice = (K + nslices - 1) / nslices;
    nslices = (K + kslice - 1) / kslice;
    for (int tile = blockIdx.x; tile < xb * nslices; tile += gridDim.x) {
        const int bx = tile % xb, sl = tile / xb;
        const size_t x0 = ((size_t)bx * 256 + threadIdx.x) * 4;
        const int k0 = sl * kslice, k1 = min(k0 + kslice, K);
        if (x0 >= v) continue;
        const int nv = (int)min((size_t)4, v - x0);
        float u0[4], u1[4], u2[4], best[4] = {0.f, 0.f, 0.f, 0.f};
        int bi[4] = {-1, -1, -1, -1};
        for (int j = 0; j < 4; ++j) { const size_t x = x0 + (j < nv ? j : 0); u0[j] = u[x]; u1[j] = u[v + x]; u2[j] = u[2 * v + x]; }
        for (int k = k0; k < k1; ++k) {
            const float m0 = mesh[k], m1 = mesh[K + k], m2 = mesh[2 * K + k];
            const ST* p = ssd + (size_t)k * v + x0;
            float c4[4];
            if (vec) { const float4 q4 = SsdIO<ST>::ld4(p); c4[0] = q4.x; c4[1] = q4.y; c4[2] = q4.z; c4[3] = q4.w; }
            else for (int j = 0; j < 4; ++j) c4[j] = j < nv ? SsdIO<ST>::ld(p + j) : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nv) continue;
                const float e0 = m0 - u0[j], e1 = m1 - u1[j], e2 = m2 - u2[j];
                float q = e0 * e0;          // (..).pow(2).sum(0): sequential over the 3 components
                q += e1 * e1;
                q += e2 * e2;
                const float cost = c4[j] + coef * q;    // ssd + coeffs[j]*(...)                       (:104)
                if ((bi[j] < 0) | argmin_better(cost, best[j])) { best[j] = cost; bi[j] = k; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nv && bi[j] >= 0) atomicMin(&keys[x0 + j], pack_min_key(best[j], (unsigned)bi[j]));
    }
}

// smin[x] from the (cost, index) keys of a plain argmin pass (inverse of pack_min_key's order-preserving map)
__global__ __launch_bounds__(256) void k_keys_to_min(const unsigned long long* __restrict__ keys, size_t v, float* __restrict__ smin) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= v) return;
    const unsigned b = (unsigned)(keys[x] >> 32);
    smin[x] = __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
}

// (cost, index) keys of a plain argmin pass -> int32 winners + per-voxel minimum in one launch (whole-pair pipeline)
__global__ __launch_bounds__(256) void k_keys_to_idx_min(const unsigned long long* __restrict__ keys, size_t v, int* __restrict__ idx,
                                                         float* __restrict__ smin, Prob2 o) {
    if (blockIdx.y) { keys = shifted(keys, o.ws); idx = shifted(idx, o.ws); smin = shifted(smin, o.ws); }
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= v) return;
    const unsigned long long k = keys[x];
    const unsigned b = (unsigned)(k >> 32);
    idx[x] = (int)(unsigned)(k & 0xffffffffull);
    smin[x] = __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
}

// smin[x] = ssd[argmin[x], x]: the exact minimum over the search window (argmin is the plain argmin of the same volume)
template <typename ST>
__global__ __launch_bounds__(256) void k_gather_min(const ST* __restrict__ ssd, const int* __restrict__ idx, size_t v,
                                                    float* __restrict__ smin, Prob2 o) {
    if (blockIdx.y) { ssd = shifted(ssd, o.ssd); idx = shifted(idx, o.ws); smin = shifted(smin, o.ws); }
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < v) smin[x] = SsdIO<ST>::ld(ssd + (size_t)idx[x] * v + x);
}

__global__ __launch_bounds__(256) void k_keys_to_index(const unsigned long long* __restrict__ keys, size_t v,
                                                       int* __restrict__ idx32, int64_t* __restrict__ idx64) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= v) return;
    const unsigned k = (unsigned)(keys[x] & 0xffffffffull);
    if (idx32) idx32[x] = (int)k;
    if (idx64) idx64[x] = (int64_t)k;
}
__global__ __launch_bounds__(256) void k_index64_to_32(const int64_t* __restrict__ in, size_t v, int* __restrict__ out, Prob2 o) {
    if (blockIdx.y) { in = shifted(in, o.argmin); out = shifted(out, o.ws); }
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < v) out[x] = (int)in[x];
}

// stand-alone smoothing step; `reset` (optional) is a key buffer that this launch re-arms to all ones for a later pass
template <typename IndexT>
__global__ __launch_bounds__(256) void k_gather_box3(const IndexT* __restrict__ idx, const float* __restrict__ mesh, int K,
                                                     int h, int w, int d, float* __restrict__ out,
                                                     unsigned long long* __restrict__ reset, int* __restrict__ clear_count, Prob2 o) {
    if (blockIdx.y) {
        idx = shifted(idx, o.ws); out = shifted(out, o.out);
        if (reset) reset = shifted(reset, o.ws);
        if (clear_count) clear_count = shifted(clear_count, o.ws);
    }
    const size_t v = (size_t)h * w * d;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && clear_count) *clear_count = 0;            // list length of the pruned pass that follows
    if (i >= v) return;
    if (reset) reset[i] = ~0ull;
    float o0, o1, o2;
    smooth_winner(idx, mesh, K, h, w, d, i, o0, o1, o2);
    out[i] = o0;
    out[v + i] = o1;
    out[2 * v + i] = o2;
}

template <typename ST>
static int argmin_pass(const ST* ssd, const float* mesh, const float* u, float coef, bool coupled, int K, size_t v,
                       unsigned long long* keys, bool arm, hipStream_t s) {
    if (arm && hipMemsetAsync(keys, 0xff, sizeof(unsigned long long) * v, s) != hipSuccess)
        return fail(CVX_ERR_LAUNCH, "argmin: memset failed");
    const bool vec4 = (v % 4 == 0) && ((reinterpret_cast<uintptr_t>(ssd) | reinterpret_cast<uintptr_t>(u)) & 15) == 0;
    const int xb = (int)cdiv64((int64_t)(vec4 ? v / 4 : v), 256);
    // about 512 workgroups (2 per CU): every K-slice ends in one atomicMin per voxel, and 450-650 workgroups measured fastest on
    // the 270 MB volume (52 us = 5.2 TB/s; 1024: 64 us, 256: 56 us, 2048: 75 us)
    int nslices = cdiv(vec4 ? 512 : 1024, xb);
    if (nslices > K) nslices = K;
    if (nslices < 1) nslices = 1;
    const int kslice = cdiv(K, nslices);
    nslices = cdiv(K, kslice);
    const dim3 grid(xb, nslices);
    if (vec4) {
        if (coupled) hipLaunchKernelGGL((k_argmin4<true, ST>), grid, dim3(256), 0, s, ssd, mesh, u, coef, K, v, kslice, keys);
        else hipLaunchKernelGGL((k_argmin4<false, ST>), grid, dim3(256), 0, s, ssd, mesh, u, coef, K, v, kslice, keys);
    } else if (coupled) hipLaunchKernelGGL((k_argmin<true, ST>), grid, dim3(256), 0, s, ssd, mesh, u, coef, K, v, kslice, keys);
    else hipLaunchKernelGGL((k_argmin<false, ST>), grid, dim3(256), 0, s, ssd, mesh, u, coef, K, v, kslice, keys);
    return check_last("argmin");
}

// pruned coupled pass (smoothing step included): per-voxel candidate boxes, then one wavefront per 256-displacement chunk of the
// large boxes; kprev = int32 indices (first pass) or the key buffer of the previous pass; *list_count must be zero on entry
// (the previous pass's voxel kernel clears it through next_count; the two counters alternate)
template <typename PrevT, typename ST>
static int argmin_pass_pruned(const ST* ssd, const float* mesh, float* u, float coef, int K, int n, int h, int w, int d,
                              const float* smin, const PrevT* kprev, unsigned long long* list, int* list_count, int* next_count,
                              unsigned long long* keys, const unsigned long long* minkeys, const Prob2& o, int nprob, hipStream_t s) {
    const size_t v = (size_t)h * w * d;
    const int refine = options().prune_refine != 0 ? 8 : 0x7fffffff;     // boxes above the per-thread limit get the second bound
    hipLaunchKernelGGL((k_argmin_voxel<PrevT, ST>), dim3((unsigned)cdiv64((int64_t)v, 64), nprob), dim3(64), 0, s, ssd, mesh, u, coef, K, n, h,
                       w, d, smin, kprev, 8, list, list_count, next_count, keys, minkeys, refine, o);   // limit 8 = NB of the voxel kernel
    // worst case bounded by one coalesced scan per pass: a chunk of 256 scattered reads moves about 8 KB, the scan K * v * 4 bytes
    const long long above = options().prune_stream_above >= 0 ? options().prune_stream_above : (long long)((double)K * (double)v / 2048.0);
    const int stream_above = (int)(above > 0x7fffffff ? 0x7fffffff : above);
    const int vec = (v % 4 == 0) && ((reinterpret_cast<uintptr_t>(ssd) | (uintptr_t)(o.ssd < 0 ? -o.ssd : o.ssd)) & 15) == 0;
    // voxels whose key the voxel kernel stored plainly keep it under the scan (it finds the same winner); listed voxels were armed to ~0
    hipLaunchKernelGGL((k_argmin_wave<PrevT, ST>), dim3(512, nprob), dim3(256), 0, s, ssd, mesh, u, coef, K, n, v, smin, kprev, list,
                       list_count, keys, stream_above, vec, refine, o);
    return check_last("argmin_pruned");
}

// plain argmin pass that leaves its (cost, index) keys in `keys` (first key buffer of a coupled-convex workspace)
int launch_argmin_keys(const void* ssd, bool f16, int K, size_t v, unsigned long long* keys, bool arm, hipStream_t s) {
    if (f16) return argmin_pass(static_cast<const __half*>(ssd), nullptr, nullptr, 0.0f, false, K, v, keys, arm, s);
    return argmin_pass(static_cast<const float*>(ssd), nullptr, nullptr, 0.0f, false, K, v, keys, arm, s);
}
int* coupled_ws_counts(void* workspace, size_t workspace_bytes, int h, int w, int d, int disp_hw) {
    const int n = 2 * disp_hw + 1, K = n * n * n;
    const size_t v = (size_t)h * w * d;
    Carver cv(workspace, workspace_bytes);
    for (int i = 0; i < 3; ++i) (void)cv.take<unsigned long long>(v);
    (void)cv.take<int>(v);
    (void)cv.take<float>(v);
    const size_t list_cap = (size_t)((K + 255) / 256) * v;
    return reinterpret_cast<int*>(cv.take<unsigned long long>(list_cap + 8) + list_cap);
}

int launch_argmin(const void* ssd, bool f16, const float* mesh, const float* u, float coef, bool coupled, int K, size_t v,
                  unsigned long long* keys, int64_t* argmin_out, hipStream_t s) {
    int rc = f16 ? argmin_pass(static_cast<const __half*>(ssd), mesh, u, coef, coupled, K, v, keys, true, s)
                 : argmin_pass(static_cast<const float*>(ssd), mesh, u, coef, coupled, K, v, keys, true, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_keys_to_index, dim3((unsigned)cdiv64((int64_t)v, 256)), dim3(256), 0, s, keys, v, (int*)nullptr,
                       argmin_out);
    return check_last("argmin index");
}

// ---- inverse consistency -------------------------------------------------------------------------------
//   d1 <- 0.5*(d1 - d2 o (id + d1)),  d2 <- 0.5*(d2 - d1 o (id + d2))   simultaneously, `iters` times
__global__ __launch_bounds__(256) void k_ic_step(const float* __restrict__ a1, const float* __restrict__ a2, int h, int w,
                                                 int d, const float* __restrict__ bh, const float* __restrict__ bw,
                                                 const float* __restrict__ bd, float* __restrict__ o1,
                                                 float* __restrict__ o2) {
    const size_t v = (size_t)h * w * d;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= v) return;
    const int x = (int)(p % d), y = (int)((p / d) % w), z = (int)(p / ((size_t)d * w));
    Tri t;
    tri_setup(t, bd[x] + a1[p], bw[y] + a1[v + p], bh[z] + a1[2 * v + p], h, w, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) o1[(size_t)c * v + p] = 0.5f * (a1[(size_t)c * v + p] - tri_sample(t, a2 + (size_t)c * v, h, w, d));
    tri_setup(t, bd[x] + a2[p], bw[y] + a2[v + p], bh[z] + a2[2 * v + p], h, w, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) o2[(size_t)c * v + p] = 0.5f * (a2[(size_t)c * v + p] - tri_sample(t, a1 + (size_t)c * v, h, w, d));
}


// ---- inverse consistency in ONE launch (option ic_fused) ---------------------------------------------------------------------------
// The 15 steps are dependent and tiny (30 784 voxels, two 370 KB fields): one launch per step costs 5.7 us, of which the kernel boundary is
// half.  A device-wide barrier costs more than a boundary on this part (4 us for the counter over 256 workgroups, 16 us with the L2
// write-back / invalidate that makes one XCD's writes visible to another) -- but workgroups of ONE XCD share its L2:
// between them a barrier could be an L2 atomic and visibility would need no cache maintenance, only accesses that do not stop in the per-CU L1.  The launch has 8 x IC_NWG workgroups, dealt round-robin to the 8 XCDs; the IC_NWG workgroups with
// blockIdx % 8 == xcd do the work, the others exit at once.  That placement is how the hardware dispatches, not a guarantee: every working
// workgroup reports its XCC_ID, and if they are not all equal the kernel stops after the first step and k_ic_fallback (always enqueued,
// normally an empty launch) redoes the whole computation with one workgroup.
constexpr int IC_NWG = 32, IC_NT = 512;
struct ICSync { unsigned arrive, xcc_mask, fallback, pad; };

// Field accesses are agent-scope atomics: correct on any placement, but served behind the L2 (~2 us per dependent access).  MEASURED: 19 us per step,
// 289 us per call against 145 us for 15 launches (tools/experiments/ic_time.py) -- the option is OFF by default.  The cheaper form the idea needs --
// loads that miss the L1 but hit the XCD's L2 -- does not exist on this part: `sc0` (workgroup scope) loads may hit the L1 (a workgroup owns its CU's
// L1), so a barrier that polls with them never sees the other workgroups' arrivals (tried: the kernel hangs); `sc1` (agent scope) goes past the L2 for
// ordinary (non-coherent across XCDs) allocations.  DESIGN.md section 12.10.
template <bool AG>
struct ICMem {
    __device__ __forceinline__ float ld(const float* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void st(float* p, float v) const { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
// tri_sample with loads that are served by the L2 (same arithmetic, same order)
template <bool AG>
__device__ __forceinline__ float tri_sample_l2(const ICMem<AG>& M, const Tri& t, const float* vol, int h, int w, int d) {
    const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const bool zi0 = (unsigned)z0 < (unsigned)h, zi1 = (unsigned)z1 < (unsigned)h, yi0 = (unsigned)y0 < (unsigned)w,
               yi1 = (unsigned)y1 < (unsigned)w, xi0 = (unsigned)x0 < (unsigned)d, xi1 = (unsigned)x1 < (unsigned)d;
    const int zc0 = clampi(z0, 0, h - 1), zc1 = clampi(z1, 0, h - 1), yc0 = clampi(y0, 0, w - 1), yc1 = clampi(y1, 0, w - 1),
              xc0 = clampi(x0, 0, d - 1), xc1 = clampi(x1, 0, d - 1);
    const int r00 = (zc0 * w + yc0) * d, r01 = (zc0 * w + yc1) * d, r10 = (zc1 * w + yc0) * d, r11 = (zc1 * w + yc1) * d;
    const float v0 = M.ld(vol + r00 + xc0), v1 = M.ld(vol + r00 + xc1), v2 = M.ld(vol + r01 + xc0), v3 = M.ld(vol + r01 + xc1),
                v4 = M.ld(vol + r10 + xc0), v5 = M.ld(vol + r10 + xc1), v6 = M.ld(vol + r11 + xc0), v7 = M.ld(vol + r11 + xc1);
    float o = 0.0f, n;
    n = o + v0 * t.tnw; o = (zi0 && yi0 && xi0) ? n : o;
    n = o + v1 * t.tne; o = (zi0 && yi0 && xi1) ? n : o;
    n = o + v2 * t.tsw; o = (zi0 && yi1 && xi0) ? n : o;
    n = o + v3 * t.tse; o = (zi0 && yi1 && xi1) ? n : o;
    n = o + v4 * t.bnw; o = (zi1 && yi0 && xi0) ? n : o;
    n = o + v5 * t.bne; o = (zi1 && yi0 && xi1) ? n : o;
    n = o + v6 * t.bsw; o = (zi1 && yi1 && xi0) ? n : o;
    n = o + v7 * t.bse; o = (zi1 && yi1 && xi1) ? n : o;
    return o;
}
// one step over the voxels p = first, first + stride, ..   (`lo` = lowest address of the four field buffers, `span` = bytes they cover)
template <bool AG>
__device__ __forceinline__ void ic_sweep(const float* a1, const float* a2, int h, int w, int d, const float* __restrict__ bh, const float* __restrict__ bw,
                                         const float* __restrict__ bd, float* o1, float* o2, int first, int stride, const float* lo, unsigned span) {
    const int v = h * w * d;
    ICMem<AG> M;
    (void)lo; (void)span;
    for (int p = first; p < v; p += stride) {
        const int x = p % d, y = (p / d) % w, z = p / (d * w);
        const float a10 = M.ld(a1 + p), a11 = M.ld(a1 + v + p), a12 = M.ld(a1 + 2 * v + p);
        const float a20 = M.ld(a2 + p), a21 = M.ld(a2 + v + p), a22 = M.ld(a2 + 2 * v + p);
        Tri t, u;
        tri_setup(t, bd[x] + a10, bw[y] + a11, bh[z] + a12, h, w, d);
        tri_setup(u, bd[x] + a20, bw[y] + a21, bh[z] + a22, h, w, d);
        const float s0 = tri_sample_l2(M, t, a2, h, w, d), s1 = tri_sample_l2(M, t, a2 + v, h, w, d), s2 = tri_sample_l2(M, t, a2 + 2 * v, h, w, d);
        const float r0 = tri_sample_l2(M, u, a1, h, w, d), r1 = tri_sample_l2(M, u, a1 + v, h, w, d), r2 = tri_sample_l2(M, u, a1 + 2 * v, h, w, d);
        M.st(o1 + p, 0.5f * (a10 - s0)); M.st(o1 + v + p, 0.5f * (a11 - s1)); M.st(o1 + 2 * v + p, 0.5f * (a12 - s2));
        M.st(o2 + p, 0.5f * (a20 - r0)); M.st(o2 + v + p, 0.5f * (a21 - r1)); M.st(o2 + 2 * v + p, 0.5f * (a22 - r2));
    }
}
template <bool AG>
__global__ __launch_bounds__(IC_NT) void k_ic_persistent(const float* f1, const float* f2, int h, int w, int d, int iters, const float* __restrict__ bh,
                                                         const float* __restrict__ bw, const float* __restrict__ bd, float* t1, float* t2, float* o1,
                                                         float* o2, ICSync* sync, int xcd, int force_fallback, const float* lo, unsigned span) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    const int wg = (int)(blockIdx.x >> 3);
    constexpr int SCOPE = __HIP_MEMORY_SCOPE_AGENT;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u;
        __hip_atomic_fetch_or(&sync->xcc_mask, force_fallback ? (1u << (wg & 1)) : (1u << xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float *s1 = f1, *s2 = f2;
    for (int it = 0; it < iters; ++it) {
        const bool to_out = ((iters - 1 - it) & 1) == 0;
        float* d1 = to_out ? o1 : t1;
        float* d2 = to_out ? o2 : t2;
        ic_sweep<AG>(s1, s2, h, w, d, bh, bw, bd, d1, d2, wg * IC_NT + (int)threadIdx.x, IC_NWG * IC_NT, lo, span);
        if (it == iters - 1) break;
        // barrier between the IC_NWG workgroups: every store of this workgroup has reached the L2 (vmcnt(0) of all its wavefronts), then one arrival
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&sync->arrive, 1u, __ATOMIC_RELAXED, SCOPE);
            const unsigned want = (unsigned)(it + 1) * IC_NWG;
            while (__hip_atomic_fetch_add(&sync->arrive, 0u, __ATOMIC_RELAXED, SCOPE) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (it == 0) {                         // everyone has reported: all on one XCD?
            const unsigned m = __hip_atomic_load(&sync->xcc_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m & (m - 1)) {
                if (wg == 0 && threadIdx.x == 0) __hip_atomic_store(&sync->fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        s1 = d1; s2 = d2;
    }
}
// normally an empty launch; after a failed placement check: the whole computation by one workgroup (its wavefronts share one L1, every access
// goes to the L2 anyway)
__global__ __launch_bounds__(1024) void k_ic_fallback(const float* f1, const float* f2, int h, int w, int d, int iters, const float* __restrict__ bh,
                                                      const float* __restrict__ bw, const float* __restrict__ bd, float* t1, float* t2, float* o1,
                                                      float* o2, const ICSync* sync) {
    if (!sync->fallback) return;
    const float *s1 = f1, *s2 = f2;
    for (int it = 0; it < iters; ++it) {
        const bool to_out = ((iters - 1 - it) & 1) == 0;
        float* d1 = to_out ? o1 : t1;
        float* d2 = to_out ? o2 : t2;
        ic_sweep<true>(s1, s2, h, w, d, bh, bw, bd, d1, d2, (int)threadIdx.x, 1024, f1, 0u);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        s1 = d1; s2 = d2;
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_coupled_convex_workspace_bytes(int h, int w, int d, int disp_hw) {
    const int n = 2 * disp_hw + 1;
    const size_t v = (size_t)h * w * d;
    size_t used = 0;
    for (int i = 0; i < 3; ++i) used = carve_size(used, sizeof(unsigned long long) * v);
    used = carve_size(used, sizeof(int) * v);
    used = carve_size(used, sizeof(float) * v);       // smin
    used = carve_size(used, sizeof(unsigned long long) * ((size_t)((n * n * n + 255) / 256) * v + 8));   // work items of the pruned passes + count
    return used + 256;
}


extern "C" int cvx_coupled_convex_f32(const float* ssd, const int64_t* argmin, const float* mesh, int h, int w, int d,
                                      int disp_hw, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    // the caller's `argmin` only seeds the first smoothing step (as in the reference); the lower bound of the pruned
    // passes is taken from the volume itself
    return cvx::coupled_convex_impl(ssd, false, argmin, mesh, h, w, d, disp_hw, out, /*argmin_is_exact=*/false, workspace, workspace_bytes, stream);
}

extern "C" int cvx_coupled_convex_f16(const void* ssd_half, const int64_t* argmin, const float* mesh, int h, int w, int d,
                                      int disp_hw, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    return cvx::coupled_convex_impl(ssd_half, true, argmin, mesh, h, w, d, disp_hw, out, /*argmin_is_exact=*/false, workspace, workspace_bytes, stream);
}

// argmin_is_exact: `argmin` is the plain argmin of `ssd` (the whole-pair pipeline computes it itself), so ssd[argmin] is the
// per-voxel minimum and the extra streaming pass that determines it can be skipped.
// nprob = 2: a second, independent problem (displaced by `o`, see Prob2) is solved by the same launches.
template <typename ST>
static int coupled_core(const ST* ssd, const int64_t* argmin, const float* mesh, int h, int w, int d, int disp_hw, float* out,
                        bool argmin_is_exact, void* workspace, size_t workspace_bytes, const Prob2& o, int nprob, void* stream,
                        bool counts_zeroed = false) {
    // argmin == nullptr: the (cost, index) keys of the plain argmin pass sit in the first key buffer of the workspace
    const bool from_keys = argmin == nullptr;
    CVX_REQUIRE(ssd && mesh && out && workspace, "cvx_coupled_convex_f32: null pointer");
    CVX_REQUIRE(h > 0 && w > 0 && d > 0 && disp_hw >= 0, "cvx_coupled_convex_f32: bad arguments");
    if (workspace_bytes < cvx_coupled_convex_workspace_bytes(h, w, d, disp_hw))
        return fail(CVX_ERR_WORKSPACE, "cvx_coupled_convex_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int n = 2 * disp_hw + 1, K = n * n * n;
    const size_t v = (size_t)h * w * d;
    Carver cv(workspace, workspace_bytes);
    // three key buffers in rotation: pass p mins into keys[p % 3]; the gather that consumes pass p re-arms keys[(p+2) % 3]
    unsigned long long* keys[3];
    for (int i = 0; i < 3; ++i) keys[i] = cv.take<unsigned long long>(v);
    int* idx = cv.take<int>(v);
    float* smin = cv.take<float>(v);
    const size_t list_cap = (size_t)((K + 255) / 256) * v;
    unsigned long long* list = cv.take<unsigned long long>(list_cap + 8);
    int* list_count = reinterpret_cast<int*>(list + list_cap);
    const dim3 gv((unsigned)cdiv64((int64_t)v, 256), nprob);
    const bool no_prune_env = options().no_prune != 0;
    if (from_keys && (no_prune_env || !argmin_is_exact)) return fail(CVX_ERR_INVALID_ARG, "coupled_convex: key input needs the pruned path");
    if (!from_keys) hipLaunchKernelGGL(k_index64_to_32, gv, dim3(256), 0, s, argmin, v, idx, o);
    // exact pruning needs smin[x] = min_k ssd[k,x].  CVX_NO_PRUNE=1 streams every pass instead.
    const bool no_prune = options().no_prune != 0;
    const bool prune = !no_prune;
    if (prune) {
        int* counts = list_count;                                       // two alternating list lengths
        for (int q = 0; q < nprob && !counts_zeroed; ++q)
            if (hipMemsetAsync(reinterpret_cast<char*>(counts) + (q ? o.ws : 0), 0, 2 * sizeof(int), s) != hipSuccess)
                return fail(CVX_ERR_LAUNCH, "coupled_convex: memset failed");
        if (from_keys) hipLaunchKernelGGL(k_keys_to_idx_min, gv, dim3(256), 0, s, keys[0], v, idx, smin, o);
        else if (argmin_is_exact) hipLaunchKernelGGL(k_gather_min<ST>, gv, dim3(256), 0, s, ssd, idx, v, smin, o);
        else {
            int rc = argmin_pass(ssd, nullptr, nullptr, 0.0f, false, K, v, keys[0], true, s);   // per-voxel minimum of the volume
            if (rc) return rc;
            hipLaunchKernelGGL(k_keys_to_min, dim3(gv.x), dim3(256), 0, s, keys[0], v, smin);
        }
        static const float coeffs[6] = {0.003f, 0.01f, 0.03f, 0.1f, 0.3f, 1.0f};   // torch.tensor([...]) float32 (:98)
        for (int it = 0; it < 6; ++it) {
            // smoothing of the previous winners + pruned argmin; keys[1], keys[2] alternate (keys[0] may hold the minimum pass)
            unsigned long long* kc = keys[1 + (it & 1)];
            int rc;
            // keys[0] holds the library's own minimum pass unless the caller vouched for `argmin` (then idx IS the plain argmin)
            const unsigned long long* minkeys = (from_keys || !argmin_is_exact) ? keys[0] : nullptr;
            if (it == 0) rc = argmin_pass_pruned<int, ST>(ssd, mesh, out, coeffs[it], K, n, h, w, d, smin, idx, list, counts + (it & 1), counts + ((it + 1) & 1), kc, minkeys, o, nprob, s);
            else rc = argmin_pass_pruned<unsigned long long, ST>(ssd, mesh, out, coeffs[it], K, n, h, w, d, smin, keys[1 + ((it - 1) & 1)], list, counts + (it & 1), counts + ((it + 1) & 1), kc, minkeys, o, nprob, s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_gather_box3<unsigned long long>, gv, dim3(256), 0, s, keys[1 + (5 & 1)], mesh, K, h, w, d, out, (unsigned long long*)nullptr, (int*)nullptr, o);
        return check_last("coupled_convex");
    }
    // streaming path: one problem per call
    if (hipMemsetAsync(keys[0], 0xff, sizeof(unsigned long long) * v, s) != hipSuccess) return fail(CVX_ERR_LAUNCH, "coupled_convex: memset failed");
    hipLaunchKernelGGL(k_gather_box3<int>, dim3(gv.x), dim3(256), 0, s, idx, mesh, K, h, w, d, out, keys[1], (int*)nullptr, o);
    static const float coeffs[6] = {0.003f, 0.01f, 0.03f, 0.1f, 0.3f, 1.0f};   // torch.tensor([...]) float32 (:98)
    for (int it = 0; it < 6; ++it) {
        int rc = argmin_pass(ssd, mesh, out, coeffs[it], true, K, v, keys[it % 3], false, s);
        if (rc) return rc;
        // the streamed passes need their key buffer re-armed two passes ahead
        hipLaunchKernelGGL(k_gather_box3<unsigned long long>, dim3(gv.x), dim3(256), 0, s, keys[it % 3], mesh, K, h, w, d, out,
                           it < 4 ? keys[(it + 2) % 3] : nullptr, (int*)nullptr, o);
    }
    return check_last("coupled_convex");
}

int cvx::coupled_convex_impl(const void* ssd, bool f16, const int64_t* argmin, const float* mesh, int h, int w, int d, int disp_hw, float* out,
                             bool argmin_is_exact, void* workspace, size_t workspace_bytes, void* stream) {
    if (f16) return coupled_core(static_cast<const __half*>(ssd), argmin, mesh, h, w, d, disp_hw, out, argmin_is_exact, workspace, workspace_bytes, Prob2{0, 0, 0, 0}, 1, stream);
    return coupled_core(static_cast<const float*>(ssd), argmin, mesh, h, w, d, disp_hw, out, argmin_is_exact, workspace, workspace_bytes, Prob2{0, 0, 0, 0}, 1, stream);
}

// Forward and reverse direction of a pair in the same launches (the per-pass kernels are latency-bound at 30 000 voxels, so two
// problems cost about as much as one).  Both argmins must be the plain argmins of their volumes; both workspaces have the size
// cvx_coupled_convex_workspace_bytes.  Falls back to two sequential solves when pruning is switched off.  argminA == argminB ==
// nullptr: the plain argmin passes left their (cost, index) keys at the start of the respective workspace (launch_argmin_keys).
template <typename ST>
static int coupled_dual_t(const ST* ssdA, const int64_t* argminA, float* outA, void* wsA, const ST* ssdB,
                          const int64_t* argminB, float* outB, void* wsB, const float* mesh, int h, int w, int d, int disp_hw,
                          size_t workspace_bytes, void* stream, bool counts_zeroed) {
    const bool no_prune = options().no_prune != 0;
    if (no_prune || !ssdB || !outB || !wsB) {
        int rc = coupled_core(ssdA, argminA, mesh, h, w, d, disp_hw, outA, true, wsA, workspace_bytes, Prob2{0, 0, 0, 0}, 1, stream, counts_zeroed);
        if (rc || !ssdB) return rc;
        return coupled_core(ssdB, argminB, mesh, h, w, d, disp_hw, outB, true, wsB, workspace_bytes, Prob2{0, 0, 0, 0}, 1, stream, counts_zeroed);
    }
    auto diff = [](const void* b, const void* a) { return (ptrdiff_t)(reinterpret_cast<uintptr_t>(b) - reinterpret_cast<uintptr_t>(a)); };
    // the carve-up of a workspace depends on its alignment modulo 256: equal residues give equal layouts
    if (((reinterpret_cast<uintptr_t>(wsA) ^ reinterpret_cast<uintptr_t>(wsB)) & 255) != 0)
        return fail(CVX_ERR_INVALID_ARG, "coupled_convex_dual: workspaces must share their alignment modulo 256");
    const Prob2 o{diff(ssdB, ssdA), diff(argminB, argminA), diff(outB, outA), diff(wsB, wsA)};
    return coupled_core(ssdA, argminA, mesh, h, w, d, disp_hw, outA, true, wsA, workspace_bytes, o, 2, stream, counts_zeroed);
}
int cvx::coupled_convex_dual_impl(const void* ssdA, const int64_t* argminA, float* outA, void* wsA, const void* ssdB, bool f16,
                                  const int64_t* argminB, float* outB, void* wsB, const float* mesh, int h, int w, int d, int disp_hw,
                                  size_t workspace_bytes, void* stream, bool counts_zeroed) {
    if (f16) return coupled_dual_t(static_cast<const __half*>(ssdA), argminA, outA, wsA, static_cast<const __half*>(ssdB), argminB, outB, wsB, mesh, h, w, d, disp_hw, workspace_bytes, stream, counts_zeroed);
    return coupled_dual_t(static_cast<const float*>(ssdA), argminA, outA, wsA, static_cast<const float*>(ssdB), argminB, outB, wsB, mesh, h, w, d, disp_hw, workspace_bytes, stream, counts_zeroed);
}

extern "C" size_t cvx_inverse_consistency_workspace_bytes(int h, int w, int d) {
    return 2 * (256 + sizeof(float) * 3 * (size_t)h * w * d) + 256 + 256;      // two ping-pong fields + the synchronisation words of k_ic_persistent
}

extern "C" int cvx_inverse_consistency_f32(const float* f1, const float* f2, int h, int w, int d, int iters,
                                           const float* base_h, const float* base_w, const float* base_d, float* o1,
                                           float* o2, void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(f1 && f2 && o1 && o2 && base_h && base_w && base_d, "cvx_inverse_consistency_f32: null pointer");
    CVX_REQUIRE(h > 0 && w > 0 && d > 0 && iters >= 0, "cvx_inverse_consistency_f32: bad arguments");
    CVX_REQUIRE(o1 != f1 && o2 != f2 && o1 != f2 && o2 != f1, "cvx_inverse_consistency_f32: outputs must not alias inputs");
    if (workspace_bytes < cvx_inverse_consistency_workspace_bytes(h, w, d) || !workspace)
        return fail(CVX_ERR_WORKSPACE, "cvx_inverse_consistency_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const size_t v = (size_t)h * w * d, bytes = sizeof(float) * 3 * v;
    Carver cv(workspace, workspace_bytes);
    float* t1 = cv.take<float>(3 * v);
    float* t2 = cv.take<float>(3 * v);
    if (iters == 0) {
        if (o1 != f1) (void)hipMemcpyAsync(o1, f1, bytes, hipMemcpyDeviceToDevice, s);
        if (o2 != f2) (void)hipMemcpyAsync(o2, f2, bytes, hipMemcpyDeviceToDevice, s);
        return check_last("inverse_consistency");
    }
    if (options().ic_fused && iters >= 2 && v * 3 < ((size_t)1 << 30)) {
        ICSync* sync = cv.take<ICSync>(1);
        if (hipMemsetAsync(sync, 0, sizeof(ICSync), s) != hipSuccess) return fail(CVX_ERR_LAUNCH, "inverse_consistency: memset failed");
        static std::atomic<unsigned> turn{0};
        const int xcd = (int)(turn.fetch_add(1) & 7u);                     // concurrent pairs (register_pairs streams) take different XCDs
        hipLaunchKernelGGL(k_ic_persistent<true>, dim3(8 * IC_NWG), dim3(IC_NT), 0, s, f1, f2, h, w, d, iters, base_h, base_w, base_d, t1, t2, o1, o2, sync, xcd,
                           (int)(options().ic_fused == 2), f1, 0u);
        hipLaunchKernelGGL(k_ic_fallback, dim3(1), dim3(1024), 0, s, f1, f2, h, w, d, iters, base_h, base_w, base_d, t1, t2, o1, o2, sync);
        return check_last("inverse_consistency");
    }
    // ping-pong (tmp <-> out) arranged so that the last iteration writes o1/o2
    const float *s1 = f1, *s2 = f2;
    const dim3 gv((unsigned)cdiv64((int64_t)v, 256));
    for (int it = 0; it < iters; ++it) {
        const bool to_out = ((iters - 1 - it) & 1) == 0;
        float* d1 = to_out ? o1 : t1;
        float* d2 = to_out ? o2 : t2;
        hipLaunchKernelGGL(k_ic_step, dim3((unsigned)cdiv64((int64_t)v, 64)), dim3(64), 0, s, s1, s2, h, w, d, base_h, base_w, base_d, d1, d2);   // one wave per workgroup: the 30 000 voxels spread over all CUs
        s1 = d1; s2 = d2;
    }
    return check_last("inverse_consistency");
}
