// mindmarch.hip -- the MIND-SSC stencil pass as a z-marching kernel (radius 1, dilation 2: the registration pipeline's setting;
// reference: convex_adam_utils.py:24-58).  Same results, bit for bit, as the tiled k_mind<1> of mind.hip, which stays as the
// general path (other radii / dilations, ragged rows).
//
// A workgroup owns 8 rows (y) x 64 columns (x) x a chunk of planes (z) and marches along z.  LDS holds a ring of the five
// image planes around the current one (rows/columns with halo 3, replicate addressing = the reference's two paddings) and
// one exchange buffer; NO squared-difference volume is staged: a thread evaluates the 3 x 6 window of squared differences of
// its 4 output columns straight from the image ring (8-byte reads, bank-conflict free) for each of its 4 channels -- three
// groups of two wavefronts share the 12 channels -- and keeps, per channel and column, two running raster-order sums in
// registers:  P(z) = the 9 taps of plane z added to +0.0 (the prefix of the 27-tap sum of output plane z+1) and
// A(z) = P(z-1) + 9 taps of plane z; output plane z-1 = A(z-1) + 9 taps of plane z, one exact division.  Every tap is
// therefore computed once per plane instead of three times, nothing but the image is read from HBM, and the per-tile
// prologue / 12 barrier phases of the tiled kernel disappear (one barrier per plane).
// The 12 patch SSDs of a voxel meet in the exchange buffer for the order-independent exact variance statistics (as in mind.hip).
#include "cvx_common.h"
#include "mind_common.h"

namespace cvx {

constexpr int MM_NT = 512, MM_NS = 4, MM_CPS = 3;     // 4 wave groups x 3 channels
// Tile shapes: 8 rows x 64 columns, or 16 x 32 when the last 64-column tile of a row would be at most half full (D = 224: 7 tiles
// of 32 instead of 3.5 of 64).  A wave group (128 threads) covers the TY x TX/4 quads of the tile either way.
template <int TY_, int TX_>
struct MMGeo {
    static constexpr int TY = TY_, TX = TX_, TXQ = TX_ / 4;
    // ring row pitch (floats).  TX = 64: 74 = 2 (mod 4) -> the 8-byte reads of the two rows of a 32-lane group hit disjoint banks.
    // TX = 32: a 32-lane group reads FOUR rows (8 quads each); pitch 48 puts consecutive rows 48 = -16 banks apart and odd rows are skewed by
    // two floats, so the four rows' (4q, 4q+1) pairs tile the 64 banks exactly (round 5: pitch 42, 2-way conflicts on every read, 13.1 M
    // conflict cycles per launch)
    // TX = 84 (single-pass pooled kernel, 6 x 84 tiles): 21 quads per row are mapped as 16 + 5 (quad_of): the main part as for TX = 64, the five
    // trailing quads of the six rows as one 32-lane group -- rows of equal parity are 2 * 106 = 20 (mod 64) banks apart, their 20-bank spans disjoint
    static constexpr int RP = TX_ == 32 ? 48 : (TX_ == 84 ? 106 : TX_ + 10);
    static constexpr int SK = TX_ == 32 ? 2 : 0;
    __device__ static constexpr int rowbase(int r) { return r * RP + SK * (r & 1); }      // (row + 2k keeps its skew: the stencil's row offsets are even)
    static constexpr int ROWS = TY_ + 6;          // ring row r = volume row clamp(y0 - 3 + r); ring column k = volume column clamp(x0 - 5 + k)
    static constexpr int PLANE = ROWS * RP + SK;
    static constexpr int LQ = TXQ + 2;            // loader quads per row: columns x0-4 .. x0+TX+3
    static_assert(TY_ * TXQ <= 128 && TY_ * TX_ <= MM_NT && ROWS * LQ <= MM_NT, "tile shape");
    // (row, quad) of thread t128 of a wave group; `on` false: the thread has no quad
    __device__ static void quad_of(int t128, int& row, int& q, bool& on) {
        if (TX_ == 84) {
            const int u = t128 - 96;
            on = u < 30;
            row = t128 < 96 ? t128 >> 4 : (on ? u / 5 : 0);
            q = t128 < 96 ? (t128 & 15) : (on ? 16 + u % 5 : 0);
        } else { row = t128 / TXQ; q = t128 % TXQ; on = true; }
    }
};
__device__ constexpr int MM_SETS[MM_NS][MM_CPS] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {9, 10, 11}};   // pre-permutation channels of the wave groups

struct MMLoader {
    bool on, fast;
    int gy, gx;               // clamped row, first column (may be < 0)
    float* dst;               // ring position of the first element inside a slot
    float4 pre;
};

__device__ __forceinline__ void mm_fetch(const float* __restrict__ img, int H, int W, int D, const MMLoader& L, int l, float4& v) {
    if (!L.on) return;
    const int gz = clampi(l, 0, H - 1);
    const float* row = img + ((size_t)gz * W + L.gy) * D;
    if (L.fast) v = *reinterpret_cast<const float4*>(row + L.gx);
    else {
        v.x = row[clampi(L.gx, 0, D - 1)];
        v.y = row[clampi(L.gx + 1, 0, D - 1)];
        v.z = row[clampi(L.gx + 2, 0, D - 1)];
        v.w = row[clampi(L.gx + 3, 0, D - 1)];
    }
}
template <int PLANE>
__device__ __forceinline__ void mm_publish(const MMLoader& L, float* ring, int l, const float4& v) {
    if (!L.on) return;
    float* p = L.dst + ((l + 12) % 6) * PLANE;           // odd index: b32 + b64 + b32
    // the 8-byte store needs an even-aligned register pair and the middle of a 16-byte load is an odd one: without re-defining the two
    // values HERE the compiler carries the pair through the loop and copies into it right after the load is issued -- i.e. it waits
    // for the prefetched plane in the step that requested it (same fix as in boxmarch.hip)
    float my = v.y, mz = v.z;
    asm volatile("" : "+v"(my), "+v"(mz));
    p[0] = v.x;
    const f32x2 mid = {my, mz};
    lds_store2(p + 1, mid);
    p[3] = v.w;
}

// one plane of one wave group: taps of squared-difference plane zc -> running sums; EMIT: output plane gz is due
template <typename G, int SET, bool EMIT>
__device__ __forceinline__ void mm_box_step(const float* __restrict__ ring, float* __restrict__ X, float* __restrict__ out, size_t V /* channel stride of `out` */,
                                            size_t lin, bool store_ok, int zc, const int (&rowoff)[3], int colbase, bool left,
                                            bool right, int row, int q, float (&A)[MM_CPS][4], float (&P)[MM_CPS][4], bool x_ok = true) {
    constexpr MindOffsets MO{};
    const float* sb[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) sb[o] = ring + ((zc + 2 * (o - 1) + 12) % 6) * G::PLANE + colbase;
#pragma unroll
    for (int k = 0; k < MM_CPS; ++k) {
        const int c = MM_SETS[SET][k];
        float t[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float* p1 = sb[MO.o1[c][0] + 1] + rowoff[i] + 2 * MO.o1[c][1] * G::RP + 2 * MO.o1[c][2];
            const float* p2 = sb[MO.o2[c][0] + 1] + rowoff[i] + 2 * MO.o2[c][1] * G::RP + 2 * MO.o2[c][2];
            const f32x2 a0 = lds_load2(p1), a1 = lds_load2(p1 + 2), a2 = lds_load2(p1 + 4);
            const f32x2 b0 = lds_load2(p2), b1 = lds_load2(p2 + 2), b2 = lds_load2(p2 + 4);
            float d0 = a0.x - b0.x, d1 = a0.y - b0.y, d2 = a1.x - b1.x, d3 = a1.y - b1.y, d4 = a2.x - b2.x, d5 = a2.y - b2.y;
            t[i][0] = d0 * d0; t[i][1] = d1 * d1; t[i][2] = d2 * d2; t[i][3] = d3 * d3; t[i][4] = d4 * d4; t[i][5] = d5 * d5;
            // the box clamps the POSITION first: column -1 is column 0, column D is column D-1
            t[i][0] = left ? t[i][1] : t[i][0];
            t[i][5] = right ? t[i][4] : t[i][5];
        }
        float dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o = A[k][j], a = P[k][j], pn = t[0][j];        // pn: 0.0 + t = t exactly (squares are never -0.0)
            if (EMIT) o += t[0][j];
            a += t[0][j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    if (i == 0 && cc == 0) continue;
                    if (EMIT) o += t[i][j + cc];
                    a += t[i][j + cc];
                    pn += t[i][j + cc];
                }
            dv[j] = EMIT ? div_exact<27>(o) : 0.0f;
            A[k][j] = a;
            P[k][j] = pn;
        }
        if (EMIT) {
            const f32x4 r = {dv[0], dv[1], dv[2], dv[3]};
            if (x_ok) lds_store4(X + c * (G::TY * G::TX) + row * G::TX + 4 * q, r);
            if (store_ok) *reinterpret_cast<float4*>(out + (size_t)MIND_INV[c] * V + lin) = make_float4(r.x, r.y, r.z, r.w);
        }
    }
}

template <typename G, int SET>
__device__ __forceinline__ void mm_run(const float* __restrict__ img, float* __restrict__ out, MindStats* __restrict__ st, float* ring,
                                       float* X, double (*red)[MM_NT / 64], int H, int W, int D, int z0, int z1, int y0, int x0,
                                       MMLoader& L, const MindRawLayout& lay) {
    const int tid = threadIdx.x, t128 = tid & 127;
    const int row = t128 / G::TXQ, q = t128 % G::TXQ;
    const int gy = y0 + row, gx0 = x0 + 4 * q;
    const size_t V = (size_t)H * W * D;
    // blocked output (see MindRawLayout): the part of the index that does not depend on the plane
    const int gyc = gy < W ? gy : 0, gxc = gx0 < D ? gx0 : 0;
    const size_t out_a = lay.T ? ((size_t)(gyc / lay.T) * lay.ntx + (size_t)(gxc / 24)) * lay.tile_floats + (size_t)(gyc % lay.T) * 24 + (size_t)(gxc % 24) : 0;
    const size_t out_cs = lay.T ? lay.chan_floats : V;
    int rowoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rowoff[i] = G::rowbase(min(clampi(gy + i - 1, 0, W - 1) - y0 + 3, G::ROWS - 3));   // (min: rows beyond the volume of an
                                                                                                                // overhanging tile stay inside the ring)
    const int colbase = 4 * q + 4;
    const bool left = gx0 == 0, right = gx0 + 4 == D;
    const bool store_ok = gy < W && gx0 < D;
    float A[MM_CPS][4], P[MM_CPS][4];
#pragma unroll
    for (int k = 0; k < MM_CPS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { A[k][j] = 0.0f; P[k][j] = 0.0f; }

    // statistics: every thread owns one voxel of the plane
    const int srow = tid / G::TX, scol = tid % G::TX;
    const double m1 = st->m1, m2 = st->m2, m3 = st->m3;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const size_t tail_from = (V / 32) * 32;

    // variance statistics of the plane whose 12 patch SSDs sit in exchange buffer `Xb`
    auto stats = [&](const float* Xb, int gz) {
        float r[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c] = Xb[c * (G::TY * G::TX) + srow * G::TX + scol];
        const int sy = y0 + srow, sx = x0 + scol;
        float mn = r[0];
#pragma unroll
        for (int c = 1; c < 12; ++c) mn = fminf(mn, r[c]);
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c] = r[c] - mn;
        const size_t vl = ((size_t)gz * W + sy) * D + sx;
        float sum;
        if (vl >= tail_from) sum = outer_sum_ilp<12>(r);          // only the last < 32 voxels of the volume
        else sum = cascade_seq<12>(r);
        const float var = fdiv(sum, 12.0f);
        if (sy < W && sx < D) {
            const double v = (double)var;
            const double q1 = (v + m1) - m1, r1 = v - q1;
            const double q2 = (r1 + m2) - m2, r2 = r1 - q2;
            const double q3 = (r2 + m3) - m3;
            a1 += q1; a2 += q2; a3 += q3;
        }
    };
    // One barrier per plane: step s publishes ring plane zc+2 into the slot that no reader of step s-1 touches (6 slots), the
    // box pass writes exchange buffer s & 1 and the statistics read the buffer of the previous step.
    constexpr int XSZ = 12 * G::TY * G::TX;
    const int nsteps = (z1 - z0) + 2;
    int zc_prev = clampi(z0 - 1, 0, H - 1);
    for (int s = 0; s < nsteps; ++s) {
        const int zc = clampi(z0 - 1 + s, 0, H - 1);
        if (zc != zc_prev) mm_publish<G::PLANE>(L, ring, zc + 2, L.pre);
        mm_fetch(img, H, W, D, L, zc + 3, L.pre);
        zc_prev = zc;
        cvx_barrier();
        const int gz = z0 + s - 2;
        const int gzc = gz < 0 ? 0 : gz;
        const size_t lin = lay.T ? out_a + ((size_t)(gzc / lay.T) * lay.nty * lay.ntx * lay.tile_floats + (size_t)(gzc % lay.T) * lay.T * 24)
                                 : ((size_t)gzc * W + (gy < W ? gy : 0)) * D + (gx0 < D ? gx0 : 0);
        float* Xs = X + (s & 1) * XSZ;
        if (s >= 2) mm_box_step<G, SET, true>(ring, Xs, out, out_cs, lin, store_ok, zc, rowoff, colbase, left, right, row, q, A, P);
        else mm_box_step<G, SET, false>(ring, Xs, out, out_cs, lin, store_ok, zc, rowoff, colbase, left, right, row, q, A, P);
        if (s >= 3) stats(X + ((s - 1) & 1) * XSZ, gz - 1);
    }
    cvx_barrier();
    stats(X + ((nsteps - 1) & 1) * XSZ, z1 - 1);
    // every partial sum is exactly representable -> any reduction order gives the same bits
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o); a3 += __shfl_down(a3, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a1; red[1][tid >> 6] = a2; red[2][tid >> 6] = a3; }
    cvx_barrier();
    if (tid == 0) {
        for (int i = 1; i < MM_NT / 64; ++i) { a1 += red[0][i]; a2 += red[1][i]; a3 += red[2][i]; }
        atomicAdd(&st->a1, a1); atomicAdd(&st->a2, a2); atomicAdd(&st->a3, a3);
    }
}

template <typename G>
__global__ __launch_bounds__(MM_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mind_march(const float* __restrict__ img, int H, int W, int D, int zc_len, int nzc, int nyt,
                                                      int nxt, MindStats* __restrict__ st, float* __restrict__ out, MindRawLayout lay) {
    __shared__ __attribute__((aligned(16))) float ring[6 * G::PLANE];      // 5 live planes + the one being replaced
    __shared__ __attribute__((aligned(16))) float X[2 * 12 * G::TY * G::TX];   // double buffered
    __shared__ double red[3][MM_NT / 64];
    // XCD-aware order: XCD k (workgroups k, k+8, ..) takes the k-th contiguous run of (z chunk, y tile, x tile) triples
    const int nblk = nzc * nyt * nxt;
    const int b = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (b >= nblk) return;
    const int xi = b % nxt, yi = (b / nxt) % nyt, zi = b / (nxt * nyt);
    const int x0 = xi * G::TX, y0 = yi * G::TY, z0 = zi * zc_len, z1 = min(H, z0 + zc_len);
    const int tid = threadIdx.x;

    MMLoader L;
    L.on = tid < G::ROWS * G::LQ;
    const int lr = tid / G::LQ, lq = tid - lr * G::LQ;
    L.gy = clampi(y0 - 3 + lr, 0, W - 1);
    L.gx = x0 - 4 + 4 * lq;
    L.fast = L.gx >= 0 && L.gx + 3 <= D - 1;
    L.dst = ring + G::rowbase(lr) + 4 * lq + 1;
    L.pre = make_float4(0.f, 0.f, 0.f, 0.f);
    // ring around the first centre plane
    const int zc0 = clampi(z0 - 1, 0, H - 1);
    {
        float4 v[5];
#pragma unroll
        for (int o = 0; o < 5; ++o) { v[o] = make_float4(0.f, 0.f, 0.f, 0.f); mm_fetch(img, H, W, D, L, zc0 - 2 + o, v[o]); }
#pragma unroll
        for (int o = 0; o < 5; ++o) mm_publish<G::PLANE>(L, ring, zc0 - 2 + o, v[o]);
    }
    // (the first barrier of the march makes the ring visible)
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (grp == 0) mm_run<G, 0>(img, out, st, ring, X, red, H, W, D, z0, z1, y0, x0, L, lay);
    else if (grp == 1) mm_run<G, 1>(img, out, st, ring, X, red, H, W, D, z0, z1, y0, x0, L, lay);
    else if (grp == 2) mm_run<G, 2>(img, out, st, ring, X, red, H, W, D, z0, z1, y0, x0, L, lay);
    else mm_run<G, 3>(img, out, st, ring, X, red, H, W, D, z0, z1, y0, x0, L, lay);
}

// ---- the pipeline's descriptor in ONE pass: stencil + normalisation + exp + both stride poolings (launch_mind_pooled, mind.hip) -----------------------
// The registration pipeline consumes MIND-SSC only through avg_pool3d(., g, stride g) (convex_adam_MIND.py:118-119, 149-150).  The two-pass path
// writes the 12 raw patch SSDs of every voxel (48 B) and reads them back because the variance clamp needs the GLOBAL mean of the per-voxel
// variances (convex_adam_utils.py:60-62).  Here the marching kernel normalises with the UNCLAMPED variance right away -- which is what the
// reference computes wherever the clamp does not bind -- and pools in ATen's raster order while it marches; it records per GA^3 block the smallest
// variance of a voxel that is not all-zero, the largest variance, and whether a voxel has twelve zero distances (exp(-0 / v) = 1 for every
// positive clamp result: such voxels need no mean).  When the mean is known, k_mind_repair (mind.hip) recomputes the pooled cells of exactly those
// blocks in which the clamp binds on some voxel, from the image, with the clamped variance.  Same bits as the two-pass path for every input.
//
// Step of the march (two barriers):  A | prefetch image plane, pool plane gz-1 from E, stencil plane gz -> X | B | normalise plane gz: X -> E.
// A 6 x 84 tile (windows 6 with 2 / 3 / 6) or an 8 x 64 tile (windows 4 and 2); z chunks are multiples of the larger window, so every block and
// every pooling window belongs to ONE workgroup: plain stores, no atomics on the outputs, nothing to initialise.
#ifdef CVX_RACE_JITTER
__device__ __forceinline__ void ms_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); cvx_jitter(); __builtin_amdgcn_s_barrier(); cvx_jitter(); asm volatile("" ::: "memory"); }
#else
// LDS traffic only: the image prefetch and the pooled stores in flight are NOT waited for (__syncthreads would drain them at every barrier)
__device__ __forceinline__ void ms_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#endif

// partial sum of one plane of a GW^3 window: GW x GW taps in raster order added to `acc` (E: [channel][TY][TX] of the tile's plane)
template <typename G, int GW>
__device__ __forceinline__ void ms_chain_plane(const float* __restrict__ base, float& acc) {
    if (GW % 2 == 0) {
        f32x2 v[GW][GW / 2];
#pragma unroll
        for (int y = 0; y < GW; ++y)
#pragma unroll
            for (int x = 0; x < GW / 2; ++x) v[y][x] = lds_load2(base + y * G::TX + 2 * x);
#pragma unroll
        for (int y = 0; y < GW; ++y)
#pragma unroll
            for (int x = 0; x < GW / 2; ++x) { acc += v[y][x].x; acc += v[y][x].y; }
    } else {
        float v[GW][GW];
#pragma unroll
        for (int y = 0; y < GW; ++y)
#pragma unroll
            for (int x = 0; x < GW; ++x) v[y][x] = base[y * G::TX + x];
#pragma unroll
        for (int y = 0; y < GW; ++y)
#pragma unroll
            for (int x = 0; x < GW; ++x) acc += v[y][x];
    }
}

struct MSArgs {
    const float* img;
    float* out1;              // GA pooling, planar [12][H/GA][W/GA][D/GA]
    void* out2;               // GB pooling: planar floats, or feature records (REC; half precision if rec_half), or null
    MindStats* st;
    unsigned* blk;            // [3][nbz nby nbx]: min variance of the not-all-zero voxels (float bits), max variance (bits), any all-zero voxel
    int H, W, D, zc_len, nzc, nyt, nxt, rec_half, nby, nbx;
};

template <int GW, typename G> struct MSPlan {
    static constexpr int ncy = G::TY / GW, ncx = G::TX / GW, ncell = ncy * ncx;
    static_assert(G::TY % GW == 0 && G::TX % GW == 0, "windows tile the tile");
};
constexpr int ms_cdiv(int a, int b) { return (a + b - 1) / b; }

// (one instance for the four wave groups: only the stencil step depends on the group's channel set -- a wave-uniform switch around mm_box_step --, the
// normalisation and the pooling are shared code: 66 KB of instructions per kernel became 35 KB, inside the 64 KB instruction cache of a CU pair)
template <typename G, int GA, int GB, bool REC>
__device__ __forceinline__ void ms_run(const MSArgs& a, float* ring, float* X, float* E, unsigned* cst, double (*red)[MM_NT / 64], int z0, int z1, int y0, int x0, MMLoader& L, int grp) {
    const int H = a.H, W = a.W, D = a.D;
    const int tid = threadIdx.x, t128 = tid & 127;
    int row, q; bool qon;
    G::quad_of(t128, row, q, qon);
    const int gy = y0 + row, gx0 = x0 + 4 * q;
    int rowoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rowoff[i] = G::rowbase(min(clampi(gy + i - 1, 0, W - 1) - y0 + 3, G::ROWS - 3));
    const int colbase = 4 * q + 4;
    const bool left = gx0 == 0, right = gx0 + 4 == D;
    float A[MM_CPS][4], P[MM_CPS][4];
#pragma unroll
    for (int k = 0; k < MM_CPS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { A[k][j] = 0.0f; P[k][j] = 0.0f; }

    // normalisation: one voxel of the plane per thread
    constexpr int NVOX = G::TY * G::TX;
    const bool von = tid < NVOX;
    const int srow = von ? tid / G::TX : 0, scol = von ? tid % G::TX : 0;
    const int sy = y0 + srow, sx = x0 + scol;
    const bool vin = von && sy < W && sx < D;
    const double m1 = a.st->m1, m2 = a.st->m2, m3 = a.st->m3;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const size_t V = (size_t)H * W * D, tail_from = (V / 32) * 32;
    typedef MSPlan<GA, G> PA;
    typedef MSPlan<GB, G> PB;
    const int cellA = (srow / GA) * PA::ncx + scol / GA;
    unsigned vmin = 0x7f800000u, vmax = 0u, anyz = 0u;

    // pooling units of this thread.  GA: planar chains (channel, cell); GB: planar chains, or record units (4 channels of a cell) when REC.
    // The GB units are dealt from the last thread downwards: the few long GA chains and the many short GB ones share as few wavefronts as possible.
    constexpr int NA = 12 * PA::ncell, KA = ms_cdiv(NA, MM_NT);
    constexpr int NB = REC ? 3 * PB::ncell : 12 * PB::ncell, KB = ms_cdiv(NB, MM_NT), CB = REC ? 4 : 1;
    const int HoA = H / GA, WoA = W / GA, DoA = D / GA, HoB = H / GB, WoB = W / GB, DoB = D / GB;
    int eA[KA], eB[KB];               // LDS offset of the window's first tap (-1: no unit)
    long long oA[KA], oB[KB];         // output offset without the plane term (-1: outside the pooled extent)
    float accA[KA], accB[KB][CB];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int u = tid + k * MM_NT;
        const bool on = u < NA;
        const int c = on ? u / PA::ncell : 0, cell = on ? u % PA::ncell : 0, cy = cell / PA::ncx, cx = cell % PA::ncx;
        eA[k] = on ? (c * G::TY + cy * GA) * G::TX + cx * GA : -1;
        const int oy = y0 / GA + cy, ox = x0 / GA + cx;
        oA[k] = (on && oy < WoA && ox < DoA) ? (long long)c * HoA * WoA * DoA + (long long)oy * DoA + ox : -1;
        accA[k] = 0.0f;
    }
    const bool has2 = a.out2 != nullptr;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int u = (MM_NT - 1 - tid) + k * MM_NT;
        const bool on = has2 && u < NB;
        const int c = on ? u / PB::ncell : 0, cell = on ? u % PB::ncell : 0, cy = cell / PB::ncx, cx = cell % PB::ncx;     // (c: channel, or record chunk)
        eB[k] = on ? ((REC ? 4 * c : c) * G::TY + cy * GB) * G::TX + cx * GB : -1;
        const int oy = y0 / GB + cy, ox = x0 / GB + cx;
        const long long chunk = REC ? (long long)HoB * WoB * DoB + 1 : (long long)HoB * WoB * DoB;
        oB[k] = (on && oy < WoB && ox < DoB) ? (long long)c * chunk + (long long)oy * DoB + ox : -1;
#pragma unroll
        for (int j = 0; j < CB; ++j) accB[k][j] = 0.0f;
    }
    const int nblk_plane = a.nby * a.nbx;

    // pooling of plane gz (E holds its descriptor), and the block statistics once a block is complete
    auto pool = [&](int gz) {
        const int rz = gz - z0;
        const bool lastA = rz % GA == GA - 1, lastB = rz % GB == GB - 1;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            if (eA[k] < 0) continue;
            ms_chain_plane<G, GA>(E + eA[k], accA[k]);
            if (lastA) {
                const float val = fdiv(accA[k], (float)(GA * GA * GA));
                accA[k] = 0.0f;
                if (oA[k] >= 0) a.out1[oA[k] + (long long)(gz / GA) * WoA * DoA] = val;
            }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (eB[k] < 0) continue;
#pragma unroll
            for (int j = 0; j < CB; ++j) ms_chain_plane<G, GB>(E + eB[k] + j * (G::TY * G::TX), accB[k][j]);
            if (lastB) {
                float val[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) { val[j] = fdiv(accB[k][j], (float)(GB * GB * GB)); accB[k][j] = 0.0f; }
                if (oB[k] >= 0) {
                    const long long at = oB[k] + (long long)(gz / GB) * WoB * DoB;
                    if (!REC) static_cast<float*>(a.out2)[at] = val[0];
                    else if (a.rec_half) {
                        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                        const h16x4 o = {(_Float16)val[0], (_Float16)val[CB > 1 ? 1 : 0], (_Float16)val[CB > 2 ? 2 : 0], (_Float16)val[CB > 3 ? 3 : 0]};     // round to nearest even
                        static_cast<uint2*>(a.out2)[at] = __builtin_bit_cast(uint2, o);
                    } else static_cast<float4*>(a.out2)[at] = make_float4(val[0], val[CB > 1 ? 1 : 0], val[CB > 2 ? 2 : 0], val[CB > 3 ? 3 : 0]);
                }
            }
        }
        if ((lastA || gz == z1 - 1) && tid < PA::ncell) {
            const int by = y0 / GA + tid / PA::ncx, bx = x0 / GA + tid % PA::ncx;
            if (by < a.nby && bx < a.nbx) {
                const size_t b = (size_t)(gz / GA) * nblk_plane + (size_t)by * a.nbx + bx, nb = (size_t)ms_cdiv(H, GA) * nblk_plane;
                a.blk[b] = cst[tid]; a.blk[nb + b] = cst[PA::ncell + tid]; a.blk[2 * nb + b] = cst[2 * PA::ncell + tid];
            }
            cst[tid] = 0x7f800000u; cst[PA::ncell + tid] = 0u; cst[2 * PA::ncell + tid] = 0u;
        }
    };
    // normalisation of plane gz: X (12 patch SSDs per voxel, pre-permutation order) -> E (descriptor, final channel order)
    auto norm = [&](int gz) {
        if (!von) return;
        float r[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c] = X[c * NVOX + tid];
        float mn = r[0];
#pragma unroll
        for (int c = 1; c < 12; ++c) mn = fminf(mn, r[c]);
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c] = r[c] - mn;
        const size_t vl = ((size_t)gz * W + sy) * D + sx;
        float sum;
        if (vl >= tail_from) sum = outer_sum_ilp<12>(r);          // only the last < 32 voxels of the volume
        else sum = cascade_seq<12>(r);
        const float var = fdiv(sum, 12.0f);
        const bool allz = sum == 0.0f;                             // a sum of non-negative terms: zero iff every term is
        if (vin) {
            const double v = (double)var;
            const double q1 = (v + m1) - m1, r1 = v - q1;
            const double q2 = (r1 + m2) - m2, r2 = r1 - q2;
            const double q3 = (r2 + m3) - m3;
            a1 += q1; a2 += q2; a3 += q3;
            const unsigned vb = __float_as_uint(var);
            anyz |= allz ? 1u : 0u;
            vmin = allz ? vmin : min(vmin, vb);
            vmax = max(vmax, vb);
        }
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const float e = cvx_expf(-fdiv(r[c], var));
            E[MIND_INV[c] * NVOX + tid] = allz ? 1.0f : e;
        }
        const int rz = gz - z0;
        if (rz % GA == GA - 1 || gz == z1 - 1) {
            if (vin) {
                atomicMin(&cst[cellA], vmin);
                atomicMax(&cst[PA::ncell + cellA], vmax);
                if (anyz) atomicOr(&cst[2 * PA::ncell + cellA], 1u);
            }
            vmin = 0x7f800000u; vmax = 0u; anyz = 0u;
        }
    };

    const int nsteps = (z1 - z0) + 2;
    int zc_prev = clampi(z0 - 1, 0, H - 1);
    for (int s = 0; s < nsteps; ++s) {
        const int zc = clampi(z0 - 1 + s, 0, H - 1);
        if (zc != zc_prev) mm_publish<G::PLANE>(L, ring, zc + 2, L.pre);
        zc_prev = zc;
        ms_barrier();                                                  // A
        mm_fetch(a.img, H, W, D, L, zc + 3, L.pre);
        const int gz = z0 + s - 2;
        if (s >= 3) pool(gz - 1);
        if (s >= 2) {
            if (grp == 0) mm_box_step<G, 0, true>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else if (grp == 1) mm_box_step<G, 1, true>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else if (grp == 2) mm_box_step<G, 2, true>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else mm_box_step<G, 3, true>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
        } else {
            if (grp == 0) mm_box_step<G, 0, false>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else if (grp == 1) mm_box_step<G, 1, false>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else if (grp == 2) mm_box_step<G, 2, false>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
            else mm_box_step<G, 3, false>(ring, X, nullptr, 0, 0, false, zc, rowoff, colbase, left, right, row, q, A, P, qon);
        }
        ms_barrier();                                                  // B
        if (s >= 2) norm(gz);
    }
    ms_barrier();
    pool(z1 - 1);
    // every partial sum is exactly representable -> any reduction order gives the same bits
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o); a3 += __shfl_down(a3, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a1; red[1][tid >> 6] = a2; red[2][tid >> 6] = a3; }
    ms_barrier();
    if (tid == 0) {
        for (int i = 1; i < MM_NT / 64; ++i) { a1 += red[0][i]; a2 += red[1][i]; a3 += red[2][i]; }
        atomicAdd(&a.st->a1, a1); atomicAdd(&a.st->a2, a2); atomicAdd(&a.st->a3, a3);
    }
}

template <int GA> struct MSTile { typedef MMGeo<8, 64> G; };
template <> struct MSTile<6> { typedef MMGeo<6, 84> G; };

template <int GA, int GB, bool REC>
__global__ __launch_bounds__(MM_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mind_march_pool(MSArgs a) {
    typedef typename MSTile<GA>::G G;
    __shared__ __attribute__((aligned(16))) float ring[6 * G::PLANE];
    __shared__ __attribute__((aligned(16))) float X[12 * G::TY * G::TX];
    __shared__ __attribute__((aligned(16))) float E[12 * G::TY * G::TX];
    __shared__ unsigned cst[3 * MSPlan<GA, G>::ncell];
    __shared__ double red[3][MM_NT / 64];
    const int nblk = a.nzc * a.nyt * a.nxt;
    const int b = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (b >= nblk) return;
    const int xi = b % a.nxt, yi = (b / a.nxt) % a.nyt, zi = b / (a.nxt * a.nyt);
    const int x0 = xi * G::TX, y0 = yi * G::TY, z0 = zi * a.zc_len, z1 = min(a.H, z0 + a.zc_len);
    const int tid = threadIdx.x;
    if (tid < MSPlan<GA, G>::ncell) { cst[tid] = 0x7f800000u; cst[MSPlan<GA, G>::ncell + tid] = 0u; cst[2 * MSPlan<GA, G>::ncell + tid] = 0u; }

    MMLoader L;
    L.on = tid < G::ROWS * G::LQ;
    const int lr = tid / G::LQ, lq = tid - lr * G::LQ;
    L.gy = clampi(y0 - 3 + lr, 0, a.W - 1);
    L.gx = x0 - 4 + 4 * lq;
    L.fast = L.gx >= 0 && L.gx + 3 <= a.D - 1;
    L.dst = ring + G::rowbase(lr) + 4 * lq + 1;
    L.pre = make_float4(0.f, 0.f, 0.f, 0.f);
    const int zc0 = clampi(z0 - 1, 0, a.H - 1);
    {
        float4 v[5];
#pragma unroll
        for (int o = 0; o < 5; ++o) { v[o] = make_float4(0.f, 0.f, 0.f, 0.f); mm_fetch(a.img, a.H, a.W, a.D, L, zc0 - 2 + o, v[o]); }
#pragma unroll
        for (int o = 0; o < 5; ++o) mm_publish<G::PLANE>(L, ring, zc0 - 2 + o, v[o]);
    }
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 7);
    ms_run<G, GA, GB, REC>(a, ring, X, E, cst, red, z0, z1, y0, x0, L, grp);
}

template <int GA, int GB, bool REC>
static void launch_ms(const MSArgs& a0, hipStream_t s) {
    typedef typename MSTile<GA>::G G;
    MSArgs a = a0;
    a.nyt = cdiv(a.W, G::TY); a.nxt = cdiv(a.D, G::TX);
    // two workgroups per CU (LDS and registers); z chunks are multiples of the larger window.  Option ms_zlen fixes the chunk length; the default
    // takes the shortest chunks whose count fits the resident set (mm_slots), at least two windows long
    int zc_len = (int)options().ms_zlen;
    if (zc_len <= 0) {
        int nzc = (int)options().mm_slots / (a.nyt * a.nxt);
        if (nzc < 1) nzc = 1;
        zc_len = cdiv(cdiv(a.H, nzc), GA) * GA;
        if (zc_len < 12) zc_len = 12;
    }
    zc_len = cdiv(zc_len, GA) * GA;
    a.zc_len = zc_len;
    a.nzc = cdiv(a.H, zc_len);
    const unsigned grid = (unsigned)((a.nzc * a.nyt * a.nxt + 7) / 8 * 8);
    hipLaunchKernelGGL((k_mind_march_pool<GA, GB, REC>), dim3(grid), dim3(MM_NT), 0, s, a);
}

bool mind_single_supported(int ga, int gb) {
    return (ga == 6 && (gb == 2 || gb == 3 || gb == 6)) || (ga == 4 && (gb == 2 || gb == 4)) || (ga == 2 && gb == 2);
}
// ga >= gb, gb divides ga (mind_single_supported); out2 null: single pooling; records: out2 receives feature records (1 float32, 2 half precision)
void launch_mind_march_pool(const float* img, int H, int W, int D, int ga, float* out1, int gb, void* out2, int records, MindStats* st, unsigned* blk, hipStream_t s) {
    MSArgs a = {img, out1, out2, st, blk, H, W, D, 0, 0, 0, 0, records == 2 ? 1 : 0, cdiv(W, ga), cdiv(D, ga)};
    const bool rec = records != 0;
#define CVX_MS(GA, GB) do { if (rec) launch_ms<GA, GB, true>(a, s); else launch_ms<GA, GB, false>(a, s); } while (0)
    if (ga == 6 && gb == 2) CVX_MS(6, 2);
    else if (ga == 6 && gb == 3) CVX_MS(6, 3);
    else if (ga == 6 && gb == 6) CVX_MS(6, 6);
    else if (ga == 4 && gb == 2) CVX_MS(4, 2);
    else if (ga == 4 && gb == 4) CVX_MS(4, 4);
    else CVX_MS(2, 2);
#undef CVX_MS
}

bool mind_march_supported(const float* img, const float* out, int H, int W, int D, int radius, int dilation) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    (void)H; (void)W;
    return radius == 1 && dilation == 2 && (D & 3) == 0 && al(img) && al(out);
}

template <typename G>
static void launch_mind_march_g(const float* img, int H, int W, int D, MindStats* st, float* out, hipStream_t s, const MindRawLayout& lay) {
    const int nyt = cdiv(W, G::TY), nxt = cdiv(D, G::TX);
    // two workgroups per CU (register bound): at most 512 workgroups so that all of them are resident at once -- a second,
    // partly filled round costs more than the longer chunks; chunks of at least 8 planes keep the 2-plane fill below 25 %
    const int slots = (int)options().mm_slots;
    int nzc = slots / (nyt * nxt);
    if (nzc < 1) nzc = 1;
    int zc_len = cdiv(H, nzc);
    if (zc_len < 8) zc_len = 8;
    nzc = cdiv(H, zc_len);
    const unsigned grid = (unsigned)((nzc * nyt * nxt + 7) / 8 * 8);
    hipLaunchKernelGGL(k_mind_march<G>, dim3(grid), dim3(MM_NT), 0, s, img, H, W, D, zc_len, nzc, nyt, nxt, st, out, lay);
}

void launch_mind_march(const float* img, int H, int W, int D, MindStats* st, float* out, hipStream_t s, MindRawLayout lay) {
    const int force = (int)options().mm_tx;
    const int rem = D % 64;
    const bool narrow = force ? force == 32 : (rem != 0 && rem <= 32);
    if (narrow) launch_mind_march_g<MMGeo<16, 32>>(img, H, W, D, st, out, s, lay);
    else launch_mind_march_g<MMGeo<8, 64>>(img, H, W, D, st, out, s, lay);
}

}  // namespace cvx
