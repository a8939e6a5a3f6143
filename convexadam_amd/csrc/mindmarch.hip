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
    static constexpr int RP = TX_ == 32 ? 48 : TX_ + 10;
    static constexpr int SK = TX_ == 32 ? 2 : 0;
    __device__ static constexpr int rowbase(int r) { return r * RP + SK * (r & 1); }      // (row + 2k keeps its skew: the stencil's row offsets are even)
    static constexpr int ROWS = TY_ + 6;          // ring row r = volume row clamp(y0 - 3 + r); ring column k = volume column clamp(x0 - 5 + k)
    static constexpr int PLANE = ROWS * RP + SK;
    static constexpr int LQ = TXQ + 2;            // loader quads per row: columns x0-4 .. x0+TX+3
    static_assert(TY_ * TXQ == 128 && ROWS * LQ <= MM_NT, "tile shape");
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
                                            bool right, int row, int q, float (&A)[MM_CPS][4], float (&P)[MM_CPS][4]) {
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
            lds_store4(X + c * (G::TY * G::TX) + row * G::TX + 4 * q, r);
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
