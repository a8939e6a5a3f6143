// mind.hip -- MIND-SSC descriptors (reference: src/convexAdam/convex_adam_utils.py:24-68).
//
// out_c(x) = exp( -(D_c(x) - min_c D_c(x)) / clamp(mean_c(D - min), 0.001*mu, 1000*mu) )
// D_c(x)   = box_{(2r+1)^3}[ (I(P + o1_c*d) - I(P + o2_c*d))^2 ](x)      replicate borders twice
// mu       = mean over the volume of mean_c(D - min)
//
// Two passes (the global mean mu is a grid-wide dependency), after a min/max pre-pass that sizes the exact accumulation:
//   stencil pass  : the 12 patch-SSDs D_c(x) -> out (raw), per-voxel variance -> order-independent exact sum (three
//                   power-of-two split grids, double atomics)   [reads V*4, writes 12*V*4 B]
//                   k_mind_march (mindmarch.hip) for radius 1 / dilation 2 / rows of 4k voxels, the tiled k_mind<R, TX> below otherwise
//   k_mind_finish : streaming, in place -- min, mean, clamp, exp per voxel        [reads + writes 12*V*4 B]
//   k_mind_finish_pool : the pipeline's variant of the second pass -- normalisation and both stride poolings in one go
// (recomputing the stencil in the second pass instead costs 2.2 x the time of streaming the 12 channels once)
// Tiled stencil: 4 x 8 x 64 voxels (H x W x D) per 512-thread workgroup, image tile with halo r+d staged in
// LDS once, squared-difference tile per channel double-buffered in LDS, each thread owns 4
// consecutive D-voxels and keeps 12 x 4 results in registers.  Roofline: HBM (385 MB per image when
// the full-resolution descriptor is materialised); the 27-tap raster-order sums (ATen avg_pool3d
// order, one exact division) make it VALU/LDS-bound in practice -- see DESIGN.md.
#include "cvx_common.h"
#include "mind_common.h"

namespace cvx {

constexpr int TZ = 4, TY = 8, RUN = 4;

// ---- min / max of the image (bound for the exact accumulation) -----------------------------------
__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ img, size_t V, float* part) {
    float mn = INFINITY, mx = -INFINITY;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if ((reinterpret_cast<uintptr_t>(img) & 15) == 0) {          // 16-byte loads over the aligned bulk
        const size_t n4 = V / 4;
        const float4* img4 = reinterpret_cast<const float4*>(img);
        for (size_t i = tid; i < n4; i += nthr) {
            const float4 v = img4[i];
            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
            mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
        done = n4 * 4;
    }
    for (size_t i = done + tid; i < V; i += nthr) {
        const float v = img[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o));
        mx = fmaxf(mx, __shfl_down(mx, o));
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    cvx_barrier();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        part[2 * blockIdx.x] = mn;
        part[2 * blockIdx.x + 1] = mx;
    }
}
__global__ __launch_bounds__(256) void k_mind_stats_init(const float* part, int nparts, double count, MindStats* st) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_down(mn, o)); mx = fmaxf(mx, __shfl_down(mx, o)); }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    cvx_barrier();
    if (threadIdx.x != 0) return;
    for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
    const double range = (double)mx - (double)mn;
    double bound = range * range;
    if (!(bound > 0.0)) bound = 1e-300;
    int e;
    (void)frexp(bound * count, &e);
    const double top = ldexp(1.0, e + 1);
    st->m1 = 1.5 * top;
    st->m2 = st->m1 * 0x1p-30;
    st->m3 = st->m2 * 0x1p-30;
    st->a1 = st->a2 = st->a3 = 0.0;
    st->imin = mn;
    st->imax = mx;
    st->mean_override = 0.0f;
    st->use_override = 0;
    st->n_repair = 0u;
}
// clamp bounds of the normalisation from the exact partial sums (every consumer evaluates them itself: a handful of scalar
// operations instead of a one-thread launch between the two passes)
__device__ __forceinline__ void mind_bounds(const MindStats* __restrict__ st, double count, float& lo, float& hi) {
    const float gm = st->use_override ? st->mean_override : (float)((st->a1 + (st->a2 + st->a3)) / count);
    lo = (float)((double)gm * 0.001);      // python: mind_var.mean().item()*0.001   (:61)
    hi = (float)((double)gm * 1000.0);
}

// ---- the tiled stencil ---------------------------------------------------------------------------
// TX = tile width (64, or 32 when radius + dilation is too large for the 160 KB of LDS), NT = TX / RUN * TY * TZ threads
template <int R, int TX>
__global__ __launch_bounds__(TX / RUN * TY * TZ) void k_mind(const float* __restrict__ img, int H, int W, int D, int dil, int nbuf,
                                              MindStats* __restrict__ st, float* __restrict__ out) {
    constexpr int K = 2 * R + 1, NT = TX / RUN * TY * TZ;
    constexpr int SZ = TZ + 2 * R, SY = TY + 2 * R, SX = TX + 2 * R;
    constexpr int SXP = (SX + 3) / 4 * 4 + 2;        // row pitch = 2 (mod 4) floats: the 8-byte reads of lanes 16 B apart in
                                                     // adjacent rows land on disjoint banks
    const int halo = R + dil;
    const int IZ = TZ + 2 * halo, IY = TY + 2 * halo, IX = TX + 2 * halo;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* simg = smem;
    float* ssq = smem + ((IZ * IY * IX + 3) / 4) * 4;   // two buffers of SZ*SY*SXP

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;

    // image tile with replicate (clamp) addressing: simg[q] = I(clamp(origin - halo + q))
    // (i / IX, i / (IX*IY) through float reciprocals: exact for i < 2^16 -- (i + 0.5) / n is at least 0.5 / n away from an
    // integer, the rounding error of the product is below 2^-7 of that -- and an order of magnitude cheaper than the
    // integer division by a run-time extent, which used to cost more than the stencil itself)
    const float inv_ix = 1.0f / (float)IX, inv_iy = 1.0f / (float)IY;
    constexpr int LB = 10;                               // loads in flight per thread (r = 1, d = 2: 20 elements each)
    for (int i0 = tid; i0 < IZ * IY * IX; i0 += LB * NT) {
        float v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int i = i0 + u * NT;
            const int row = (int)(((float)i + 0.5f) * inv_ix);
            const int ix = i - __mul24(row, IX);
            const int iz = (int)(((float)row + 0.5f) * inv_iy);
            const int iy = row - __mul24(iz, IY);
            const int gz = clampi(z0 - halo + iz, 0, H - 1), gy = clampi(y0 - halo + iy, 0, W - 1),
                      gx = clampi(x0 - halo + ix, 0, D - 1);
            v[u] = img[(size_t)(unsigned)(gz * W + gy) * (unsigned)D + gx];     // clamped: in range even past the tile's end
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (i0 + u * NT < IZ * IY * IX) simg[i0 + u * NT] = v[u];
    }
    cvx_barrier();

    const int trun = tid % (TX / RUN), ty = (tid / (TX / RUN)) % TY, tz = tid / ((TX / RUN) * TY);
    const int tx0 = trun * RUN;
    float res[12][RUN];

    // squared-difference stage: every thread owns NSQ fixed positions of the (tile + R) region; their clamped
    // source index in the image tile and their destination index are the same for all 12 channels
    constexpr int NSQ = (SZ * SY * SX + NT - 1) / NT;
    int sq_src[NSQ], sq_dst[NSQ];
#pragma unroll
    for (int e = 0; e < NSQ; ++e) {
        const int i = tid + e * NT;
        const int sx = i % SX, sy = (i / SX) % SY, sz = i / (SX * SY);
        // the box sees the clamped POSITION (rpad2), the shifts clamp again (rpad1): I(clamp(clamp(P)+o*d))
        const int pz = clampi(z0 - R + sz, 0, H - 1), py = clampi(y0 - R + sy, 0, W - 1), px = clampi(x0 - R + sx, 0, D - 1);
        sq_src[e] = ((pz - (z0 - halo)) * IY + (py - (y0 - halo))) * IX + (px - (x0 - halo));
        sq_dst[e] = (i < SZ * SY * SX) ? (sz * SY + sy) * SXP + sx : -1;
        if (i >= SZ * SY * SX) sq_src[e] = (halo * IY + halo) * IX + halo;      // any in-range position; never stored
    }

    constexpr MindOffsets MO{};
#pragma unroll
    for (int c = 0; c < 12; ++c) {          // fully unrolled: res[c][] must stay in registers
        float* sq = ssq + (c & (nbuf - 1)) * (SZ * SY * SXP);
        const int off1 = ((MO.o1[c][0] * IY + MO.o1[c][1]) * IX + MO.o1[c][2]) * dil;
        const int off2 = ((MO.o2[c][0] * IY + MO.o2[c][1]) * IX + MO.o2[c][2]) * dil;
#pragma unroll
        for (int e = 0; e < NSQ; ++e) {
            const float df = simg[sq_src[e] + off1] - simg[sq_src[e] + off2];
            if (sq_dst[e] >= 0) sq[sq_dst[e]] = df * df;
        }
        cvx_barrier();
        // raster-order box sums (z slowest, x fastest) for 4 adjacent outputs from aligned 8-byte LDS reads,
        // one exact division by K^3
        float s[RUN];
#pragma unroll
        for (int j = 0; j < RUN; ++j) s[j] = 0.0f;
#pragma unroll 1
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
                const float* row = sq + ((tz + a) * SY + (ty + b)) * SXP + tx0;
                float rv[RUN + 2 * R];
#pragma unroll
                for (int j = 0; j < (RUN + 2 * R) / 2; ++j) {
                    const f32x2 q = lds_load2(row + 2 * j);
                    rv[2 * j] = q.x; rv[2 * j + 1] = q.y;
                }
#pragma unroll
                for (int j = 0; j < RUN; ++j)
#pragma unroll
                    for (int cc = 0; cc < K; ++cc) s[j] += rv[j + cc];
            }
#pragma unroll
        for (int j = 0; j < RUN; ++j) res[c][j] = div_exact<K * K * K>(s[j]);
        // double-buffered sq (nbuf = 2): the next channel writes the other buffer, one barrier per channel;
        // large radius/dilation tiles only fit one buffer and need a second barrier
        if (nbuf == 1) cvx_barrier();
    }

    const int gz = z0 + tz, gy = y0 + ty;     // threads of an overhanging tile still join the reduction
    const size_t V = (size_t)H * W * D;
    const size_t tail_from = (V / 32) * 32;          // ATen outer-sum tail columns (interleaved order)
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double m1 = st->m1, m2 = st->m2, m3 = st->m3;

#pragma unroll
    for (int j = 0; j < RUN; ++j) {
        const int gx = x0 + tx0 + j;
        const bool valid = (gz < H) && (gy < W) && (gx < D);
        float mc[12];
        float mn = res[0][j];
#pragma unroll
        for (int c = 1; c < 12; ++c) mn = fminf(mn, res[c][j]);   // no NaN handling needed: NaN stays NaN below
#pragma unroll
        for (int c = 0; c < 12; ++c) mc[c] = res[c][j] - mn;
        const size_t lin = ((size_t)gz * W + gy) * D + gx;
        const float sum = (lin >= tail_from) ? outer_sum_ilp<12>(mc) : cascade_seq<12>(mc);
        const float var = fdiv(sum, 12.0f);
        if (valid) {
            const double v = (double)var;
            const double q1 = (v + m1) - m1, r1 = v - q1;
            const double q2 = (r1 + m2) - m2, r2 = r1 - q2;
            const double q3 = (r2 + m3) - m3;
            a1 += q1; a2 += q2; a3 += q3;
        }
    }

    // every partial sum is exactly representable -> any reduction order gives the same bits
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o); a3 += __shfl_down(a3, o);
    }
    __shared__ double red[3][NT / 64];
    if ((tid & 63) == 0) { red[0][tid >> 6] = a1; red[1][tid >> 6] = a2; red[2][tid >> 6] = a3; }
    cvx_barrier();
    if (tid == 0) {
        for (int i = 1; i < NT / 64; ++i) { a1 += red[0][i]; a2 += red[1][i]; a3 += red[2][i]; }
        atomicAdd(&st->a1, a1); atomicAdd(&st->a2, a2); atomicAdd(&st->a3, a3);
    }
    // raw patch-SSDs -> out, already in the final channel order (normalised in place by k_mind_finish)
    if (gz < H && gy < W) {
        const int gx0 = x0 + tx0;
        const bool vec = ((D & 3) == 0) && (gx0 + RUN <= D);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            float* dst = out + (size_t)MIND_INV[c] * V + ((size_t)gz * W + gy) * D + gx0;
            if (vec) {
                *reinterpret_cast<float4*>(dst) = make_float4(res[c][0], res[c][1], res[c][2], res[c][3]);
            } else {
#pragma unroll
                for (int j = 0; j < RUN; ++j)
                    if (gx0 + j < D) dst[j] = res[c][j];
            }
        }
    }
}

// the 12 raw patch SSDs of one voxel (pre-permutation channel order) -> descriptor values, in place:
//   exp(-(D_c - min_c D) / clamp(mean_c(D - min), lo, hi)); `tail` selects ATen's interleaved order of the channel sum
__device__ __forceinline__ void mind_normalise(float (&r)[12], float lo, float hi, bool tail, const ExpTable& et) {
    float mn = r[0];
#pragma unroll
    for (int c = 1; c < 12; ++c) mn = fminf(mn, r[c]);
#pragma unroll
    for (int c = 0; c < 12; ++c) r[c] = r[c] - mn;
    const float sum = tail ? outer_sum_ilp<12>(r) : cascade_seq<12>(r);
    float var = fdiv(sum, 12.0f);
    var = var < lo ? lo : var;
    var = var > hi ? hi : var;
    if (!et.tbl) {                                   // kernel argument: wave-uniform
#pragma unroll
        for (int c = 0; c < 12; ++c) r[c] = cvx_expf(-fdiv(r[c], var));
        return;
    }
    // reference-bits mode: the 12 table bytes are requested together (clamped index, no branch around the loads) and applied
    // afterwards -- one memory round trip per voxel instead of twelve dependent ones
    unsigned key[12], byte[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const float q = fdiv(r[c], var);
        r[c] = cvx_expf(-q);
        const unsigned b = __float_as_uint(q) & 0x7fffffffu, k = b - et.first;
        key[c] = (b >= et.first && k < et.count) ? k : 0xffffffffu;
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) byte[c] = et.tbl[key[c] != 0xffffffffu ? (key[c] >> 2) : 0u];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const unsigned code = key[c] != 0xffffffffu ? (byte[c] >> ((key[c] & 3u) * 2u)) & 3u : 0u;
        if (code) r[c] = __uint_as_float(__float_as_uint(r[c]) + (code == 1u ? 1u : 0xffffffffu));
    }
}

// ---- reference-bits mode: `mind_var.mean()` exactly as torch evaluates it (option mind_mean_threads = T > 0) -------------------
// The default global mean is the exactly rounded one (order independent).  The reference's own value is ATen's float32 sum of the
// V per-voxel variances with T threads (convex_adam_utils.py:61): TensorIteratorReduce's two_pass_reduction splits the elements into
// nt = min(T, ceil(V / 32768)) chunks of ceil(V / nt); a chunk is summed by SumKernel's vectorized_inner_sum -- 8-float vectors,
// vector i of the chunk to one of 4 interleaved accumulators (i mod 4) which are cascade sums (4 levels, 16 or more rows per level),
// leftover vectors to accumulator 0, ((p0 + p1) + p2) + p3 per lane, then the scalar tail and the 8 lanes in order; the T slots are
// reduced by the same routine.  Restated here (and in oracle/cvx_oracle.c::orc_torch_sum, which is pinned against torch.sum for
// 1..128 threads): it only matters for voxels whose variance is clamped to its bounds.
//   k_mind_var          var[x] exactly as mind_normalise computes it
//   k_torch_sum_level   the cascade levels as a parallel radix-16 tree: chain L < 32 of a chunk = (accumulator L / 8, vector lane
//                       L % 8) owns elements 32 i + L; k_torch_sum_chunks folds the top level, the remainders and the lanes
//   k_torch_sum_final   the slots -> mean -> MindStats::mean_override
__device__ __forceinline__ int ceil_log2_ll(long long x) { int r = 0; long long v = 1; while (v < x) { v <<= 1; ++r; } return r; }
// ATen multi_row_sum for one row set: elements x[i * stride], i < size, 4 cascade levels
__device__ float cascade_sum_dev(const float* __restrict__ x, long long stride, long long size) {
    int level_power = ceil_log2_ll(size) / 4;
    if (level_power < 4) level_power = 4;
    const long long level_step = 1ll << level_power, level_mask = level_step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long long i = 0;
    for (; i + level_step <= size;) {
        for (long long j = 0; j < level_step; j += 16) {                  // level_step is a multiple of 16: 16 loads in flight
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = x[(i + u) * stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[0] += v[u];
            i += 16;
        }
        for (int l = 1; l < 4; ++l) {
            acc[l] += acc[l - 1];
            acc[l - 1] = 0.f;
            const long long mask = level_mask << (l * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] += x[i * stride];
    for (int l = 1; l < 4; ++l) acc[0] += acc[l];
    return acc[0];
}
// one thread: SumKernel's inner sum of n contiguous floats (any n; used for the T slots and for tiny volumes)
__device__ float torch_inner_sum_serial(const float* __restrict__ x, long long n) {
    if (n < 8) {
        const long long n4 = n / 4;
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = cascade_sum_dev(x + k, 4, n4);
        for (long long i = n4 * 4; i < n; ++i) p[0] += x[i];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        return p[0];
    }
    const long long nv = n / 8, nv4 = nv / 4;
    float fin = 0.0f;
    for (long long k = nv * 8; k < n; ++k) fin += x[k];
    for (int lane = 0; lane < 8; ++lane) {
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = cascade_sum_dev(x + k * 8 + lane, 32, nv4);
        for (long long i = nv4 * 4; i < nv; ++i) p[0] += x[i * 8 + lane];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        fin += p[0];
    }
    return fin;
}
__global__ __launch_bounds__(256) void k_mind_var(const float* __restrict__ raw, size_t V, float* __restrict__ var) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= V) return;
    float r[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) r[c] = raw[(size_t)MIND_INV[c] * V + x];
    float mn = r[0];
#pragma unroll
    for (int c = 1; c < 12; ++c) mn = fminf(mn, r[c]);
#pragma unroll
    for (int c = 0; c < 12; ++c) r[c] = r[c] - mn;
    const float sum = x >= (V / 32) * 32 ? outer_sum_ilp<12>(r) : cascade_seq<12>(r);
    var[x] = fdiv(sum, 12.0f);
}
// geometry of chunk c of the two-pass reduction: n elements from b; nv4 rows of 32 floats feed the 32 cascade chains
struct TSChunk { long long b, n, nv, nv4; int power; long long step, n0, n1, n2; };
__device__ __forceinline__ TSChunk ts_chunk(long long V, long long chunk, int c) {
    TSChunk t;
    t.b = (long long)c * chunk;
    t.n = (t.b + chunk < V ? t.b + chunk : V) - t.b;
    if (t.n < 0) t.n = 0;
    t.nv = t.n / 8;
    t.nv4 = t.nv / 4;
    t.power = ceil_log2_ll(t.nv4) / 4;
    if (t.power < 4) t.power = 4;
    t.step = 1ll << t.power;
    t.n0 = t.nv4 / t.step; t.n1 = t.n0 / t.step; t.n2 = t.n1 / t.step;
    return t;
}
// One cascade level for every chain of every chunk in parallel: the cascade is a radix-`step` tree (level l+1 adds `step` level-l sums
// in order, starting from 0), so out[j][L] = ((0 + in[j*step][L]) + in[j*step+1][L]) + ...  LEVEL 0 reads the variances themselves
// (row i of chunk c = var[b + 32 i + L]), levels 1, 2 read the previous level's rows; rows are 32 floats = one coalesced access.
template <int LEVEL>
__global__ __launch_bounds__(256) void k_torch_sum_level(const float* __restrict__ in, float* __restrict__ out, long long V, long long chunk,
                                                         long long in_stride, long long out_stride) {
    const TSChunk t = ts_chunk(V, chunk, blockIdx.y);
    const long long nout = LEVEL == 0 ? t.n0 : (LEVEL == 1 ? t.n1 : t.n2);
    const long long j = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int L = threadIdx.x & 31;
    if (j >= nout) return;
    const float* src = (LEVEL == 0 ? in + t.b : in + (long long)blockIdx.y * in_stride) + j * t.step * 32 + L;
    float acc = 0.0f;
    for (long long u = 0; u < t.step; u += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(u + q) * 32];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    out[(long long)blockIdx.y * out_stride + j * 32 + L] = acc;
}
// per chunk: the top level and the ragged remainders of every level in order, ((acc0 + acc1) + acc2) + acc3 per chain, leftover
// vectors, the four accumulators of a vector lane, the scalar tail and the eight lanes -> the chunk's slot
__global__ __launch_bounds__(64) void k_torch_sum_chunks(const float* __restrict__ var, const float* __restrict__ B0, const float* __restrict__ B1,
                                                         const float* __restrict__ B2, long long V, long long chunk, long long s0, long long s1,
                                                         long long s2, float* __restrict__ slots) {
    const TSChunk t = ts_chunk(V, chunk, blockIdx.x);
    const int L = threadIdx.x;
    __shared__ float part[32];
    if (t.n <= 0) return;                                                // (slot keeps the identity)
    if (t.n < 64) {                                                      // tiny chunk: one thread
        if (L == 0) slots[blockIdx.x] = 0.0f + torch_inner_sum_serial(var + t.b, t.n);
        return;
    }
    if (L < 32) {
        const float* x = var + t.b + L;
        const float* b0 = B0 + (long long)blockIdx.x * s0 + L;
        const float* b1 = B1 + (long long)blockIdx.x * s1 + L;
        const float* b2 = B2 + (long long)blockIdx.x * s2 + L;
        float a3 = 0.f, a2 = 0.f, a1 = 0.f, a0 = 0.f;
        for (long long j = 0; j < t.n2; ++j) a3 += b2[j * 32];
        for (long long j = t.n2 * t.step; j < t.n1; ++j) a2 += b1[j * 32];
        for (long long j = t.n1 * t.step; j < t.n0; ++j) a1 += b0[j * 32];
        for (long long i = t.n0 * t.step; i < t.nv4; ++i) a0 += x[i * 32];
        float p = ((a0 + a1) + a2) + a3;
        if (L < 8) for (long long i = t.nv4 * 4; i < t.nv; ++i) p += var[t.b + i * 8 + L];   // leftover vectors -> accumulator 0
        part[L] = p;
    }
    cvx_barrier();
    if (L == 0) {
        float fin = 0.0f;
        for (long long k = t.nv * 8; k < t.n; ++k) fin += var[t.b + k];
        for (int lane = 0; lane < 8; ++lane) fin += ((part[lane] + part[8 + lane]) + part[16 + lane]) + part[24 + lane];
        slots[blockIdx.x] = 0.0f + fin;
    }
}
__global__ void k_torch_sum_final(const float* __restrict__ slots, int nslots, int two_pass, long long V, MindStats* __restrict__ st) {
    const float sum = two_pass ? 0.0f + torch_inner_sum_serial(slots, nslots) : slots[0];
    st->mean_override = fdiv(sum, (float)V);                             // sum_out(..).div_(numel)
    st->use_override = 1;
}

// out_c(x) in place; NV voxels per thread (4 = 16-byte access).
// The channel mean runs over the reference's PRE-permutation channel order (the permutation is applied last, :66).
template <int NV>
__global__ __launch_bounds__(256) void k_mind_finish(float* __restrict__ out, size_t V, const MindStats* __restrict__ st, ExpTable et) {
    float lo, hi;
    mind_bounds(st, (double)V, lo, hi);
    const size_t x = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * NV;
    if (x >= V) return;
    const size_t tail_from = (V / 32) * 32;
    float r[NV][12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const float* src = out + (size_t)MIND_INV[c] * V + x;
        if (NV == 4) {
            const float4 q = *reinterpret_cast<const float4*>(src);
            r[0][c] = q.x; r[1 % NV][c] = q.y; r[2 % NV][c] = q.z; r[3 % NV][c] = q.w;
        } else r[0][c] = src[0];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) mind_normalise(r[j], lo, hi, x + j >= tail_from, et);
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        float* dst = out + (size_t)MIND_INV[c] * V + x;
        if (NV == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[0][c], r[1 % NV][c], r[2 % NV][c], r[3 % NV][c]);
        else dst[0] = r[0][c];
    }
}

// ---- pipeline variant: normalise + exp + BOTH stride poolings in one pass over the raw SSDs -----------------------------
// The registration pipeline consumes the descriptor only through avg_pool3d(g, stride g) (convex_adam_MIND.py:118-119,
// 149-150), so the full-resolution descriptor is never written: a workgroup normalises a T x T x 24 voxel tile (T = the
// larger window, a multiple of the smaller one) into LDS and evaluates the pooling windows of both sizes from there in
// ATen's raster order (sum of g^3 taps, one division).  HBM: 12*V*4 B read + the pooled outputs, instead of
// 2 x 12*V*4 (finish) + 2 x 12*V*4 (two pooling passes).
constexpr int MP_TX = 24, MP_NT = 512;

// raster-order window sum of G^3 taps from the LDS tile and its store; (c, wz, wy, wx) = window inside the tile
template <int T, int G>
__device__ __forceinline__ float mp_window_mean(const float* __restrict__ E, int c, int wz, int wy, int wx) {
    const float* base = E + ((c * T + wz * G) * T + wy * G) * MP_TX + wx * G;
    float s = 0.0f;
#pragma unroll
    for (int z = 0; z < G; ++z) {
        // all loads of a z-slice are issued before its adds: the sequential sum then waits for LDS once per slice
        if (G % 2 == 0) {
            f32x2 v[G][G / 2];
#pragma unroll
            for (int y = 0; y < G; ++y)
#pragma unroll
                for (int x = 0; x < G / 2; ++x) v[y][x] = lds_load2(base + (z * T + y) * MP_TX + 2 * x);
#pragma unroll
            for (int y = 0; y < G; ++y)
#pragma unroll
                for (int x = 0; x < G / 2; ++x) { s += v[y][x].x; s += v[y][x].y; }
        } else {
            float v[G][G];
#pragma unroll
            for (int y = 0; y < G; ++y)
#pragma unroll
                for (int x = 0; x < G; ++x) v[y][x] = base[(z * T + y) * MP_TX + x];
#pragma unroll
            for (int y = 0; y < G; ++y)
#pragma unroll
                for (int x = 0; x < G; ++x) s += v[y][x];
        }
    }
    return fdiv(s, (float)(G * G * G));
}
template <int T, int G>
__device__ __forceinline__ void mp_window(const float* __restrict__ E, int c, int wz, int wy, int wx, int z0, int y0, int x0, int H,
                                          int W, int D, float* __restrict__ out) {
    const int Ho = H / G, Wo = W / G, Do = D / G;
    const int oz = z0 / G + wz, oy = y0 / G + wy, ox = x0 / G + wx;
    if (oz >= Ho || oy >= Wo || ox >= Do) return;
    out[(size_t)c * Ho * Wo * Do + ((size_t)oz * Wo + oy) * Do + ox] = mp_window_mean<T, G>(E, c, wz, wy, wx);
}

// T = GA >= GB, GB divides GA; out2 may be null (single pooling).  512 threads: phase 1 gives every thread 2 adjacent voxels
// (12 x 8-byte loads, normalise, 12 x 8-byte LDS stores); in phase 2 the first wavefronts evaluate the few large windows
// (one long sequential sum each) while the others sweep the many small ones.
// rec2 (optional, replaces out2): the small-window output as the Adam loop's feature RECORDS (warp.hip::k_to_chunked layout:
// [3][V2 + 1][4 channels], record V2 of a chunk all zero; rec_half: four half-precision values per record, rounded to nearest even as
// k_to_chunked_h does) -- the planar pooled copy and the re-packing pass over it disappear from the pipeline.
template <int GA, int GB>
__global__ __launch_bounds__(MP_NT) void k_mind_finish_pool(const float* __restrict__ raw, int H, int W, int D,
                                                            const MindStats* __restrict__ st, float* __restrict__ out1,
                                                            float* __restrict__ out2, void* __restrict__ rec2, int rec_half, ExpTable et, MindRawLayout lay) {
    constexpr int T = GA;
    __shared__ __attribute__((aligned(16))) float E[12 * T * T * MP_TX];          // [c][z][y][x], final channel order
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * MP_TX, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    const size_t V = (size_t)H * W * D;
    float lo, hi;
    mind_bounds(st, (double)V, lo, hi);
    const size_t tail_from = (V / 32) * 32;
    constexpr int NP = MP_TX / 2;
    // blocked raw SSDs (MindRawLayout): this workgroup's tile is one contiguous piece
    const float* tile_raw = lay.T ? raw + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * lay.tile_floats : raw;
    auto voxel_pair = [&](int z, int y, int xp) {
        const int gz = z0 + z, gy = y0 + y, gx = x0 + 2 * xp;
        const size_t lin = ((size_t)gz * W + gy) * D + gx;
        const size_t at = lay.T ? (size_t)((z * T + y) * MP_TX + 2 * xp) : lin, cs = lay.T ? lay.chan_floats : V;
        float r[2][12];
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const float2 q = *reinterpret_cast<const float2*>(tile_raw + (size_t)MIND_INV[c] * cs + at);
            r[0][c] = q.x; r[1][c] = q.y;
        }
        mind_normalise(r[0], lo, hi, lin >= tail_from, et);
        mind_normalise(r[1], lo, hi, lin + 1 >= tail_from, et);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const f32x2 v = {r[0][c], r[1][c]};
            lds_store2(E + ((MIND_INV[c] * T + z) * T + y) * MP_TX + 2 * xp, v);
        }
    };
    // voxels of the tile inside the volume (D is even); a tile that overhangs the volume numbers only those, so that its idle lanes
    // fill whole wavefronts that skip the pass (the last x tile of a 224-voxel row holds 8 of 24 columns) -- no complete window reaches
    // the others and the windows that do exist never read them
    const int nz = min(T, H - z0), ny = min(T, W - y0), ncol = min(MP_TX, D - x0) / 2;
    if (nz == T && ny == T && ncol == NP) {
        for (int i = tid; i < T * T * NP; i += MP_NT) voxel_pair(i / (NP * T), (i / NP) % T, i % NP);
    } else {
        for (int i = tid; i < nz * ny * ncol; i += MP_NT) voxel_pair(i / (ncol * ny), (i / ncol) % ny, i % ncol);
    }
    cvx_barrier();
    constexpr int NA = 12 * (MP_TX / GA);                                          // large windows of the tile (T / GA = 1)
    constexpr int WA = (NA + 63) / 64 * 64;                                        // threads reserved for them (whole wavefronts)
    if (tid < WA) {
        if (tid < NA) mp_window<T, GA>(E, tid / (MP_TX / GA), 0, 0, tid % (MP_TX / GA), z0, y0, x0, H, W, D, out1);
    } else if (rec2) {
        constexpr int nb = T / GB, nx = MP_TX / GB;
        const int Ho = H / GB, Wo = W / GB, Do = D / GB;
        const size_t V2 = (size_t)Ho * Wo * Do;
        for (int i = tid - WA; i < 3 * nb * nb * nx; i += MP_NT - WA) {
            const int wx = i % nx, wy = (i / nx) % nb, wz = (i / (nx * nb)) % nb, cq = i / (nx * nb * nb);
            const int oz = z0 / GB + wz, oy = y0 / GB + wy, ox = x0 / GB + wx;
            if (oz >= Ho || oy >= Wo || ox >= Do) continue;
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = mp_window_mean<T, GB>(E, 4 * cq + j, wz, wy, wx);
            const size_t at = (size_t)cq * (V2 + 1) + ((size_t)oz * Wo + oy) * Do + ox;
            if (rec_half) {
                typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                const h16x4 o = {(_Float16)r[0], (_Float16)r[1], (_Float16)r[2], (_Float16)r[3]};          // round to nearest even
                static_cast<uint2*>(rec2)[at] = __builtin_bit_cast(uint2, o);
            } else static_cast<float4*>(rec2)[at] = make_float4(r[0], r[1], r[2], r[3]);
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid >= WA && tid < WA + 3) {       // the zero record of every chunk
            const size_t at = (size_t)(tid - WA) * (V2 + 1) + V2;
            if (rec_half) static_cast<uint2*>(rec2)[at] = make_uint2(0u, 0u);
            else static_cast<float4*>(rec2)[at] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if (out2) {
        constexpr int nb = T / GB, nx = MP_TX / GB;
        for (int i = tid - WA; i < 12 * nb * nb * nx; i += MP_NT - WA) {
            const int wx = i % nx, wy = (i / nx) % nb, wz = (i / (nx * nb)) % nb, c = i / (nx * nb * nb);
            mp_window<T, GB>(E, c, wz, wy, wx, z0, y0, x0, H, W, D, out2);
        }
    }
}

// ---- single-pass pooled path (mindmarch.hip::k_mind_march_pool): the blocks where the variance clamp binds -------------------------------------
// The marching kernel normalised with the unclamped variance.  That is the reference's value unless var < lo or var > hi (lo, hi = 0.001 x and
// 1000 x the global mean, convex_adam_utils.py:61) on a voxel whose twelve distances are not all zero (those give exp(-0 / v) = 1 for every
// v > 0), or lo == 0 (a mean of zero: 0 / 0).  Per T^3 block the march left the smallest such variance, the largest variance and the all-zero mark; this
// kernel tests every block against the exact bounds -- every workgroup looks at the blocks congruent to its index, 64 per round -- and recomputes the
// pooled cells of a block that holds a clamped voxel from the image: the 12^3 neighbourhood in LDS, the 27-tap raster-order sums of the stencil
// (mindmarch.hip::mm_box_step's order), mind_normalise with the clamp, the windows in ATen's order (mp_window_mean).
__device__ constexpr int MIND_O1[12][3] = {{0,0,-1},{0,-1,0},{0,-1,0},{0,0,1},{0,0,1},{1,0,0},{1,0,0},{1,0,0},{0,1,0},{0,1,0},{0,1,0},{0,1,0}};
__device__ constexpr int MIND_O2[12][3] = {{-1,0,0},{-1,0,0},{0,0,-1},{-1,0,0},{0,-1,0},{0,0,-1},{0,-1,0},{0,0,1},{-1,0,0},{0,0,-1},{0,0,1},{1,0,0}};
constexpr int MR_NT = 256;

template <int GA, int GB>
__global__ __launch_bounds__(MR_NT) void k_mind_repair(const float* __restrict__ img, int H, int W, int D, MindStats* __restrict__ st, const unsigned* __restrict__ blk,
                                                       float* __restrict__ out1, float* __restrict__ out2, void* __restrict__ rec2, int rec_half, ExpTable et, int force) {
    constexpr int T = GA, RE = T + 6;                        // region edge: the block and three voxels around it (dilation 2 + box radius 1)
    __shared__ float R[RE * RE * RE];
    __shared__ __attribute__((aligned(16))) float E[12 * T * T * MP_TX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nbz = (H + T - 1) / T, nby = (W + T - 1) / T, nbx = (D + T - 1) / T, nblk = nbz * nby * nbx;
    const size_t V = (size_t)H * W * D, tail_from = (V / 32) * 32;
    float lo, hi;
    mind_bounds(st, (double)V, lo, hi);
    for (int base = blockIdx.x; base < nblk; base += 64 * gridDim.x) {
        const int mine = base + lane * gridDim.x;
        bool bad = false;
        if (mine < nblk) {
            const unsigned bmin = blk[mine], bmax = blk[nblk + mine], bz = blk[2 * (size_t)nblk + mine];
            const float fmin = __uint_as_float(bmin), fmax = __uint_as_float(bmax);
            // (NaN variances: bits above infinity -- recomputed, whatever the bounds are)
            bad = force != 0 || fmin < lo || fmax > hi || bmax > 0x7f800000u || (bz != 0u && !(lo > 0.0f));
        }
        unsigned long long todo = __ballot(bad);              // the same in every wavefront of the workgroup
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int b = base + l * gridDim.x;
            const int bx = b % nbx, by = (b / nbx) % nby, bzi = b / (nbx * nby);
            const int x0 = bx * T, y0 = by * T, z0 = bzi * T;
            cvx_barrier();                                    // the previous block's windows are done with E, its voxels with R
            for (int i = tid; i < RE * RE * RE; i += MR_NT) {
                const int rx = i % RE, ry = (i / RE) % RE, rz = i / (RE * RE);
                R[i] = img[((size_t)clampi(z0 - 3 + rz, 0, H - 1) * W + clampi(y0 - 3 + ry, 0, W - 1)) * D + clampi(x0 - 3 + rx, 0, D - 1)];
            }
            cvx_barrier();
            if (tid < T * T * T) {
                const int vx = tid % T, vy = (tid / T) % T, vz = tid / (T * T);
                const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
                if (gz < H && gy < W && gx < D) {
                    float r[12];
                    for (int c = 0; c < 12; ++c) {
                        float sum = 0.0f;
                        for (int dz = -1; dz <= 1; ++dz)
                            for (int dy = -1; dy <= 1; ++dy)
                                for (int dx = -1; dx <= 1; ++dx) {
                                    // the box clamps the POSITION first, the shifts clamp again (the reference's two replication pads)
                                    const int pz = clampi(gz + dz, 0, H - 1), py = clampi(gy + dy, 0, W - 1), px = clampi(gx + dx, 0, D - 1);
                                    const int az = clampi(pz + 2 * MIND_O1[c][0], 0, H - 1) - z0 + 3, ay = clampi(py + 2 * MIND_O1[c][1], 0, W - 1) - y0 + 3,
                                              ax = clampi(px + 2 * MIND_O1[c][2], 0, D - 1) - x0 + 3;
                                    const int bz2 = clampi(pz + 2 * MIND_O2[c][0], 0, H - 1) - z0 + 3, by2 = clampi(py + 2 * MIND_O2[c][1], 0, W - 1) - y0 + 3,
                                              bx2 = clampi(px + 2 * MIND_O2[c][2], 0, D - 1) - x0 + 3;
                                    const float d = R[(az * RE + ay) * RE + ax] - R[(bz2 * RE + by2) * RE + bx2];
                                    sum += d * d;
                                }
                        r[c] = div_exact<27>(sum);
                    }
                    const size_t lin = ((size_t)gz * W + gy) * D + gx;
                    mind_normalise(r, lo, hi, lin >= tail_from, et);
#pragma unroll
                    for (int c = 0; c < 12; ++c) E[((MIND_INV[c] * T + vz) * T + vy) * MP_TX + vx] = r[c];
                }
            }
            cvx_barrier();
            if (tid < 12) mp_window<T, GA>(E, tid, 0, 0, 0, z0, y0, x0, H, W, D, out1);
            constexpr int nb = T / GB;
            if (rec2) {
                const int Ho = H / GB, Wo = W / GB, Do = D / GB;
                const size_t V2 = (size_t)Ho * Wo * Do;
                for (int i = tid; i < 3 * nb * nb * nb; i += MR_NT) {
                    const int wx = i % nb, wy = (i / nb) % nb, wz = (i / (nb * nb)) % nb, cq = i / (nb * nb * nb);
                    const int oz = z0 / GB + wz, oy = y0 / GB + wy, ox = x0 / GB + wx;
                    if (oz >= Ho || oy >= Wo || ox >= Do) continue;
                    float q[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = mp_window_mean<T, GB>(E, 4 * cq + j, wz, wy, wx);
                    const size_t at = (size_t)cq * (V2 + 1) + ((size_t)oz * Wo + oy) * Do + ox;
                    if (rec_half) {
                        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                        const h16x4 o = {(_Float16)q[0], (_Float16)q[1], (_Float16)q[2], (_Float16)q[3]};
                        static_cast<uint2*>(rec2)[at] = __builtin_bit_cast(uint2, o);
                    } else static_cast<float4*>(rec2)[at] = make_float4(q[0], q[1], q[2], q[3]);
                }
            } else if (out2) {
                for (int i = tid; i < 12 * nb * nb * nb; i += MR_NT) {
                    const int wx = i % nb, wy = (i / nb) % nb, wz = (i / (nb * nb)) % nb, c = i / (nb * nb * nb);
                    mp_window<T, GB>(E, c, wz, wy, wx, z0, y0, x0, H, W, D, out2);
                }
            }
            if (tid == 0) atomicAdd(&st->n_repair, 1u);
        }
    }
    // the zero record that closes every chunk of feature records (k_to_chunked layout)
    if (rec2 && blockIdx.x == 0 && tid < 3) {
        const size_t V2 = (size_t)(H / GB) * (W / GB) * (D / GB), at = (size_t)tid * (V2 + 1) + V2;
        if (rec_half) static_cast<uint2*>(rec2)[at] = make_uint2(0u, 0u);
        else static_cast<float4*>(rec2)[at] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

static size_t mind_lds_bytes(int R, int dil, int nbuf, int TX) {
    const int halo = R + dil;
    const int IZ = TZ + 2 * halo, IY = TY + 2 * halo, IX = TX + 2 * halo;
    const int SZ = TZ + 2 * R, SY = TY + 2 * R, SX = TX + 2 * R, SXP = (SX + 3) / 4 * 4 + 2;
    return sizeof(float) * ((size_t)((IZ * IY * IX + 3) / 4) * 4 + (size_t)nbuf * SZ * SY * SXP);
}

template <int R>
static int mind_launch_r(const float* img, int H, int W, int D, int dil, MindStats* st, float* out, hipStream_t s, const MindRawLayout& lay) {
    const bool tiled_only = options().mind_tiled != 0;
    if (!tiled_only && mind_march_supported(img, out, H, W, D, R, dil)) {
        launch_mind_march(img, H, W, D, st, out, s, lay);
        return check_last("mindssc");
    }
    if (lay.T) return fail(CVX_ERR_UNSUPPORTED, "mindssc: the blocked layout needs the marching stencil");
    if (mind_lds_bytes(R, dil, 1, 64) <= 160 * 1024) {
        const dim3 grid(cdiv(D, 64), cdiv(W, TY), cdiv(H, TZ));
        const int nbuf = mind_lds_bytes(R, dil, 2, 64) <= 160 * 1024 ? 2 : 1;
        const size_t lds = mind_lds_bytes(R, dil, nbuf, 64);
        static size_t granted0 = 0;
        ensure_dynamic_lds(&k_mind<R, 64>, lds, granted0);
        hipLaunchKernelGGL((k_mind<R, 64>), grid, dim3(64 / RUN * TY * TZ), lds, s, img, H, W, D, dil, nbuf, st, out);
    } else {                                        // radius 3 with dilation 4: tiles of 32 columns
        const dim3 grid(cdiv(D, 32), cdiv(W, TY), cdiv(H, TZ));
        const int nbuf = mind_lds_bytes(R, dil, 2, 32) <= 160 * 1024 ? 2 : 1;
        const size_t lds = mind_lds_bytes(R, dil, nbuf, 32);
        static size_t granted1 = 0;
        ensure_dynamic_lds(&k_mind<R, 32>, lds, granted1);
        hipLaunchKernelGGL((k_mind<R, 32>), grid, dim3(32 / RUN * TY * TZ), lds, s, img, H, W, D, dil, nbuf, st, out);
    }
    return check_last("mindssc");
}

// min/max -> split grids -> stencil pass: raw patch SSDs in `raw` [12][V] (final channel order), statistics in *st
static int mind_stencil(const float* img, int H, int W, int D, int radius, int dilation, float* raw, void* workspace,
                        size_t workspace_bytes, MindStats** st_out, hipStream_t s, const MindRawLayout& lay = MindRawLayout{0, 0, 0, 0, 0}) {
    Carver cv(workspace, workspace_bytes);
    float* part = cv.take<float>(2 * 1024);
    MindStats* st = cv.take<MindStats>(1);
    *st_out = st;
    const size_t V = (size_t)H * W * D;
    const int nb = (int)(V / 4096 + 1 < 1024 ? V / 4096 + 1 : 1024);
    hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(256), 0, s, img, V, part);
    hipLaunchKernelGGL(k_mind_stats_init, dim3(1), dim3(256), 0, s, part, nb, (double)V, st);
    int rc;
    switch (radius) {
        case 1: rc = mind_launch_r<1>(img, H, W, D, dilation, st, raw, s, lay); break;
        case 2: rc = mind_launch_r<2>(img, H, W, D, dilation, st, raw, s, lay); break;
        default: rc = mind_launch_r<3>(img, H, W, D, dilation, st, raw, s, lay); break;
    }
    const long long T = options().mind_mean_threads;
    if (rc || T <= 0) return rc;
    // reference-bits mode: torch's own mean instead of the exactly rounded one
    const int threads = (int)(T > 1024 ? 1024 : T);
    float* var = cv.take<float>(V);
    float* slots = cv.take<float>(1024);
    const bool two_pass = V >= 32768 && threads > 1;
    long long nt = 1, chunk = (long long)V;
    if (two_pass) {
        nt = ((long long)V + 32767) / 32768;
        if (nt > threads) nt = threads;
        chunk = ((long long)V + nt - 1) / nt;
    }
    // level buffers: at most nv4 / 16, / 256, / 4096 rows of 32 floats per chunk
    const long long rows = chunk / 32 + 1;
    const long long s0 = (rows / 16 + 1) * 32, s1 = (rows / 256 + 1) * 32, s2 = (rows / 4096 + 1) * 32;
    float* B0 = cv.take<float>((size_t)(s0 * nt));
    float* B1 = cv.take<float>((size_t)(s1 * nt));
    float* B2 = cv.take<float>((size_t)(s2 * nt));
    if (!cv.ok()) return fail(CVX_ERR_WORKSPACE, "mindssc: workspace too small for mind_mean_threads (query the size after setting the option)");
    hipLaunchKernelGGL(k_mind_var, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, s, raw, V, var);
    if (two_pass) (void)hipMemsetAsync(slots, 0, 1024 * sizeof(float), s);              // unused slots keep the identity
    auto blocks = [](long long nrows) { return (unsigned)((nrows * 32 + 255) / 256 > 0 ? (nrows * 32 + 255) / 256 : 1); };
    hipLaunchKernelGGL(k_torch_sum_level<0>, dim3(blocks(rows / 16 + 1), (unsigned)nt), dim3(256), 0, s, var, B0, (long long)V, chunk, 0ll, s0);
    hipLaunchKernelGGL(k_torch_sum_level<1>, dim3(blocks(rows / 256 + 1), (unsigned)nt), dim3(256), 0, s, B0, B1, (long long)V, chunk, s0, s1);
    hipLaunchKernelGGL(k_torch_sum_level<2>, dim3(blocks(rows / 4096 + 1), (unsigned)nt), dim3(256), 0, s, B1, B2, (long long)V, chunk, s1, s2);
    hipLaunchKernelGGL(k_torch_sum_chunks, dim3((unsigned)nt), dim3(64), 0, s, var, B0, B1, B2, (long long)V, chunk, s0, s1, s2, slots);
    hipLaunchKernelGGL(k_torch_sum_final, dim3(1), dim3(1), 0, s, slots, threads, two_pass ? 1 : 0, (long long)V, st);
    return check_last("mind_mean");
}

static int mind_check(const float* img, const float* out, const void* workspace, int H, int W, int D, int radius, int dilation,
                      size_t workspace_bytes) {
    CVX_REQUIRE(img && out && workspace, "cvx_mindssc_f32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_mindssc_f32: bad extent %dx%dx%d", H, W, D);
    CVX_REQUIRE(radius >= 1 && radius <= 3, "cvx_mindssc_f32: radius %d not in 1..3", radius);
    CVX_REQUIRE(dilation >= 1 && dilation <= 4, "cvx_mindssc_f32: dilation %d not in 1..4", dilation);
    if (workspace_bytes < cvx_mindssc_workspace_bytes(H, W, D, radius, dilation))
        return fail(CVX_ERR_WORKSPACE, "cvx_mindssc_f32: workspace too small");
    if (mind_lds_bytes(radius, dilation, 1, 32) > 160 * 1024)
        return fail(CVX_ERR_UNSUPPORTED, "cvx_mindssc_f32: radius %d dilation %d exceeds the LDS tile", radius, dilation);
    return CVX_OK;
}

// tile edge of the fused finish + pooling pass for window sizes (g1, g2), 0 if it does not apply
static int mind_pool_tile(int H, int W, int D, int g1, int g2) {
    const int T = g1 > g2 ? g1 : g2, g = g1 > g2 ? g2 : g1;
    (void)H; (void)W;
    if ((D & 1) != 0) return 0;                                       // 8-byte row segments
    const bool ok = (T == 6 && (g == 2 || g == 3 || g == 6)) || (T == 4 && (g == 2 || g == 4)) || (T == 2 && g == 2);
    return ok ? T : 0;                                                // 12 * T * T * 24 floats of LDS: 41 KB for T = 6
}
bool mind_pooled_supported(int H, int W, int D, int g1, int g2) { return mind_pool_tile(H, W, D, g1, g2 > 0 ? g2 : g1) != 0; }

// MIND-SSC of `img` delivered only as avg_pool3d(., g1, stride g1) -> out1 and (g2 > 0) avg_pool3d(., g2, stride g2) -> out2;
// `raw` is a 12*V float scratch (the raw patch SSDs).  Same values as cvx_mindssc_f32 followed by cvx_avgpool_f32.
// The stencil pass of the pooled path writes its raw SSDs blocked by the second pass's tiles (MindRawLayout) when the marching kernel runs it,
// the global mean is the exactly rounded one (the reference-bits mean walks the planar volume) and option mind_blocked is on.
// mind_pooled_raw_floats: size of the `raw` scratch the caller must provide (the tiles overhang the volume).
static bool mind_pooled_blocked(const float* img, const float* raw, int H, int W, int D, int radius, int dilation, int g1, int g2) {
    return options().mind_blocked != 0 && options().mind_tiled == 0 && options().mind_mean_threads <= 0 && mind_pool_tile(H, W, D, g1, g2 > 0 ? g2 : g1) != 0 &&
           mind_march_supported(img, raw, H, W, D, radius, dilation);
}
size_t mind_pooled_raw_floats(int H, int W, int D, int g1, int g2) {
    const int T = mind_pool_tile(H, W, D, g1, g2 > 0 ? g2 : g1);
    const size_t planar = (size_t)12 * H * W * D;
    if (T == 0) return planar;
    const size_t blocked = (size_t)cdiv(D, MP_TX) * cdiv(W, T) * cdiv(H, T) * 12 * T * T * MP_TX;
    return blocked > planar ? blocked : planar;
}
// the single-pass kernel applies: the marching stencil's geometry, a window pair it is instantiated for, the library's own exp and the exactly
// rounded mean (reference-bits mode walks the planar variance volume and corrects exp through a table)
static bool mind_single_pass(const float* img, int H, int W, int D, int radius, int dilation, int ga, int gb) {
    return options().mind_single != 0 && options().mind_tiled == 0 && options().mind_mean_threads <= 0 && mind_exp_table().tbl == nullptr &&
           mind_single_supported(ga, gb) && mind_march_supported(img, img, H, W, D, radius, dilation);
}
// records (with out2): out2 receives the g2-pooled descriptor as feature records instead of planar channels (needs g2 <= g1); see
// mind_pooled_records_supported
bool mind_pooled_records_supported(int H, int W, int D, int g1, int g2) { return g2 > 0 && g2 <= g1 && mind_pool_tile(H, W, D, g1, g2) != 0; }
int launch_mind_pooled(const float* img, int H, int W, int D, int radius, int dilation, int g1, float* out1, int g2, float* out2,
                       float* raw, void* workspace, size_t workspace_bytes, hipStream_t s, int records) {
    int rc = mind_check(img, raw, workspace, H, W, D, radius, dilation, workspace_bytes);
    if (rc) return rc;
    const int T = mind_pool_tile(H, W, D, g1, g2 > 0 ? g2 : g1);
    if (T == 0 || !out1) return fail(CVX_ERR_UNSUPPORTED, "mind_pooled: window sizes %d, %d do not tile", g1, g2);
    MindStats* st = nullptr;
    // (ga, gb) = (larger, smaller) window; the larger one goes to the matching output
    const bool swap = g2 > g1;
    const int ga = swap ? g2 : g1, gb = g2 > 0 ? (swap ? g1 : g2) : g1;
    float* oa = swap ? out2 : out1;
    float* ob = g2 > 0 ? (swap ? out1 : out2) : nullptr;
    if (records && (swap || !out2)) return fail(CVX_ERR_UNSUPPORTED, "mind_pooled: feature records need 0 < g2 <= g1");
    void* rec = records ? out2 : nullptr;             // records: 1 = float32, 2 = half precision
    if (records) ob = nullptr;
    if (mind_single_pass(img, H, W, D, radius, dilation, ga, gb)) {
        // ONE pass over the image (mindmarch.hip::k_mind_march_pool) + the repair of the blocks where the variance clamp binds; `raw` is not touched
        Carver cv(workspace, workspace_bytes);
        float* part = cv.take<float>(2 * 1024);
        st = cv.take<MindStats>(1);
        const size_t nblk = (size_t)cdiv(H, ga) * cdiv(W, ga) * cdiv(D, ga);
        unsigned* blk = cv.take<unsigned>(3 * nblk);
        if (!cv.ok()) return fail(CVX_ERR_WORKSPACE, "mind_pooled: workspace too small");
        const size_t V = (size_t)H * W * D;
        const int nb = (int)(V / 4096 + 1 < 1024 ? V / 4096 + 1 : 1024);
        hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(256), 0, s, img, V, part);
        hipLaunchKernelGGL(k_mind_stats_init, dim3(1), dim3(256), 0, s, part, nb, (double)V, st);
        launch_mind_march_pool(img, H, W, D, ga, oa, gb, records ? rec : static_cast<void*>(ob), records, st, blk, s);
        const unsigned rgrid = (unsigned)(nblk / 64 + 1 < 1024 ? nblk / 64 + 1 : 1024);
        const int force = options().mind_single == 2 ? 1 : 0;           // 2: every block through the repair kernel (test of the exact recomputation)
#define CVX_MR(GA, GB) hipLaunchKernelGGL((k_mind_repair<GA, GB>), dim3(rgrid), dim3(MR_NT), 0, s, img, H, W, D, st, blk, oa, ob, rec, records == 2 ? 1 : 0, mind_exp_table(), force)
        if (ga == 6 && gb == 2) CVX_MR(6, 2);
        else if (ga == 6 && gb == 3) CVX_MR(6, 3);
        else if (ga == 6 && gb == 6) CVX_MR(6, 6);
        else if (ga == 4 && gb == 2) CVX_MR(4, 2);
        else if (ga == 4 && gb == 4) CVX_MR(4, 4);
        else CVX_MR(2, 2);
#undef CVX_MR
        return check_last("mind_march_pool");
    }
    const dim3 grid(cdiv(D, MP_TX), cdiv(W, T), cdiv(H, T));
    const MindRawLayout lay = mind_pooled_blocked(img, raw, H, W, D, radius, dilation, g1, g2)
                                  ? MindRawLayout{T, (int)grid.x, (int)grid.y, (size_t)12 * T * T * MP_TX, (size_t)T * T * MP_TX} : MindRawLayout{0, 0, 0, 0, 0};
    if ((rc = mind_stencil(img, H, W, D, radius, dilation, raw, workspace, workspace_bytes, &st, s, lay))) return rc;
#define CVX_MP(GA, GB) hipLaunchKernelGGL((k_mind_finish_pool<GA, GB>), grid, dim3(MP_NT), 0, s, raw, H, W, D, st, oa, ob, rec, records == 2 ? 1 : 0, mind_exp_table(), lay)
    if (ga == 6 && gb == 2) CVX_MP(6, 2);
    else if (ga == 6 && gb == 3) CVX_MP(6, 3);
    else if (ga == 6 && gb == 6) CVX_MP(6, 6);
    else if (ga == 4 && gb == 2) CVX_MP(4, 2);
    else if (ga == 4 && gb == 4) CVX_MP(4, 4);
    else CVX_MP(2, 2);
#undef CVX_MP
    return check_last("mind_finish_pool");
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_mindssc_workspace_bytes(int H, int W, int D, int radius, int dilation) {
    (void)radius; (void)dilation;
    size_t n = 256 + 2 * 1024 * sizeof(float) + 256 + sizeof(MindStats) + 256;
    n += 256 + 3 * sizeof(unsigned) * (size_t)cdiv(H, 2) * cdiv(W, 2) * cdiv(D, 2);             // block statistics of the single-pass pooled path (smallest window: 2)
    if (options().mind_mean_threads > 0)                                       // var, thread slots, three cascade levels (< V / 14 floats)
        n += 256 + (size_t)H * W * D * sizeof(float) + 256 + 1024 * sizeof(float) + (size_t)H * W * D / 14 * sizeof(float) + 3 * (256 + 1024 * 64 * sizeof(float));
    return n;
}

extern "C" size_t cvx_mindssc_pooled_scratch_bytes(int H, int W, int D, int radius, int dilation, int g1, int g2) {
    if (H <= 0 || W <= 0 || D <= 0) return 0;
    const bool swap = g2 > g1;
    const int ga = swap ? g2 : g1, gb = g2 > 0 ? (swap ? g1 : g2) : g1;
    // (alignment is the caller's hipMalloc: the predicate is asked with an aligned pointer)
    if (mind_single_pass(reinterpret_cast<const float*>(uintptr_t(256)), H, W, D, radius, dilation, ga, gb)) return 0;
    return mind_pooled_raw_floats(H, W, D, g1, g2) * sizeof(float);
}

extern "C" int cvx_mindssc_pooled_f32(const float* img, int H, int W, int D, int radius, int dilation, int g1, float* out1, int g2, float* out2,
                                      void* scratch, size_t scratch_bytes, void* workspace, size_t workspace_bytes, int* repaired_host, void* stream) {
    CVX_REQUIRE(img && out1 && workspace, "cvx_mindssc_pooled_f32: null pointer");
    CVX_REQUIRE(H > 0 && W > 0 && D > 0, "cvx_mindssc_pooled_f32: bad extent %dx%dx%d", H, W, D);
    CVX_REQUIRE(g1 >= 1 && g2 >= 0, "cvx_mindssc_pooled_f32: bad windows %d, %d", g1, g2);
    CVX_REQUIRE(g2 == 0 || out2, "cvx_mindssc_pooled_f32: second output missing");
    if (!mind_pooled_supported(H, W, D, g1, g2)) return fail(CVX_ERR_UNSUPPORTED, "cvx_mindssc_pooled_f32: windows %d, %d do not tile", g1, g2);
    const size_t need = cvx_mindssc_pooled_scratch_bytes(H, W, D, radius, dilation, g1, g2);
    const bool single = need == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0;
    if (!single) {
        const size_t two = mind_pooled_raw_floats(H, W, D, g1, g2) * sizeof(float);
        if (!scratch || scratch_bytes < two) return fail(CVX_ERR_WORKSPACE, "cvx_mindssc_pooled_f32: scratch %zu < %zu bytes", scratch_bytes, two);
    }
    hipStream_t s = as_stream(stream);
    // (launch_mind_pooled validates through `raw`: the image itself stands in when the single pass needs no scratch)
    int rc = launch_mind_pooled(img, H, W, D, radius, dilation, g1, out1, g2, g2 > 0 ? out2 : nullptr, single ? const_cast<float*>(img) : static_cast<float*>(scratch),
                                workspace, workspace_bytes, s, 0);
    if (rc || !repaired_host) return rc;
    *repaired_host = 0;
    if (single) {
        Carver cv(workspace, workspace_bytes);
        (void)cv.take<float>(2 * 1024);
        const MindStats* st = cv.take<MindStats>(1);
        unsigned n = 0;
        if (hipMemcpyAsync(&n, &st->n_repair, sizeof(n), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return check_last("cvx_mindssc_pooled_f32: repair count");
        *repaired_host = (int)n;
    }
    return CVX_OK;
}

extern "C" int cvx_mindssc_f32(const float* img, int H, int W, int D, int radius, int dilation, float* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    int rc = mind_check(img, out, workspace, H, W, D, radius, dilation, workspace_bytes);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    MindStats* st = nullptr;
    if ((rc = mind_stencil(img, H, W, D, radius, dilation, out, workspace, workspace_bytes, &st, s))) return rc;
    const size_t V = (size_t)H * W * D;
    if (V % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(k_mind_finish<4>, dim3((unsigned)cdiv64((int64_t)(V / 4), 256)), dim3(256), 0, s, out, V, st, mind_exp_table());
    else
        hipLaunchKernelGGL(k_mind_finish<1>, dim3((unsigned)cdiv64((int64_t)V, 256)), dim3(256), 0, s, out, V, st, mind_exp_table());
    return check_last("mind_finish");
}
