// surfdist.hip -- surface distances of cupy_hd95 without volume-sized distance transforms (hyper_util.py:32-51).
//
// The reference builds, per label, four Euclidean distance transforms of whole volumes (inside / outside of the label in both maps)
// and then reads them on the SURFACE of the other map only (dist1[surf2], dist2[surf1], :48): a few per cent of the voxels.
// Here the distances are computed at those voxels alone:
//   k_label_bits         one pass over a label map -> one bit per (label, voxel): bits[label-1][h*W + w][ceil(D/64)] (64 voxels along D
//                        per word; 13 labels at 160x192x224: 12.8 MB instead of 715 MB of int32 transforms)
//   k_surface_dist_hist  one pass over the OTHER map: a voxel of label l with an in-bounds 6-neighbour of another value is a surface voxel
//                        (its inside distance is exactly 1, :41/:45); the wavefront that found it searches the bit planes of l for the
//                        exact squared distance to the nearest voxel outside l (if the voxel lies inside l in the first map) or
//                        inside l (if it lies outside): rows (h', w') in square rings of growing radius r around (h, w), 64 rows
//                        per step, the nearest set bit along D in each by count-leading / trailing-zeros; every row of ring r is
//                        at least r away, so the search stops at the first ring with r*r >= best.  Integer arithmetic: the result
//                        is the exact squared distance, equal to distance_transform_edt(..)**2 whatever the order of ties.
// The squared distances are counted into the per-label histogram that k_hist_order_stats (edt.hip) reads the percentile from.
#include <limits.h>

#include "cvx_common.h"

namespace cvx {

struct ActiveLabels { unsigned long long m[4]; };       // bit l: label l (1 .. 255) is scored

__device__ __forceinline__ int label_of(float v, int nl) { return (v >= 1.0f && v <= (float)nl && v == floorf(v)) ? (int)v : 0; }

// one wavefront per row of D voxels
__global__ __launch_bounds__(256) void k_label_bits(const float* __restrict__ seg, int nrows, int D, int nseg, int nl,
                                                    unsigned long long* __restrict__ bits) {
    const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    for (int sg = 0; sg < nseg; ++sg) {
        const int i = sg * 64 + lane;
        const int l = i < D ? label_of(seg[(size_t)row * D + i], nl) : 0;
        for (int q = 1; q <= nl; ++q) {
            const unsigned long long m = __ballot(l == q);
            if (lane == ((q - 1) & 63)) bits[((size_t)(q - 1) * nrows + row) * nseg + sg] = m;
        }
    }
}

// distance along D from voxel z to the nearest set bit of a row of `nseg` words (the complement of the row if `invert`); INT_MAX if none
__device__ __forceinline__ int nearest_in_word(unsigned long long m, int sg, int zw, int zb, int z) {
    if (!m) return INT_MAX;
    if (sg < zw) return z - (sg * 64 + 63 - __builtin_clzll(m));
    if (sg > zw) return sg * 64 + __builtin_ctzll(m) - z;
    const unsigned long long left = m & ((2ull << zb) - 1ull), right = m & ~((1ull << zb) - 1ull);
    const int dl = left ? zb - (63 - __builtin_clzll(left)) : INT_MAX, dr = right ? __builtin_ctzll(right) - zb : INT_MAX;
    return min(dl, dr);
}
__device__ __forceinline__ int nearest_in_row(const unsigned long long* __restrict__ rowbits, int nseg, int D, int z, bool invert) {
    const int zw = z >> 6, zb = z & 63;
    const unsigned long long tail = (D & 63) ? (1ull << (D & 63)) - 1ull : ~0ull;
    int best = INT_MAX;
    if (nseg <= 4) {                                         // all words in flight before the first is used (D <= 256: the common case)
        unsigned long long m[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) m[sg] = rowbits[min(sg, nseg - 1)];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            unsigned long long v = invert ? ~m[sg] : m[sg];
            if (sg == nseg - 1) v &= tail;
            if (sg < nseg) best = min(best, nearest_in_word(v, sg, zw, zb, z));
        }
        return best;
    }
    for (int sg = 0; sg < nseg; ++sg) {
        unsigned long long m = rowbits[sg];
        if (invert) m = ~m;
        if (sg == nseg - 1) m &= tail;
        best = min(best, nearest_in_word(m, sg, zw, zb, z));
    }
    return best;
}

// one wavefront per 64 voxels along D of map b
__global__ __launch_bounds__(256) void k_surface_dist_hist(const float* __restrict__ segb, const unsigned long long* __restrict__ bits_a, int H,
                                                           int W, int D, int nseg, int nl, ActiveLabels act, int nbins,
                                                           unsigned long long* __restrict__ hist_all, size_t hist_stride,
                                                           int* __restrict__ overflow_all, int overflow_stride, int max_radius) {
    // surface voxels are close to the other surface: almost every count lands in a few low bins, accumulated per workgroup in LDS
    constexpr int LL = 64, LB = 64;
    __shared__ unsigned int low[LL * LB];
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) low[i] = 0;
    cvx_barrier();
    const int nrows = H * W;
    const int lane = threadIdx.x & 63;
    // a bounded grid walks the volume (the low bins are flushed once per workgroup: with one workgroup per 256 voxels the flush was
    // a million global atomics on some forty addresses, most of the launch)
    const int nwaves = nrows * nseg, wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    for (int wave = wave0; wave < nwaves; wave += (int)((gridDim.x * blockDim.x) >> 6)) {
        const int row = wave / nseg, sg = wave - row * nseg;
        const int h = row / W, w = row - h * W, i = sg * 64 + lane;
        const float* p = segb + (size_t)row * D + i;
        bool surf = false;
        int l = 0;
        if (i < D) {
            const float v = *p;
            l = label_of(v, nl);
            if (l && ((act.m[l >> 6] >> (l & 63)) & 1ull))
                surf = (i > 0 && p[-1] != v) || (i + 1 < D && p[1] != v) || (w > 0 && p[-D] != v) || (w + 1 < W && p[D] != v) ||
                       (h > 0 && p[-(ptrdiff_t)((size_t)W * D)] != v) || (h + 1 < H && p[(size_t)W * D] != v);
        }
        // fast path, one surface voxel per lane:
        //   step 1: the 3x3 rows around (h, w), bits z-1 .. z+1 of each: squared distances 1, 2, 3 -- where most surface voxels of a
        //           registered pair end
        //   step 2: the 5x5 rows, bits z-2 .. z+2: every voxel at squared distance <= 8 lies in this cube (3*3 > 8), so a minimum <= 8
        //           is final; what is left goes to the wavefront-wide ring search below
        // (the lanes of a wavefront share the row segment, so lanes of the same label read the same words: one cache line per load.
        // Reading them through the scalar unit instead -- a loop over the labels present, uniform addresses -- was 2-4 x slower.)
        int d2 = 0, seed = INT_MAX;
        if (surf) {
            const unsigned long long* plane = bits_a + (size_t)(l - 1) * nrows * nseg;
            const int zb = lane;
            const unsigned long long tail = (D & 63) ? (1ull << (D & 63)) - 1ull : ~0ull;
            const bool inside = (plane[(size_t)row * nseg + sg] >> zb) & 1ull;            // inside l in map a: distance to the complement
            // window of 2 R + 1 bits around voxel z (bit R = voxel z) of the (complemented) plane in row (hh, ww); 0 outside the volume
            auto window = [&](int hh, int ww, int R) -> unsigned {
                if (hh < 0 || hh >= H || ww < 0 || ww >= W) return 0u;
                const unsigned long long* rb = plane + ((size_t)hh * W + ww) * nseg;
                auto word = [&](int q) { unsigned long long m = rb[q]; if (inside) m = ~m; if (q == nseg - 1) m &= tail; return m; };
                const unsigned long long m = word(sg);
                unsigned long long win = zb >= R ? m >> (zb - R) : m << (R - zb);
                if (zb < R && sg > 0) win |= word(sg - 1) >> (64 - R + zb);
                if (zb > 63 - R && sg + 1 < nseg) win |= word(sg + 1) << (64 + R - zb);
                return (unsigned)win & ((2u << (2 * R)) - 1u);
            };
            const unsigned c = window(h, w, 1), f = window(h - 1, w, 1) | window(h + 1, w, 1) | window(h, w - 1, 1) | window(h, w + 1, 1);
            if ((c & 5u) || (f & 2u)) d2 = 1;
            else {
                const unsigned g = window(h - 1, w - 1, 1) | window(h - 1, w + 1, 1) | window(h + 1, w - 1, 1) | window(h + 1, w + 1, 1);
                d2 = ((f & 5u) || (g & 2u)) ? 2 : (g & 5u) ? 3 : 0;
            }
            if (d2 == 0) {
                int best = INT_MAX;
                for (int dh = -2; dh <= 2; ++dh)
                    for (int dw = -2; dw <= 2; ++dw) {
                        const unsigned win = window(h + dh, w + dw, 2);
                        if (win) best = min(best, dh * dh + dw * dw + ((win & 4u) ? 0 : (win & 10u) ? 1 : 4));
                    }
                if (best <= 8) d2 = best;
                else seed = best;                                                          // 9 .. 12: an upper bound for the ring search
            }
        }
        {   // one LDS / global atomic per distinct (label, distance) of the wavefront
            const int key = d2 ? (l - 1) * 16 + d2 : -1;
            unsigned long long pend = __ballot(key >= 0);
            while (pend) {
                const int leader = __builtin_ctzll(pend);
                const int k = __shfl(key, leader);
                const unsigned long long same = __ballot(key == k);
                if (lane == leader) {
                    const int q = k >> 4, bin = k & 15;
                    const unsigned cnt = (unsigned)__builtin_popcountll(same);
                    if (bin >= nbins) atomicMax(&overflow_all[(size_t)q * overflow_stride], 1);
                    else if (q < LL) atomicAdd(&low[q * LB + bin], cnt);
                    else atomicAdd(&hist_all[(size_t)q * hist_stride + bin], (unsigned long long)cnt);
                }
                pend &= ~same;
            }
        }
        unsigned long long todo = __ballot(surf && d2 == 0);
        const int rmax = max(max(h, H - 1 - h), max(w, W - 1 - w));
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int ql = __shfl(l, b), z = sg * 64 + b;
            // a label that has already exceeded the radius is void for the caller (flag 2: the whole call goes to the transforms): the
            // other far voxels of that label are not searched any more -- a badly registered pair costs little before it is handed over
            if (max_radius > 0 && __hip_atomic_load(&overflow_all[(size_t)(ql - 1) * overflow_stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) continue;
            const unsigned long long* plane = bits_a + (size_t)(ql - 1) * nrows * nseg;
            const bool inside = (plane[(size_t)row * nseg + sg] >> b) & 1ull;          // inside l in map a: distance to the complement
            int best = __shfl(seed, b);
            bool gave_up = false;
            for (int i0 = 0;; i0 += 64) {
                const int idx = i0 + lane;
                int s = (int)sqrtf((float)idx);                                       // floor(sqrt(idx)); idx < 2^24
                if (s * s > idx) --s;
                if ((s + 1) * (s + 1) <= idx) ++s;
                const int r = (s + 1) >> 1;                                           // ring of cell idx: smallest r with (2r+1)^2 > idx
                const int r0 = __shfl(r, 0);
                if (r0 > rmax || (long long)r0 * r0 >= best) break;
                if (max_radius > 0 && r0 > max_radius) { gave_up = true; break; }
                int dh = 0, dw = 0;
                if (r > 0) {
                    const int t = idx - (2 * r - 1) * (2 * r - 1), side = t / (2 * r), pos = t - side * 2 * r;
                    dh = side == 0 ? -r + pos : side == 1 ? r : side == 2 ? r - pos : -r;
                    dw = side == 0 ? -r : side == 1 ? -r + pos : side == 2 ? r : r - pos;
                }
                const int hh = h + dh, ww = w + dw, base2 = dh * dh + dw * dw;
                int cand = INT_MAX;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W && base2 < best) {
                    const int g = nearest_in_row(plane + ((size_t)hh * W + ww) * nseg, nseg, D, z, inside);
                    if (g != INT_MAX) cand = base2 + g * g;
                }
                for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
                best = min(best, cand);
            }
            if (lane == 0) {
                if (gave_up) atomicMax(&overflow_all[(size_t)(ql - 1) * overflow_stride], 2);
                else if (best < 0 || best >= nbins) atomicMax(&overflow_all[(size_t)(ql - 1) * overflow_stride], 1);   // no voxel of the wanted kind in map a
                else if (ql <= LL && best < LB) atomicAdd(&low[(ql - 1) * LB + best], 1u);
                else atomicAdd(&hist_all[(size_t)(ql - 1) * hist_stride + best], 1ull);
            }
        }
    }
    cvx_barrier();
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) {
        const int q = i / LB, bin = i - q * LB;
        if (low[i] && q < nl && bin < nbins) atomicAdd(&hist_all[(size_t)q * hist_stride + bin], (unsigned long long)low[i]);
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" size_t cvx_label_bits_bytes(int H, int W, int D, int num_labels) {
    if (H <= 0 || W <= 0 || D <= 0 || num_labels <= 0) return 0;
    return sizeof(unsigned long long) * (size_t)num_labels * H * W * ((D + 63) / 64);
}

extern "C" int cvx_label_bits_u64(const float* seg, int H, int W, int D, int num_labels, uint64_t* bits, void* stream) {
    CVX_REQUIRE(seg && bits && H > 0 && W > 0 && D > 0 && num_labels > 0 && num_labels <= 255, "cvx_label_bits_u64: bad arguments (1 .. 255 labels)");
    CVX_REQUIRE((int64_t)H * W <= INT_MAX / 64, "cvx_label_bits_u64: too many rows");
    const int nrows = H * W;
    hipLaunchKernelGGL(k_label_bits, dim3((unsigned)cdiv64(nrows, 4)), dim3(256), 0, as_stream(stream), seg, nrows, D, (D + 63) / 64, num_labels,
                       reinterpret_cast<unsigned long long*>(bits));
    return check_last("label_bits");
}

extern "C" int cvx_surface_distance_hist_i64(const float* seg_b, const uint64_t* bits_a, int H, int W, int D, int num_labels, const uint64_t* active4,
                                             int nbins, int64_t* hist, int64_t hist_stride, int* overflow, int overflow_stride, int max_radius, void* stream) {
    CVX_REQUIRE(seg_b && bits_a && hist && overflow && active4 && H > 0 && W > 0 && D > 0 && num_labels > 0 && num_labels <= 255 && nbins > 0 &&
                    hist_stride >= nbins && overflow_stride >= 1 && max_radius >= 0,
                "cvx_surface_distance_hist_i64: bad arguments (1 .. 255 labels)");
    CVX_REQUIRE(H <= 2047 && W <= 2047 && D <= 32768, "cvx_surface_distance_hist_i64: extent too large (H, W <= 2047, D <= 32768)");
    ActiveLabels act;
    for (int i = 0; i < 4; ++i) act.m[i] = active4[i];
    const int nseg = (D + 63) / 64;
    const int64_t waves = (int64_t)H * W * nseg;
    CVX_REQUIRE(waves <= INT_MAX - 65536, "cvx_surface_distance_hist_i64: volume too large");      // the kernel's wave index is an int that steps past the end by up to one grid
    const int64_t wgs = cdiv64(waves, 4) < 4096 ? cdiv64(waves, 4) : 4096;
    hipLaunchKernelGGL(k_surface_dist_hist, dim3((unsigned)wgs), dim3(256), 0, as_stream(stream), seg_b,
                       reinterpret_cast<const unsigned long long*>(bits_a), H, W, D, nseg, num_labels, act, nbins,
                       reinterpret_cast<unsigned long long*>(hist), (size_t)hist_stride, overflow, overflow_stride, max_radius);
    return check_last("surface_distance_hist");
}
