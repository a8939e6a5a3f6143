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


// ---- the same histogram from the bit planes of BOTH maps, 64 voxels per thread (round 5) ----------------------------------------------
// k_surface_dist_hist spends one lane per voxel on window extraction and, beyond squared distance 8, one wavefront per voxel on a ring
// search: 275 us on a well registered 160x192x224 pair of 17 labels, 1.07 ms at an HD95 of 6 (30 % of the two-stage sweep).  With the
// planes of map b at hand as well (cupy_hd95 builds both for the two directions anyway) the whole thing is word arithmetic:
//   k_surf_words   one thread per word of b's planes: surface = S & ~(the six in-bounds neighbours are all in S); the non-zero words of the
//                  active labels go to a compact list (a few per cent of the 2 M words)
//   k_surf_levels  one thread per listed word.  The 64 surface bits split into those outside the label in map a (target: a's set bits)
//                  and inside (target: the zero bits of a that are voxels).  Stage A: the 3 x 3 rows around the word's row, shifts by
//                  0 .. 1 along D -> one mask per squared distance 1, 2, 3 ("some target lies exactly there"); the bits are counted at the
//                  FIRST level that covers them.  Stage B (compacted: only the words with bits left): the offsets (dh, dw, dz) with squared
//                  distance 4 .. 8 (a table), one level mask per distance.  A cube of radius R holds every voxel at squared distance
//                  < (R + 1)^2, so every level is exact.  The bits left after level 8 become one list entry per voxel.
//   k_surf_voxels  a quad of lanes per such voxel: rows in order of their distance, the 63 bits around z of each; two compacted rounds
//                  (256 rows, then all 3 969 of the square of radius 31); exact below squared distance 1 024.
//   k_surf_far     one wavefront per voxel that is farther (or was handed over): the ring search of k_surface_dist_hist (64 rows per step,
//                  exact, stops at r^2 >= best).
// Same counts as k_surface_dist_hist bit for bit (tests/test_gpu_parity.py::test_hd95_*: equal to the transform method).
struct SurfWord { unsigned long long surf; unsigned at; unsigned q; };          // at = row * nseg + sg, q = label
// Both work lists are SURF_NL sub-lists with a counter each: appends to ONE counter serialise in the L2 (measured: ~10 ns per returning
// atomic on the same address -- 19 000 wavefront appends took 207 us), 256 counters on different lines do not
constexpr int SURF_NL = 256;
struct SurfLists {            // n_*: [SURF_NL] counters, cap_*: per sub-list
    SurfWord* words[2]; unsigned* n_words[2];          // words[s] = input of k_surf_levels<s>
    unsigned long long* vox[2]; unsigned* n_vox[2];     // voxels left after level 8: input of the two rounds of k_surf_voxels   (label << 56 | z << 32 | row)
    unsigned long long* far; unsigned* n_far;           // voxels left after that: input of k_surf_far
    unsigned cap_words, cap_vox, cap_far;
};

__global__ __launch_bounds__(256) void k_surf_words(const unsigned long long* __restrict__ bits_b, int H, int W, int D, int nseg, int nl,
                                                    ActiveLabels act, SurfLists L) {
    const size_t per = (size_t)H * W * nseg;              // (< 2^31: checked by the caller)
    const int q = (int)blockIdx.y + 1;                    // grid.y = label plane
    if (!((act.m[q >> 6] >> (q & 63)) & 1ull)) return;
    const unsigned at = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long surf = 0ull;
    if (at < per) {
        const int row = (int)(at / (unsigned)nseg), sg = (int)(at - (unsigned)row * nseg), h = row / W, w = row - h * W;
        const unsigned long long* B = bits_b + (size_t)(q - 1) * per;
        const unsigned long long S = B[at];
        if (S) {
            // bit z of a neighbour mask: the neighbour of voxel z in that direction carries the label too -- or does not exist (the
            // reference compares with in-bounds neighbours only)
            const unsigned long long zm = (S << 1) | (sg > 0 ? B[at - 1] >> 63 : 1ull);
            unsigned long long zp = (S >> 1) | (sg + 1 < nseg ? B[at + 1] << 63 : 0ull);
            if (sg == nseg - 1) zp |= 1ull << ((D - 1) & 63);
            const unsigned long long wm = w > 0 ? B[at - nseg] : ~0ull, wp = w + 1 < W ? B[at + nseg] : ~0ull;
            const unsigned long long hm = h > 0 ? B[at - (size_t)W * nseg] : ~0ull, hp = h + 1 < H ? B[at + (size_t)W * nseg] : ~0ull;
            surf = S & ~(zm & zp & wm & wp & hm & hp);
        }
    }
    // one counter update per wavefront; its entries stay together and in word order (the lanes of k_surf_levels then read neighbouring words)
    const unsigned long long has = __ballot(surf != 0ull);
    if (!has) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0u;
    const unsigned sub = (blockIdx.x * 4u + (threadIdx.x >> 6)) % (unsigned)SURF_NL;
    if (lane == __builtin_ctzll(has)) base = atomicAdd(&L.n_words[0][sub], (unsigned)__builtin_popcountll(has));
    base = __shfl(base, __builtin_ctzll(has));
    if (surf) {
        const unsigned slot = base + (unsigned)__builtin_popcountll(has & ((1ull << lane) - 1ull));
        if (slot < L.cap_words) L.words[0][(size_t)sub * L.cap_words + slot] = SurfWord{surf, at, (unsigned)q};
        // (a full sub-list: k_surf_levels sees the counter above the capacity and flags every active label)
    }
}

// target words (previous, own, next segment of one row) shifted by dz in both directions: bit z set <=> a target at z - dz or z + dz
__device__ __forceinline__ unsigned long long surf_hit(unsigned long long tp, unsigned long long t, unsigned long long tn, int dz) {
    if (dz == 0) return t;
    return (t << dz) | (tp >> (64 - dz)) | (t >> dz) | (tn << (64 - dz));
}

// stages B and C as a table: the (dh, dw, dz >= 0) inside the cube of radius SURF_R with SURF_K0 <= dh^2 + dw^2 + dz^2 < (SURF_R + 1)^2, ordered by
// squared distance (counting sort at compile time) -- a rolled loop that ends as soon as the word has no bits left
constexpr int SURF_R = 2, SURF_K0 = 4, SURF_K1 = (SURF_R + 1) * (SURF_R + 1) - 1, SURF_TMAX = 1280;
struct SurfTable { unsigned e[SURF_TMAX]; int n; };      // entry = k << 24 | dz << 16 | (dw + 8) << 8 | (dh + 8): one scalar load per entry
constexpr SurfTable surf_make_table() {
    SurfTable t{};
    int first[SURF_K1 + 2] = {};
    for (int dh = -SURF_R; dh <= SURF_R; ++dh)
        for (int dw = -SURF_R; dw <= SURF_R; ++dw)
            for (int dz = 0; dz <= SURF_R; ++dz) {
                const int k = dh * dh + dw * dw + dz * dz;
                if (k >= SURF_K0 && k <= SURF_K1) ++first[k + 1];
            }
    for (int k = 1; k <= SURF_K1 + 1; ++k) first[k] += first[k - 1];
    t.n = first[SURF_K1 + 1];
    for (int dh = -SURF_R; dh <= SURF_R; ++dh)
        for (int dw = -SURF_R; dw <= SURF_R; ++dw)
            for (int dz = 0; dz <= SURF_R; ++dz) {
                const int k = dh * dh + dw * dw + dz * dz;
                if (k >= SURF_K0 && k <= SURF_K1) {
                    const int at = first[k]++;
                    t.e[at] = ((unsigned)k << 24) | ((unsigned)dz << 16) | ((unsigned)(dw + 8) << 8) | (unsigned)(dh + 8);
                }
            }
    return t;
}
static_assert(surf_make_table().n <= SURF_TMAX, "table size");
constexpr int surf_first_ge(int k) { const SurfTable t = surf_make_table(); int i = 0; while (i < t.n && (int)(t.e[i] >> 24) < k) ++i; return i; }
constexpr int SURF_T9 = surf_first_ge(9);          // entries [0, SURF_T9): squared distances 4 .. 8 (the cube of radius 2)
__constant__ SurfTable SURF_T = surf_make_table();

// level masks of one cube: rows |dh|, |dw| <= R, shifts 0 .. R, squared distances KLO .. KHI (all loops unrolled: the level index of a
// (row, shift) is a compile-time constant, combinations outside KLO .. KHI vanish).  LO: targets = set bits, LI: targets = zero bits.
template <int R, int KLO, int KHI>
__device__ __forceinline__ void surf_cube(const unsigned long long* __restrict__ PA, int H, int W, int nseg, int h, int w, int sg,
                                          unsigned long long vmp, unsigned long long vm, unsigned long long vmn,
                                          unsigned long long (&LO)[KHI - KLO + 1], unsigned long long (&LI)[KHI - KLO + 1]) {
#pragma unroll
    for (int dh = -R; dh <= R; ++dh)
#pragma unroll
        for (int dw = -R; dw <= R; ++dw) {
            const int base2 = dh * dh + dw * dw;
            if (base2 > KHI) continue;
            // (a row that contributed every shift it can to the levels below KLO in an earlier stage still has larger shifts to give)
            if (base2 + R * R < KLO) continue;
            const int hh = h + dh, ww = w + dw;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const unsigned long long* rb = PA + ((size_t)(ok ? hh : h) * W + (ok ? ww : w)) * nseg + sg;
            unsigned long long m = rb[0], mp = sg > 0 ? rb[-1] : 0ull, mn = sg + 1 < nseg ? rb[1] : 0ull;
            const unsigned long long okm = ok ? ~0ull : 0ull;
            const unsigned long long c = ~m & vm & okm, cp = ~mp & vmp & okm, cn = ~mn & vmn & okm;
            m &= okm; mp &= okm; mn &= okm;
#pragma unroll
            for (int dz = 0; dz <= R; ++dz) {
                const int k = base2 + dz * dz;
                if (k < KLO || k > KHI) continue;
                LO[k - KLO] |= surf_hit(mp, m, mn, dz);
                LI[k - KLO] |= surf_hit(cp, c, cn, dz);
            }
        }
}

// STAGE 0: levels 1 .. 3 (unrolled 3 x 3 cube) of every surface word; 1: levels 4 .. 8 of the words that still hold bits, then one list entry
// per voxel that is left.  Two launches with a compaction in between: a wavefront waits for its slowest word.
template <int STAGE>
__global__ __launch_bounds__(256) void k_surf_levels(const unsigned long long* __restrict__ bits_a, int H, int W, int D, int nseg, int nl,
                                                     int nbins, unsigned long long* __restrict__ hist_all, size_t hist_stride,
                                                     int* __restrict__ overflow_all, int overflow_stride, SurfLists L) {
    constexpr int LL = 64, LB = 64;
    __shared__ unsigned int low[LL * LB];
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) low[i] = 0;
    cvx_barrier();
    const size_t per = (size_t)H * W * nseg;
    // workgroup (sub-list, part): grid = SURF_NL x parts
    const unsigned sub = blockIdx.x % (unsigned)SURF_NL, part = blockIdx.x / (unsigned)SURF_NL, parts = gridDim.x / (unsigned)SURF_NL;
    const unsigned n_all = L.n_words[STAGE][sub], n = min(n_all, L.cap_words);
    if (n_all > L.cap_words && part == 0 && threadIdx.x < (unsigned)nl) atomicMax(&overflow_all[(size_t)threadIdx.x * overflow_stride], 2);   // lost words: hand over
    const unsigned long long tail = (D & 63) ? (1ull << (D & 63)) - 1ull : ~0ull;
    for (unsigned e0 = part * blockDim.x; e0 < n; e0 += parts * blockDim.x) {          // (uniform trip count: the body holds wavefront-wide ballots)
        const unsigned e = e0 + threadIdx.x;
        const bool live = e < n;
        const SurfWord wd = live ? L.words[STAGE][(size_t)sub * L.cap_words + e] : SurfWord{0ull, 0u, 1u};
        const int q = (int)wd.q, row = (int)(wd.at / (unsigned)nseg), sg = (int)(wd.at - (unsigned)row * nseg), h = row / W, w = row - h * W;
        const unsigned long long* PA = bits_a + (size_t)(q - 1) * per;
        const unsigned long long A0 = PA[wd.at];
        unsigned long long rem_o = wd.surf & ~A0, rem_i = wd.surf & A0;         // outside l in map a: nearest set bit; inside: nearest zero bit
        // which bit positions of the three segments are voxels
        const unsigned long long vm = sg == nseg - 1 ? tail : ~0ull, vmp = sg > 0 ? ~0ull : 0ull,
                                 vmn = sg + 1 < nseg ? (sg + 1 == nseg - 1 ? tail : ~0ull) : 0ull;
        auto settle = [&](int k, unsigned long long lo, unsigned long long li) {
            const unsigned c = (unsigned)(__builtin_popcountll(rem_o & lo) + __builtin_popcountll(rem_i & li));
            rem_o &= ~lo; rem_i &= ~li;
            if (!c) return;
            if (k >= nbins) atomicMax(&overflow_all[(size_t)(q - 1) * overflow_stride], 1);
            else if (q <= LL && k < LB) atomicAdd(&low[(q - 1) * LB + k], c);
            else atomicAdd(&hist_all[(size_t)(q - 1) * hist_stride + k], (unsigned long long)c);
        };
        if (STAGE == 0) {
            unsigned long long LO[3] = {0, 0, 0}, LI[3] = {0, 0, 0};
            surf_cube<1, 1, 3>(PA, H, W, nseg, h, w, sg, vmp, vm, vmn, LO, LI);
#pragma unroll
            for (int k = 1; k <= 3; ++k) settle(k, LO[k - 1], LI[k - 1]);
        } else {
            // batches of NB table entries: their 3 NB loads are in flight together (one entry at a time the loop ran at one memory round
            // trip per entry: ~100 us for a word that goes through the whole table)
            constexpr int NB = 8;
            const int nt = SURF_T9;
            int cur_k = SURF_K0;
            unsigned long long lo = 0ull, li = 0ull;
            for (int t0 = 0; t0 < nt && (rem_o | rem_i); t0 += NB) {
                unsigned long long m[NB], mp[NB], mn[NB];
                unsigned ent[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    ent[j] = SURF_T.e[min(t0 + j, nt - 1)];
                    const int hh = h + (int)(ent[j] & 255u) - 8, ww = w + (int)((ent[j] >> 8) & 255u) - 8;
                    const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;           // (a row outside the volume is skipped below: any valid address will do)
                    const unsigned long long* rb = PA + ((size_t)(ok ? hh : h) * W + (ok ? ww : w)) * nseg + sg;
                    m[j] = rb[0];
                    mp[j] = sg > 0 ? rb[-1] : 0ull;
                    mn[j] = sg + 1 < nseg ? rb[1] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (t0 + j >= nt) break;
                    const int k = (int)(ent[j] >> 24), dz = (int)((ent[j] >> 16) & 255u);
                    const int hh = h + (int)(ent[j] & 255u) - 8, ww = w + (int)((ent[j] >> 8) & 255u) - 8;
                    const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                    if (k != cur_k) { settle(cur_k, lo, li); lo = 0ull; li = 0ull; cur_k = k; }
                    if (!ok) continue;
                    lo |= surf_hit(mp[j], m[j], mn[j], dz);
                    li |= surf_hit(~mp[j] & vmp, ~m[j] & vm, ~mn[j] & vmn, dz);
                }
            }
            settle(cur_k, lo, li);
        }
        unsigned long long rest = rem_o | rem_i;
        if (STAGE == 0) {                    // words with bits left: input of the next stage (one counter update per wavefront)
            const unsigned long long has = __ballot(rest != 0ull);
            if (has) {
                const int lane = threadIdx.x & 63, leader = __builtin_ctzll(has);
                unsigned base = 0u;
                if (lane == leader) base = atomicAdd(&L.n_words[1][sub], (unsigned)__builtin_popcountll(has));
                base = __shfl(base, leader);
                if (rest) {
                    const unsigned slot = base + (unsigned)__builtin_popcountll(has & ((1ull << lane) - 1ull));
                    if (slot < L.cap_words) L.words[1][(size_t)sub * L.cap_words + slot] = SurfWord{rest, wd.at, wd.q};
                }
            }
        } else {                             // voxels beyond squared distance 8: one list entry each (label, row, z); one counter update per wavefront
            const int lane = threadIdx.x & 63;
            const unsigned mine = (unsigned)__builtin_popcountll(rest);
            unsigned incl = mine;                                              // inclusive prefix sum over the lanes
            for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            const unsigned total = __shfl(incl, 63);
            if (total) {
                const unsigned vsub = (sub * 7u + (threadIdx.x >> 6) + (e0 >> 8)) % (unsigned)SURF_NL;
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(&L.n_vox[0][vsub], total);
                unsigned slot = __shfl(base, 0) + incl - mine;
                while (rest) {
                    const int b = __builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    if (slot < L.cap_vox) L.vox[0][(size_t)vsub * L.cap_vox + slot] = ((unsigned long long)q << 56) | ((unsigned long long)(sg * 64 + b) << 32) | (unsigned)row;
                    else atomicMax(&overflow_all[(size_t)(q - 1) * overflow_stride], 2);         // list full: the label is handed to the transforms
                    ++slot;
                }
            }
        }
    }
    cvx_barrier();
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) {
        const int q = i / LB, bin = i - q * LB;
        if (low[i] && q < nl && bin < nbins) atomicAdd(&hist_all[(size_t)q * hist_stride + bin], (unsigned long long)low[i]);
    }
}

// rows (dh, dw) of the square of radius SURF_VR around a voxel's row, ordered by dh^2 + dw^2 (counting sort at compile time)
constexpr int SURF_VR = 31, SURF_VROWS = (2 * SURF_VR + 1) * (2 * SURF_VR + 1);
struct SurfRows { unsigned e[SURF_VROWS]; };             // entry = base2 << 16 | (dw + 32) << 8 | (dh + 32)
constexpr SurfRows surf_make_rows() {
    SurfRows t{};
    int first[2 * SURF_VR * SURF_VR + 2] = {};
    for (int dh = -SURF_VR; dh <= SURF_VR; ++dh)
        for (int dw = -SURF_VR; dw <= SURF_VR; ++dw) ++first[dh * dh + dw * dw + 1];
    for (int k = 1; k <= 2 * SURF_VR * SURF_VR + 1; ++k) first[k] += first[k - 1];
    for (int dh = -SURF_VR; dh <= SURF_VR; ++dh)
        for (int dw = -SURF_VR; dw <= SURF_VR; ++dw) {
            const int k = dh * dh + dw * dw;
            t.e[first[k]++] = ((unsigned)k << 16) | ((unsigned)(dw + 32) << 8) | (unsigned)(dh + 32);
        }
    return t;
}
__constant__ SurfRows SURF_ROWS = surf_make_rows();
// round 0 of k_surf_voxels scans the SURF_VROWS0 nearest rows: every other row is at least sqrt(SURF_FINAL0) away
constexpr int SURF_VROWS0 = 256, SURF_FINAL0 = (int)(surf_make_rows().e[SURF_VROWS0] >> 16);
static_assert(SURF_FINAL0 <= (SURF_VR + 1) * (SURF_VR + 1), "round 0 is bounded by its rows, not by the window");

// One LANE per voxel that is farther than squared distance 8 from its target (the compacted leftovers of k_surf_levels<1>): the rows around
// it in order of dh^2 + dw^2; of each row the 63 bits z-31 .. z+31 (own word and a neighbour), nearest target bit by count-leading /
// trailing-zeros.  A lane is done when the next row is at least sqrt(best) away; the cube of radius 31 holds every voxel at squared
// distance < 1024, so a minimum below that is final.  What is left (or has no target inside the cube) goes to the ring search.
// (The same loop inside k_surface_dist_hist, round 5, was SLOWER than its wavefront-per-voxel search: there a few far lanes kept whole
// wavefronts of settled voxels waiting.  Here every lane is a far voxel.  The search radius option does not apply: it bounds the ring search.)
// Two rounds with a compaction in between (a wavefront waits for its slowest lane): ROUND 0 -- the 256 nearest rows, final below the squared
// distance of the first row it does not scan; ROUND 1 -- what is left, all 3 969 rows from the start, final below 1 024; once fewer than
// SURF_MIN_LANES lanes of a wavefront are still searching beyond the first 256 rows they are handed to the ring search (one wavefront per
// voxel, better for the few).
constexpr int SURF_MIN_LANES = 8;
template <int ROUND>
__global__ __launch_bounds__(256) void k_surf_voxels(const unsigned long long* __restrict__ bits_a, int H, int W, int D, int nseg, int nbins,
                                                     unsigned long long* __restrict__ hist_all, size_t hist_stride, int* __restrict__ overflow_all,
                                                     int overflow_stride, SurfLists L) {
    constexpr int NROWS = ROUND == 0 ? SURF_VROWS0 : SURF_VROWS, FINAL_BELOW = ROUND == 0 ? SURF_FINAL0 : (SURF_VR + 1) * (SURF_VR + 1);
    unsigned long long* const out_list = ROUND == 0 ? L.vox[1] : L.far;
    unsigned* const out_n = ROUND == 0 ? L.n_vox[1] : L.n_far;
    const unsigned out_cap = ROUND == 0 ? L.cap_vox : L.cap_far;
    // a QUAD of lanes per voxel: lane s of the quad takes rows t0 + 4 j + s of a step (32 rows per step and voxel), the quad shares its best
    // value after every step -- four times fewer dependent steps per voxel than one lane each (a step is a memory round trip), and a
    // wavefront waits for the slowest of 16 voxels instead of 64
    constexpr int LL = 64, LB = 128, R = SURF_VR, NB = 8, LPV = 4, VPW = 256 / LPV;
    __shared__ unsigned int low[LL * LB];
    __shared__ unsigned rows[NROWS];                   // the row table in LDS: the lanes of a quad read different entries (from constant memory: one more memory round trip per step)
    const unsigned sub = blockIdx.x % (unsigned)SURF_NL, part = blockIdx.x / (unsigned)SURF_NL, parts = gridDim.x / (unsigned)SURF_NL;
    const unsigned n_all = L.n_vox[ROUND][sub], n = min(n_all, L.cap_vox);
    if (part * VPW >= n) return;
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) low[i] = 0;
    for (int i = threadIdx.x; i < NROWS; i += blockDim.x) rows[i] = SURF_ROWS.e[i];
    cvx_barrier();
    const size_t per = (size_t)H * W * nseg;
    const int sl = threadIdx.x & (LPV - 1);
    for (unsigned e0 = part * VPW; e0 < n; e0 += parts * VPW) {
        const unsigned e = e0 + (threadIdx.x >> 2);
        const bool live = e < n;
        const unsigned long long it = live ? L.vox[ROUND][(size_t)sub * L.cap_vox + e] : (1ull << 56);
        const int q = (int)(it >> 56), z = (int)((it >> 32) & 0xffffffu), row = (int)(unsigned)(it & 0xffffffffull);
        const int h = row / W, w = row - h * W, sg = z >> 6, zb = z & 63;
        const unsigned long long* PA = bits_a + (size_t)(q - 1) * per;
        const bool inside = live && ((PA[(size_t)row * nseg + sg] >> zb) & 1ull);       // inside the label in map a: the target is the nearest voxel outside
        // window bit j <-> voxel z - R + j; valid voxels only (the complement of a row must not count positions outside the volume)
        const int jlo = max(0, R - z), jhi = min(2 * R + 1, D - z + R);              // valid window bits [jlo, jhi)
        const unsigned long long vwin = ((jhi >= 64 ? ~0ull : (1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull)) & ((1ull << (2 * R + 1)) - 1ull);
        int best = live ? INT_MAX : 0;
        bool handed = false;
        for (int t0 = 0; t0 < NROWS; t0 += NB * LPV) {
            const int base2_0 = (int)(rows[t0] >> 16);
            const unsigned long long busy = __ballot(base2_0 < best);
            if (!busy) break;                                                      // every voxel done (rows are ordered: nothing closer can follow)
            if (ROUND == 1 && t0 >= SURF_VROWS0 && __builtin_popcountll(busy) < SURF_MIN_LANES * LPV) { handed = base2_0 < best; break; }
            unsigned long long m[NB], mp[NB], mn[NB];
            unsigned ent[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                ent[j] = rows[min(t0 + LPV * j + sl, NROWS - 1)];
                const int hh = h + (int)(ent[j] & 255u) - 32, ww = w + (int)((ent[j] >> 8) & 255u) - 32;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                const unsigned long long* rb = PA + ((size_t)(ok ? hh : h) * W + (ok ? ww : w)) * nseg + sg;
                m[j] = rb[0];
                mp[j] = (zb < R && sg > 0) ? rb[-1] : 0ull;
                mn[j] = (zb > 63 - R && sg + 1 < nseg) ? rb[1] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int base2 = (int)(ent[j] >> 16);
                const int hh = h + (int)(ent[j] & 255u) - 32, ww = w + (int)((ent[j] >> 8) & 255u) - 32;
                if (t0 + LPV * j + sl >= NROWS || base2 >= best || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                unsigned long long win = zb >= R ? m[j] >> (zb - R) : m[j] << (R - zb);
                if (zb < R) win |= mp[j] >> (64 - R + zb);
                if (zb > 63 - R) win |= mn[j] << (64 + R - zb);
                win = (inside ? ~win : win) & vwin;
                if (!win) continue;
                const unsigned long long left = win & ((2ull << R) - 1ull), right = win >> R;
                const int dl = left ? R - (63 - __builtin_clzll(left)) : INT_MAX, dr = right ? __builtin_ctzll(right) : INT_MAX;
                const int g = min(dl, dr);
                best = min(best, base2 + g * g);
            }
            best = min(best, __builtin_amdgcn_update_dpp(best, best, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
            best = min(best, __builtin_amdgcn_update_dpp(best, best, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]: the quad agrees
        }
        const bool mine = live && sl == 0;                                          // one lane of the quad reports
        const bool settled = mine && !handed && best < FINAL_BELOW;
        if (settled) {
            if (best >= nbins) atomicMax(&overflow_all[(size_t)(q - 1) * overflow_stride], 1);
            else if (q <= LL && best < LB) atomicAdd(&low[(q - 1) * LB + best], 1u);
            else atomicAdd(&hist_all[(size_t)(q - 1) * hist_stride + best], 1ull);
        }
        // the others (beyond the cube, no target in it, or handed over): next list, one counter update per wavefront
        const unsigned long long pass = __ballot(mine && !settled);
        if (pass) {
            const int lane = threadIdx.x & 63, leader = __builtin_ctzll(pass);
            const unsigned osub = (sub * 7u + (threadIdx.x >> 6) + (e0 >> 6)) % (unsigned)SURF_NL;
            unsigned base = 0u;
            if (lane == leader) base = atomicAdd(&out_n[osub], (unsigned)__builtin_popcountll(pass));
            base = __shfl(base, leader);
            if (mine && !settled) {
                const unsigned slot = base + (unsigned)__builtin_popcountll(pass & ((1ull << lane) - 1ull));
                if (slot < out_cap) out_list[(size_t)osub * out_cap + slot] = it;
                else atomicMax(&overflow_all[(size_t)(q - 1) * overflow_stride], 2);
            }
        }
    }
    cvx_barrier();
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) {
        const int q = i / LB, bin = i - q * LB;
        if (low[i] && bin < nbins) atomicAdd(&hist_all[(size_t)q * hist_stride + bin], (unsigned long long)low[i]);
    }
}

// minimum over the 64 lanes (in every lane): four DPP steps inside the rows of 16 lanes, then the four row results through readlane
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));     // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));     // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// one wavefront per far voxel (squared distance >= 64): rows in square rings around its row, 64 rows per step
__global__ __launch_bounds__(256) void k_surf_far(const unsigned long long* __restrict__ bits_a, int H, int W, int D, int nseg, int nbins,
                                                  unsigned long long* __restrict__ hist_all, size_t hist_stride, int* __restrict__ overflow_all,
                                                  int overflow_stride, int max_radius, SurfLists L) {
    constexpr int LL = 64, LB = 128;
    __shared__ unsigned int low[LL * LB];
    const int nrows = H * W, lane = threadIdx.x & 63;
    const unsigned sub = blockIdx.x % (unsigned)SURF_NL, part = blockIdx.x / (unsigned)SURF_NL, parts = gridDim.x / (unsigned)SURF_NL;
    const unsigned n = min(L.n_far[sub], L.cap_far);
    if (part * (blockDim.x >> 6) >= n) return;             // nothing for this workgroup (before the 32 KB of LDS are cleared and flushed)
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) low[i] = 0;
    cvx_barrier();
    const unsigned wave0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(part * (blockDim.x >> 6) + (threadIdx.x >> 6))), nw = parts * (blockDim.x >> 6);
    for (unsigned e = wave0; e < n; e += nw) {
        const unsigned long long it = L.far[(size_t)sub * L.cap_far + e];
        const int ql = (int)(it >> 56), z = (int)((it >> 32) & 0xffffffu), row = (int)(unsigned)(it & 0xffffffffull);
        const int h = row / W, w = row - h * W;
        // a label that has already exceeded the radius is void for the caller (flag 2): its other far voxels are skipped.  A plain load: a stale
        // value only costs the searches it would have saved (the agent-scope atomic load here was a memory round trip per voxel)
        if (max_radius > 0 && overflow_all[(size_t)(ql - 1) * overflow_stride] == 2) continue;
        const unsigned long long* plane = bits_a + (size_t)(ql - 1) * nrows * nseg;
        const bool inside = (plane[(size_t)row * nseg + (z >> 6)] >> (z & 63)) & 1ull;
        const int rmax = max(max(h, H - 1 - h), max(w, W - 1 - w));
        int best = INT_MAX;
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
            best = min(best, wave_min_i32(cand));
        }
        if (lane == 0) {
            if (gave_up) atomicMax(&overflow_all[(size_t)(ql - 1) * overflow_stride], 2);
            else if (best < 0 || best >= nbins) atomicMax(&overflow_all[(size_t)(ql - 1) * overflow_stride], 1);   // no voxel of the wanted kind in map a
            else if (ql <= LL && best < LB) atomicAdd(&low[(ql - 1) * LB + best], 1u);
            else atomicAdd(&hist_all[(size_t)(ql - 1) * hist_stride + best], 1ull);
        }
    }
    cvx_barrier();
    for (int i = threadIdx.x; i < LL * LB; i += blockDim.x) {
        const int q = i / LB, bin = i - q * LB;
        if (low[i] && bin < nbins) atomicAdd(&hist_all[(size_t)q * hist_stride + bin], (unsigned long long)low[i]);
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

// capacity of one of the SURF_NL sub-lists: at most 2^14 entries (4 M surface words / far voxels in all), and no more than a small volume
// can fill.  The sub-list of an entry follows from the block and wavefront that found it, so a volume of a few blocks fills only a few
// sub-lists: the bound for ONE sub-list is every word of the maps (the workspace is grow-only per thread, device and stream;
// tiny label maps no longer pin 224 MB.  Round 6: the first version divided by the number of sub-lists and overflowed on 12 x 10 x 64 maps)
static size_t surf_list_cap(int H, int W, int D, int num_labels) {
    const size_t words = (size_t)num_labels * H * W * ((D + 63) / 64) + 64;
    return words < ((size_t)1 << 14) ? words : ((size_t)1 << 14);
}
// the voxel lists (near / far surface voxels) hold VOXELS: a voxel carries one label, so H W D bounds all of them together
static size_t surf_vox_cap(int H, int W, int D) {
    const size_t vox = (size_t)H * W * D + 64;
    return vox < ((size_t)1 << 14) ? vox : ((size_t)1 << 14);
}
// bits_b / bits_a = cvx_label_bits_u64 of the two maps; otherwise as cvx_surface_distance_hist_i64 (same counts, same flags)
extern "C" size_t cvx_surface_distance_hist_bits_workspace_bytes(int H, int W, int D, int num_labels) {
    if (H <= 0 || W <= 0 || D <= 0 || num_labels <= 0) return 0;
    const size_t cap_words = surf_list_cap(H, W, D, num_labels), cap_far = surf_vox_cap(H, W, D);       // per sub-list; beyond that the call reports flag 2
    return 256 + sizeof(unsigned) * 5 * SURF_NL + 2 * (256 + sizeof(SurfWord) * cap_words * SURF_NL) + 3 * (256 + sizeof(unsigned long long) * cap_far * SURF_NL) + 256;
}

extern "C" int cvx_surface_distance_hist_bits_i64(const uint64_t* bits_b, const uint64_t* bits_a, int H, int W, int D, int num_labels,
                                                  const uint64_t* active4, int nbins, int64_t* hist, int64_t hist_stride, int* overflow,
                                                  int overflow_stride, int max_radius, void* workspace, size_t workspace_bytes, void* stream) {
    CVX_REQUIRE(bits_b && bits_a && hist && overflow && active4 && workspace && H > 0 && W > 0 && D > 0 && num_labels > 0 && num_labels <= 255 &&
                    nbins > 0 && hist_stride >= nbins && overflow_stride >= 1 && max_radius >= 0,
                "cvx_surface_distance_hist_bits_i64: bad arguments (1 .. 255 labels)");
    CVX_REQUIRE(H <= 2047 && W <= 2047 && D <= 32768, "cvx_surface_distance_hist_bits_i64: extent too large (H, W <= 2047, D <= 32768)");
    if (workspace_bytes < cvx_surface_distance_hist_bits_workspace_bytes(H, W, D, num_labels))
        return fail(CVX_ERR_WORKSPACE, "cvx_surface_distance_hist_bits_i64: workspace too small");
    ActiveLabels act;
    for (int i = 0; i < 4; ++i) act.m[i] = active4[i];
    const int nseg = (D + 63) / 64;
    const size_t words = (size_t)num_labels * H * W * nseg;
    CVX_REQUIRE((int64_t)H * W * nseg <= INT_MAX - 65536 && words < ((size_t)1 << 32), "cvx_surface_distance_hist_bits_i64: volume too large");
    hipStream_t s = as_stream(stream);
    Carver cv(workspace, workspace_bytes);
    unsigned* counters = cv.take<unsigned>(5 * SURF_NL);
    SurfLists L;
    L.cap_words = (unsigned)surf_list_cap(H, W, D, num_labels);
    L.cap_far = (unsigned)surf_vox_cap(H, W, D);
    L.cap_vox = L.cap_far;
    for (int i = 0; i < 2; ++i) { L.words[i] = cv.take<SurfWord>((size_t)L.cap_words * SURF_NL); L.n_words[i] = counters + i * SURF_NL; }
    for (int i = 0; i < 2; ++i) { L.vox[i] = cv.take<unsigned long long>((size_t)L.cap_vox * SURF_NL); L.n_vox[i] = counters + (2 + i) * SURF_NL; }
    L.far = cv.take<unsigned long long>((size_t)L.cap_far * SURF_NL);
    L.n_far = counters + 4 * SURF_NL;
    if (hipMemsetAsync(counters, 0, 5 * SURF_NL * sizeof(unsigned), s) != hipSuccess) return fail(CVX_ERR_LAUNCH, "surface_distance_hist_bits: memset failed");
    const unsigned long long* Bb = reinterpret_cast<const unsigned long long*>(bits_b);
    const unsigned long long* Ba = reinterpret_cast<const unsigned long long*>(bits_a);
    unsigned long long* hh = reinterpret_cast<unsigned long long*>(hist);
    hipLaunchKernelGGL(k_surf_words, dim3((unsigned)cdiv64((int64_t)H * W * nseg, 256), (unsigned)num_labels), dim3(256), 0, s, Bb, H, W, D, nseg, num_labels, act, L);
    hipLaunchKernelGGL(k_surf_levels<0>, dim3(SURF_NL * 4), dim3(256), 0, s, Ba, H, W, D, nseg, num_labels, nbins, hh, (size_t)hist_stride, overflow, overflow_stride, L);
    hipLaunchKernelGGL(k_surf_levels<1>, dim3(SURF_NL * 4), dim3(256), 0, s, Ba, H, W, D, nseg, num_labels, nbins, hh, (size_t)hist_stride, overflow, overflow_stride, L);
    hipLaunchKernelGGL(k_surf_voxels<0>, dim3(SURF_NL * 8), dim3(256), 0, s, Ba, H, W, D, nseg, nbins, hh, (size_t)hist_stride, overflow, overflow_stride, L);
    hipLaunchKernelGGL(k_surf_voxels<1>, dim3(SURF_NL * 8), dim3(256), 0, s, Ba, H, W, D, nseg, nbins, hh, (size_t)hist_stride, overflow, overflow_stride, L);
    hipLaunchKernelGGL(k_surf_far, dim3(SURF_NL * 8), dim3(256), 0, s, Ba, H, W, D, nseg, nbins, hh, (size_t)hist_stride, overflow, overflow_stride, max_radius, L);
    return check_last("surface_distance_hist_bits");
}

