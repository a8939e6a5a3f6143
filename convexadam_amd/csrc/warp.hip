// warp.hip -- data term of the Adam iteration: trilinear warp of the moving features and its gradient with
// respect to the sampling displacement (reference: convex_adam_MIND.py:170-178 through autograd; ATen
// grid_sampler_3d forward/backward, GridSampler.cpp).  Compiled without SLP vectorisation: the packed-math
// version needs 160 VGPRs (3 waves/SIMD), the scalar one 93 (5 waves/SIMD); packed fp32 has no throughput
// advantage on gfx950 and the kernel is bound by latency hiding.
#include "cvx_common.h"

namespace cvx {

// chunked copies of the pooled features: [C][V] -> [CP/4][V+1][4] (CP = C rounded up to 4, zero filled): one trilinear
// corner of a 4-channel chunk is one 16-byte load, and the 16 consecutive voxels of a tile row read 256 contiguous
// bytes (every byte of the two cache lines is used; a [V][CP] record layout touches CP/4 times as many lines).
// Record V of every chunk is all zero: corners outside the volume are gathered from it.
__global__ __launch_bounds__(256) void k_to_chunked(const float* __restrict__ in, int C, size_t V, float* __restrict__ out) {
    // one thread = one record: 4 coalesced channel reads, one 16-byte store
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > V) return;
    const int c0 = 4 * (int)blockIdx.y;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < V) {
        const float* src = in + (size_t)c0 * V + p;
        r.x = src[0];
        if (c0 + 1 < C) r.y = src[V];
        if (c0 + 2 < C) r.z = src[2 * V];
        if (c0 + 3 < C) r.w = src[3 * V];
    }
    reinterpret_cast<float4*>(out)[(size_t)blockIdx.y * (V + 1) + p] = r;
}

// fp16 STORAGE of the pooled features (the reference's GPU default dtype, convex_adam_MIND.py:79): the same records with four
// half-precision values (8 bytes, rounded to nearest even here); the warp kernel widens them to float32 -- exact -- on load.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_to_chunked_h(const float* __restrict__ in, int C, size_t V, uint2* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > V) return;
    const int c0 = 4 * (int)blockIdx.y;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < V) {
        const float* src = in + (size_t)c0 * V + p;
        r.x = src[0];
        if (c0 + 1 < C) r.y = src[V];
        if (c0 + 2 < C) r.z = src[2 * V];
        if (c0 + 3 < C) r.w = src[3 * V];
    }
    const h16x4 o = {(_Float16)r.x, (_Float16)r.y, (_Float16)r.z, (_Float16)r.w};          // round to nearest even
    out[(size_t)blockIdx.y * (V + 1) + p] = __builtin_bit_cast(uint2, o);
}
// (plain _Float16 vectors: a __builtin_bit_cast to hip_fp16.h's __half2 made the compiler drop the second dword of the load)
__device__ __forceinline__ float4 buffer_load8h(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uni_off) {
    const h16x4 v = __builtin_bit_cast(h16x4, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)uni_off, 0));
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}

// HALF: records of four half-precision values (k_to_chunked_h) instead of four floats; buffer loads only
template <bool BUF, bool HALF = false>
__global__ __launch_bounds__(256) void k_warp_grad(const float* __restrict__ F2, const float* __restrict__ M2, int C, int CP,
                                                   int h, int w, int d, const float* __restrict__ U,
                                                   const float* __restrict__ bh, const float* __restrict__ bw,
                                                   const float* __restrict__ bd, float gsc, float cH, float cW, float cD,
                                                   float* __restrict__ gU, unsigned long long* __restrict__ census, FastDiv dx, FastDiv dy,
                                                   float sc0, float sc1, float sc2) {
    const size_t V = (size_t)h * w * d;
    if (census && threadIdx.x == 0) {                   // debugging aid (option census_ptr, tools/adam_census.py)
        census[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        census[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
    // 4 x 4 x 16 voxel tile per workgroup: the 8-corner footprints of a tile overlap in L1 (each moving-feature
    // record is fetched from L2 about 1.7x instead of 4x with a linear mapping)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so XCD q takes the q-th contiguous slab of
    // tiles and the overlapping footprints of neighbouring tiles meet in the same 4 MB L2
    const int ntx = (d + 15) / 16, nty = (w + 3) / 4, ntz = (h + 3) / 4;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int bid = __builtin_amdgcn_readfirstlane((int)blockIdx.x);                 // (the census branch above otherwise drags the index into a vector register)
    const int tile = (bid & 7) * per_xcd + (bid >> 3);
    if (tile >= ntx * nty * ntz) return;
    const int trow = fastdiv(tile, dx), tbz = fastdiv(trow, dy);                     // scalar unit: tile / ntx, (tile / ntx) / nty
    const int tbx = tile - trow * ntx, tby = trow - tbz * nty;
    const int x = tbx * 16 + (threadIdx.x & 15), y = tby * 4 + ((threadIdx.x >> 4) & 3), z = tbz * 4 + (threadIdx.x >> 6);
    if (x >= d || y >= w || z >= h) return;
    const unsigned p = (unsigned)((z * w + y) * d + x);
    // sc0 = (float)((h - 1) / 2.0), sc1, sc2 likewise: evaluated on the host (launch_warp_grad)                  (:171)
    const float uH = U[p], uW = U[V + p], uD = U[2 * V + p];
    Tri t;
    tri_setup(t, bd[x] + fdiv(uD, sc2), bw[y] + fdiv(uW, sc1), bh[z] + fdiv(uH, sc0), h, w, d);
    const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float fx0 = (float)x0, fy0 = (float)y0, fz0 = (float)z0, fx1 = (float)x1, fy1 = (float)y1, fz1 = (float)z1;
    // Branch-free gathers: a corner outside the volume reads the all-zero record V.  ATen skips such corners; adding
    // their exact-zero products instead is bit-identical here because the features and the trilinear factors are
    // non-negative (products are +0, never -0).  Byte offsets inside one chunk fit 32 bits (checked by the launcher).
    const bool zin0 = (unsigned)z0 < (unsigned)h, zin1 = (unsigned)z1 < (unsigned)h, yin0 = (unsigned)y0 < (unsigned)w,
               yin1 = (unsigned)y1 < (unsigned)w, xin0 = (unsigned)x0 < (unsigned)d, xin1 = (unsigned)x1 < (unsigned)d;
    const int r00 = (z0 * w + y0) * d, r01 = (z0 * w + y1) * d, r10 = (z1 * w + y0) * d, r11 = (z1 * w + y1) * d;
    constexpr unsigned REC = HALF ? 8u : 16u;                             // bytes per record
    const unsigned zero_rec = (unsigned)V * REC;
    unsigned off[8];
    off[0] = (zin0 && yin0 && xin0) ? (unsigned)(r00 + x0) * REC : zero_rec; off[1] = (zin0 && yin0 && xin1) ? (unsigned)(r00 + x1) * REC : zero_rec;
    off[2] = (zin0 && yin1 && xin0) ? (unsigned)(r01 + x0) * REC : zero_rec; off[3] = (zin0 && yin1 && xin1) ? (unsigned)(r01 + x1) * REC : zero_rec;
    off[4] = (zin1 && yin0 && xin0) ? (unsigned)(r10 + x0) * REC : zero_rec; off[5] = (zin1 && yin0 && xin1) ? (unsigned)(r10 + x1) * REC : zero_rec;
    off[6] = (zin1 && yin1 && xin0) ? (unsigned)(r11 + x0) * REC : zero_rec; off[7] = (zin1 && yin1 && xin1) ? (unsigned)(r11 + x1) * REC : zero_rec;
    // forward weights in ATen's corner order and the backward factor pairs per corner (GridSampler.cpp)
    const float wgt[8] = {t.tnw, t.tne, t.tsw, t.tse, t.bnw, t.bne, t.bsw, t.bse};
    const float ax[8] = {fy1 - t.iy, fy1 - t.iy, t.iy - fy0, t.iy - fy0, fy1 - t.iy, fy1 - t.iy, t.iy - fy0, t.iy - fy0};
    const float bx[8] = {fz1 - t.iz, fz1 - t.iz, fz1 - t.iz, fz1 - t.iz, t.iz - fz0, t.iz - fz0, t.iz - fz0, t.iz - fz0};
    const float ay[8] = {fx1 - t.ix, t.ix - fx0, fx1 - t.ix, t.ix - fx0, fx1 - t.ix, t.ix - fx0, fx1 - t.ix, t.ix - fx0};
    const float bz[8] = {fy1 - t.iy, fy1 - t.iy, t.iy - fy0, t.iy - fy0, fy1 - t.iy, fy1 - t.iy, t.iy - fy0, t.iy - fy0};
    float gix = 0.f, giy = 0.f, giz = 0.f;
    const unsigned foff = p * REC;
    const unsigned chunk_bytes = (unsigned)(V + 1) * REC;
    // buffer descriptors: per-lane 32-bit offsets + the chunk offset in a scalar register (9 address registers instead of 18, no
    // 64-bit vector adds in the loop); the launcher guarantees CP/4 * (V + 1) * 16 < 2^32
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M2), 0, (int)(chunk_bytes * (unsigned)(CP / 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F2), 0, (int)(chunk_bytes * (unsigned)(CP / 4)), 0x00020000);
    unsigned coff = 0;
    for (int c0 = 0; c0 < CP / 4; ++c0, coff += chunk_bytes) {
        float vv[8][4], fv[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = HALF ? buffer_load8h(mr, off[k], coff) : BUF ? buffer_load16(mr, off[k], coff) : *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(M2) + (size_t)c0 * chunk_bytes + off[k]);
            vv[k][0] = q.x; vv[k][1] = q.y; vv[k][2] = q.z; vv[k][3] = q.w;
        }
        const float4 fq = HALF ? buffer_load8h(fr, foff, coff) : BUF ? buffer_load16(fr, foff, coff) : *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(F2) + (size_t)c0 * chunk_bytes + foff);
        fv[0] = fq.x; fv[1] = fq.y; fv[2] = fq.z; fv[3] = fq.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // channels beyond C are zero-padded in both volumes: df = 0, gOut = 0, all updates are exact no-ops
            float wv = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) wv += vv[k][j] * wgt[k];
            const float df = wv - fv[j];
            const float gOut = gsc * (2.0f * df);                    // PowBackward0: grad * (2 * self)
            // corner order tnw,tne,tsw,tse,bnw,bne,bsw,bse ; signs from GridSampler.cpp
            gix -= vv[0][j] * ax[0] * bx[0] * gOut; giy -= vv[0][j] * ay[0] * bx[0] * gOut; giz -= vv[0][j] * ay[0] * bz[0] * gOut;
            gix += vv[1][j] * ax[1] * bx[1] * gOut; giy -= vv[1][j] * ay[1] * bx[1] * gOut; giz -= vv[1][j] * ay[1] * bz[1] * gOut;
            gix -= vv[2][j] * ax[2] * bx[2] * gOut; giy += vv[2][j] * ay[2] * bx[2] * gOut; giz -= vv[2][j] * ay[2] * bz[2] * gOut;
            gix += vv[3][j] * ax[3] * bx[3] * gOut; giy += vv[3][j] * ay[3] * bx[3] * gOut; giz -= vv[3][j] * ay[3] * bz[3] * gOut;
            gix -= vv[4][j] * ax[4] * bx[4] * gOut; giy -= vv[4][j] * ay[4] * bx[4] * gOut; giz += vv[4][j] * ay[4] * bz[4] * gOut;
            gix += vv[5][j] * ax[5] * bx[5] * gOut; giy -= vv[5][j] * ay[5] * bx[5] * gOut; giz += vv[5][j] * ay[5] * bz[5] * gOut;
            gix -= vv[6][j] * ax[6] * bx[6] * gOut; giy += vv[6][j] * ay[6] * bx[6] * gOut; giz += vv[6][j] * ay[6] * bz[6] * gOut;
            gix += vv[7][j] * ax[7] * bx[7] * gOut; giy += vv[7][j] * ay[7] * bx[7] * gOut; giz += vv[7][j] * ay[7] * bz[7] * gOut;
        }
    }
    // grad wrt the normalised grid (x,y,z) = (size/2)*gi ; flip ; / scale -> grad wrt U (H,W,D)
    float g[3];
    g[0] = fdiv(((float)h / 2.0f) * giz, sc0);
    g[1] = fdiv(((float)w / 2.0f) * giy, sc1);
    g[2] = fdiv(((float)d / 2.0f) * gix, sc2);
    // Diffusion regulariser: the 18 neighbour values are fetched in one batch from clamped (always valid) addresses
    // and the one-sided terms are selected afterwards -- one memory round trip instead of 18 dependent ones.
    // (32-bit element offsets from the uniform channel base: the launcher guarantees 16 * (V + 1) < 2^32)
    const unsigned sH = (unsigned)(w * d);
    const unsigned pxp = x < d - 1 ? p + 1 : p, pxm = x > 0 ? p - 1 : p, pzp = z < h - 1 ? p + sH : p, pzm = z > 0 ? p - sH : p,
                   pyp = y < w - 1 ? p + (unsigned)d : p, pym = y > 0 ? p - (unsigned)d : p;
    float nb[3][6];
    // (buffer loads: 6 per-lane byte offsets + the channel offset in a scalar register instead of 18 64-bit vector address additions)
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)(12u * (unsigned)V), 0x00020000);
    const unsigned nboff[6] = {4u * pxp, 4u * pxm, 4u * pzp, 4u * pzm, 4u * pyp, 4u * pym};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 6; ++k) nb[a][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ur, (int)nboff[k], (int)(4u * (unsigned)V) * a, 0));
    const float uc3[3] = {uH, uW, uD};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float uc = uc3[a];
        float acc = g[a], t;
        t = acc + -(cD * (2.0f * (nb[a][0] - uc))); acc = x < d - 1 ? t : acc;
        t = acc +  (cD * (2.0f * (uc - nb[a][1]))); acc = x > 0 ? t : acc;
        t = acc + -(cH * (2.0f * (nb[a][2] - uc))); acc = z < h - 1 ? t : acc;
        t = acc +  (cH * (2.0f * (uc - nb[a][3]))); acc = z > 0 ? t : acc;
        t = acc + -(cW * (2.0f * (nb[a][4] - uc))); acc = y < w - 1 ? t : acc;
        t = acc +  (cW * (2.0f * (uc - nb[a][5]))); acc = y > 0 ? t : acc;
        (gU + (size_t)a * V)[p] = acc;
    }
    if (census && threadIdx.x == 0) census[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
}


int launch_to_chunked(const float* in, int C, size_t V, float* out, bool half, hipStream_t s) {
    const int CP = (C + 3) / 4 * 4;
    if ((size_t)(CP / 4) * (V + 1) * 16 >= ((size_t)1 << 31)) return fail(CVX_ERR_UNSUPPORTED, "adam_run: control grid too large (%zu voxels x %d channels)", V, C);
    if (half) hipLaunchKernelGGL(k_to_chunked_h, dim3((unsigned)cdiv64((int64_t)(V + 1), 256), CP / 4), dim3(256), 0, s, in, C, V, reinterpret_cast<uint2*>(out));
    else hipLaunchKernelGGL(k_to_chunked, dim3((unsigned)cdiv64((int64_t)(V + 1), 256), CP / 4), dim3(256), 0, s, in, C, V, out);
    return check_last("to_chunked");
}

int launch_warp_grad(const float* Fcl, const float* Mcl, int C, int h, int w, int d, const float* U, const float* bh,
                     const float* bw, const float* bd, float gsc, float cH, float cW, float cD, float* gU, bool half, hipStream_t s) {
    const int CP = (C + 3) / 4 * 4;
    const dim3 gv((unsigned)((cdiv(d, 16) * cdiv(w, 4) * cdiv(h, 4) + 7) / 8 * 8));     // multiple of the 8 XCDs
    unsigned long long* census = reinterpret_cast<unsigned long long*>(options().census_ptr);       // debugging aid: slots [8192, ..)
    if (census) census += 8 * 1024;
    const FastDiv dx = fastdiv_make(cdiv(d, 16)), dy = fastdiv_make(cdiv(w, 4));
    const float sc0 = (float)((h - 1) / 2.0), sc1 = (float)((w - 1) / 2.0), sc2 = (float)((d - 1) / 2.0);
    if (half) hipLaunchKernelGGL((k_warp_grad<true, true>), gv, dim3(256), 0, s, Fcl, Mcl, C, CP, h, w, d, U, bh, bw, bd, gsc, cH, cW, cD, gU, census, dx, dy, sc0, sc1, sc2);
    else if (options().warp_flat) hipLaunchKernelGGL(k_warp_grad<false>, gv, dim3(256), 0, s, Fcl, Mcl, C, CP, h, w, d, U, bh, bw, bd, gsc, cH, cW, cD, gU, census, dx, dy, sc0, sc1, sc2);
    else hipLaunchKernelGGL(k_warp_grad<true>, gv, dim3(256), 0, s, Fcl, Mcl, C, CP, h, w, d, U, bh, bw, bd, gsc, cH, cW, cD, gU, census, dx, dy, sc0, sc1, sc2);
    return check_last("warp_grad");
}

}  // namespace cvx
