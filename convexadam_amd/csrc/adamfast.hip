// adamfast.hip -- the throughput ("fast") arithmetic of the Adam instance optimisation (reference: convex_adam_MIND.py:163-179).
//
// adam_mode = "fast" keeps the mathematics of the loop and relaxes the EVALUATION ORDER where the reference's own order buys
// nothing: the default build is not bit-identical to the reference anyway (library expf / IEEE sqrt instead of the reference
// host's MKL calls), so the result is graded by end-point error against the reference's field.  What changes, per iteration:
//   * k_warp_grad_fast: per-voxel set-up (coordinates, floor, the eight corner weights) exactly as ATen's grid_sampler_3d, but per
//     channel an FMA chain for the warped value and eight corner accumulators A_k += df * v_k; the three gradient components are
//     combined from the A_k once per voxel: 17 instead of 113 operations per channel and voxel.
//   * k_box3_fast: the ADJOINT of the three chained 3^3 boxes (a symmetric operator) as separable sums -- per axis three chained
//     1-D stages t[i] = (t[i-1] + t[i]) + t[i+1] with zeros outside the volume after every stage (each avg_pool3d zero-pads its
//     own input), one multiplication by 1/19683 -- 18 additions per output instead of 78 + 3 divisions, no z-marching pipeline:
//     independent tiles, three barriers per workgroup; the Adam update runs in the epilogue.
//   * round 5: the regulariser gradient (autograd's arrival order) and the Adam update (sqrt / bc2_sqrt + eps) are ATen's again --
//     ~50 instructions per voxel in two memory-bound kernels; with them the mode follows the reference more closely on every
//     full-size capture at 20 / 40 iterations and on average at 80.
//   * the FORWARD boxes stay in ATen's order (boxtile.hip / boxmarch.hip): the diffusion regulariser differentiates U twice, so the rounding
//     pattern of U itself is what keeps the trajectory next to the reference's (measured on the CPU restatement: fast forward boxes
//     alone move the 80-iteration field 2.2e-3 voxels away, everything else together 1.3e-3 -- the exact mode's own distance).
// Every operation is a correctly rounded IEEE operation in a fixed order; oracle/cvx_oracle.c::orc_adam_run_fast restates it, and the
// GPU tests compare bit for bit.  Roofline (profiles/r04_v3_*): every kernel boundary empties the per-XCD L2s, so both kernels stream
// their operands from the Infinity Cache / HBM each iteration -- k_warp_grad_fast 125 MB in 22 us (5.7 TB/s, and at the same time at
// the vector L1's limit: 27 gathers of 16 bytes per voxel = 371 MB at 19.5 TB/s of L1 hits), k_box3_fast 76 MB in 16 us (4.8 TB/s; a
// device copy reaches 5.1): memory-bound, no MFMA (stencil + gather).
#include "cvx_common.h"

namespace cvx {

// ---- warp + data-term gradient + regulariser gradient -----------------------------------------------------------------------
// HALF: records of four half-precision values (8 bytes; warp.hip::k_to_chunked_h -- fp16 STORAGE of the pooled features, the reference's
// GPU default dtype, convex_adam_MIND.py:79) widened to float32 on load: half the gather traffic of a memory-bound kernel.
typedef _Float16 h16x4f __attribute__((ext_vector_type(4)));
template <bool HALF>
__device__ __forceinline__ float4 wf_load_rec(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uni_off) {
    if (HALF) {
        const h16x4f v = __builtin_bit_cast(h16x4f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)uni_off, 0));
        return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    }
    return buffer_load16(rsrc, lane_off, uni_off);
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_warp_grad_fast(const float* __restrict__ F2, const float* __restrict__ M2, int CP,
                                                        int h, int w, int d, const float* __restrict__ U,
                                                        const float* __restrict__ bh, const float* __restrict__ bw,
                                                        const float* __restrict__ bd, float gsc2, float cH, float cW, float cD,
                                                        float* __restrict__ gU, int octant) {
    const size_t V = (size_t)h * w * d;
    // 4 x 4 x 16 voxel tile per workgroup.  XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; XCD q = (qz, qy, qx) takes
    // the q-th OCTANT of the tile grid (2 x 2 x 2 split) and walks it x, y, z: the tiles that share a face in y or z are then
    // (ntx/2) resp. (ntx/2)(nty/2) positions apart instead of ntx resp. ntx nty, close enough in time to find the shared moving-feature
    // records in the XCD's 4 MB L2 (octant == 0: the slab order of k_warp_grad, warp.hip)
    const int ntx = (d + 15) / 16, nty = (w + 3) / 4, ntz = (h + 3) / 4;
    int tbx, tby, tbz;
    if (octant == 1) {
        const int q = (int)(blockIdx.x & 7), i = (int)(blockIdx.x >> 3);
        const int hx = (ntx + 1) >> 1, hy = (nty + 1) >> 1, hz = (ntz + 1) >> 1;
        if (i >= hx * hy * hz) return;
        tbx = (q & 1) * hx + i % hx; tby = ((q >> 1) & 1) * hy + (i / hx) % hy; tbz = (q >> 2) * hz + i / (hx * hy);
        if (tbx >= ntx || tby >= nty || tbz >= ntz) return;
    } else if (octant == 0) {
        const int per_xcd = (int)(gridDim.x >> 3);
        const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
        if (tile >= ntx * nty * ntz) return;
        tbx = tile % ntx; tby = (tile / ntx) % nty; tbz = tile / (ntx * nty);
    } else {
        // z-groups: x fastest, then G = `octant` z-adjacent tiles, then y: the tiles that share planes in z follow each other within ntx
        // launches (their records are still in the XCD's L2), the ones that share rows in y within G ntx
        const int per_xcd = (int)(gridDim.x >> 3), G = octant;
        const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
        const int ntzq = (ntz + G - 1) / G;
        if (tile >= ntx * nty * ntzq * G) return;
        const int zq = tile / (ntx * nty * G), r1 = tile - zq * (ntx * nty * G);
        tby = r1 / (G * ntx);
        const int r2 = r1 - tby * G * ntx, zi = r2 / ntx;
        tbx = r2 - zi * ntx; tbz = G * zq + zi;
        if (tbz >= ntz) return;
    }
    const int x = tbx * 16 + (threadIdx.x & 15), y = tby * 4 + ((threadIdx.x >> 4) & 3), z = tbz * 4 + (threadIdx.x >> 6);
    if (x >= d || y >= w || z >= h) return;
    const unsigned p = (unsigned)((z * w + y) * d + x);
    const float sc0 = (float)((h - 1) / 2.0), sc1 = (float)((w - 1) / 2.0), sc2 = (float)((d - 1) / 2.0);   // (:171)
    const float uH = U[p], uW = U[V + p], uD = U[2 * V + p];
    Tri t;
    tri_setup(t, bd[x] + fdiv(uD, sc2), bw[y] + fdiv(uW, sc1), bh[z] + fdiv(uH, sc0), h, w, d);
    const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float wx[2] = {(float)x1 - t.ix, t.ix - (float)x0}, wy[2] = {(float)y1 - t.iy, t.iy - (float)y0},
                wz[2] = {(float)z1 - t.iz, t.iz - (float)z0};
    const bool zin0 = (unsigned)z0 < (unsigned)h, zin1 = (unsigned)z1 < (unsigned)h, yin0 = (unsigned)y0 < (unsigned)w,
               yin1 = (unsigned)y1 < (unsigned)w, xin0 = (unsigned)x0 < (unsigned)d, xin1 = (unsigned)x1 < (unsigned)d;
    const int r00 = (z0 * w + y0) * d, r01 = (z0 * w + y1) * d, r10 = (z1 * w + y0) * d, r11 = (z1 * w + y1) * d;
    constexpr unsigned REC = HALF ? 8u : 16u;                     // bytes per record (four channels)
    const unsigned zero_rec = (unsigned)V * REC;                  // record V of every chunk is all zero (corners outside the volume)
    unsigned off[8];
    off[0] = (zin0 && yin0 && xin0) ? (unsigned)(r00 + x0) * REC : zero_rec; off[1] = (zin0 && yin0 && xin1) ? (unsigned)(r00 + x1) * REC : zero_rec;
    off[2] = (zin0 && yin1 && xin0) ? (unsigned)(r01 + x0) * REC : zero_rec; off[3] = (zin0 && yin1 && xin1) ? (unsigned)(r01 + x1) * REC : zero_rec;
    off[4] = (zin1 && yin0 && xin0) ? (unsigned)(r10 + x0) * REC : zero_rec; off[5] = (zin1 && yin0 && xin1) ? (unsigned)(r10 + x1) * REC : zero_rec;
    off[6] = (zin1 && yin1 && xin0) ? (unsigned)(r11 + x0) * REC : zero_rec; off[7] = (zin1 && yin1 && xin1) ? (unsigned)(r11 + x1) * REC : zero_rec;
    const float wgt[8] = {t.tnw, t.tne, t.tsw, t.tse, t.bnw, t.bne, t.bsw, t.bse};          // corner k: bit 0 = x1, bit 1 = y1, bit 2 = z1
    const unsigned foff = p * REC;
    const unsigned chunk_bytes = (unsigned)(V + 1) * REC;
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M2), 0, (int)(chunk_bytes * (unsigned)(CP / 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F2), 0, (int)(chunk_bytes * (unsigned)(CP / 4)), 0x00020000);
    float A[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned coff = 0;
    for (int c0 = 0; c0 < CP / 4; ++c0, coff += chunk_bytes) {
        float vv[8][4], fv[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = wf_load_rec<HALF>(mr, off[k], coff);
            vv[k][0] = q.x; vv[k][1] = q.y; vv[k][2] = q.z; vv[k][3] = q.w;
        }
        const float4 fq = wf_load_rec<HALF>(fr, foff, coff);
        fv[0] = fq.x; fv[1] = fq.y; fv[2] = fq.z; fv[3] = fq.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // channels beyond C are zero in both volumes: df = 0 and every update is an exact no-op
            float wv = vv[0][j] * wgt[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) wv = __builtin_fmaf(vv[k][j], wgt[k], wv);
            const float df = wv - fv[j];
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = __builtin_fmaf(df, vv[k][j], A[k]);
        }
    }
    // regulariser neighbours: one batch of loads from clamped (always valid) addresses
    const unsigned sH = (unsigned)(w * d);
    const unsigned pxp = x < d - 1 ? p + 1 : p, pxm = x > 0 ? p - 1 : p, pzp = z < h - 1 ? p + sH : p, pzm = z > 0 ? p - sH : p,
                   pyp = y < w - 1 ? p + (unsigned)d : p, pym = y > 0 ? p - (unsigned)d : p;
    float nb[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* Ua = U + (size_t)a * V;
        nb[a][0] = Ua[pxp]; nb[a][1] = Ua[pxm]; nb[a][2] = Ua[pzp]; nb[a][3] = Ua[pzm]; nb[a][4] = Ua[pyp]; nb[a][5] = Ua[pym];
    }
    // d warp / d ix = sum_k (+-) wy wz v_k (sign: + for the x1 corners), likewise iy, iz
    float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int kx = k & 1, ky = (k >> 1) & 1, kz = (k >> 2) & 1;
        const float cx = wy[ky] * wz[kz], cy = wx[kx] * wz[kz], cz = wx[kx] * wy[ky];
        gix = __builtin_fmaf(kx ? cx : -cx, A[k], gix);
        giy = __builtin_fmaf(ky ? cy : -cy, A[k], giy);
        giz = __builtin_fmaf(kz ? cz : -cz, A[k], giz);
    }
    gix = gix * gsc2; giy = giy * gsc2; giz = giz * gsc2;
    float g[3];
    g[0] = fdiv(((float)h / 2.0f) * giz, sc0);
    g[1] = fdiv(((float)w / 2.0f) * giy, sc1);
    g[2] = fdiv(((float)d / 2.0f) * gix, sc2);
    const float uc3[3] = {uH, uW, uD};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float uc = uc3[a];
        float acc = g[a], tt;
        // autograd's arrival order and ATen's expressions (as k_warp_grad, warp.hip): data, D[:-1], D[1:], H[:-1], H[1:], W[:-1], W[1:]
        tt = acc + -(cD * (2.0f * (nb[a][0] - uc))); acc = x < d - 1 ? tt : acc;
        tt = acc +  (cD * (2.0f * (uc - nb[a][1]))); acc = x > 0 ? tt : acc;
        tt = acc + -(cH * (2.0f * (nb[a][2] - uc))); acc = z < h - 1 ? tt : acc;
        tt = acc +  (cH * (2.0f * (uc - nb[a][3]))); acc = z > 0 ? tt : acc;
        tt = acc + -(cW * (2.0f * (nb[a][4] - uc))); acc = y < w - 1 ? tt : acc;
        tt = acc +  (cW * (2.0f * (uc - nb[a][5]))); acc = y > 0 ? tt : acc;
        (gU + (size_t)a * V)[p] = acc;
    }
}

int launch_warp_grad_fast(const float* Fcl, const float* Mcl, int C, int h, int w, int d, const float* U, const float* bh,
                          const float* bw, const float* bd, float gsc, float cH, float cW, float cD, float* gU, bool half, hipStream_t s) {
    const int CP = (C + 3) / 4 * 4;
    // 32-bit byte offsets into the chunked feature volumes (buffer descriptors): same limit as launch_warp_grad (warp.hip)
    if ((size_t)(CP / 4) * ((size_t)h * w * d + 1) * 16 >= ((size_t)1 << 31)) return fail(CVX_ERR_UNSUPPORTED, "warp_grad_fast: control grid too large (%zu voxels x %d channels)", (size_t)h * w * d, C);
    const int ntx = cdiv(d, 16), nty = cdiv(w, 4), ntz = cdiv(h, 4);
    const int oc = (int)options().warp_octant;
    const int octant = oc >= 2 && oc <= 64 ? oc : (oc == 1 && ntx >= 2 && nty >= 2 && ntz >= 2) ? 1 : 0;       // >= 2: z-groups of that many tiles
    const dim3 gv(octant == 1 ? (unsigned)(8 * ((ntx + 1) / 2) * ((nty + 1) / 2) * ((ntz + 1) / 2))
                  : octant >= 2 ? (unsigned)((ntx * nty * ((ntz + octant - 1) / octant) * octant + 7) / 8 * 8) : (unsigned)((ntx * nty * ntz + 7) / 8 * 8));     // multiple of the 8 XCDs
    if (half) hipLaunchKernelGGL(k_warp_grad_fast<true>, gv, dim3(256), 0, s, Fcl, Mcl, CP, h, w, d, U, bh, bw, bd, 2.0f * gsc, cH, cW, cD, gU, octant);
    else hipLaunchKernelGGL(k_warp_grad_fast<false>, gv, dim3(256), 0, s, Fcl, Mcl, CP, h, w, d, U, bh, bw, bd, 2.0f * gsc, cH, cW, cD, gU, octant);
    return check_last("warp_grad_fast");
}

// ---- three chained 1-D stages along one axis, in registers ------------------------------------------------------------------------
// a[j] holds the value at coordinate g0 + j (j = 0 .. N+5); afterwards a[3 .. N+2] hold the third stage at g0 + 3 .. g0 + N + 2.
// Stages 1 and 2 are zero outside [0, n): every avg_pool3d of the reference zero-pads its own input.
template <int N>
__device__ __forceinline__ void chain3(float (&a)[N + 6], int g0, int n) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float prev = a[s];
#pragma unroll
        for (int j = s + 1; j <= N + 4 - s; ++j) {
            const float cur = a[j];
            float t = (prev + cur) + a[j + 1];
            if (s < 2) { const int g = g0 + j; t = (g >= 0 && g < n) ? t : 0.0f; }
            prev = cur;
            a[j] = t;
        }
    }
}

// torch.optim.Adam's own update (cvx_common.h::adam_update: sqrt / bc2_sqrt + eps, two IEEE divisions) with the IEEE square root.
// (Round 4 used one division -- den = fma(sqrt(v), 1 / sqrt(bc2), eps); on the four full-size captures of the reference that form moved
// the 20- and 40-iteration fields 1.5-5 x further from the reference than ATen's, for ~10 instructions per element in a memory-bound
// epilogue: DESIGN.md section 11.)
typedef AdamConsts AdamFastConsts;
__device__ __forceinline__ void adam_update_fast(float g, float& P, float& m, float& v, const AdamFastConsts& ac) { adam_update(g, P, m, v, ac); }
static AdamFastConsts adam_fast_consts(double bc1, double bc2) {
    const double beta1 = 0.9, beta2 = 0.999;
    return AdamFastConsts{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)(-(1.0 / bc1)), nullptr};
}

// One workgroup = one channel x one TZ x TY x TX output tile (TX = 4 TXQ - 8); input region: 3 planes / rows of halo and one aligned
// quad of columns per side.  Phase Z: a thread owns one (row, column) of the region, loads its TZ + 6 planes from global memory
// (coalesced along x) and runs the three z stages in registers; phase Y: one thread per (plane, column), in place in LDS (a column
// is private to its thread); phase X: one thread per (plane, row, output quad) reads three aligned quads, runs the x stages, scales
// and either stores G or applies the Adam update to P, m, v (16-byte accesses when rows are 16-byte aligned; the P / m / v quads of
// a round are requested one round ahead -- those of the first round before phase Z -- so that their latency hides behind the boxes).
// Tile order: z fastest, and XCD q (workgroups are dealt round-robin to the 8 XCDs) takes the q-th contiguous slab of tiles: the
// halo planes that z-neighbours share are re-read from the XCD's own L2, not from the Infinity Cache.
template <int TZ, int TY, int TXQ, bool ADAM>
__global__ __launch_bounds__(256) void k_box3_fast(const float* __restrict__ in, float* __restrict__ out, int h, int w, int d,
                                                   float* __restrict__ P, float* __restrict__ m, float* __restrict__ v,
                                                   AdamFastConsts ac, float* __restrict__ gsave, int ntiles) {
    constexpr int NT = 256, PX = 4 * TXQ, TX = PX - 8, RY = TY + 6;
    __shared__ __attribute__((aligned(16))) float S[TZ * RY * PX];
    const int ntx = (d + TX - 1) / TX, nty = (w + TY - 1) / TY, ntz = (h + TZ - 1) / TZ;
    const int per_xcd = (int)(gridDim.x >> 3);
    int b = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (b >= ntiles) return;
    const int tz = b % ntz; b /= ntz;
    const int tx = b % ntx; b /= ntx;
    const int ty = b % nty; const int c = b / nty;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const size_t V = (size_t)h * w * d;
    const float* ic = in + (size_t)c * V;
    const bool vec = (d & 3) == 0;
    float* oc = out ? out + (size_t)c * V : nullptr;
    float* Pc = ADAM ? P + (size_t)c * V : nullptr;
    float* mc = ADAM ? m + (size_t)c * V : nullptr;
    float* vc = ADAM ? v + (size_t)c * V : nullptr;
    float* gs = gsave ? gsave + (size_t)c * V : nullptr;
    constexpr int NQ = TXQ - 2, NI = TZ * TY * NQ, NR = (NI + NT - 1) / NT;
    // item `it` of phase X: output quad q of row r of plane j
    auto item = [&](int it, int& j, int& r, int& q, size_t& i0, bool& live, bool& full) {
        q = it % NQ + 1; r = (it / NQ) % TY; j = it / (NQ * TY);
        const int gz = z0 + j, gy = y0 + r, gx = x0 - 4 + 4 * q;
        live = it < NI && gz < h && gy < w && gx < d;
        full = live && vec && gx + 3 < d;
        i0 = live ? ((size_t)gz * w + gy) * d + gx : 0;
    };
    float4 Pn = make_float4(0.f, 0.f, 0.f, 0.f), mn = Pn, vn = Pn;
    if (ADAM) {
        int j, r, q; size_t i0; bool live, full;
        item((int)threadIdx.x, j, r, q, i0, live, full);
        if (full) { Pn = *reinterpret_cast<const float4*>(Pc + i0); mn = *reinterpret_cast<const float4*>(mc + i0); vn = *reinterpret_cast<const float4*>(vc + i0); }
    }
    // ---- phase Z
    for (int col = threadIdx.x; col < RY * PX; col += NT) {
        const int r = col / PX, cx = col % PX;
        const int gy = y0 - 3 + r, gx = x0 - 4 + cx;
        float a[TZ + 6];
        const bool colin = gy >= 0 && gy < w && gx >= 0 && gx < d;
        const float* src = ic + (size_t)(colin ? gy : 0) * d + (colin ? gx : 0);
#pragma unroll
        for (int j = 0; j < TZ + 6; ++j) {
            const int gz = z0 - 3 + j;
            a[j] = (colin && gz >= 0 && gz < h) ? src[(size_t)gz * w * d] : 0.0f;
        }
        chain3<TZ>(a, z0 - 3, h);
#pragma unroll
        for (int j = 0; j < TZ; ++j) S[(j * RY + r) * PX + cx] = a[j + 3];
    }
    cvx_barrier();
    // ---- phase Y (in place: rows 0 .. TY-1 of the column receive the outputs y0 .. y0+TY-1)
    for (int col = threadIdx.x; col < TZ * PX; col += NT) {
        const int j = col / PX, cx = col % PX;
        float a[TY + 6];
        float* colp = S + (size_t)j * RY * PX + cx;
#pragma unroll
        for (int r = 0; r < TY + 6; ++r) a[r] = colp[r * PX];
        chain3<TY>(a, y0 - 3, w);
#pragma unroll
        for (int r = 0; r < TY; ++r) colp[r * PX] = a[r + 3];
    }
    cvx_barrier();
    // ---- phase X + epilogue
    const float rs = (float)(1.0 / 19683.0);
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {
        int j, r, q; size_t i0; bool live, full;
        item((int)threadIdx.x + rd * NT, j, r, q, i0, live, full);
        const float4 Pq = Pn, mq = mn, vq = vn;
        if (ADAM && rd + 1 < NR) {
            int j2, r2, q2; size_t i2; bool live2, full2;
            item((int)threadIdx.x + (rd + 1) * NT, j2, r2, q2, i2, live2, full2);
            if (full2) { Pn = *reinterpret_cast<const float4*>(Pc + i2); mn = *reinterpret_cast<const float4*>(mc + i2); vn = *reinterpret_cast<const float4*>(vc + i2); }
        }
        if (!live) continue;
        const int gx = x0 - 4 + 4 * q;                                                 // first of the four output columns
        const float* row = S + ((size_t)j * RY + r) * PX + 4 * (q - 1);
        const f32x4 l0 = lds_load4(row), l1 = lds_load4(row + 4), l2 = lds_load4(row + 8);
        // a[jj] at column gx - 3 + jj: the quad before starts at gx - 4
        float a[10] = {l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z};
        chain3<4>(a, gx - 3, d);
        float g4[4] = {a[3] * rs, a[4] * rs, a[5] * rs, a[6] * rs};
        if (!ADAM) {
            if (full) *reinterpret_cast<float4*>(oc + i0) = make_float4(g4[0], g4[1], g4[2], g4[3]);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gx + e < d) oc[i0 + e] = g4[e];
        } else if (full) {
            float Pv[4] = {Pq.x, Pq.y, Pq.z, Pq.w}, mv[4] = {mq.x, mq.y, mq.z, mq.w}, vv[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) adam_update_fast(g4[e], Pv[e], mv[e], vv[e], ac);
            *reinterpret_cast<float4*>(Pc + i0) = make_float4(Pv[0], Pv[1], Pv[2], Pv[3]);
            *reinterpret_cast<float4*>(mc + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(vc + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            if (gs) *reinterpret_cast<float4*>(gs + i0) = make_float4(g4[0], g4[1], g4[2], g4[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gx + e < d) {
                    float Pv = Pc[i0 + e], mv = mc[i0 + e], vv = vc[i0 + e];
                    adam_update_fast(g4[e], Pv, mv, vv, ac);
                    Pc[i0 + e] = Pv; mc[i0 + e] = mv; vc[i0 + e] = vv;
                    if (gs) gs[i0 + e] = g4[e];
                }
        }
    }
}

// ---- separable restatement of a CHAIN of zero-padded boxes (kovesi_spline of the sweep; oracle: orc_fast_boxchain) --------------------
// Per axis (H, W, D) the boxes of the chain one after the other as 1-D sums out[i] = ((in[i-r] + in[i-r+1]) + ...) + in[i+r], zeros
// outside the line; `reverse` = the adjoint's order (clipped boxes of different sizes do not commute at the borders); one
// multiplication by 1 / prod k^3 at the end of the last pass.  One launch per axis: a workgroup stages 64 lines in LDS (two buffers),
// every thread evaluates single outputs of a stage, one barrier per box.  Lines along H / W are 64 adjacent x columns (coalesced),
// lines along D are 64 consecutive rows.  A pass may run in place (a workgroup reads its lines completely before it writes them).
struct BoxChainArg { int n; int r[4]; float scale; };

// one box of half-width R over the NL staged lines: a thread takes FOUR consecutive positions of a line and reads the 4 + 2 R taps they
// share once; every output is still its own left-to-right sum ((t[i-R] + t[i-R+1]) + ...) + t[i+R] with zeros outside the line
template <bool STRIDED, int NL, int R>
__device__ __forceinline__ void bc_stage(const float* __restrict__ b0, float* __restrict__ b1, int len, int tid) {
    const int nq = (len + 3) >> 2;                                  // quads per line
    constexpr int st = STRIDED ? NL : 1;
    for (int item = tid; item < nq * NL; item += 256) {
        // strided: the NL lines are the fast index (adjacent lanes = adjacent lines: conflict-free); contiguous: quads of a row are adjacent
        const int line = STRIDED ? item % NL : item / nq, i0 = 4 * (STRIDED ? item / NL : item % nq);
        const float* p = b0 + (STRIDED ? (size_t)i0 * NL + line : (size_t)line * len + i0);
        float v[4 + 2 * R];
#pragma unroll
        for (int u = 0; u < 4 + 2 * R; ++u) {
            const int ii = i0 - R + u;
            v[u] = (ii >= 0 && ii < len) ? p[(u - R) * st] : 0.0f;
        }
        float* o = b1 + (STRIDED ? (size_t)i0 * NL + line : (size_t)line * len + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sacc = v[j];
#pragma unroll
            for (int u = 1; u <= 2 * R; ++u) sacc += v[j + u];
            if (i0 + j < len) o[j * st] = sacc;
        }
    }
}

// NL = lines per workgroup (64 / 32 / 16 by line length: two buffers of len * NL floats must leave several workgroups per CU -- with 64
// lines of 224 voxels a workgroup took 115 KB and a full-resolution pass ran at 0.3 TB/s)
template <bool STRIDED, int NL>
__global__ __launch_bounds__(256) void k_boxchain_pass(const float* __restrict__ in, float* __restrict__ out, int len, int ninner, size_t line_stride,
                                                       size_t A, size_t B, int n_io, size_t nrows, BoxChainArg ch) {
    extern __shared__ float bc_lds[];                     // two buffers of len * NL floats
    float* b0 = bc_lds;
    float* b1 = bc_lds + (size_t)len * NL;
    const int tid = threadIdx.x, n = len * NL;
    size_t base = 0;
    int nact = NL;                                        // lines of this tile inside the volume
    // contiguous variant: element e = row * len + i; (row, i) of e = tid advance by 256 without a division per element
    const int i_first = STRIDED ? 0 : tid % len, r_first = STRIDED ? 0 : tid / len, di = STRIDED ? 0 : 256 % len, dr = STRIDED ? 0 : 256 / len;
    if (STRIDED) {
        const int x0 = (int)blockIdx.x * NL, o = (int)blockIdx.y;
        base = (size_t)(o / n_io) * A + (size_t)(o % n_io) * B + x0;
        nact = min(NL, ninner - x0);
        for (int e = tid; e < n; e += 256) {
            const int i = e / NL, t = e % NL;
            b0[e] = t < nact ? in[base + t + (size_t)i * line_stride] : 0.0f;
        }
    } else {
        const size_t row0 = (size_t)blockIdx.x * NL;
        base = row0 * (size_t)len;
        nact = (int)min((size_t)NL, nrows - row0);
        int row = r_first, i = i_first;
        for (int e = tid; e < n; e += 256) {
            b0[e] = row < nact ? in[base + e] : 0.0f;   // [row][i], rows back to back
            i += di; row += dr;
            if (i >= len) { i -= len; ++row; }
        }
    }
    cvx_barrier();
    for (int q = 0; q < ch.n; ++q) {
        const int r = ch.r[q];
        if (r == 1) bc_stage<STRIDED, NL, 1>(b0, b1, len, tid);
        else if (r == 2) bc_stage<STRIDED, NL, 2>(b0, b1, len, tid);
        else if (r == 3) bc_stage<STRIDED, NL, 3>(b0, b1, len, tid);
        else if (r == 4) bc_stage<STRIDED, NL, 4>(b0, b1, len, tid);
        else bc_stage<STRIDED, NL, 0>(b0, b1, len, tid);
        cvx_barrier();
        float* t = b0; b0 = b1; b1 = t;
    }
    if (STRIDED) {
        for (int e = tid; e < n; e += 256) {
            const int i = e / NL, t = e % NL;
            if (t < nact) out[base + t + (size_t)i * line_stride] = b0[e] * ch.scale;
        }
    } else {
        int row = r_first, i = i_first;
        for (int e = tid; e < n; e += 256) {
            if (row < nact) out[base + e] = b0[e] * ch.scale;
            i += di; row += dr;
            if (i >= len) { i -= len; ++row; }
        }
    }
}

bool boxchain_fast_supported(const cvx_smoother& sm, int h, int w, int d) {
    if (sm.kind != 0 || sm.n_boxes < 1 || sm.n_boxes > 4) return false;
    for (int i = 0; i < sm.n_boxes; ++i)
        if (sm.box_k[i] < 1 || !(sm.box_k[i] & 1) || sm.box_k[i] > 9) return false;
    const int lmax = h > w ? (h > d ? h : d) : (w > d ? w : d);
    return lmax <= 320;                                  // 16 lines of 320 voxels in two buffers: 41 KB
}

// out = chain(in) for [3][h][w][d] (in == out allowed); reverse = adjoint order of the boxes
int launch_boxchain_fast(const float* in, float* out, int h, int w, int d, const cvx_smoother& sm, bool reverse, hipStream_t s) {
    if (!boxchain_fast_supported(sm, h, w, d)) return fail(CVX_ERR_UNSUPPORTED, "fast box chain: odd box sizes <= 9, at most 4 boxes, lines of at most 320 voxels");
    BoxChainArg ch{};
    ch.n = sm.n_boxes;
    double prod = 1.0;
    for (int i = 0; i < sm.n_boxes; ++i) {
        const int k = sm.box_k[reverse ? sm.n_boxes - 1 - i : i];
        ch.r[i] = k / 2;
        prod *= (double)k * k * k;
    }
    const size_t V = (size_t)h * w * d;
    // lines per workgroup; measured at full resolution (lines of 160-224 voxels, rocprofv3): strided passes 102 / 71 / 89 us with 8 / 16 / 32 lines,
    // the contiguous pass 69 / 85 / 144 us
    auto nl_of = [](int len, bool strided) { return len <= 64 ? 64 : len <= 128 ? 32 : strided ? 16 : 8; };
    auto lds = [](int len, int nl) { return (size_t)len * nl * 2 * sizeof(float); };
    const size_t nrows = (size_t)3 * h * w;
    // one pass: strided (lines along H or W: NL adjacent x columns per workgroup) or contiguous (lines along D: NL consecutive rows)
    auto pass = [&](bool strided, const float* src, float* dst, int len, int gy, size_t line_stride, size_t A, size_t B, int n_io) {
        const int nl = nl_of(len, strided);
        const size_t bytes = lds(len, nl);
#define CVX_BC(S, N)                                                                                                                      \
        do {                                                                                                                              \
            static size_t granted = 0;                                                                                                    \
            ensure_dynamic_lds(&k_boxchain_pass<S, N>, bytes, granted);                                                                   \
            if (S) hipLaunchKernelGGL((k_boxchain_pass<S, N>), dim3((unsigned)cdiv(d, N), (unsigned)gy), dim3(256), bytes, s, src, dst, len, d, line_stride, A, B, n_io, (size_t)0, ch); \
            else hipLaunchKernelGGL((k_boxchain_pass<S, N>), dim3((unsigned)cdiv64((int64_t)nrows, N)), dim3(256), bytes, s, src, dst, len, 0, (size_t)1, (size_t)0, (size_t)0, 1, nrows, ch); \
        } while (0)
        if (strided) { if (nl == 64) CVX_BC(true, 64); else if (nl == 32) CVX_BC(true, 32); else if (nl == 8) CVX_BC(true, 8); else CVX_BC(true, 16); }
        else { if (nl == 64) CVX_BC(false, 64); else if (nl == 32) CVX_BC(false, 32); else if (nl == 8) CVX_BC(false, 8); else CVX_BC(false, 16); }
#undef CVX_BC
    };
    ch.scale = 1.0f;
    pass(true, in, out, h, 3 * w, (size_t)w * d, V, (size_t)d, w);           // along H: lines (c, y, x): base = c V + y d + x, elements w d apart
    pass(true, out, out, w, 3 * h, (size_t)d, (size_t)w * d, (size_t)0, 1);  // along W: lines (c, z, x): base = (c h + z) w d + x, elements d apart
    ch.scale = (float)(1.0 / prod);
    pass(false, out, out, d, 0, (size_t)1, (size_t)0, (size_t)0, 1);         // along D: rows back to back
    return check_last("boxchain_fast");
}

__global__ __launch_bounds__(256) void k_adam_update_fast(const float* __restrict__ G, float* __restrict__ P, float* __restrict__ m,
                                                          float* __restrict__ v, size_t n, AdamFastConsts ac) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float Pv = P[i], mv = m[i], vv = v[i];
    adam_update_fast(G[i], Pv, mv, vv, ac);
    P[i] = Pv; m[i] = mv; v[i] = vv;
}
int launch_adam_update_fast(const float* G, float* P, float* m, float* v, size_t n, double bc1, double bc2, hipStream_t s) {
    const AdamFastConsts ac = adam_fast_consts(bc1, bc2);
    hipLaunchKernelGGL(k_adam_update_fast, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, s, G, P, m, v, n, ac);
    return check_last("adam_update_fast");
}

template <int TZ, int TY, int TXQ>
static int launch_box3_fast_t(const float* in, float* out, int h, int w, int d, float* P, float* m, float* v, AdamFastConsts ac,
                              float* gsave, hipStream_t s) {
    constexpr int TX = 4 * TXQ - 8;
    const int ntiles = cdiv(d, TX) * cdiv(w, TY) * cdiv(h, TZ) * 3;
    const unsigned nb = (unsigned)((ntiles + 7) / 8 * 8);                        // multiple of the 8 XCDs
    if (P) hipLaunchKernelGGL((k_box3_fast<TZ, TY, TXQ, true>), dim3(nb), dim3(256), 0, s, in, out, h, w, d, P, m, v, ac, gsave, ntiles);
    else hipLaunchKernelGGL((k_box3_fast<TZ, TY, TXQ, false>), dim3(nb), dim3(256), 0, s, in, out, h, w, d, P, m, v, ac, gsave, ntiles);
    return check_last("box3_fast");
}

// out = fastbox(in) (P == nullptr) or the Adam update of P, m, v with G = fastbox(in) (gsave optionally receives G); 3 channels.
// Tile shapes <TZ, TY, TXQ> (option fbox_tile; all bit-identical): 1 = 8 x 10 x 24, 2 = 8 x 10 x 56, 3 = 16 x 10 x 24, 4 = 16 x 10 x 56,
// 5 = 8 x 8 x 32, 6 = 4 x 10 x 24; 0 = automatic = 5 (measured on the benchmark grid 80 x 96 x 112: 5.65 ms per pair against 5.76 - 6.20
// for the others -- the kernel moves 75 MB at ~4.3 TB/s whatever the tile, what differs is the tail of the last dispatch round).
int launch_box3_fast(const float* in, float* out, int h, int w, int d, float* P, float* m, float* v, double bc1, double bc2,
                     float* gsave, hipStream_t s) {
    const AdamFastConsts ac = adam_fast_consts(bc1, bc2);
    long long shape = options().fbox_tile;
    if (shape <= 0 || shape > 6) shape = 5;
    switch (shape) {
        case 1: return launch_box3_fast_t<8, 10, 8>(in, out, h, w, d, P, m, v, ac, gsave, s);
        case 2: return launch_box3_fast_t<8, 10, 16>(in, out, h, w, d, P, m, v, ac, gsave, s);
        case 3: return launch_box3_fast_t<16, 10, 8>(in, out, h, w, d, P, m, v, ac, gsave, s);
        case 4: return launch_box3_fast_t<16, 10, 16>(in, out, h, w, d, P, m, v, ac, gsave, s);
        case 5: return launch_box3_fast_t<8, 8, 10>(in, out, h, w, d, P, m, v, ac, gsave, s);
        default: return launch_box3_fast_t<4, 10, 8>(in, out, h, w, d, P, m, v, ac, gsave, s);
    }
}

}  // namespace cvx
