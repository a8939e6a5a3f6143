// cvx_common.h -- shared host/device helpers of libconvexadam_hip.so (gfx950 only).
//
// All kernels are compiled with -ffp-contract=off: a multiply-add is fused only where the code
// says fmaf()/__builtin_fmaf(), because the evaluation order (and the absence or presence of a
// fused rounding) is part of the contract with the reference's ATen CPU kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <mutex>

#include "convexadam_hip.h"

namespace cvx {

// ---- error plumbing ------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_last(const char* what);   // hipGetLastError -> CVX_ERR_LAUNCH

#define CVX_REQUIRE(cond, ...)                                      \
    do {                                                            \
        if (!(cond)) return ::cvx::fail(CVX_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// opt in to more than 64 KiB of dynamic LDS for `kernel` (idempotent; never leaves a sticky error)
template <typename KernelT>
static inline void ensure_dynamic_lds(KernelT kernel, size_t bytes, size_t& granted) {
    static std::mutex mu;                       // `granted` is a function-local static of the caller shared by all threads
    std::lock_guard<std::mutex> lock(mu);
    if (bytes > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        (void)hipGetLastError();
        granted = bytes;
    }
}

// bump allocator over the caller's workspace (256-byte aligned carves)
struct Carver {
    char* base;
    size_t size, used;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n), used(0) {}
    template <typename T>
    T* take(size_t count) {
        size_t off = align_up(used, 256);
        used = off + count * sizeof(T);
        return reinterpret_cast<T*>(base + off);
    }
    bool ok() const { return used <= size && (base != nullptr || used == 0); }
};
static inline size_t carve_size(size_t used, size_t bytes) { return align_up(used, 256) + bytes; }

// ---- run-time switches between kernel variants ---------------------------------------------------------------------------------
// One table per context (cvx_context_*); the default context is initialised once from the environment (CVX_<NAME IN CAPITALS>, e.g.
// CVX_MIND_TILED=1) and changeable through cvx_set_option(); every selectable path except mind_mean_threads is bit-identical, the
// switches exist for A/B timing and so that the test-suite can run every variant (tests/test_gpu_parity.py::test_kernel_variants_agree).
struct Options {
    long long mind_tiled;          // 1: tiled MIND stencil instead of the z-marching one
    long long mind_overlap;        // 1: the whole-pair pipeline runs the moving image's descriptor pass on a side stream beside the fixed one's (measured: no gain)
    long long mm_tx;               // 32 / 64: tile width of the marching MIND stencil (0 = automatic)
    long long mm_slots;            // workgroup budget of the marching MIND stencil (512)
    long long box_tiled;           // 1: tiled three-box kernels of the Adam loop instead of the z-marching ones
    long long no_prune;            // 1: streaming coupled-convex passes instead of branch and bound
    long long corr_unfused;        // 1: k_corr_raw + k_corr_box2 instead of the fused correlation kernel
    long long corr_fused_all;      // 1: the fused correlation kernel also for C >= 16 (default there: the round-1 kernels, which are faster)
    long long prune_stream_above;  // pruned pass falls back to a coalesced scan above this many 256-displacement chunks (-1 = K*v/2048)
    long long cf_census;           // 1: the fused correlation kernel records per-workgroup residency in its workspace
    long long cf_prio;             // fused correlation kernel: issue priorities (s_setprio, 0..3) as four base-4 digits -- first-round workgroup raw / box,
                                   //    second-round workgroup raw / box (two workgroups share a CU; 136 = 2,0,2,0: the raw stage above the boxes)
    long long warp_flat;           // 1: flat 64-bit gathers in k_warp_grad instead of buffer loads
    long long box_yt;              // rows per tile of the marching three-box kernels: 8 (default) or 4
    long long box_wg_target;       // workgroups the z-marching three-box kernels of the Adam loop aim for (z-chunk length follows); 0 = automatic
    long long box_xsplit;          // x tiles of the marching three-box kernels: -1 automatic (rows > 62 columns: tiles of <= 56), 0 off
    long long box_cpt;             // output columns per thread of the marching three-box kernels: 4, or 2 (x tiles / rows <= 62 columns only)
    long long box_uneven;          // marching three-box kernels with two workgroups per CU: length ratio (percent) of the z chunks given to the first and
                                   //    to the second dispatch round (boxmarch.hip, BMTable); <= 100: equal chunks
    long long box_adam_role;       // adjoint + Adam kernel: 1 = the Adam update runs as a fourth role (two extra wavefronts) instead of inside every wavefront's step
    long long box_dpp;             // marching three-box kernels, x-tile path: 1 = halo columns through DPP lane shifts instead of a second LDS read per row
    long long box_pk;              // marching three-box kernels: 1 = the two running sums of a column as one register pair (v_pk_add_f32 with a broadcast tap)
    long long box_prio;            // marching three-box kernels: 1 / 2 = the workgroups sharing a CU alternate their issue priority step by step
    long long label_pow_block;     // cvx_label_weights_host: elements per vectorised block of the reference host's torch.pow (32: AVX-512 build, the golden host; 16: AVX2)
    long long census_ptr;          // debugging aid: device address of a uint64 buffer; the Adam-loop kernels record per workgroup
                                   //    {start, first data, end} in 100 MHz ticks (s_memrealtime) + placement there (0 = off)
    long long mind_mean_threads;   // 0: exactly rounded global mean in MINDSSC (default); T > 0: torch's own float sum with T threads
                                   //    (reference-bits mode; NOT a bit-identical variant -- it changes the clamp bounds by ulps)
    long long edt_sequential;      // 1: squared distance transform with the sequential lower-envelope passes (one thread per line) instead of the tiled outward search
    long long warp_octant;         // adam_mode "fast" warp kernel, tile order inside an XCD's share: G >= 2 (default 4) = x fastest, then G z-adjacent tiles, then y (the tiles that share planes follow each other: FETCH_SIZE -13 %, 5.64 -> 5.59 ms per pair); 0 = plain slabs (x, y, z); 1 = one octant of the tile grid per XCD (measured: no gain)
    long long box_fwd_tile;        // forward three-box pass of the Adam loop: -1 = automatic (tiles of boxtile.hip where they fill the chip), 0 = z-marching pipeline (boxmarch.hip), kind * 1000 + segments = a tile kernel variant (boxtile.hip; bit-identical)
    long long box_bwd_tile;        // adjoint three-box pass (+ Adam update) of the exact Adam loop: as box_fwd_tile (-1 automatic, 0 = z-marching pipeline, kind * 1000 + segments)
    long long box_walk;            // 1 (default): single zero-padded box filters (sweep smoothers, final smoothing) through the z-walking kernel; 0 = one thread per output (bit-identical)
    long long corr_dual;           // 1: the whole-pair pipeline evaluates both directions' cost volumes in ONE launch of the fused correlation kernel (bit-identical; measured: the
                                   //    correlation stage 0.374 -> 0.360 ms for both directions, frac 0.183 -> 0.190, but the plain argmin of the first volume then reads it from HBM instead of the Infinity
                                   //    Cache -- 0.106 -> 0.157 ms for both -- so the pair is 0.02 ms SLOWER); 0 (default) = one launch per direction
    long long prune_refine;        // 1 (default): a candidate box too large for one thread is closed again with the cost of the displacement nearest to the smoothed field as the
                                   //    bound (ties at the minimum -- zero background -- otherwise keep whole windows; bit-identical); 0 = previous winner's cost only
    long long mind_records;        // 1 (default): the whole-pair pipeline's MIND pass writes the Adam-grid pooling directly as the loop's feature records (no planar copy, no
                                   //    k_to_chunked pass: -31 us per pair); 0 = planar pooled features + re-packing (bit-identical)
    long long resize_up2;          // 1 (default): exact factor-2 up-sampling of a 3-channel field through k_resize_up2 (2 x 2 x 2 outputs per thread from one 27-tap
                                   //    neighbourhood; bit-identical); 0 = one thread per output
    long long mind_blocked;        // 1 (default): the pipeline's MIND stencil writes its raw patch SSDs blocked by the tiles of the normalise + pool pass (contiguous reads there:
                                   //    half as many L1-miss requests for the same bytes); 0 = planar (bit-identical)
    long long corr_cert;           // 1 (default) / 2: the whole-pair pipeline evaluates its cost volumes in the certified-fast arithmetic and takes the argmin decisions with
                                   //    certification (certify.hip) where the geometry allows: SAME winners, SAME field bits as the exact kernels; 1 = the role kernel of
                                   //    corrfused.hip (fast arithmetic, unscaled), 2 = the staged kernel of corrcert.hip; 0 = exact volumes
    long long cc_debug;            // timing experiments on the certified-fast correlation kernel (bits: 1 no stores, 2 no staging after the first chunk, 4 no channel
                                   //    sums, 8 no y pass, 16 no x / z pass): WRONG results, never set outside tools/experiments
    long long ic_fused;            // inverse consistency: 1 = all iterations in ONE launch by 32 workgroups of one XCD with a barrier between the iterations
                                   //    (convex.hip::k_ic_persistent; 2 = its device-side fallback forced).  MEASURED SLOWER (289 vs 145 us per call: the field accesses
                                   //    must be agent-scope and are served behind the L2); 0 (default) = one launch per iteration (bit-identical)
    long long mind_single;         // 1: the whole-pair pipeline's descriptor in ONE stencil pass (normalisation with the unclamped variance and both poolings inside the
                                   //    marching kernel, k_mind_repair for the blocks where the variance clamp binds; no raw-SSD round trip: HBM traffic 7.8 x -> ~1.5 x of
                                   //    the algorithmic bytes; bit-identical).  MEASURED SLOWER (0.546 vs 0.499 ms for both images: both forms are bound by instruction
                                   //    issue, not by memory); 2 = every block through the repair kernel (test); 0 (default) = two passes
    long long ms_zlen;             // planes per z chunk of the single-pass kernel (0 = automatic)
    long long cf_map;              // fused correlation kernel, item -> workgroup order: 1 (default) = the two adjacent D-shift groups of a (dH, dW) pair in the two workgroup
                                   //    slots of ONE CU (blocks b and b + 256 share a CU: 251 of 251 in the census), so that their moving rows meet in that CU's L1
                                   //    (154 -> 152 us, three alternating rounds; bit-identical); 0 = large groups first
    long long cert_unfused;        // C >= 16: the round-1 pair of kernels in the certified-fast arithmetic (correlate.hip: FMA channel chain, separable boxes without divisions)
                                   //    1 = from K v C >= 1e9 on, 2 = whenever the geometry allows (tests), 0 (default) = never.  MEASURED: configs[3] 462 vs 503 us per
                                   //    direction -- the raw kernel takes 290 us with either arithmetic (not issue-bound), only the boxes gain (191 -> 145 us) -- which the certified
                                   //    passes' 0.08 ms per pair takes back
    long long fbox_tile;           // adam_mode "fast": tile shape of the separable adjoint-box + Adam kernel (adamfast.hip): 0 = automatic, 1 = 8x10x24,
                                   //    2 = 8x10x56, 3 = 16x10x24, 4 = 16x10x56, 5 = 8x8x32, 6 = 4x10x24 (bit-identical)
};
// All three read the context bound to the calling thread (cvx_context_bind / cvx_pair_params.ctx), else the process default context;
// launchers copy what they need into kernel arguments at enqueue time (api.hip).
Options& options();
const unsigned* adam_sqrt_table();       // device table of the current context (nullptr: IEEE sqrt)
// binds a context to the calling thread for the lifetime of the object (no-op for nullptr)
class ContextScope {
public:
    explicit ContextScope(const cvx_context* c);
    ~ContextScope();
    ContextScope(const ContextScope&) = delete;
    ContextScope& operator=(const ContextScope&) = delete;
private:
    void* prev_;
    bool active_;
};

// ---- workgroup barrier ---------------------------------------------------------------------------------------------------------
// Every barrier of the library goes through cvx_barrier().  The race-stress build (python -m convexadam_amd.csrc.build --jitter ->
// libconvexadam_hip_jitter.so, tools/race_stress.sh) defines CVX_RACE_JITTER: each wavefront then sleeps a pseudo-random time (0..15
// x 64 clocks, from the shader clock and its hardware wave id) on both sides of every barrier, which shuffles the arrival order of the
// specialised wavefronts of the marching pipelines; a missing barrier or a ring slot reused too early shows up as a result that is no
// longer bit-identical to the oracle.
#ifdef CVX_RACE_JITTER
__device__ __forceinline__ void cvx_jitter() {
    unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
    t ^= (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) * 2654435761u;       // HW_REG_HW_ID
    t ^= t >> 13; t *= 0x9E3779B1u; t ^= t >> 15;
    const int n = (int)(t & 15);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);                                          // 64 clocks each
}
__device__ __forceinline__ void cvx_barrier() { cvx_jitter(); __syncthreads(); cvx_jitter(); }
#else
__device__ __forceinline__ void cvx_barrier() { __syncthreads(); }
#endif

// ---- exact device math ---------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// IEEE correctly rounded division / sqrt (never the fast approximations)
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// NB: HIP's __fsqrt_rn() maps to the APPROXIMATE native sqrt; sqrtf() is the correctly rounded one
// (-fhip-fp32-correctly-rounded-divide-sqrt, on by default and passed explicitly by build.py).
__device__ __forceinline__ float fsqrt(float a) { return sqrtf(a); }

// exp(): SLEEF-style expf (Cody-Waite reduction, degree-6 FMA polynomial, two-step ldexp).
// Bit-identical to oracle/cvx_oracle.c::orc_expf; <= 1 ulp from the reference's MKL vsExp.
__device__ __forceinline__ float cvx_expf(float d) {
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    const float qf = rintf(d * R_LN2f);
    const int q = (int)qf;
    float s = __builtin_fmaf(qf, -L2Uf, d);
    s = __builtin_fmaf(qf, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    // scaling by 2^q: one v_ldexp_f32 (single rounding, denormals included) == the oracle's two exact-then-rounded
    // multiplications u * 2^(q>>1) * 2^(q-(q>>1)) for every argument (tools/expf_ldexp_check.hip, all 2^32 floats)
    u = __builtin_ldexpf(u, q);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = __int_as_float(0x7f800000);
    return u;
}

// Optional correction of cvx_expf to the exp of one particular reference BUILD (torch CPU -> MKL vsExp differs from cvx_expf by at
// most one ulp, position independent): two bits per argument x <= 0, keyed by the bit pattern of |x| minus `first`
// (0 = equal, 1 = one ulp above, 2 = one ulp below); installed by cvx_set_mind_exp_table, tbl == nullptr (default) = cvx_expf.
struct ExpTable { const unsigned char* tbl; unsigned first, count; };
ExpTable mind_exp_table();
// (applied in mind.hip::mind_normalise: the 12 table bytes of a voxel are requested together)

// 16-byte LDS/global vector access that the compiler must keep as ONE b128 instruction: hipcc otherwise
// re-splits an aligned float4 LDS load into two ds_read2_b32 (4-way bank conflicts for 16-byte lane strides).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ f32x4 lds_load4(const float* p) { return *(const volatile lds_f32x4*)p; }   // p must point into LDS
__device__ __forceinline__ void lds_store4(float* p, f32x4 v) { *(volatile lds_f32x4*)p = v; }

// 16-byte load through a buffer descriptor: address = descriptor base + per-lane byte offset (VGPR) + wave-uniform byte offset (SGPR);
// no 64-bit vector address arithmetic and half the address registers of the flat form
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uni_off) {
    const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)uni_off, 0));
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
__device__ __forceinline__ f32x2 lds_load2(const float* p) { return *(const volatile lds_f32x2*)p; }   // one ds_read_b64
__device__ __forceinline__ void lds_store2(float* p, f32x2 v) { *(volatile lds_f32x2*)p = v; }

// Exact x / D for the divisors of the box filters and poolings (D in {8, 27, 64, 125, 343}):
//   q = x*r ; e = fma(-D, q, x) ; q' = fma(e, r, q)      with r = RN(1/D)
// equals the correctly rounded IEEE quotient for EVERY float x (verified exhaustively over all 2^32 bit
// patterns, scratch/div27.c in the build log) except that -0.0 maps to +0.0, which no caller can produce:
// the dividends are sums that start from +0.0.  3 instructions instead of the ~15 of the IEEE sequence.
template <int D>
__device__ __forceinline__ float div_exact(float x) {
    static_assert(D == 8 || D == 27 || D == 64 || D == 125 || D == 343, "divisor not verified");
    constexpr float r = 1.0f / (float)D;
    const float q = x * r;
    const float e = __builtin_fmaf(-(float)D, q, x);
    return __builtin_fmaf(e, r, q);
}

// ATen outer-dimension sum order over `n` values held in registers (see oracle outer_sum_rows):
// plain sequential cascade (level step 16) or, for the last (ncols mod 32) columns, the 4-way
// interleaved `row_sum` order.  n is a compile-time constant at every call site that matters.
template <int N>
__device__ __forceinline__ float cascade_seq(const float (&v)[N]) {
    static_assert(N < 256, "two cascade levels");
    float a0 = 0.f, a1 = 0.f;
    int i = 0;
#pragma unroll
    for (; i + 16 <= N; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a0 += v[i + j];
        a1 += a0;
        a0 = 0.f;
    }
#pragma unroll
    for (; i < N; ++i) a0 += v[i];
    a0 += a1;      // + acc[2], acc[3] are zero: exact no-ops
    return a0;
}
template <int N>
__device__ __forceinline__ float outer_sum_ilp(const float (&v)[N]) {
    static_assert(N / 4 < 16, "single cascade level inside the interleaved sums");
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] += v[4 * i + k];
#pragma unroll
    for (int i = (N / 4) * 4; i < N; ++i) p[0] += v[i];
    p[0] += p[1];
    p[0] += p[2];
    p[0] += p[3];
    return p[0];
}

// order-preserving key for floats: packs (value, index) so that a 64-bit atomicMin returns the smallest value and, among equals, the
// smallest index.  A NaN sorts BEFORE everything (value part 0), like torch.argmin, whose scan stops at the first NaN (round 3).
__device__ __forceinline__ unsigned long long pack_min_key(float v, unsigned idx) {
    unsigned b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // total order for any sign
    if (v != v) b = 0u;
    return ((unsigned long long)b << 32) | idx;
}

// trilinear sampling set-up shared by grid_sample-like kernels (ATen GridSampler.cpp order)
struct Tri {
    float ix, iy, iz;
    int x0, y0, z0;
    float tnw, tne, tsw, tse, bnw, bne, bsw, bse;
};
__device__ __forceinline__ float unnormalize(float g, int S) { return ((g + 1.0f) * (float)S - 1.0f) * 0.5f; }  // /2 is exact
__device__ __forceinline__ void tri_setup(Tri& t, float gx, float gy, float gz, int h, int w, int d) {
    t.ix = unnormalize(gx, d);
    t.iy = unnormalize(gy, w);
    t.iz = unnormalize(gz, h);
    const float fx = floorf(t.ix), fy = floorf(t.iy), fz = floorf(t.iz);
    t.x0 = (int)fmaxf(fminf(fx, 1.0e9f), -1.0e9f);
    t.y0 = (int)fmaxf(fminf(fy, 1.0e9f), -1.0e9f);
    t.z0 = (int)fmaxf(fminf(fz, 1.0e9f), -1.0e9f);
    const float x1 = (float)(t.x0 + 1), y1 = (float)(t.y0 + 1), z1 = (float)(t.z0 + 1);
    const float x0f = (float)t.x0, y0f = (float)t.y0, z0f = (float)t.z0;
    t.tnw = (x1 - t.ix) * (y1 - t.iy) * (z1 - t.iz);
    t.tne = (t.ix - x0f) * (y1 - t.iy) * (z1 - t.iz);
    t.tsw = (x1 - t.ix) * (t.iy - y0f) * (z1 - t.iz);
    t.tse = (t.ix - x0f) * (t.iy - y0f) * (z1 - t.iz);
    t.bnw = (x1 - t.ix) * (y1 - t.iy) * (t.iz - z0f);
    t.bne = (t.ix - x0f) * (y1 - t.iy) * (t.iz - z0f);
    t.bsw = (x1 - t.ix) * (t.iy - y0f) * (t.iz - z0f);
    t.bse = (t.ix - x0f) * (t.iy - y0f) * (t.iz - z0f);
}
__device__ __forceinline__ bool inb3(int z, int y, int x, int h, int w, int d) {
    return z >= 0 && z < h && y >= 0 && y < w && x >= 0 && x < d;
}
// ATen adds the products of the in-range corners only, in corner order.  Branch-free form: all 8 corners are loaded
// at once from clamped addresses (one memory round trip instead of up to 8 dependent ones) and an out-of-range corner
// leaves the accumulator untouched through a select -- bit-identical to skipping the addition.
__device__ __forceinline__ float tri_sample(const Tri& t, const float* __restrict__ vol, int h, int w, int d) {
    const int x0 = t.x0, y0 = t.y0, z0 = t.z0, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const bool zi0 = (unsigned)z0 < (unsigned)h, zi1 = (unsigned)z1 < (unsigned)h, yi0 = (unsigned)y0 < (unsigned)w,
               yi1 = (unsigned)y1 < (unsigned)w, xi0 = (unsigned)x0 < (unsigned)d, xi1 = (unsigned)x1 < (unsigned)d;
    const int zc0 = clampi(z0, 0, h - 1), zc1 = clampi(z1, 0, h - 1), yc0 = clampi(y0, 0, w - 1), yc1 = clampi(y1, 0, w - 1),
              xc0 = clampi(x0, 0, d - 1), xc1 = clampi(x1, 0, d - 1);
    const size_t r00 = ((size_t)zc0 * w + yc0) * d, r01 = ((size_t)zc0 * w + yc1) * d, r10 = ((size_t)zc1 * w + yc0) * d,
                 r11 = ((size_t)zc1 * w + yc1) * d;
    const float v0 = vol[r00 + xc0], v1 = vol[r00 + xc1], v2 = vol[r01 + xc0], v3 = vol[r01 + xc1], v4 = vol[r10 + xc0],
                v5 = vol[r10 + xc1], v6 = vol[r11 + xc0], v7 = vol[r11 + xc1];
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

// Division of a WAVE-UNIFORM index by a launch constant on the scalar unit: q = mulhi(n, ceil(2^32 / d)) is exact whenever n * d < 2^32
// (error term n (m d - 2^32) / (d 2^32) < 1 / d).  The compiler's own expansion of `tile / ntx` runs ~25 vector instructions per division
// in every lane (the scalar unit has no divide) -- three of them opened k_warp_grad.
struct FastDiv { unsigned d, m; };
static inline FastDiv fastdiv_make(int d) { FastDiv f; f.d = (unsigned)d; f.m = d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + (unsigned)d - 1) / (unsigned)d); return f; }
__device__ __forceinline__ int fastdiv(int n, FastDiv f) { return f.d <= 1 ? n : (int)__umulhi((unsigned)n, f.m); }

// pipeline.hip: event mark behind a kernel of the Adam loop (no-op unless cvx_set_profiling(3))
void profile_mark_kernel(const char* name, hipStream_t s);

// ---- internal (non-ABI) launchers shared between translation units -------------------------------
int launch_box_zero(const float* in, float* out, int C, int H, int W, int D, int k, bool backward, hipStream_t s);
int launch_smoother(const float* in, float* out, float* tmp, int C, int H, int W, int D, const cvx_smoother& sm, bool backward,
                    hipStream_t s);
// (ssd: float32 cost volume, or half precision when f16 -- fp16 storage)
int launch_argmin_keys(const void* ssd, bool f16, int K, size_t v, unsigned long long* keys, bool arm, hipStream_t s);   // arm = false: the caller set the keys to all ones
// the two alternating list lengths of the pruned passes inside a coupled-convex workspace (same carve-up as coupled_core)
int* coupled_ws_counts(void* workspace, size_t workspace_bytes, int h, int w, int d, int disp_hw);
int launch_argmin(const void* ssd, bool f16, const float* mesh, const float* u, float coef, bool coupled, int K, size_t v,
                  unsigned long long* keys, int64_t* argmin_out, hipStream_t s);
// out = interp(in * pre_mul) / post_div   (pre_mul, post_div = 1 -> plain F.interpolate)
int launch_resize(const float* in, int C, int h, int w, int d, float* out, int H, int W, int D, float pre_mul,
                  float post_div, hipStream_t s);
int launch_resize2(const float* in, int C, int h, int w, int d, int H, int W, int D, float* scratch, float* out, int h2, int w2,
                   int d2, float post_div, hipStream_t s);
// Adam step constants of one iteration (torch.optim.Adam, lr = 1, eps = 1e-8) and the in-place update of one element
// sqrt_tbl: optional restatement of the reference build's sqrt (torch CPU -> MKL vsSqrt = the correctly rounded root, or one ulp
// beside it, as a function of (exponent parity, mantissa)): two bits per class, 0 = IEEE root, 1 = one ulp above, 2 = one ulp below;
// entries 0 .. 2^24-1 normal inputs, key = exponent parity << 23 | mantissa; entries 2^24 .. 2^24+2^23-1 denormal inputs, key =
// mantissa; 16 entries per 32-bit word, low bits first (cvx_set_adam_sqrt_table).  nullptr (default): IEEE sqrt.
struct AdamConsts { float w1, b2, omb2, bc2s, neg_step; const unsigned* sqrt_tbl; };
__device__ __forceinline__ float adam_sqrt(float x, const unsigned* __restrict__ tbl) {
    float r = fsqrt(x);
    if (tbl) {
        const unsigned b = __float_as_uint(x), e = b >> 23, mant = b & 0x7fffffu;
        if (b != 0 && e < 255) {
            const unsigned key = e ? (((e & 1u) << 23) | mant) : ((1u << 24) | mant);
            const unsigned code = (tbl[key >> 4] >> ((key & 15u) * 2u)) & 3u;
            if (code) r = __uint_as_float(__float_as_uint(r) + (code == 1u ? 1u : 0xffffffffu));
        }
    }
    return r;
}
// torch.argmin's comparison inside a sequential scan: a candidate replaces the running best when it is smaller, or when it is the first NaN
// (bitwise operators: no short-circuit branches inside the unrolled streaming loops)
__device__ __forceinline__ bool argmin_better(float cost, float best) { return (cost < best) | ((cost != cost) & (best == best)); }
__device__ __forceinline__ void adam_update(float g, float& P, float& m, float& v, const AdamConsts& ac) {
    const float mm = __builtin_fmaf(ac.w1, g - m, m);            // exp_avg.lerp_(grad, 1-beta1)
    float vv = v * ac.b2;                                         // exp_avg_sq.mul_(beta2)
    vv = __builtin_fmaf(ac.omb2 * g, g, vv);                      // .addcmul_(grad, grad, value=1-beta2)
    const float den = fdiv(adam_sqrt(vv, ac.sqrt_tbl), ac.bc2s) + 1e-8f;   // (sqrt / bias_correction2_sqrt).add_(eps)
    P = P + fdiv(ac.neg_step * mm, den);                          // addcdiv_(exp_avg, denom, value=-step_size)
    m = mm;
    v = vv;
}
// adam.hip: the Adam loop behind cvx_adam_run_f32 / cvx_adam_run_smoother_f32; keep_state = false lets the whole-pair pipeline
// drop the final (unobserved) gradient + update
int adam_run_impl(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m, float* v, float lambda_weight,
                  int niter, int step0, float cost_scale, const float* base_h, const float* base_w, const float* base_d, float* U,
                  float* grad_out, const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm,
                  bool keep_state, bool f16_features, int fast, void* workspace, size_t workspace_bytes, void* stream,
                  bool features_are_records = false);   // fast: 0 exact, 1 fast, 2 fast_all; features_are_records: F2 / M2 already hold the chunked records
// convex.hip: coupled convex regularisation behind cvx_coupled_convex_f32 (argmin_is_exact: see there)
int coupled_convex_impl(const void* ssd, bool f16, const int64_t* argmin, const float* mesh, int h, int w, int d, int disp_hw, float* out,
                        bool argmin_is_exact, void* workspace, size_t workspace_bytes, void* stream);
int coupled_convex_dual_impl(const void* ssdA, const int64_t* argminA, float* outA, void* wsA, const void* ssdB, bool f16, const int64_t* argminB,
                             float* outB, void* wsB, const float* mesh, int h, int w, int d, int disp_hw, size_t workspace_bytes,
                             void* stream, bool counts_zeroed = false);   // counts_zeroed: the caller cleared coupled_ws_counts of both workspaces
// mind.hip: MIND-SSC delivered only through its stride poolings (pipeline path, no full-resolution descriptor)
bool mind_pooled_supported(int H, int W, int D, int g1, int g2);
int launch_mind_pooled(const float* img, int H, int W, int D, int radius, int dilation, int g1, float* out1, int g2, float* out2,
                       float* raw, void* workspace, size_t workspace_bytes, hipStream_t s, int records = 0);   // records 1 / 2: out2 = float32 / half feature records
bool mind_pooled_records_supported(int H, int W, int D, int g1, int g2);
size_t mind_pooled_raw_floats(int H, int W, int D, int g1, int g2);      // floats of launch_mind_pooled's `raw` scratch (>= 12 H W D: blocked tiles overhang)
// corrbox.hip: the two box filters of the SSD volume (z-marching pipeline); raw [K][h][w][px] -> ssd [K][h][w][d]
bool corr_box2_supported(int h, int w, int d, int px);
int launch_corr_box2(const float* raw, int K, int h, int w, int d, int px, float* ssd, hipStream_t s, bool fast = false);   // fast: separable running sums, no divisions (certified-fast arithmetic)
// corrfused.hip: raw SSD + both boxes in one kernel (C < 16, planes of at most 320 quads); else the unfused path above
bool corr_fused_supported(int C, int h, int w, int d, int hw);
void corr_fused_set_prep_hook(void (*hook)(hipStream_t));         // profiling: called between the feature copies and the fused kernel (per thread; nullptr = off)
int corr_fused_items(int C, int h, int w, int d, int hw);
bool corr_fused_tiled(int C, int h, int w, int d, int hw);          // planes cut into y tiles (halo rows recomputed)
void corr_call_prep_hook(hipStream_t s);                            // the hook of corr_fused_set_prep_hook, for the other correlation paths
bool corr_certfast_pays(int C, int h, int w, int d, int hw);        // (correlate.hip) the whole-pair pipeline's rule for taking the certified path            // work items of one launch of the fused kernel (0: unsupported geometry)
size_t corr_fused_workspace_bytes(int C, int h, int w, int d, int hw);
int launch_corr_fused(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box, int fast, int f16,
                      void* ssd, void* workspace, size_t workspace_bytes, hipStream_t s);
// both directions of a pair in ONE launch (ssd_rev = correlate(mov, fix); workspace_rev: a second workspace of the same size)
int launch_corr_fused_dual(const float* fix, const float* mov, int C, int h, int w, int d, int hw, int cost, int n_box, int fast, int f16,
                           void* ssd, void* ssd_rev, void* workspace, size_t workspace_bytes, void* workspace_rev, hipStream_t s);
// corrcert.hip: the cost volume in the certified-fast arithmetic -- UNSCALED sums over the 729 taps, |ssdu / 729 - ssd| <= 2^-16 ssd
bool corr_cert_supported(int C, int h, int w, int d, int hw);
size_t corr_cert_workspace_bytes(int C, int h, int w, int d, int hw);
int launch_corr_cert(const float* fix, const float* mov, int C, int h, int w, int d, int hw, float* ssdu, void* workspace, size_t workspace_bytes,
                     hipStream_t s);
// correlate.hip: the certified-fast volume from whichever kernel option corr_cert selects
bool corr_certfast_supported(int C, int h, int w, int d, int hw);
size_t corr_certfast_workspace_bytes(int C, int h, int w, int d, int hw);
int launch_corr_certfast(const float* fix, const float* mov, int C, int h, int w, int d, int hw, float* ssdu, void* workspace, size_t workspace_bytes,
                         hipStream_t s);
// certify.hip: decisions on the certified-fast volume that equal the reference's on the exact one
size_t corr_certify_workspace_bytes(int C, int h, int w, int d, int hw, bool plain = false);     // plain: the plain argmin only
int corr_certified_argmin(const float* ssdu, const float* fix, const float* mov, int C, int h, int w, int d, int hw, int64_t* argmin,
                          void* workspace, size_t workspace_bytes, hipStream_t s);
// both directions (ssduB == nullptr: one) from the certified-fast volumes to the smoothed fields: plain argmin + six coupled passes
int coupled_convex_cert_impl(const float* ssduA, const float* fixA, const float* movA, float* outA, void* wsA, const float* ssduB, const float* fixB,
                             const float* movB, float* outB, void* wsB, const float* mesh, int C, int h, int w, int d, int hw, size_t workspace_bytes,
                             hipStream_t s, int stage = 0);      // stage 0 = everything, 1..5 = arm / stream A / stream B / certify the plain argmin / coupled passes
// correlate.hip: would cvx_correlate_ex_f32 take the unfused round-1 kernels for this problem?
bool corr_use_unfused(int C, int h, int w, int d, int hw, bool variant);
// boxmarch.hip: three chained 3^3 boxes (forward / adjoint / adjoint + Adam) for rows of at most 126 voxels
bool box3_march_supported(int d);
int launch_box3_march(const float* in, float* out, int h, int w, int d, bool backward, float* P, float* m, float* v,
                      AdamConsts ac, float* gsave, hipStream_t s);
// boxtile.hip: the forward boxes as independent tiles (two barriers per workgroup instead of one per plane)
bool box3_tile_fwd_supported(const float* in, const float* out, int h, int w, int d);
int launch_box3_tile_fwd(const float* in, float* out, int h, int w, int d, int variant, hipStream_t s);
int box3_tile_fwd_auto(int h, int w, int d);       // variant for this grid, 0 = keep the marching kernel
// the same tiles for the exact adjoint boxes (ATen's avg_pool3d_backward order), optionally with the Adam update in the last pass
bool box3_tile_supported(const float* in, const float* out, int h, int w, int d, const float* P, const float* m, const float* v, const float* gsave);
int launch_box3_tile(const float* in, float* out, int h, int w, int d, int variant, bool backward, float* P, float* m, float* v, AdamConsts ac,
                     float* gsave, hipStream_t s);
// warp.hip: [C][V] -> [CP/4][V][4] feature copies and the warp + data-term gradient of one Adam iteration
// (half: records of four half-precision values -- fp16 storage of the pooled features -- instead of four floats)
int launch_to_chunked(const float* in, int C, size_t V, float* out, bool half, hipStream_t s);
int launch_warp_grad(const float* Fcl, const float* Mcl, int C, int h, int w, int d, const float* U, const float* bh,
                     const float* bw, const float* bd, float gsc, float cH, float cW, float cD, float* gU, bool half, hipStream_t s);
// adamfast.hip: adam_mode "fast" -- FMA / factored warp + gradient (float32 records) and the separable adjoint boxes with the Adam
// update in the epilogue (P != nullptr: in-place update of P, m, v with G = box(in); P == nullptr: out = box(in)); bc1, bc2 = the bias
// corrections 1 - beta^step of this iteration
int launch_warp_grad_fast(const float* Fcl, const float* Mcl, int C, int h, int w, int d, const float* U, const float* bh,
                          const float* bw, const float* bd, float gsc, float cH, float cW, float cD, float* gU, bool half, hipStream_t s);
int launch_box3_fast(const float* in, float* out, int h, int w, int d, float* P, float* m, float* v, double bc1, double bc2,
                     float* gsave, hipStream_t s);
// the same arithmetic for a chain of boxes (the sweep's kovesi splines): three 1-D passes, [3][h][w][d], in == out allowed
bool boxchain_fast_supported(const cvx_smoother& sm, int h, int w, int d);
int launch_boxchain_fast(const float* in, float* out, int h, int w, int d, const cvx_smoother& sm, bool reverse, hipStream_t s);
int launch_adam_update_fast(const float* G, float* P, float* m, float* v, size_t n, double bc1, double bc2, hipStream_t s);

}  // namespace cvx
