// pipeline.hip -- one registration of an image pair, end to end on one stream
// (reference: convex_adam_pt, src/convexAdam/convex_adam_MIND.py:64-202; multi-channel variant
// convex_adam_nnUNet.py:41-159).  Every stage is one of the C-ABI operators; nothing touches the host
// between the upload of the two images and the finished displacement field.
#include <vector>

#include "cvx_common.h"

namespace cvx {

// identity coordinate of F.affine_grid(align_corners=False) and the search mesh of
// convex_adam_MIND.py:127, built on the device with the same float ops as the host helpers
// (torch.linspace restated: start + step*i / end - step*(S-1-i), each a single fused rounding).
__device__ __forceinline__ float linspace_pm1_at(int S, int i) {
    if (S == 1) return -1.0f;
    const float step = fdiv(1.0f - (-1.0f), (float)(S - 1));
    return (i < S / 2) ? __builtin_fmaf(step, (float)i, -1.0f) : __builtin_fmaf(-step, (float)(S - 1 - i), 1.0f);
}
// all six identity tables of a pair in one launch: (extent, destination) x 6, one workgroup row per table
struct BaseTables { int S[6]; float* out[6]; };
__global__ void k_affine_bases(BaseTables t) {
    const int S = t.S[blockIdx.y];
    float* out = t.out[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; out && i < S; i += gridDim.x * blockDim.x)
        out[i] = fdiv(linspace_pm1_at(S, i) * (float)(S - 1), (float)S);
}
__global__ void k_disp_mesh(int hw, float* out) {
    const int n = 2 * hw + 1, K = n * n * n;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int c = k % n, b = (k / n) % n, a = k / (n * n);
    const float la = n == 1 ? 0.f : linspace_pm1_at(n, a), lb = n == 1 ? 0.f : linspace_pm1_at(n, b),
                lc = n == 1 ? 0.f : linspace_pm1_at(n, c);
    out[k] = lc * (float)hw;
    out[K + k] = lb * (float)hw;
    out[2 * K + k] = la * (float)hw;
}

// Everything of a pair that depends on nothing but its geometry, in ONE launch instead of seven (two table kernels, the two all-ones fills
// of the plain argmins' key buffers, the list counters of both coupled-convex solves, the zero Adam state): the identity tables, the
// search mesh, keys = ~0, counts = 0, m = v = 0.
struct PairSetup {
    BaseTables t;
    int hw; float* mesh;
    unsigned long long* keys[2]; size_t v;
    int* counts[2];
    float4* zero[2]; size_t nzero4;
};
__global__ __launch_bounds__(256) void k_pair_setup(PairSetup a) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int S = a.t.S[t];
        float* out = a.t.out[t];
        for (size_t i = gid; out && i < (size_t)S; i += gsz) out[i] = fdiv(linspace_pm1_at(S, (int)i) * (float)(S - 1), (float)S);
    }
    const int n = 2 * a.hw + 1, K = n * n * n;
    for (size_t k = gid; k < (size_t)K; k += gsz) {
        const int c = (int)k % n, b = ((int)k / n) % n, aa = (int)k / (n * n);
        const float la = n == 1 ? 0.f : linspace_pm1_at(n, aa), lb = n == 1 ? 0.f : linspace_pm1_at(n, b), lc = n == 1 ? 0.f : linspace_pm1_at(n, c);
        a.mesh[k] = lc * (float)a.hw;
        a.mesh[K + k] = lb * (float)a.hw;
        a.mesh[2 * K + k] = la * (float)a.hw;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (a.keys[q]) for (size_t x = gid; x < a.v; x += gsz) a.keys[q][x] = ~0ull;
        if (a.counts[q] && gid < 2) a.counts[q][gid] = 0;
        if (a.zero[q]) for (size_t i = gid; i < a.nzero4; i += gsz) a.zero[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// (disp_soft / scale).flip(1)                                             convex_adam_MIND.py:134,139
// (both directions of a pair in one launch: blockIdx.y selects the field)
__global__ __launch_bounds__(256) void k_ic_prepare(const float* __restrict__ soft, const float* __restrict__ soft_rev, int h, int w, int d,
                                                    float* __restrict__ out, float* __restrict__ out_rev) {
    const size_t v = (size_t)h * w * d;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= v) return;
    if (blockIdx.y) { soft = soft_rev; out = out_rev; }
    const float sc[3] = {(float)(h - 1) / 2.0f, (float)(w - 1) / 2.0f, (float)(d - 1) / 2.0f};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)c * v + p] = fdiv(soft[(size_t)(2 - c) * v + p], sc[2 - c]);
}
// disp_ice.flip(1) * scale * grid_sp                                        convex_adam_MIND.py:141
__global__ __launch_bounds__(256) void k_ic_finish(const float* __restrict__ ice, int h, int w, int d, float gs, float* __restrict__ out) {
    const size_t v = (size_t)h * w * d;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= v) return;
    const float sc[3] = {(float)(h - 1) / 2.0f, (float)(w - 1) / 2.0f, (float)(d - 1) / 2.0f};
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * v + p] = (ice[(size_t)(2 - a) * v + p] * sc[a]) * gs;
}

struct PairLayout {
    // extents
    int H, W, D, h, w, d, h2, w2, d2, C, K;
    size_t V, v, V2;
    // byte offsets into the workspace (0 = not used)
    size_t featF, featM, mind_ws, mind_ws2, fs, ms, corr_ws, corr_ws2, ssd, argmin, mesh, conv_ws, ssd2, argmin2, conv_ws2, cert_ws, cert_ws2, soft, soft2, in1, in2, ic1, ic2, ic_ws,
        upin, disp_hr, F2, M2, P, m, v_, U, adam_ws, smooth_ws, snaps, bh, bw, bd, bh2, bw2, bd2, total;
};

static size_t take(size_t& used, size_t bytes) {
    const size_t off = align_up(used, 256);
    used = off + bytes;
    return off;
}

static PairLayout pair_layout(const cvx_pair_params& p, int n_snap = 0, int max_smooth = 0) {
    PairLayout L{};
    L.H = p.H; L.W = p.W; L.D = p.D;
    L.h = p.H / p.grid_sp; L.w = p.W / p.grid_sp; L.d = p.D / p.grid_sp;
    L.h2 = p.H / p.grid_sp_adam; L.w2 = p.W / p.grid_sp_adam; L.d2 = p.D / p.grid_sp_adam;
    L.C = p.n_feat > 0 ? p.n_feat : 12;
    const int n = 2 * p.disp_hw + 1;
    L.K = n * n * n;
    L.V = (size_t)p.H * p.W * p.D; L.v = (size_t)L.h * L.w * L.d; L.V2 = (size_t)L.h2 * L.w2 * L.d2;
    size_t u = 256;      // offset 0 is reserved as "unused"
    const size_t f = sizeof(float);
    if (p.n_feat == 0) {
        // (the pooled path's raw patch SSDs may be blocked by tiles that overhang the volume: mind_pooled_raw_floats >= 12 V)
        const size_t raw_floats = mind_pooled_raw_floats(p.H, p.W, p.D, p.grid_sp, p.lambda_weight > 0 ? p.grid_sp_adam : 0);
        L.featF = take(u, f * raw_floats);
        L.featM = take(u, f * raw_floats);
        L.mind_ws = take(u, cvx_mindssc_workspace_bytes(p.H, p.W, p.D, p.mind_r, p.mind_d));
        L.mind_ws2 = take(u, cvx_mindssc_workspace_bytes(p.H, p.W, p.D, p.mind_r, p.mind_d));      // the moving image's pass runs beside the fixed one's
    }
    L.fs = take(u, f * L.C * L.v);
    L.ms = take(u, f * L.C * L.v);
    L.corr_ws = take(u, cvx_correlate_workspace_bytes(L.C, L.h, L.w, L.d, p.disp_hw));
    // the reverse direction's padded copies (both directions in one launch): only where that path can run (option corr_dual, off by default)
    L.corr_ws2 = (p.ic && options().corr_dual != 0 && corr_fused_supported(L.C, L.h, L.w, L.d, p.disp_hw)) ? take(u, corr_fused_workspace_bytes(L.C, L.h, L.w, L.d, p.disp_hw)) : 0;
    // fp16 storage: the cost volumes hold __half (half the bytes written by the correlation kernel and read by every argmin pass)
    const size_t ssd_elem = p.fp16_storage ? 2 : f;
    L.ssd = take(u, ssd_elem * (size_t)L.K * L.v);
    L.argmin = take(u, sizeof(int64_t) * L.v);
    L.mesh = take(u, f * 3 * (size_t)L.K);
    L.conv_ws = take(u, cvx_coupled_convex_workspace_bytes(L.h, L.w, L.d, p.disp_hw));
    // certified decisions on the fast cost volume (certify.hip): one small workspace per direction
    L.cert_ws = take(u, corr_certify_workspace_bytes(L.C, L.h, L.w, L.d, p.disp_hw));
    if (p.ic) L.cert_ws2 = take(u, corr_certify_workspace_bytes(L.C, L.h, L.w, L.d, p.disp_hw));
    L.soft = take(u, f * 3 * L.v);
    L.bh = take(u, f * L.h); L.bw = take(u, f * L.w); L.bd = take(u, f * L.d);
    if (p.ic) {
        // the reverse direction keeps its own cost volume: both coupled-convex solves share their launches
        L.ssd2 = take(u, ssd_elem * (size_t)L.K * L.v);
        L.argmin2 = take(u, sizeof(int64_t) * L.v);
        L.conv_ws2 = take(u, cvx_coupled_convex_workspace_bytes(L.h, L.w, L.d, p.disp_hw));
        L.soft2 = take(u, f * 3 * L.v);
        L.in1 = take(u, f * 3 * L.v); L.in2 = take(u, f * 3 * L.v);
        L.ic1 = take(u, f * 3 * L.v); L.ic2 = take(u, f * 3 * L.v);
        L.ic_ws = take(u, cvx_inverse_consistency_workspace_bytes(L.h, L.w, L.d));
        L.upin = take(u, f * 3 * L.v);
        L.disp_hr = take(u, f * 3 * L.V);
    }
    if (p.lambda_weight > 0) {
        // planar pooled features, or (MIND path, option mind_records) the Adam loop's records: [CP/4][V2 + 1][4]
        const size_t f2_bytes = f * (size_t)((L.C + 3) / 4 * 4) * (L.V2 + 1);
        L.F2 = take(u, f2_bytes);
        L.M2 = take(u, f2_bytes);
        L.P = take(u, f * 3 * L.V2); L.m = take(u, f * 3 * L.V2); L.v_ = take(u, f * 3 * L.V2); L.U = take(u, f * 3 * L.V2);
        L.adam_ws = take(u, cvx_adam_workspace_bytes(L.C, L.h2, L.w2, L.d2));
        L.bh2 = take(u, f * L.h2); L.bw2 = take(u, f * L.w2); L.bd2 = take(u, f * L.d2);
        if (p.selected_smooth > 0 || max_smooth > 0) L.smooth_ws = take(u, 2 * (256 + f * 3 * L.V));
        if (n_snap > 0) L.snaps = take(u, f * 3 * L.V2 * (size_t)n_snap);
    }
    L.total = u + 256;
    return L;
}

// ---- optional per-stage timing -------------------------------------------------------------------------
static thread_local int g_profiling = 0;       // 0 off, 1 last call only, 2 accumulate over calls, 3 = 2 + one mark per kernel of the Adam loop
struct StageMark { const char* name; hipEvent_t ev; };
static thread_local std::vector<StageMark> g_marks;
static thread_local std::vector<hipEvent_t> g_pool;
static thread_local size_t g_pool_used = 0;

static void mark(const char* name, hipStream_t s) {
    if (!g_profiling) return;
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_pool.push_back(e);
    }
    hipEvent_t e = g_pool[g_pool_used++];
    (void)hipEventRecord(e, s);
    g_marks.push_back({name, e});
}
// (adam.hip) one mark behind every kernel of the Adam loop when cvx_set_profiling(3) is on: per-kernel durations from events on the
// launch stream, the iteration's three launches named "adam.forward_boxes", "adam.warp_gradient", "adam.adjoint_update"
void profile_mark_kernel(const char* name, hipStream_t s) {
    if (g_profiling == 3) mark(name, s);
}

// ---- side stream for the independent half of a stage (the two images' descriptors) ---------------------------------------
// One per (thread, slot): slot = the internal stream index of cvx_register_pairs_f32 (0 for a single pair), so that concurrent pairs do
// not share a side stream.  Forks from / joins into the pair's stream with events; created once per thread and device.
struct SidePool {
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> fork, join;
    int device = -1;
};
static thread_local SidePool g_side;
static thread_local int g_side_slot = 0;
static bool side_stream(hipStream_t* side, hipEvent_t* fork, hipEvent_t* join) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    SidePool& P = g_side;
    if (P.device != dev) { P.streams.clear(); P.fork.clear(); P.join.clear(); P.device = dev; }        // leaked on device switch (rare)
    while ((int)P.streams.size() <= g_side_slot) {
        hipStream_t st; hipEvent_t a, b;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        P.streams.push_back(st); P.fork.push_back(a); P.join.push_back(b);
    }
    *side = P.streams[g_side_slot]; *fork = P.fork[g_side_slot]; *join = P.join[g_side_slot];
    return true;
}

static int validate(const cvx_pair_params* p) {
    CVX_REQUIRE(p, "cvx_register_pair: null params");
    CVX_REQUIRE(p->H > 0 && p->W > 0 && p->D > 0, "cvx_register_pair: bad extent");
    CVX_REQUIRE(p->grid_sp >= 1 && p->grid_sp_adam >= 1, "cvx_register_pair: grid spacing must be >= 1");
    CVX_REQUIRE(p->H / p->grid_sp >= 2 && p->W / p->grid_sp >= 2 && p->D / p->grid_sp >= 2,
                "cvx_register_pair: volume too small for grid_sp %d", p->grid_sp);
    CVX_REQUIRE(p->disp_hw >= 0 && p->disp_hw <= CVX_MAX_DISP_HW, "cvx_register_pair: disp_hw %d not in 0..%d", p->disp_hw, CVX_MAX_DISP_HW);
    CVX_REQUIRE(p->n_feat >= 0 && p->n_feat < 256, "cvx_register_pair: n_feat out of range");
    CVX_REQUIRE(p->selected_smooth == 0 || (p->selected_smooth & 1), "cvx_register_pair: selected_smooth must be odd "
                "(an even kernel changes the volume size in the reference, convex_adam_MIND.py:185-191)");
    CVX_REQUIRE((p->cost == 0 || p->cost == 1) && (p->n_box == 0 || p->n_box == 1 || p->n_box == 2) &&
                (p->n_spline_pools == 0 || p->n_spline_pools == 2 || p->n_spline_pools == 3) && (p->corr_fast == 0 || p->corr_fast == 1) &&
                (p->fp16_storage == 0 || p->fp16_storage == 1), "cvx_register_pair: bad variant fields (cost, n_box, n_spline_pools, corr_fast, fp16_storage)");
    CVX_REQUIRE((p->adam_fast == 0 || p->adam_fast == 1 || p->adam_fast == 2) && p->reserved_[0] == 0 && p->reserved_[1] == 0 && p->reserved_[2] == 0,
                "cvx_register_pair: adam_fast must be 0, 1 or 2 and the reserved fields zero (struct laid out by an older header? check cvx_version())");
    CVX_REQUIRE(!p->adam_fast || p->n_spline_pools != 2, "cvx_register_pair: adam_fast needs the packaged smoother (three 3^3 boxes)");
    if (p->lambda_weight > 0) {
        CVX_REQUIRE(p->selected_niter >= 1, "cvx_register_pair: selected_niter must be >= 1 when lambda_weight > 0 "
                    "(the reference raises UnboundLocalError, convex_adam_MIND.py:181)");
        CVX_REQUIRE(p->H / p->grid_sp_adam >= 2 && p->W / p->grid_sp_adam >= 2 && p->D / p->grid_sp_adam >= 2,
                    "cvx_register_pair: volume too small for grid_sp_adam %d", p->grid_sp_adam);
    }
    return CVX_OK;
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_disp_mesh_f32(int disp_hw, float* out_device, void* stream) {
    CVX_REQUIRE(out_device && disp_hw >= 0 && disp_hw <= CVX_MAX_DISP_HW, "cvx_disp_mesh_f32: bad arguments (disp_hw 0 .. %d)", CVX_MAX_DISP_HW);
    const int n = 2 * disp_hw + 1, K = n * n * n;
    hipLaunchKernelGGL(k_disp_mesh, dim3(cdiv(K, 256)), dim3(256), 0, as_stream(stream), disp_hw, out_device);
    return check_last("disp_mesh");
}
extern "C" int cvx_affine_base_f32(int S, float* out_device, void* stream) {
    CVX_REQUIRE(out_device && S >= 1, "cvx_affine_base_f32: bad arguments");
    BaseTables t = {{S, 0, 0, 0, 0, 0}, {out_device, nullptr, nullptr, nullptr, nullptr, nullptr}};
    hipLaunchKernelGGL(k_affine_bases, dim3(cdiv(S, 64) > 4 ? cdiv(S, 64) : 4, 6), dim3(64), 0, as_stream(stream), t);
    return check_last("affine_base");
}

extern "C" void cvx_set_profiling(int enabled) {
    g_profiling = enabled < 0 ? 0 : (enabled > 3 ? 3 : enabled);
    g_marks.clear();
    g_pool_used = 0;
}

extern "C" int cvx_last_pair_profile(const char** names_host, float* ms_host, int max_stages) {
    if (g_marks.size() < 2) return 0;
    (void)hipEventSynchronize(g_marks.back().ev);
    int n = 0;
    for (size_t i = 1; i < g_marks.size() && n < max_stages; ++i) {
        if (g_marks[i].name[0] == 's' && g_marks[i].name[1] == 't' && g_marks[i].name[2] == 'a') continue;   // "start" of the next pair
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_marks[i - 1].ev, g_marks[i].ev);
        names_host[n] = g_marks[i].name;
        ms_host[n] = ms;
        ++n;
    }
    return n;
}

extern "C" size_t cvx_register_pair_workspace_bytes(const cvx_pair_params* p) {
    if (validate(p) != CVX_OK) return 0;
    const ContextScope scope(p->ctx);
    return pair_layout(*p).total;
}

namespace cvx {
static int smooth_max(const int* smooth_host, int n_smooth) {
    int m = 0;
    for (int i = 0; i < n_smooth; ++i) m = smooth_host[i] > m ? smooth_host[i] : m;
    return m;
}
// out_field: the field of the packaged pipeline ([3][H][W][D]) when n_snap == 0; otherwise [n_snap][n_smooth][3][H][W][D]: for every
// listed Adam iteration the up-sampled disp_sample of that iteration, once per listed final smoothing (0 = none, k = three k^3 boxes)
static int register_pair_core(const float* img_fixed, const float* img_moving, const float* feat_fixed, const float* feat_moving,
                              const cvx_pair_params* p, float* out_field, int* out_dims_host, const int* snap_iters_host, int n_snap,
                              const int* smooth_host, int n_smooth, void* workspace, size_t workspace_bytes, void* stream);
}  // namespace cvx

extern "C" int cvx_register_pair_f32(const float* img_fixed, const float* img_moving, const float* feat_fixed,
                                     const float* feat_moving, const cvx_pair_params* p, float* out_field,
                                     int* out_dims_host, void* workspace, size_t workspace_bytes, void* stream) {
    return register_pair_core(img_fixed, img_moving, feat_fixed, feat_moving, p, out_field, out_dims_host, nullptr, 0, nullptr, 0, workspace,
                              workspace_bytes, stream);
}

extern "C" size_t cvx_register_pair_snapshots_workspace_bytes(const cvx_pair_params* p, int n_snap, const int* smooth_host, int n_smooth) {
    if (validate(p) != CVX_OK || n_snap < 1 || n_smooth < 1 || !smooth_host) return 0;
    const ContextScope scope(p->ctx);
    return pair_layout(*p, n_snap, smooth_max(smooth_host, n_smooth)).total;
}

extern "C" int cvx_register_pair_snapshots_f32(const float* img_fixed, const float* img_moving, const float* feat_fixed,
                                               const float* feat_moving, const cvx_pair_params* p, const int* snapshot_iters_host, int n_snap,
                                               const int* smooth_host, int n_smooth, float* out_fields, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    int rc = validate(p);
    if (rc) return rc;
    CVX_REQUIRE(snapshot_iters_host && n_snap >= 1 && smooth_host && n_smooth >= 1, "cvx_register_pair_snapshots_f32: snapshot / smoothing lists missing");
    CVX_REQUIRE(p->lambda_weight > 0, "cvx_register_pair_snapshots_f32: snapshots exist only with the Adam stage (lambda_weight > 0)");
    for (int i = 0; i < n_snap; ++i)
        CVX_REQUIRE(snapshot_iters_host[i] >= 1 && snapshot_iters_host[i] <= p->selected_niter && (i == 0 || snapshot_iters_host[i] > snapshot_iters_host[i - 1]),
                    "cvx_register_pair_snapshots_f32: snapshot iterations must be ascending and within 1..selected_niter");
    for (int i = 0; i < n_smooth; ++i)
        CVX_REQUIRE(smooth_host[i] == 0 || (smooth_host[i] & 1), "cvx_register_pair_snapshots_f32: smoothing sizes must be 0 or odd");
    return register_pair_core(img_fixed, img_moving, feat_fixed, feat_moving, p, out_fields, nullptr, snapshot_iters_host, n_snap, smooth_host,
                              n_smooth, workspace, workspace_bytes, stream);
}

static int cvx::register_pair_core(const float* img_fixed, const float* img_moving, const float* feat_fixed, const float* feat_moving,
                                   const cvx_pair_params* p, float* out_field, int* out_dims_host, const int* snap_iters_host, int n_snap,
                                   const int* smooth_host, int n_smooth, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = validate(p);
    if (rc) return rc;
    const ContextScope scope(p->ctx);           // switches and tables of this call (cvx_pair_params.ctx; nullptr keeps the thread's)
    CVX_REQUIRE(out_field && workspace, "cvx_register_pair_f32: null pointer");
    if (p->n_feat == 0) CVX_REQUIRE(img_fixed && img_moving, "cvx_register_pair_f32: images missing");
    else CVX_REQUIRE(feat_fixed && feat_moving, "cvx_register_pair_f32: feature volumes missing");
    const PairLayout L = pair_layout(*p, n_snap, n_snap ? smooth_max(smooth_host, n_smooth) : 0);
    if (workspace_bytes < L.total) return fail(CVX_ERR_WORKSPACE, "cvx_register_pair_f32: workspace %zu < %zu", workspace_bytes, L.total);
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    if (g_profiling < 2) { g_marks.clear(); g_pool_used = 0; }
    mark("start", s);
    // stage intervals "correlate" / "correlate_rev" = the fused kernel alone; the padded feature copies (and the certification set-up before them) are
    // "correlate_prep" (only while profiling: the hook records an event)
    corr_fused_set_prep_hook(g_profiling ? +[](hipStream_t st) { mark("correlate_prep", st); } : nullptr);
    struct HookReset { ~HookReset() { corr_fused_set_prep_hook(nullptr); } } hook_reset;

    // 1. features                                                              (:106-116)
    const float *featF = feat_fixed, *featM = feat_moving;
    // MIND path: the descriptor is consumed only through its two stride poolings, so when the window sizes tile it is never
    // written at full resolution (launch_mind_pooled: raw patch SSDs -> normalise + exp + both poolings in one pass)
    const bool adam = p->lambda_weight > 0;
    const bool pooled_mind = p->n_feat == 0 && mind_pooled_supported(p->H, p->W, p->D, p->grid_sp, adam ? p->grid_sp_adam : 0);
    // the Adam-grid pooling of the descriptor written as the loop's feature records (no planar copy, no re-packing pass)
    const bool mind_records = pooled_mind && adam && options().mind_records != 0 && mind_pooled_records_supported(p->H, p->W, p->D, p->grid_sp, p->grid_sp_adam);
    const int rec_kind = mind_records ? (p->fp16_storage ? 2 : 1) : 0;
    if (p->n_feat == 0) {
        const size_t mws = cvx_mindssc_workspace_bytes(p->H, p->W, p->D, p->mind_r, p->mind_d);
        if (pooled_mind) {
            const int g2 = adam ? p->grid_sp_adam : 0;
            // The two images are independent until `correlate`.  Option mind_overlap = 1 runs the moving image's pass on a side stream (one
            // image's VALU-bound stencil beside the other's memory-bound normalise + pool pass); measured on the benchmark pair: 0.52 vs
            // 0.53 ms for the stage and a SLOWER pair (7.92 vs 7.76 ms) -- every one of these kernels fills the chip on its own -- so the
            // default keeps them in order.
            hipStream_t side = s; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
            const bool overlap = options().mind_overlap != 0 && side_stream(&side, &ev_fork, &ev_join);
            if (overlap) { (void)hipEventRecord(ev_fork, s); (void)hipStreamWaitEvent(side, ev_fork, 0); }
            else side = s;
            if ((rc = launch_mind_pooled(img_fixed, p->H, p->W, p->D, p->mind_r, p->mind_d, p->grid_sp, F(L.fs), g2, adam ? F(L.F2) : nullptr,
                                         F(L.featF), ws + L.mind_ws, mws, s, rec_kind))) return rc;
            rc = launch_mind_pooled(img_moving, p->H, p->W, p->D, p->mind_r, p->mind_d, p->grid_sp, F(L.ms), g2, adam ? F(L.M2) : nullptr,
                                    F(L.featM), ws + (overlap ? L.mind_ws2 : L.mind_ws), mws, side, rec_kind);
            if (overlap) { (void)hipEventRecord(ev_join, side); (void)hipStreamWaitEvent(s, ev_join, 0); }      // (joined even after an error)
            if (rc) return rc;
        } else {
            if ((rc = cvx_mindssc_f32(img_fixed, p->H, p->W, p->D, p->mind_r, p->mind_d, F(L.featF), ws + L.mind_ws, mws, stream))) return rc;
            if ((rc = cvx_mindssc_f32(img_moving, p->H, p->W, p->D, p->mind_r, p->mind_d, F(L.featM), ws + L.mind_ws, mws, stream))) return rc;
        }
        featF = F(L.featF); featM = F(L.featM);
    }
    mark("mind", s);
    // 2. coarse features                                                       (:118-119)
    if (!pooled_mind) {
        if ((rc = cvx_avgpool_f32(featF, L.C, p->H, p->W, p->D, p->grid_sp, F(L.fs), stream))) return rc;
        if ((rc = cvx_avgpool_f32(featM, L.C, p->H, p->W, p->D, p->grid_sp, F(L.ms), stream))) return rc;
    }
    const size_t vws = cvx_coupled_convex_workspace_bytes(L.h, L.w, L.d, p->disp_hw);
    // first key buffer of the coupled-convex workspace (carved exactly as coupled_core does): the plain argmin leaves its keys there
    unsigned long long* keys = Carver(ws + L.conv_ws, vws).take<unsigned long long>(L.v);
    unsigned long long* keys2 = p->ic ? Carver(ws + L.conv_ws2, vws).take<unsigned long long>(L.v) : nullptr;
    const bool no_prune = options().no_prune != 0;        // streaming coupled passes need int64 winners
    {
        const bool adam_tables = p->lambda_weight > 0;
        // (m, v zeroed here only if 16-byte stores fit: 3 * V2 floats from a 256-byte aligned offset)
        const bool zero_state = adam_tables && (3 * L.V2) % 4 == 0;
        PairSetup a = {{{L.h, L.w, L.d, L.h2, L.w2, L.d2},
                        {F(L.bh), F(L.bw), F(L.bd), adam_tables ? F(L.bh2) : nullptr, adam_tables ? F(L.bw2) : nullptr, adam_tables ? F(L.bd2) : nullptr}},
                       p->disp_hw, F(L.mesh), {keys, keys2}, L.v,
                       {no_prune ? nullptr : coupled_ws_counts(ws + L.conv_ws, vws, L.h, L.w, L.d, p->disp_hw),
                        (no_prune || !p->ic) ? nullptr : coupled_ws_counts(ws + L.conv_ws2, vws, L.h, L.w, L.d, p->disp_hw)},
                       {zero_state ? reinterpret_cast<float4*>(ws + L.m) : nullptr, zero_state ? reinterpret_cast<float4*>(ws + L.v_) : nullptr}, 3 * L.V2 / 4};
        hipLaunchKernelGGL(k_pair_setup, dim3(zero_state ? 1024 : 64), dim3(256), 0, s, a);
        if (adam_tables && !zero_state) {
            (void)hipMemsetAsync(F(L.m), 0, sizeof(float) * 3 * L.V2, s);
            (void)hipMemsetAsync(F(L.v_), 0, sizeof(float) * 3 * L.V2, s);
        }
    }
    mark("pool", s);
    // 3. forward correlation + coupled convex                                  (:124-130)
    const size_t cws = cvx_correlate_workspace_bytes(L.C, L.h, L.w, L.d, p->disp_hw);
    int64_t* am = reinterpret_cast<int64_t*>(ws + L.argmin);
    const bool f16 = p->fp16_storage != 0;
    const cvx_corr_opts copt = {p->cost, p->n_box == 1 ? 1 : 2, p->corr_fast, f16 ? 2 : 0};
    const bool variant = copt.cost || copt.n_box == 1 || copt.fast || copt.f16;
    if (p->fp16_storage) {                      // features are stored in half precision by the reference's GPU default (MIND:79)
        if ((rc = cvx_round_f16_f32(F(L.fs), (int64_t)L.C * L.v, stream)) || (rc = cvx_round_f16_f32(F(L.ms), (int64_t)L.C * L.v, stream))) return rc;
    }
    // Certified-fast path (option corr_cert, default): the cost volumes in the fast arithmetic (unscaled, 2^-16 relative to ATen's), every
    // argmin decision certified against the exact arithmetic or evaluated exactly (certify.hip) -- the SAME winners, hence the same field bits,
    // as the exact kernels below; packaged operator only (SSD, two boxes, float32, pruned passes).
    // (C >= 16: the role kernel carries the channel sums in a third of its wavefronts.  In the fast arithmetic it needs no cascade -- 64 registers, two
    // workgroups per CU -- and beats the exact kernels up to 32 channels where its items fill the chip: C = 32 at 26x32x37 hw 6 0.36 vs 0.45 ms,
    // C = 16 0.21 vs 0.33; with 162 items (hw 4) or 64 channels it loses -- C = 64 hw 4: 0.56 vs 0.23 ms --, tools/experiments/corr_time_c.py;
    // corr_cert = 2 keeps the staged kernel selectable for every supported C)
    const bool cert = options().corr_cert != 0 && !variant && !no_prune && corr_certfast_supported(L.C, L.h, L.w, L.d, p->disp_hw) &&
                      (options().corr_cert == 2 || corr_certfast_pays(L.C, L.h, L.w, L.d, p->disp_hw));
    int64_t* am2 = p->ic ? reinterpret_cast<int64_t*>(ws + L.argmin2) : nullptr;
    if (cert) {
        const size_t fws = corr_certfast_workspace_bytes(L.C, L.h, L.w, L.d, p->disp_hw), qws = corr_certify_workspace_bytes(L.C, L.h, L.w, L.d, p->disp_hw);
        auto cert_stage = [&](int stage) {
            return coupled_convex_cert_impl(F(L.ssd), F(L.fs), F(L.ms), F(L.soft), ws + L.cert_ws, p->ic ? F(L.ssd2) : nullptr, F(L.ms), F(L.fs),
                                            p->ic ? F(L.soft2) : nullptr, p->ic ? ws + L.cert_ws2 : nullptr, F(L.mesh), L.C, L.h, L.w, L.d, p->disp_hw, qws, s, stage);
        };
        if ((rc = cert_stage(1))) return rc;                                    // keys, counters, tail values
        if ((rc = launch_corr_certfast(F(L.fs), F(L.ms), L.C, L.h, L.w, L.d, p->disp_hw, F(L.ssd), ws + L.corr_ws, fws, s))) return rc;
        mark("correlate", s);
        if ((rc = cert_stage(2))) return rc;                                    // the plain argmin streams the volume while the Infinity Cache holds it
        mark("argmin", s);
        if (p->ic) {
            if ((rc = launch_corr_certfast(F(L.ms), F(L.fs), L.C, L.h, L.w, L.d, p->disp_hw, F(L.ssd2), ws + L.corr_ws, fws, s))) return rc;
            mark("correlate_rev", s);
            if ((rc = cert_stage(3))) return rc;
            mark("argmin_rev", s);
        }
        if ((rc = cert_stage(4)) || (rc = cert_stage(5))) return rc;
    } else {
    // Both directions' cost volumes in ONE launch of the fused kernel when the pair is inverse consistent (option corr_dual): the stage
    // interval "correlate" then covers both directions and "correlate_rev" is not recorded.
    const bool dual = p->ic && options().corr_dual != 0 && !corr_use_unfused(L.C, L.h, L.w, L.d, p->disp_hw, variant) && p->disp_hw <= CVX_MAX_DISP_HW;
    if (dual) {
        const size_t fws = corr_fused_workspace_bytes(L.C, L.h, L.w, L.d, p->disp_hw);
        if ((rc = launch_corr_fused_dual(F(L.fs), F(L.ms), L.C, L.h, L.w, L.d, p->disp_hw, copt.cost, copt.n_box, copt.fast, copt.f16, F(L.ssd), F(L.ssd2),
                                         ws + L.corr_ws, fws, ws + L.corr_ws2, s))) return rc;
        mark("correlate", s);
        if (no_prune) rc = launch_argmin(F(L.ssd), f16, nullptr, nullptr, 0.0f, false, L.K, L.v, keys, am, s);
        else rc = launch_argmin_keys(F(L.ssd), f16, L.K, L.v, keys, /*arm=*/false, s);
        if (rc) return rc;
        mark("argmin", s);
        if (no_prune) rc = launch_argmin(F(L.ssd2), f16, nullptr, nullptr, 0.0f, false, L.K, L.v, keys2, am2, s);
        else rc = launch_argmin_keys(F(L.ssd2), f16, L.K, L.v, keys2, /*arm=*/false, s);
        if (rc) return rc;
        mark("argmin_rev", s);
    } else {
    if ((rc = cvx_correlate_ex_f32(F(L.fs), F(L.ms), L.C, L.h, L.w, L.d, p->disp_hw, variant ? &copt : nullptr, F(L.ssd), nullptr, ws + L.corr_ws, cws, stream))) return rc;
    mark("correlate", s);
    if (no_prune) rc = launch_argmin(F(L.ssd), f16, nullptr, nullptr, 0.0f, false, L.K, L.v, keys, am, s);
    else rc = launch_argmin_keys(F(L.ssd), f16, L.K, L.v, keys, /*arm=*/false, s);            // keys stay in the coupled workspace's first buffer
    if (rc) return rc;
    mark("argmin", s);
    if (p->ic) {                                // reverse direction (:136-138): same operators with the roles swapped
        if ((rc = cvx_correlate_ex_f32(F(L.ms), F(L.fs), L.C, L.h, L.w, L.d, p->disp_hw, variant ? &copt : nullptr, F(L.ssd2), nullptr, ws + L.corr_ws, cws, stream))) return rc;
        mark("correlate_rev", s);
        if (no_prune) rc = launch_argmin(F(L.ssd2), f16, nullptr, nullptr, 0.0f, false, L.K, L.v, keys2, am2, s);
        else rc = launch_argmin_keys(F(L.ssd2), f16, L.K, L.v, keys2, /*arm=*/false, s);
        if (rc) return rc;
        mark("argmin_rev", s);
    }
    }
    // both coupled-convex solves in the same launches (ic) or the forward one alone
    if ((rc = coupled_convex_dual_impl(F(L.ssd), no_prune ? am : nullptr, F(L.soft), ws + L.conv_ws, p->ic ? F(L.ssd2) : nullptr, f16, no_prune ? am2 : nullptr,
                                       p->ic ? F(L.soft2) : nullptr, p->ic ? ws + L.conv_ws2 : nullptr, F(L.mesh), L.h, L.w, L.d,
                                       p->disp_hw, vws, stream, /*counts_zeroed=*/!no_prune))) return rc;
    }
    mark("coupled_convex", s);

    const float* disp_hr = F(L.soft);          // ic=False: coarse field, coarse units (:143-144)
    const float* coarse_src = nullptr;         // ic=True + Adam: the coarse field whose up-sampling is folded into the next resize
    int hh = L.h, hw_ = L.w, hd = L.d;
    if (p->ic) {                                // (:133-141)
        const dim3 gv((unsigned)cdiv64((int64_t)L.v, 256));
        hipLaunchKernelGGL(k_ic_prepare, dim3(gv.x, 2), dim3(256), 0, s, F(L.soft), F(L.soft2), L.h, L.w, L.d, F(L.in1), F(L.in2));
        if ((rc = cvx_inverse_consistency_f32(F(L.in1), F(L.in2), L.h, L.w, L.d, 15, F(L.bh), F(L.bw), F(L.bd), F(L.ic1), F(L.ic2),
                                              ws + L.ic_ws, cvx_inverse_consistency_workspace_bytes(L.h, L.w, L.d), stream))) return rc;
        hipLaunchKernelGGL(k_ic_finish, gv, dim3(256), 0, s, F(L.ic1), L.h, L.w, L.d, (float)p->grid_sp, F(L.upin));
        // with the Adam stage following, disp_hr is only ever read by the down-sampling to the Adam grid: the two resizes are
        // folded into one there (launch_resize2) and the full-resolution field is never written
        if (p->lambda_weight > 0) { coarse_src = F(L.upin); }
        else {
            if ((rc = launch_resize(F(L.upin), 3, L.h, L.w, L.d, out_field, p->H, p->W, p->D, 1.0f, 1.0f, s))) return rc;
            disp_hr = out_field;
        }
        hh = p->H; hw_ = p->W; hd = p->D;
        mark("inverse_consistency", s);
    }

    if (p->lambda_weight > 0) {                 // (:147-191)
        if (!pooled_mind) {
            if ((rc = cvx_avgpool_f32(featF, L.C, p->H, p->W, p->D, p->grid_sp_adam, F(L.F2), stream))) return rc;
            if ((rc = cvx_avgpool_f32(featM, L.C, p->H, p->W, p->D, p->grid_sp_adam, F(L.M2), stream))) return rc;
        }
        // disp_lr = interpolate(disp_hr, (H2,W2,D2)); weight = disp_lr / grid_sp_adam       (:153,156)
        if (coarse_src) {
            if ((rc = launch_resize2(coarse_src, 3, L.h, L.w, L.d, hh, hw_, hd, F(L.disp_hr), F(L.P), L.h2, L.w2, L.d2, (float)p->grid_sp_adam, s))) return rc;
        } else if ((rc = launch_resize(disp_hr, 3, hh, hw_, hd, F(L.P), L.h2, L.w2, L.d2, 1.0f, (float)p->grid_sp_adam, s))) return rc;
        // (m = v = 0: k_pair_setup)
        // (fp16 storage: the Adam loop keeps its feature records in half precision -- rounded when the records are built)
        mark("adam_setup", s);
        const cvx_smoother two_pools = {0, 2, {3, 3, 0, 0}, {0.f, 0.f, 0.f, 0.f, 0.f}};            // task3_docker.py:191
        if ((rc = adam_run_impl(F(L.F2), F(L.M2), L.C, L.h2, L.w2, L.d2, F(L.P), F(L.m), F(L.v_), p->lambda_weight,
                                p->selected_niter, 0, p->cost_scale, F(L.bh2), F(L.bw2), F(L.bd2), F(L.U), nullptr, snap_iters_host, n_snap,
                                n_snap ? F(L.snaps) : nullptr, p->n_spline_pools == 2 ? &two_pools : nullptr, /*keep_state=*/false, f16, p->adam_fast, ws + L.adam_ws,
                                cvx_adam_workspace_bytes(L.C, L.h2, L.w2, L.d2), stream, mind_records))) return rc;
        mark("adam", s);
        // disp_hr = interpolate(fitted_grid * grid_sp_adam, (H,W,D))                            (:182)
        if (n_snap > 0) {                       // self_configuring/convex_adam_MIND.py:115-139: every snapshot x every final smoothing
            float* tmp = F(L.smooth_ws);
            float* tmp2 = L.smooth_ws ? reinterpret_cast<float*>(ws + align_up(L.smooth_ws + sizeof(float) * 3 * L.V, 256)) : nullptr;
            for (int i = 0; i < n_snap; ++i)
                for (int j = 0; j < n_smooth; ++j) {
                    float* dst = out_field + ((size_t)i * n_smooth + j) * 3 * L.V;
                    const float* snap = F(L.snaps) + (size_t)i * 3 * L.V2;
                    const int k = smooth_host[j];
                    if (k > 0) {
                        if ((rc = launch_resize(snap, 3, L.h2, L.w2, L.d2, tmp, p->H, p->W, p->D, (float)p->grid_sp_adam, 1.0f, s))) return rc;
                        if ((rc = launch_box_zero(tmp, tmp2, 3, p->H, p->W, p->D, k, false, s))) return rc;
                        if ((rc = launch_box_zero(tmp2, tmp, 3, p->H, p->W, p->D, k, false, s))) return rc;
                        if ((rc = launch_box_zero(tmp, dst, 3, p->H, p->W, p->D, k, false, s))) return rc;
                    } else if ((rc = launch_resize(snap, 3, L.h2, L.w2, L.d2, dst, p->H, p->W, p->D, (float)p->grid_sp_adam, 1.0f, s))) return rc;
                }
        } else if (p->selected_smooth > 0) {
            float* tmp = F(L.smooth_ws);
            float* tmp2 = reinterpret_cast<float*>(ws + align_up(L.smooth_ws + sizeof(float) * 3 * L.V, 256));
            if ((rc = launch_resize(F(L.U), 3, L.h2, L.w2, L.d2, tmp, p->H, p->W, p->D, (float)p->grid_sp_adam, 1.0f, s))) return rc;
            if ((rc = launch_box_zero(tmp, tmp2, 3, p->H, p->W, p->D, p->selected_smooth, false, s))) return rc;
            if ((rc = launch_box_zero(tmp2, tmp, 3, p->H, p->W, p->D, p->selected_smooth, false, s))) return rc;
            if ((rc = launch_box_zero(tmp, out_field, 3, p->H, p->W, p->D, p->selected_smooth, false, s))) return rc;
        } else {
            if ((rc = launch_resize(F(L.U), 3, L.h2, L.w2, L.d2, out_field, p->H, p->W, p->D, (float)p->grid_sp_adam, 1.0f, s))) return rc;
        }
        hh = p->H; hw_ = p->W; hd = p->D;
        mark("upsample", s);
    } else if (!p->ic) {
        (void)hipMemcpyAsync(out_field, F(L.soft), sizeof(float) * 3 * L.v, hipMemcpyDeviceToDevice, s);
    }
    if (out_dims_host) { out_dims_host[0] = hh; out_dims_host[1] = hw_; out_dims_host[2] = hd; }
    return check_last("register_pair");
}


// ---- several pairs at once on internal streams ------------------------------------------------------------
// The kernels of one pair are chained by data dependencies and leave issue slots idle (short grids, tails);
// independent pairs fill them.  Pairs are dealt round-robin onto `n_streams` internal non-blocking streams
// that fork from / join back into the caller's stream with events, so the call is still asynchronous and
// ordered with respect to the caller's stream.  workspace = n_streams x cvx_register_pair_workspace_bytes().
namespace cvx {
struct StreamPool {
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    hipEvent_t fork = nullptr;
    int device = -1;
};
static thread_local StreamPool g_pool_streams;
static int ensure_streams(int n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    StreamPool& P = g_pool_streams;
    if (P.device != dev) { P.streams.clear(); P.done.clear(); P.fork = nullptr; P.device = dev; }   // leaked on device switch (rare)
    if (!P.fork && hipEventCreateWithFlags(&P.fork, hipEventDisableTiming) != hipSuccess) return fail(CVX_ERR_LAUNCH, "event create failed");
    while ((int)P.streams.size() < n) {
        hipStream_t s; hipEvent_t e;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail(CVX_ERR_LAUNCH, "stream create failed");
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(CVX_ERR_LAUNCH, "event create failed");
        P.streams.push_back(s); P.done.push_back(e);
    }
    return CVX_OK;
}
}  // namespace cvx

extern "C" int cvx_register_pairs_f32(int n_pairs, const float* const* img_fixed, const float* const* img_moving,
                                      const float* const* feat_fixed, const float* const* feat_moving,
                                      const cvx_pair_params* p, float* const* out_fields, int* out_dims_host, void* workspace,
                                      size_t workspace_bytes, int n_streams, void* stream) {
    CVX_REQUIRE(n_pairs >= 1 && out_fields && workspace, "cvx_register_pairs_f32: bad arguments");
    CVX_REQUIRE(n_streams >= 1 && n_streams <= 8, "cvx_register_pairs_f32: n_streams must be 1..8");
    if (n_streams > n_pairs) n_streams = n_pairs;
    const size_t per = cvx_register_pair_workspace_bytes(p);
    if (per == 0) return CVX_ERR_INVALID_ARG;
    const size_t per_al = align_up(per, 4096);
    if (workspace_bytes < per_al * (size_t)n_streams) return fail(CVX_ERR_WORKSPACE, "cvx_register_pairs_f32: workspace %zu < %zu", workspace_bytes, per_al * (size_t)n_streams);
    int rc = ensure_streams(n_streams);
    if (rc) return rc;
    StreamPool& P = g_pool_streams;
    hipStream_t user = as_stream(stream);
    const int saved_prof = g_profiling;
    g_profiling = 0;                                  // stage marks are per stream; not recorded in batch mode
    (void)hipEventRecord(P.fork, user);
    for (int s = 0; s < n_streams; ++s) (void)hipStreamWaitEvent(P.streams[s], P.fork, 0);
    for (int i = 0; i < n_pairs && rc == CVX_OK; ++i) {
        const int s = i % n_streams;
        g_side_slot = s;
        rc = cvx_register_pair_f32(img_fixed ? img_fixed[i] : nullptr, img_moving ? img_moving[i] : nullptr,
                                   feat_fixed ? feat_fixed[i] : nullptr, feat_moving ? feat_moving[i] : nullptr, p, out_fields[i],
                                   out_dims_host, static_cast<char*>(workspace) + per_al * (size_t)s, per, P.streams[s]);
    }
    for (int s = 0; s < n_streams; ++s) {
        (void)hipEventRecord(P.done[s], P.streams[s]);
        (void)hipStreamWaitEvent(user, P.done[s], 0);
    }
    g_profiling = saved_prof;
    g_side_slot = 0;
    return rc;
}
