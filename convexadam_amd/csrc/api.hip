// api.hip -- error plumbing, version/device queries and the host-side table helpers of the C ABI.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "cvx_common.h"

namespace cvx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int check_last(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CVX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return CVX_OK;
}

static long long env_ll(const char* name, long long dflt) {
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}

// ---- contexts: variant switches + reference-build tables, per caller ------------------------------------------------------------
// Every entry point reads its switches and tables from the context bound to the CALLING THREAD (cvx_context_bind; the whole-pair
// entry points also accept one in cvx_pair_params.ctx) and passes them to its kernels by value at enqueue time, so two threads
// driving two streams with different settings never see each other's state.  No context bound = the process default context
// (what cvx_set_option and the legacy table setters modify).  Tables are COPIED into device memory the context owns: the caller
// may free its buffer as soon as the setter returns, and the copy lives until the context is destroyed or the table replaced,
// both of which wait for the device first (enqueued work keeps a valid table).
}  // namespace cvx
struct cvx_context {
    cvx::Options opt;
    unsigned* sqrt_tbl = nullptr;              // owned, 6 MiB (nullptr: IEEE sqrt)
    unsigned char* exp_tbl = nullptr;          // owned (nullptr: the library's expf)
    unsigned exp_first = 0, exp_count = 0;
    int tbl_device = -1;                       // device the owned tables live on
};
namespace cvx {
static Options env_options() {
    return {env_ll("CVX_MIND_TILED", 0),   env_ll("CVX_MIND_OVERLAP", 0),  env_ll("CVX_MM_TX", 0),        env_ll("CVX_MM_SLOTS", 512),          env_ll("CVX_BOX_TILED", 0),
            env_ll("CVX_NO_PRUNE", 0),     env_ll("CVX_CORR_UNFUSED", 0), env_ll("CVX_CORR_FUSED_ALL", 0), env_ll("CVX_PRUNE_STREAM_ABOVE", -1), env_ll("CVX_CF_CENSUS", 0), env_ll("CVX_CF_PRIO", 136),
            env_ll("CVX_WARP_FLAT", 0),    env_ll("CVX_BOX_YT", 8),       env_ll("CVX_BOX_WG_TARGET", 0),
            env_ll("CVX_BOX_XSPLIT", -1),  env_ll("CVX_BOX_CPT", 4),      env_ll("CVX_BOX_UNEVEN", 200), env_ll("CVX_BOX_ADAM_ROLE", 0), env_ll("CVX_BOX_DPP", 0),      env_ll("CVX_BOX_PK", 0),       env_ll("CVX_BOX_PRIO", 0),     env_ll("CVX_LABEL_POW_BLOCK", 32), 0,                             env_ll("CVX_MIND_MEAN_THREADS", 0),
            env_ll("CVX_EDT_SEQUENTIAL", 0), env_ll("CVX_WARP_OCTANT", 4), env_ll("CVX_BOX_FWD_TILE", -1), env_ll("CVX_BOX_BWD_TILE", -1), env_ll("CVX_BOX_WALK", 1), env_ll("CVX_CORR_DUAL", 0), env_ll("CVX_PRUNE_REFINE", 1), env_ll("CVX_MIND_RECORDS", 1), env_ll("CVX_RESIZE_UP2", 1), env_ll("CVX_MIND_BLOCKED", 1), env_ll("CVX_CORR_CERT", 1), env_ll("CVX_CC_DEBUG", 0), env_ll("CVX_IC_FUSED", 0), env_ll("CVX_MIND_SINGLE", 0), env_ll("CVX_MS_ZLEN", 0), env_ll("CVX_CF_MAP", 1), env_ll("CVX_CERT_UNFUSED", 0), env_ll("CVX_FBOX_TILE", 0)};
}
static cvx_context& default_context() {
    static cvx_context c = [] { cvx_context d; d.opt = env_options(); return d; }();
    return c;
}
static thread_local cvx_context* t_bound = nullptr;
static cvx_context& current_context() { return t_bound ? *t_bound : default_context(); }
Options& options() { return current_context().opt; }
const unsigned* adam_sqrt_table() { return current_context().sqrt_tbl; }
ExpTable mind_exp_table() {
    const cvx_context& c = current_context();
    return {c.exp_tbl, c.exp_first, c.exp_tbl ? c.exp_count : 0u};
}
ContextScope::ContextScope(const cvx_context* c) : prev_(t_bound), active_(c != nullptr) {
    if (active_) t_bound = const_cast<cvx_context*>(c);
}
ContextScope::~ContextScope() {
    if (active_) t_bound = static_cast<cvx_context*>(prev_);
}

static void drop_table(cvx_context& c, void** slot) {
    if (!*slot) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (c.tbl_device >= 0 && c.tbl_device != cur) (void)hipSetDevice(c.tbl_device);
    (void)hipDeviceSynchronize();              // work already enqueued may still read the table
    (void)hipFree(*slot);
    if (c.tbl_device >= 0 && c.tbl_device != cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    *slot = nullptr;
}
// device copy of `bytes` bytes at `src` (device memory of the current device) owned by the context
// (the new copy is allocated and filled BEFORE the old table is released: a failed setter leaves the installed table in place)
static int adopt_table(cvx_context& c, const void* src, size_t bytes, void** slot, hipStream_t s, const char* what) {
    if (!src) { drop_table(c, slot); return CVX_OK; }
    int cur = 0;
    (void)hipGetDevice(&cur);
    const bool other_table = (slot == reinterpret_cast<void**>(&c.sqrt_tbl)) ? c.exp_tbl != nullptr : c.sqrt_tbl != nullptr;
    if (c.tbl_device >= 0 && c.tbl_device != cur && other_table)
        return fail(CVX_ERR_INVALID_ARG, "%s: the context already holds a table on device %d (current device %d)", what, c.tbl_device, cur);
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(CVX_ERR_LAUNCH, "%s: cannot allocate %zu bytes for the table copy", what, bytes); }
    if (hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(p);
        return fail(CVX_ERR_LAUNCH, "%s: table copy failed", what);
    }
    drop_table(c, slot);                       // waits for enqueued work that still reads the old table
    c.tbl_device = cur;
    *slot = p;
    return CVX_OK;
}
constexpr size_t kSqrtTableBytes = ((size_t)(1u << 24) + (1u << 23)) / 4;     // two bits per class

__global__ __launch_bounds__(256) void k_expf(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = cvx_expf(x[i]);
}
struct OptName { const char* name; long long Options::*field; };
static const OptName kOptNames[] = {{"mind_tiled", &Options::mind_tiled},     {"mind_overlap", &Options::mind_overlap},   {"mm_tx", &Options::mm_tx},
                                    {"mm_slots", &Options::mm_slots},         {"box_tiled", &Options::box_tiled},
                                    {"no_prune", &Options::no_prune},         {"corr_unfused", &Options::corr_unfused}, {"corr_fused_all", &Options::corr_fused_all},
                                    {"prune_stream_above", &Options::prune_stream_above}, {"cf_census", &Options::cf_census}, {"cf_prio", &Options::cf_prio},
                                    {"warp_flat", &Options::warp_flat},       {"box_yt", &Options::box_yt},             {"box_wg_target", &Options::box_wg_target},
                                    {"box_xsplit", &Options::box_xsplit},     {"box_cpt", &Options::box_cpt},           {"box_uneven", &Options::box_uneven},     {"box_adam_role", &Options::box_adam_role}, {"box_dpp", &Options::box_dpp},           {"box_pk", &Options::box_pk},             {"box_prio", &Options::box_prio},         {"label_pow_block", &Options::label_pow_block}, {"census_ptr", &Options::census_ptr},     {"mind_mean_threads", &Options::mind_mean_threads},
                                    {"edt_sequential", &Options::edt_sequential}, {"warp_octant", &Options::warp_octant}, {"box_fwd_tile", &Options::box_fwd_tile}, {"box_bwd_tile", &Options::box_bwd_tile}, {"box_walk", &Options::box_walk}, {"corr_dual", &Options::corr_dual}, {"prune_refine", &Options::prune_refine}, {"mind_records", &Options::mind_records}, {"resize_up2", &Options::resize_up2}, {"mind_blocked", &Options::mind_blocked}, {"corr_cert", &Options::corr_cert}, {"cc_debug", &Options::cc_debug}, {"ic_fused", &Options::ic_fused}, {"mind_single", &Options::mind_single}, {"ms_zlen", &Options::ms_zlen}, {"cf_map", &Options::cf_map}, {"cert_unfused", &Options::cert_unfused}, {"fbox_tile", &Options::fbox_tile}};

}  // namespace cvx

static int set_option_of(cvx_context& c, const char* name, long long value) {
    for (const auto& o : cvx::kOptNames)
        if (name && strcmp(name, o.name) == 0) {
            if (strcmp(name, "box_xsplit") == 0 && value > 32) return cvx::fail(CVX_ERR_INVALID_ARG, "box_xsplit %lld: at most 32 x tiles", value);
            c.opt.*(o.field) = value;
            return CVX_OK;
        }
    return cvx::fail(CVX_ERR_INVALID_ARG, "cvx_set_option: unknown option '%s'", name ? name : "(null)");
}
static long long get_option_of(const cvx_context& c, const char* name) {
    for (const auto& o : cvx::kOptNames)
        if (name && strcmp(name, o.name) == 0) return c.opt.*(o.field);
    return -1;
}

extern "C" cvx_context* cvx_context_create(void) {
    cvx_context* c = new (std::nothrow) cvx_context();
    if (!c) { cvx::set_error("cvx_context_create: out of memory"); return nullptr; }
    c->opt = cvx::default_context().opt;
    return c;
}
extern "C" void cvx_context_destroy(cvx_context* c) {
    if (!c) return;
    if (cvx::t_bound == c) cvx::t_bound = nullptr;
    cvx::drop_table(*c, reinterpret_cast<void**>(&c->sqrt_tbl));
    cvx::drop_table(*c, reinterpret_cast<void**>(&c->exp_tbl));
    delete c;
}
extern "C" cvx_context* cvx_context_bind(cvx_context* c) {
    cvx_context* prev = cvx::t_bound;
    cvx::t_bound = c;
    return prev;
}
extern "C" int cvx_context_set_option(cvx_context* c, const char* name, long long value) {
    return set_option_of(c ? *c : cvx::default_context(), name, value);
}
extern "C" long long cvx_context_get_option(const cvx_context* c, const char* name) {
    return get_option_of(c ? *c : cvx::default_context(), name);
}
extern "C" int cvx_context_set_adam_sqrt_table(cvx_context* c, const void* device_table, void* stream) {
    cvx_context& x = c ? *c : cvx::default_context();
    return cvx::adopt_table(x, device_table, cvx::kSqrtTableBytes, reinterpret_cast<void**>(&x.sqrt_tbl), cvx::as_stream(stream), "cvx_context_set_adam_sqrt_table");
}
extern "C" int cvx_context_set_mind_exp_table(cvx_context* c, const void* device_table, unsigned first_key, unsigned count, void* stream) {
    cvx_context& x = c ? *c : cvx::default_context();
    if (device_table && count == 0) return cvx::fail(CVX_ERR_INVALID_ARG, "cvx_context_set_mind_exp_table: empty table");
    const int rc = cvx::adopt_table(x, device_table, ((size_t)count + 3) / 4, reinterpret_cast<void**>(&x.exp_tbl), cvx::as_stream(stream), "cvx_context_set_mind_exp_table");
    x.exp_first = device_table && rc == CVX_OK ? first_key : 0u;
    x.exp_count = device_table && rc == CVX_OK ? count : 0u;
    return rc;
}

// process default context (also what a thread without a bound context uses)
extern "C" int cvx_set_option(const char* name, long long value) { return set_option_of(cvx::default_context(), name, value); }
extern "C" long long cvx_get_option(const char* name) { return get_option_of(cvx::default_context(), name); }
extern "C" int cvx_set_adam_sqrt_table(const void* device_table) { return cvx_context_set_adam_sqrt_table(nullptr, device_table, nullptr); }
extern "C" int cvx_set_mind_exp_table(const void* device_table, unsigned first_key, unsigned count) {
    return cvx_context_set_mind_exp_table(nullptr, device_table, first_key, count, nullptr);
}
extern "C" int cvx_expf_f32(const float* x, float* out, size_t n, void* stream) {
    if (n == 0) return CVX_OK;
    if (!x || !out) return cvx::fail(CVX_ERR_INVALID_ARG, "cvx_expf_f32: null pointer");
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cvx::k_expf, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, cvx::as_stream(stream), x, out, n);
    return cvx::check_last("expf");
}

// 2: cvx_pair_params grew (ctx in round 3, adam_fast / corr_verify / struct_size in round 4) -- bindings compiled against an older header
// must not call the whole-pair entry points; they can assert on this number
extern "C" int cvx_version(void) { return CVX_ABI_VERSION; }
extern "C" const char* cvx_last_error(void) { return cvx::g_err; }
extern "C" int cvx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// torch.linspace(-1, 1, S) in float32: step = 2/(S-1); lower half start + step*i, upper half
// end - step*(S-1-i), each with a single (fused) rounding.  Checked against torch for S = 2..399.
static void linspace_pm1(int S, float* out) {
    if (S == 1) { out[0] = -1.0f; return; }
    const float step = (1.0f - (-1.0f)) / (float)(S - 1);
    const int half = S / 2;
    for (int i = 0; i < S; ++i)
        out[i] = (i < half) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(S - 1 - i), 1.0f);
}
extern "C" void cvx_affine_base_host(int S, float* out_host) {
    linspace_pm1(S, out_host);
    for (int i = 0; i < S; ++i) out_host[i] = (out_host[i] * (float)(S - 1)) / (float)S;
}
extern "C" void cvx_disp_mesh_host(int disp_hw, float* out_host) {
    const int n = 2 * disp_hw + 1;
    float lin[1024];
    if (n == 1) lin[0] = 0.0f;
    else linspace_pm1(n, lin);
    const size_t K = (size_t)n * n * n;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            for (int c = 0; c < n; ++c) {
                const size_t k = ((size_t)a * n + b) * n + c;
                out_host[0 * K + k] = lin[c] * (float)disp_hw;   // axis 0 (H) shift is the fastest index of k
                out_host[1 * K + k] = lin[b] * (float)disp_hw;
                out_host[2 * K + k] = lin[a] * (float)disp_hw;
            }
}
