// api.hip -- error plumbing, version/device queries and the host-side table helpers of the C ABI.
#include <math.h>
#include <stdlib.h>
#include <string.h>

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
Options& options() {
    static Options o = {env_ll("CVX_MIND_TILED", 0),   env_ll("CVX_MM_TX", 0),        env_ll("CVX_MM_SLOTS", 512),          env_ll("CVX_BOX_TILED", 0),
                        env_ll("CVX_NO_PRUNE", 0),     env_ll("CVX_CORR_UNFUSED", 0), env_ll("CVX_PRUNE_STREAM_ABOVE", -1), env_ll("CVX_CF_CENSUS", 0),
                        env_ll("CVX_WARP_FLAT", 0),    env_ll("CVX_BOX_YT", 8),       env_ll("CVX_BOX_WG_TARGET", 0),
                        env_ll("CVX_BOX_XSPLIT", -1),  env_ll("CVX_MIND_MEAN_THREADS", 0)};
    return o;
}
static const unsigned* g_adam_sqrt_tbl = nullptr;
const unsigned* adam_sqrt_table() { return g_adam_sqrt_tbl; }
static ExpTable g_mind_exp_tbl = {nullptr, 0u, 0u};
ExpTable mind_exp_table() { return g_mind_exp_tbl; }

__global__ __launch_bounds__(256) void k_expf(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = cvx_expf(x[i]);
}
struct OptName { const char* name; long long Options::*field; };
static const OptName kOptNames[] = {{"mind_tiled", &Options::mind_tiled},     {"mm_tx", &Options::mm_tx},
                                    {"mm_slots", &Options::mm_slots},         {"box_tiled", &Options::box_tiled},
                                    {"no_prune", &Options::no_prune},         {"corr_unfused", &Options::corr_unfused},
                                    {"prune_stream_above", &Options::prune_stream_above}, {"cf_census", &Options::cf_census},
                                    {"warp_flat", &Options::warp_flat},       {"box_yt", &Options::box_yt},             {"box_wg_target", &Options::box_wg_target},
                                    {"box_xsplit", &Options::box_xsplit},     {"mind_mean_threads", &Options::mind_mean_threads}};

}  // namespace cvx

extern "C" int cvx_set_option(const char* name, long long value) {
    for (const auto& o : cvx::kOptNames)
        if (name && strcmp(name, o.name) == 0) { cvx::options().*(o.field) = value; return CVX_OK; }
    return cvx::fail(CVX_ERR_INVALID_ARG, "cvx_set_option: unknown option '%s'", name ? name : "(null)");
}
extern "C" long long cvx_get_option(const char* name) {
    for (const auto& o : cvx::kOptNames)
        if (name && strcmp(name, o.name) == 0) return cvx::options().*(o.field);
    return -1;
}

extern "C" int cvx_set_adam_sqrt_table(const void* device_table) {
    cvx::g_adam_sqrt_tbl = static_cast<const unsigned*>(device_table);
    return CVX_OK;
}

extern "C" int cvx_set_mind_exp_table(const void* device_table, unsigned first_key, unsigned count) {
    cvx::g_mind_exp_tbl = {static_cast<const unsigned char*>(device_table), first_key, device_table ? count : 0u};
    return CVX_OK;
}
extern "C" int cvx_expf_f32(const float* x, float* out, size_t n, void* stream) {
    if (n == 0) return CVX_OK;
    if (!x || !out) return cvx::fail(CVX_ERR_INVALID_ARG, "cvx_expf_f32: null pointer");
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cvx::k_expf, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, cvx::as_stream(stream), x, out, n);
    return cvx::check_last("expf");
}

extern "C" int cvx_version(void) { return 1000 * 0 + 1; }
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
