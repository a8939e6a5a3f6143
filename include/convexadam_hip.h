/*
 * convexadam_hip.h -- C ABI of libconvexadam_hip.so, the MI355X (gfx950) engine behind the
 * convexAdam registration hot path.
 *
 * The upstream reference (multimodallearning/convexAdam) has no FFI: its hot path is a set of
 * Python functions that call ATen operators in-process.  This header declares the entry points a
 * binding for that path needs, one per reference operator, and cites the reference interface each
 * one replaces (paths relative to the reference tree).  The Python side of this repository
 * (package convexadam_amd) binds them with ctypes; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, current device) unless the name ends in `_host`;
 *   - volumes are dense float32, layout [C][H][W][D] with D fastest (torch contiguous NCDHW, N=1);
 *   - displacement fields are [3][H][W][D]; channel a = displacement along array axis a, in voxels
 *     of the grid the field lives on; fixed(x) ~ moving(x + u(x))   (apply_convex.py:22-23);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls enqueue work and
 *     return without synchronising; nothing is allocated: the caller supplies outputs and a
 *     scratch workspace whose size comes from the matching *_workspace_bytes() query;
 *   - return value 0 = ok, negative = CVX_ERR_*; cvx_last_error() gives a thread-local message;
 *   - floating-point evaluation order follows the ATen CPU kernels the reference calls, so results
 *     are reproducible bit-for-bit against the CPU oracle (oracle/cvx_oracle.c);
 *   - state kept by the library: the thread-local error message, the optional per-thread stage timing (cvx_set_profiling), the
 *     per-thread internal stream pool of the whole-pair entry points, the per-thread context binding and the default context
 *     (see "State model" below) -- nothing else; entry points are re-entrant per (thread, stream).
 */
#ifndef CONVEXADAM_HIP_H
#define CONVEXADAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* all entry points have default visibility; everything else in the library is hidden */
#pragma GCC visibility push(default)

#define CVX_OK 0
#define CVX_ERR_INVALID_ARG (-1)
#define CVX_ERR_WORKSPACE (-2)
#define CVX_ERR_LAUNCH (-3)
#define CVX_ERR_UNSUPPORTED (-4)
#define CVX_MAX_DISP_HW 15     /* largest search half-width (n = 31, 29 791 displacements); the reference itself has no limit */

/* library / device ------------------------------------------------------------------------------ */
/* ABI version of THIS header.  cvx_version() returns the number the loaded library was built with; a binding must check
 * cvx_version() == CVX_ABI_VERSION before it calls an entry point that takes a struct (cvx_pair_params grew in versions 1 -> 2:
 * a struct laid out by an older header is shorter than what the library reads). */
#define CVX_ABI_VERSION 2
int cvx_version(void);
const char* cvx_last_error(void);      /* message of the last failing call on this thread */
int cvx_device_count(void);            /* number of visible HIP devices (0 on a CPU-only host) */

/* Run-time switches between kernel variants (A/B timing, variant coverage in the tests).  Every variant is bit-identical except
 * "mind_mean_threads", which changes MINDSSC's clamp bounds by ulps (reference-bits mode).  Names:
 *   mind_tiled          1: tiled MIND stencil instead of the z-marching one
 *   mind_overlap        1: the whole-pair pipeline computes the two images' descriptors side by side on two streams (default: in order)
 *   mm_tx, mm_slots     tile width (32 / 64, 0 = automatic) and workgroup budget (512) of the marching MIND stencil
 *   box_tiled           1: tiled three-box kernels in the Adam loop instead of the z-marching ones
 *   box_yt              rows per tile of the marching three-box kernels: 8 (default) or 4
 *   box_wg_target       workgroups the marching three-box kernels aim for (z-chunk length follows); 0 = automatic
 *   box_cpt, box_pk, box_dpp, box_prio, box_uneven   measured variants of the marching three-box kernels (2 columns per thread; packed
 *                       running sums; halo columns through DPP lane shifts; box_adam_role: the Adam update as a role of its own; alternating issue priority; z-chunk length ratio between the two dispatch rounds, default 200)
 *   box_xsplit          x tiles of the marching three-box kernels: -1 automatic, 0 off, 2..32 that many (ignored when a tile would be empty)
 *   warp_flat           1: flat 64-bit gathers in the warp kernel instead of buffer loads
 *   no_prune            1: streaming coupled-convex passes instead of branch and bound
 *   prune_stream_above  chunk budget above which a pruned pass scans the volume (-1 = automatic)
 *   corr_unfused        1: separate raw-SSD and box kernels instead of the fused correlation kernel
 *   cf_prio             issue priorities of the fused correlation kernel (s_setprio 0..3), four base-4 digits: first-round workgroup raw / box,
 *                       second-round workgroup raw / box; 136 = 2,0,2,0 (default; the other settings measured equal or slower)
 *   corr_fused_all      1: the fused correlation kernel also for C >= 16 (it covers them -- cascade channel sum -- but the separate
 *                       kernels are faster there and stay the default)
 *   cf_census           1: the fused correlation kernel records per-workgroup residency in its workspace
 *   label_pow_block     cvx_label_weights_host: elements per vectorised block of the reference host's torch.pow (32 = AVX-512 build, 16 = AVX2)
 *   mind_mean_threads   0: exactly rounded global mean in MINDSSC; T > 0: torch's float32 sum with T threads
 *   corr_cert           1 (default) / 2: cvx_register_pair(s)_f32 take the argmin decisions of the convex stage on the certified-fast cost volume
 *                       (cvx_corr_opts.fast = 2; same winners and field bits as the exact volumes; below 16 channels); 1 = the role kernel of
 *                       corrfused.hip, 2 = the staged kernel of corrcert.hip (every supported channel count); 0 = exact volumes
 *   ic_fused            1: inverse consistency in one launch (measured slower, off by default; bit-identical)
 * Workspace sizes (cvx_*_workspace_bytes) depend on some switches: query them with the same context / options the call will use.
 *
 * State model.  Switches and the two reference-build tables below live in a CONTEXT.  Every entry point uses the context bound to the
 * calling thread (cvx_context_bind), the whole-pair entry points alternatively the one named in cvx_pair_params.ctx; settings are
 * read when a call enqueues its kernels and travel with them, so threads that drive different streams with different contexts are
 * independent.  A thread without a bound context uses the process default context, whose initial switch values come from CVX_<NAME>
 * environment variables; cvx_set_option / cvx_get_option and the legacy table setters address that default context (set them while no
 * other thread is inside the library).  cvx_get_option returns -1 for an unknown name.
 *   cvx_context_create     new context; switches start as a copy of the default context's, no tables
 *   cvx_context_destroy    waits for the device, frees the context and the table copies it owns.  LIFETIME: the caller must unbind the
 *                          context on EVERY thread that bound it (cvx_context_bind(NULL) there) and must not have a call in flight whose
 *                          cvx_pair_params.ctx names it; only the calling thread's own binding is cleared here.  A table setter that
 *                          fails leaves the previously installed table in place
 *   cvx_context_bind       binds ctx (NULL = default) to the calling thread, returns the previously bound one
 *   cvx_context_set_*      ctx == NULL addresses the default context */
typedef struct cvx_context cvx_context;
cvx_context* cvx_context_create(void);
void cvx_context_destroy(cvx_context* ctx);
cvx_context* cvx_context_bind(cvx_context* ctx);
int cvx_context_set_option(cvx_context* ctx, const char* name, long long value);
long long cvx_context_get_option(const cvx_context* ctx, const char* name);
int cvx_set_option(const char* name, long long value);
long long cvx_get_option(const char* name);

/* host helpers (exact restatements of torch.linspace / affine_grid tables) ----------------------- */
/* F.affine_grid(eye, size S, align_corners=False) identity coordinate along an axis of extent S.
 * replaces: convex_adam_utils.py:121, convex_adam_MIND.py:160 */
void cvx_affine_base_host(int S, float* out_host);
/* search mesh of convex_adam_MIND.py:127: out_host[3][n^3], n = 2*disp_hw+1,
 * flat k = (dD+hw)*n*n + (dW+hw)*n + (dH+hw), channel a = displacement along axis a. */
void cvx_disp_mesh_host(int disp_hw, float* out_host);
/* the same two tables as the whole-pair pipeline builds them ON THE DEVICE (same float operations as the host helpers: the tables are
 * never uploaded): out_device[3][n^3] resp. out_device[S] */
int cvx_disp_mesh_f32(int disp_hw, float* out_device, void* stream);
int cvx_affine_base_f32(int S, float* out_device, void* stream);

/* MIND-SSC descriptors ---------------------------------------------------------------------------
 * replaces MINDSSC(img, radius, dilation, device)          convex_adam_utils.py:24-68
 *   img [H][W][D] -> out [12][H][W][D] (reference channel order after the permutation at :66)
 *   radius 1..3, dilation 1..4 (CVX_ERR_INVALID_ARG outside; the reference's sweeps draw both from 1..3) */
size_t cvx_mindssc_workspace_bytes(int H, int W, int D, int radius, int dilation);
int cvx_mindssc_f32(const float* img, int H, int W, int D, int radius, int dilation, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* MIND-SSC as the registration consumes it -- only through its stride poolings ------------------------
 * replaces F.avg_pool3d(MINDSSC(img, radius, dilation), g, stride=g) for the two window sizes of a pair
 *                                                            convex_adam_utils.py:24-68 + convex_adam_MIND.py:118-119,149-150
 *   img [H][W][D] -> out1 [12][H/g1][W/g1][D/g1]; g2 > 0: out2 [12][H/g2][W/g2][D/g2] as well (out2 NULL / g2 == 0: one pooling).
 *   Same bits as cvx_mindssc_f32 followed by cvx_avgpool_f32; the full-resolution descriptor is never written.  Radius 1, dilation 2, rows of a
 *   multiple of 4 voxels and the window pairs (6; 2|3|6), (4; 2|4), (2; 2) run in ONE pass over the image (the normalisation with the
 *   unclamped variance inside the stencil kernel, the pooled cells of the few blocks where the variance clamp of :60-62 binds recomputed once
 *   the global mean is known); other settings take two passes through `scratch` (cvx_mindssc_pooled_scratch_bytes, may be 0 bytes / NULL
 *   ONLY when that function returns 0).  CVX_ERR_UNSUPPORTED when the windows do not tile (use the two separate operators).
 *   workspace: cvx_mindssc_workspace_bytes.  repaired_host (nullable): receives the number of recomputed blocks (synchronises the stream). */
size_t cvx_mindssc_pooled_scratch_bytes(int H, int W, int D, int radius, int dilation, int g1, int g2);
int cvx_mindssc_pooled_f32(const float* img, int H, int W, int D, int radius, int dilation, int g1, float* out1, int g2, float* out2,
                           void* scratch, size_t scratch_bytes, void* workspace, size_t workspace_bytes, int* repaired_host, void* stream);

/* F.avg_pool3d(x, g, stride=g)                              convex_adam_MIND.py:118-119,149-150
 *   in [C][H][W][D] -> out [C][H/g][W/g][D/g] (floor) */
int cvx_avgpool_f32(const float* in, int C, int H, int W, int D, int g, float* out, void* stream);

/* x = float(half(x)) in place: fp16 storage of a float32 buffer (round to nearest even), n elements */
int cvx_round_f16_f32(float* x, int64_t n, void* stream);

/* F.avg_pool3d(x, k, stride=1, padding=k/2) applied `passes` times (zero pad, divisor k^3)
 *                                                            convex_adam_MIND.py:166,191
 *   in/out [C][H][W][D]; workspace needed when passes > 1 (one volume of the same size) */
size_t cvx_box_smooth_workspace_bytes(int C, int H, int W, int D, int passes);
int cvx_box_smooth_f32(const float* in, int C, int H, int W, int D, int k, int passes, float* out,
                       void* workspace, size_t workspace_bytes, void* stream);
/* the same operator with an EVEN kernel k (ONE pass): padding k/2 on both sides of k taps makes every axis one voxel longer --
 * out [C][H+1][W+1][D+1].  This is what the reference does with an even `selected_smooth` (its "+1" is overwritten,
 * convex_adam_MIND.py:185-189): three such pools, the returned field is (H+3, W+3, D+3, 3).  The whole-pair entries keep refusing an
 * even `selected_smooth` (their output is [3][H][W][D]); the Python mirror runs the pair without smoothing and applies this three times. */
int cvx_box_grow_f32(const float* in, int C, int H, int W, int D, int k, float* out, void* stream);

/* output packing of convex_adam_pt                              convex_adam_MIND.py:198-202
 *   field [3][H][W][D] float32 -> out [H][W][D][3] float64, each value first passed through the caller's `dtype` (quantize 0 = float32,
 *   1 = float16 round to nearest even): `.cpu().to(dtype)` + np.stack(..., 3).astype(float).  out: device memory, or pinned host
 *   memory that is mapped into the device's address space -- the kernel then writes the result straight into the caller's array */
int cvx_pack_field_f64(const float* field, int H, int W, int D, int quantize, double* out, void* stream);

/* masked feature extraction helpers                             convex_adam_MIND.py:36-54
 *   cvx_mask_erode_f32 : out = (AvgPool3d(3)(ReplicationPad3d(1)(mask)) > threshold) ? 1 : 0      (:40,43,48)
 *   cvx_gather_f32     : out[i] = src[index[i]]  (half-resolution nearest-in-mask gather, :45,50; the indices come from
 *                        cvx_feature_transform_i32 + cvx_feature_flat_index_i64 below -- the reference calls scipy on the host)
 *   cvx_select_f32     : out[i] = mask[i] != 0 ? a[i] : b[i]                                       (:46,51) */
int cvx_mask_erode_f32(const float* mask, int H, int W, int D, float threshold, float* out, void* stream);
int cvx_gather_f32(const float* src, const int64_t* index, int64_t n, float* out, void* stream);
int cvx_select_f32(const float* mask, const float* a, const float* b, int64_t n, float* out, void* stream);

/* label-map features                                         convex_adam_nnUNet.py:19-38
 *   lab_* [V] float-valued integer labels in [0, max_label]; weights_host[C] and present_host[C]
 *   are computed on the host by the caller (cvx_label_weights_host) from the two histograms;
 *   feat [C][V] = mult * w_c * (lab == present_c). */
int cvx_label_histogram_i64(const float* lab, int64_t V, int max_label, int64_t* hist /* device [max_label+1] */,
                            void* stream);
int cvx_label_weights_host(const int64_t* hist_fix_host, const int64_t* hist_mov_host, int max_label,
                           int* present_host, float* weights_host); /* returns C */
int cvx_label_features_f32(const float* lab, int64_t V, int C, const int* present /* device [C] */,
                           const float* weights /* device [C] */, float mult, float* feat, void* stream);

/* SSD correlation volume ---------------------------------------------------------------------------
 * replaces correlate(mind_fix, mind_mov, disp_hw, grid_sp, shape, ch)   convex_adam_utils.py:72-89
 *   fix, mov [C][h][w][d] (already pooled to the coarse grid)
 *   ssd [n^3][h][w][d], argmin int64 [h][w][d] (may be NULL) */
size_t cvx_correlate_workspace_bytes(int C, int h, int w, int d, int disp_hw);
int cvx_correlate_f32(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw,
                      float* ssd, int64_t* argmin, void* workspace, size_t workspace_bytes, void* stream);
/* variants of the challenge scripts and the opt-in fast mode; opts == NULL = the packaged operator above
 *   cost  0: sum_c (f - m)^2      convex_adam_utils.py:83            1: sum_c |f - m|   l2r_2021_convexAdam_task3_docker.py:54
 *   n_box 2: two avg_pool3d       convex_adam_utils.py:84            1: one             l2r_2021_convexAdam_task2_docker.py:60, task3:56
 *   fast  0: ATen's evaluation order, bit-identical to the CPU oracle
 *         1: fused multiply-adds and separable box sums (same real-arithmetic result, last-bit differences; cost 0, n_box 2 only)
 *         2: CERTIFIED fast (cost 0, n_box 2, float32, C <= 128; CVX_ERR_UNSUPPORTED elsewhere): `ssd` receives the UNSCALED fast volume
 *            ssdu -- |ssdu / 729 - ssd_exact| <= 2^-17 ssd_exact, ssdu == 0 exactly where ssd_exact == 0 -- and `argmin` the first minimum
 *            of the EXACT volume (torch.argmin(ssd, 0), convex_adam_utils.py:87), decided from intervals and, for the few columns in doubt,
 *            by an exact evaluation of the candidates (certify.hip).  What the whole-pair entry points use internally (switch corr_cert)
 *   f16   fp16 STORAGE of the cost volume (the reference's GPU default dtype, convex_adam_MIND.py:79,89-91; float32 accumulation, one
 *         rounding to nearest even on the way out; SSD with two boxes only):
 *         2: `ssd` points to a HALF-PRECISION buffer [n^3][h][w][d] (2-byte elements) -- half the bytes written here and read by
 *            cvx_coupled_convex_f16; `argmin` is the first minimum of the stored values
 *         1: the same values in a float32 buffer (no byte saved; kept for comparisons) */
typedef struct cvx_corr_opts { int cost, n_box, fast, f16; } cvx_corr_opts;
int cvx_correlate_ex_f32(const float* fix, const float* mov, int C, int h, int w, int d, int disp_hw, const cvx_corr_opts* opts,
                         float* ssd, int64_t* argmin, void* workspace, size_t workspace_bytes, void* stream);

/* coupled convex regularisation ---------------------------------------------------------------------
 * replaces coupled_convex(ssd, ssd_argmin, disp_mesh_t, grid_sp, shape)  convex_adam_utils.py:93-109
 *   mesh [3][n^3] (device); out [3][h][w][d] in coarse-voxel units */
size_t cvx_coupled_convex_workspace_bytes(int h, int w, int d, int disp_hw);
int cvx_coupled_convex_f32(const float* ssd, const int64_t* argmin, const float* mesh, int h, int w, int d,
                           int disp_hw, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* the same solve on a half-precision cost volume (cvx_corr_opts.f16 = 2): values are widened to float32 on load (exact) */
int cvx_coupled_convex_f16(const void* ssd_half, const int64_t* argmin, const float* mesh, int h, int w, int d,
                           int disp_hw, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* inverse consistency -------------------------------------------------------------------------------
 * replaces inverse_consistency(disp_field1s, disp_field2s, iter)        convex_adam_utils.py:114-129
 *   fields [3][h][w][d], channel 0 = normalised displacement along the LAST axis (caller flips,
 *   convex_adam_MIND.py:139); base_* = affine identity tables (device, from cvx_affine_base_host) */
size_t cvx_inverse_consistency_workspace_bytes(int h, int w, int d);
int cvx_inverse_consistency_f32(const float* f1, const float* f2, int h, int w, int d, int iters,
                                const float* base_h, const float* base_w, const float* base_d, float* o1,
                                float* o2, void* workspace, size_t workspace_bytes, void* stream);

/* F.interpolate(x, size, mode='trilinear', align_corners=False)        convex_adam_MIND.py:141,153,182
 *   in [C][h][w][d] -> out [C][H][W][D] */
int cvx_resize_trilinear_f32(const float* in, int C, int h, int w, int d, float* out, int H, int W, int D,
                             void* stream);

/* F.grid_sample(vol, grid, bilinear, zeros, align_corners=False)        convex_adam_utils.py:126,134
 *   vol [C][h][w][d], grid [ho][wo][do][3] (x,y,z normalised) -> out [C][ho][wo][do] */
int cvx_grid_sample_f32(const float* vol, int C, int h, int w, int d, const float* grid, int ho, int wo, int dd,
                        float* out, void* stream);

/* smoothers of the sweep scripts' Adam loop ---------------------------------------------
 * replaces  kovesi_spline(sigma, n)  = chain of zero-padded AvgPool3d(k, stride 1, pad k/2), k in {3,5}
 *                                       self_configuring/convexAdam_hyper_util.py:475-488
 *           GaussianSmoothing(sigma) = separable 5-tap convolution, replicate padding, along H, then W, then D
 *                                       self_configuring/convexAdam_hyper_util.py:423-473
 * (selected by `avg_n` in adam_run_withconfig_shiftSpline.py:140-141,217).  backward = 0 applies the smoother,
 * backward = 1 its adjoint in autograd's evaluation order.  in/out [C][H][W][D], in != out. */
typedef struct cvx_smoother {
    int kind;            /* 0 = box chain, 1 = gaussian */
    int n_boxes;         /* 1..4 */
    int box_k[4];        /* 3 or 5 (any odd size works) */
    float gauss_w[5];    /* normalised taps, as GaussianSmoothing.weight */
} cvx_smoother;
size_t cvx_smooth_workspace_bytes(int C, int H, int W, int D);
int cvx_smooth_f32(const float* in, int C, int H, int W, int D, const cvx_smoother* sm, int backward, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Adam instance optimisation -------------------------------------------------------------------------
 * replaces the loop of convex_adam_MIND.py:155-182 (nn.Conv3d weight + torch.optim.Adam(lr=1)).
 *   F2, M2 [C][h][w][d] pooled features; P [3][h][w][d] control grid (grid units), updated in place;
 *   m, v Adam moments (caller zeroes them for a fresh run); step0 = Adam steps already taken;
 *   cost_scale = 12 in convex_adam_MIND.py:176 (n_ch in the sweep scripts);
 *   U [3][h][w][d] receives disp_sample of the LAST forward pass (what the reference returns, :181);
 *   snapshot_iters (host, ascending, may be NULL): after iteration i (1-based) copy U to
 *   snapshots[j] ([n_snap][3][h][w][d])                    self_configuring/convex_adam_MIND.py:115-139 */
size_t cvx_adam_workspace_bytes(int C, int h, int w, int d);
int cvx_adam_run_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                     float* v, float lambda_weight, int niter, int step0, float cost_scale,
                     const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                     const int* snapshot_iters_host, int n_snap, float* snapshots, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Optional: make the Adam update use the reference BUILD's square root instead of the IEEE one.  torch's CPU Adam calls Intel MKL's
 * vsSqrt (convex_adam_MIND.py:179), which returns the correctly rounded root or a neighbour of it (Xeon / AVX-512 path: one ulp below
 * for 0.6 % of all inputs; EPYC hosts: one ulp above or below for 17 %); which one is a pure function of (exponent parity, mantissa)
 * and is tabulated from torch.sqrt itself (convexadam_amd/reference_bits.py; fixture of the golden host: tests/golden/mkl_vssqrt_low.npz).
 * device_table: 6 MiB on the current device, COPIED into memory the context owns (the caller may free it after the call; the
 * copy is released when the table is replaced or the context destroyed, after a device synchronisation) -- two bits per class (0 = IEEE root, 1 = one ulp above, 2 = one ulp
 * below), four per byte, low bits first; entries 0 .. 2^24-1 normal inputs (key = parity << 23 | mantissa), entries 2^24 ..
 * 2^24+2^23-1 denormal inputs (key = mantissa); NULL restores the IEEE sqrt (default).
 * With the table the Adam operator is bit-identical to the reference for given features at any number of iterations. */
int cvx_context_set_adam_sqrt_table(cvx_context* ctx, const void* device_table, void* stream);
int cvx_set_adam_sqrt_table(const void* device_table);                 /* default context, default stream */

/* Optional, same idea for the `exp` of MINDSSC (convex_adam_utils.py:63: torch CPU -> MKL vsExp, at most one ulp from the library's
 * expf, position independent, a property of the HOST: MKL dispatches on the CPU model).  device_table: two bits per argument x <= 0,
 * entry k = key - first_key with key = bit pattern of |x| (0 = equal to the library's expf, 1 = one ulp above, 2 = one ulp below),
 * four entries per byte, low bits first; arguments outside [first_key, first_key + count) are left alone.  Copied like the sqrt
 * table; NULL restores the default.  With BOTH tables of a host the whole pipeline reproduces the reference run on that host bit for bit
 * (tests/golden/fullsize.npz, 80 Adam iterations).  convexadam_amd/reference_bits.py builds the tables from torch itself. */
int cvx_context_set_mind_exp_table(cvx_context* ctx, const void* device_table, unsigned first_key, unsigned count, void* stream);
int cvx_set_mind_exp_table(const void* device_table, unsigned first_key, unsigned count);      /* default context, default stream */
/* the library's expf (no table), elementwise: out[i] = exp(x[i]); what a table for cvx_set_mind_exp_table is the difference to */
int cvx_expf_f32(const float* x, float* out, size_t n, void* stream);

/* same loop with a pluggable smoother instead of the three 3^3 boxes (adam_run_withconfig_shiftSpline.py:214-230);
 * sm == NULL or the chain {3,3,3} selects the fused kernels of cvx_adam_run_f32. */
int cvx_adam_run_smoother_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                              float* v, float lambda_weight, int niter, int step0, float cost_scale,
                              const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                              const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm,
                              void* workspace, size_t workspace_bytes, void* stream);

/* the same loop with an explicit storage format for the loop's own feature copies: feature_storage 0 = float32 records,
 * 1 = half-precision records (fp16 storage, convex_adam_MIND.py:79: rounded once when the records are built, widened on every load;
 * halves the bytes gathered per iteration); sm as above */
int cvx_adam_run_ex_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                        float* v, float lambda_weight, int niter, int step0, float cost_scale,
                        const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                        const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm,
                        int feature_storage, void* workspace, size_t workspace_bytes, void* stream);

/* adam_mode "fast": the same loop (packaged three 3^3 boxes, float32 feature records) in throughput arithmetic -- replaces the body of
 * the reference's loop, convex_adam_MIND.py:163-179, to within rounding: the forward boxes keep ATen's order (the regulariser
 * differentiates U twice, so U's rounding pattern decides how long the trajectory stays next to the reference's), the warp / data-term
 * gradient uses FMA chains and eight corner accumulators, the adjoint boxes are separable sums with one final scale, the update has
 * one IEEE division.  Deterministic; bit-identical to oracle/cvx_oracle.c::orc_adam_run_fast.  Arguments as cvx_adam_run_f32. */
int cvx_adam_run_fast_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                          float* v, float lambda_weight, int niter, int step0, float cost_scale,
                          const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                          const int* snapshot_iters_host, int n_snap, float* snapshots,
                          void* workspace, size_t workspace_bytes, void* stream);
/* adam_mode "fast_all": as above with the FORWARD boxes in the separable arithmetic too (44 instead of 56 us per iteration at the
 * benchmark size).  Offered for callers that grade by overlap scores, NOT accepted by the criteria the fast mode meets: mean EPE against
 * the reference's capture 7.2e-5 / 1.8e-4 / 2.3e-3 after 20 / 40 / 80 iterations (the regulariser differentiates U twice, and U's rounding
 * pattern is what keeps the trajectory next to the reference's).  Bit-identical to orc_adam_run_fast(fast_forward = 1). */
int cvx_adam_run_fast_all_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                              float* v, float lambda_weight, int niter, int step0, float cost_scale,
                              const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                              const int* snapshot_iters_host, int n_snap, float* snapshots,
                              void* workspace, size_t workspace_bytes, void* stream);
/* the loop with a smoother of the sweep scripts in the arithmetic `mode` (0 exact = cvx_adam_run_smoother_f32, 1 fast, 2 fast_all):
 * a box chain (kovesi_spline) runs its adjoint -- with mode 2 also the forward pass -- through the separable passes of
 * cvx_smooth_fast_f32, a Gaussian keeps its exact 1-D convolutions; fast warp gradient and one-division update in both.  sm == NULL:
 * the packaged three 3^3 boxes.  Bit-identical to oracle/cvx_oracle.c::orc_adam_run_fast_smoother. */
int cvx_adam_run_mode_f32(const float* F2, const float* M2, int C, int h, int w, int d, float* P, float* m,
                          float* v, float lambda_weight, int niter, int step0, float cost_scale,
                          const float* base_h, const float* base_w, const float* base_d, float* U, float* grad_out,
                          const int* snapshot_iters_host, int n_snap, float* snapshots, const cvx_smoother* sm, int mode,
                          void* workspace, size_t workspace_bytes, void* stream);
/* separable restatement of a box-chain smoother (kovesi_spline, hyper_util:475-488) on a [3][h][w][d] field: per axis the boxes of the
 * chain as 1-D sums with zero padding per stage, `backward` = the adjoint (reversed box order), one final multiplication by
 * 1 / prod k^3; in == out allowed.  Equals cvx_smooth_f32 to rounding (3e-7 relative), three launches instead of one per box and 27 /
 * 125-tap sums. */
int cvx_smooth_fast_f32(const float* in, int h, int w, int d, const cvx_smoother* sm, int backward, float* out, void* stream);
/* out = fastbox(in): the separable restatement of box3(box3(box3(.))) used for the adjoint in adam_mode "fast"; [C][h][w][d], C = 3 */
int cvx_box3_fast_f32(const float* in, int h, int w, int d, float* out, void* stream);

/* whole pair ---------------------------------------------------------------------------------------
 * replaces convex_adam_pt(...)                                       convex_adam_MIND.py:64-202
 * (use_mask=False path; features are MIND-SSC of the two images, or caller-supplied feature
 *  volumes when feat_fixed/feat_moving are non-NULL: convex_adam_nnUNet.py:41-159) */
typedef struct cvx_pair_params {
    int H, W, D;
    int mind_r, mind_d;
    float lambda_weight;
    int grid_sp, disp_hw;
    int selected_niter, selected_smooth;
    int grid_sp_adam;
    int ic;
    int n_feat;          /* 0: compute MIND (12 ch) from images; >0: feat_* given with this many channels */
    float cost_scale;    /* 12 */
    /* variants; 0 everywhere = the packaged pipeline */
    int cost;            /* 0 SSD, 1 SAD                                   l2r_2021_convexAdam_task3_docker.py:54 */
    int n_box;           /* 0 or 2: two box filters on the cost volume, 1: one     task2_docker.py:60 */
    int n_spline_pools;  /* 0 or 3: three 3^3 boxes in the Adam loop, 2: two       task3_docker.py:191 */
    int corr_fast;       /* 1: fast correlation mode (see cvx_corr_opts) */
    int fp16_storage;    /* 1: half-precision STORAGE with float32 accumulation (the reference's GPU default dtype, convex_adam_MIND.py:79):
                            both cost volumes and the Adam loop's feature records are __half buffers (half the bytes written and
                            re-read), the coarse pooled features (2 x C x h x w x d values) are rounded to half precision in their
                            float32 working copies; graded by end-point error against the float32 field */
    const cvx_context* ctx;  /* switches + tables for this call; NULL: the context bound to the calling thread (else the default one) */
    /* ---- ABI version 2 (appended; zero = the behaviour of version 1) ---- */
    int adam_fast;       /* 2: adam_mode "fast_all" (cvx_adam_run_fast_all_f32: forward boxes separable too, outside the acceptance criteria);
                            1: adam_mode "fast" -- throughput arithmetic of the Adam loop (cvx_adam_run_fast_f32): same mathematics as
                            convex_adam_MIND.py:163-179, graded by end-point error against the reference's field instead of by bits.
                            Needs the packaged smoother (n_spline_pools 0 / 3); with fp16_storage the warp kernel gathers 8-byte records
                            (half the traffic of the loop's memory-bound kernel) */
    int reserved_[3];    /* must be zero */
} cvx_pair_params;

size_t cvx_register_pair_workspace_bytes(const cvx_pair_params* p);
/* out_field [3][H][W][D] float32 (for ic=0 && lambda_weight<=0 the reference returns the coarse
 * field unchanged: [3][H/gs][W/gs][D/gs], convex_adam_MIND.py:143-144; out_dims_host[3] reports it). */
int cvx_register_pair_f32(const float* img_fixed, const float* img_moving, const float* feat_fixed,
                          const float* feat_moving, const cvx_pair_params* p, float* out_field,
                          int* out_dims_host, void* workspace, size_t workspace_bytes, void* stream);

/* the same pipeline with iteration snapshots:
 * replaces the 9-field variant self_configuring/convex_adam_MIND.py:115-139 (disp_sample after iterations 40 / 60 / 80, each without
 * and with three 3^3 / 5^3 boxes) and feeds the sweep's evaluation at iterations 59/79/99/119 (adam_run_withconfig_shiftSpline.py:234).
 *   snapshot_iters_host [n_snap] ascending, 1-based, <= selected_niter; smooth_host [n_smooth] entries 0 (none) or an odd box size
 *   out_fields [n_snap][n_smooth][3][H][W][D]: interpolate(disp_sample_i * grid_sp_adam, (H,W,D)) followed by three k^3 boxes */
size_t cvx_register_pair_snapshots_workspace_bytes(const cvx_pair_params* p, int n_snap, const int* smooth_host, int n_smooth);
int cvx_register_pair_snapshots_f32(const float* img_fixed, const float* img_moving, const float* feat_fixed,
                                    const float* feat_moving, const cvx_pair_params* p, const int* snapshot_iters_host, int n_snap,
                                    const int* smooth_host, int n_smooth, float* out_fields, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* n_pairs independent pairs with the same parameters, dealt round-robin onto n_streams (1..8) internal HIP
 * streams that fork from / join into `stream` (the sweep scripts' loop over pairs, e.g.
 * self_configuring/convex_run_withconfig.py:85).  Pointer arrays live on the HOST and hold device pointers;
 * img_* or feat_* may be NULL as in cvx_register_pair_f32.  workspace = n_streams x (per-pair size rounded up to 4096). */
int cvx_register_pairs_f32(int n_pairs, const float* const* img_fixed, const float* const* img_moving,
                           const float* const* feat_fixed, const float* const* feat_moving, const cvx_pair_params* p,
                           float* const* out_fields, int* out_dims_host, void* workspace, size_t workspace_bytes,
                           int n_streams, void* stream);

/* per-stage device time of cvx_register_pair_f32 calls on this thread (hipEvents recorded on the launch
 * stream, ms).  cvx_set_profiling(0) off, (1) keep the last call only, (2) accumulate over calls until the next
 * cvx_set_profiling(); (3) = (2) plus one interval per kernel of the Adam loop ("adam.forward_boxes", "adam.warp_gradient",
 * "adam.adjoint_update", once per iteration; the stage interval "adam" then only holds what follows the last kernel);
 * names_host receives pointers to static strings; returns the number of intervals written (waits for the last recorded event). */
int cvx_last_pair_profile(const char** names_host, float* ms_host, int max_stages);
void cvx_set_profiling(int mode);

/* ---- evaluation operators of the self-configuring sweep and apply_convex --------------
 *   cvx_jacobian_det_f32        self_configuring/convexAdam_hyper_util.py:86-108 jacobian_determinant_3d(dense_flow, convert1):
 *                               flow [3][H][W][D] -> out [(H-4)][(W-4)][(D-4)]; convert1 != 0 scales by (size-1)/2 first
 *   cvx_jacobian_stats_f64      convex_run_withconfig.py:148-150: acc3 (device) = { sum (l-l0), sum (l-l0)^2, #(jac < 0) } with
 *                               l = log(clamp(jac + 3, 1e-9, 1e9)) in float64, l0 = l of element 0 (std, folding fraction: host)
 *   cvx_warp_labels_nearest_f32 convex_run_withconfig.py:96,135,141: F.grid_sample(seg, grid0 + disp.flip/scale1, mode='nearest');
 *                               disp [3][H][W][D] in voxels, base_* = cvx_affine_base_host(H/W/D) uploaded
 *   cvx_label_overlap_i64       hyper_util:53-60 dice_coeff: counts (device) [3][num_labels] = |a==l|, |b==l|, |a==l & b==l|
 *   cvx_map_coordinates_linear_f64  src/convexAdam/apply_convex.py:13-24 apply_convex: scipy map_coordinates(order=1,
 *                               mode='constant'); moving [H][W][D] float64, disp [H][W][D][3] float64 (voxels), out float64 */
int cvx_jacobian_det_f32(const float* flow, int H, int W, int D, int convert1, float* out, void* stream);
int cvx_jacobian_stats_f64(const float* jac, int64_t n, double* acc3, void* stream);
int cvx_warp_labels_nearest_f32(const float* seg, const float* disp, int H, int W, int D, const float* base_h,
                                const float* base_w, const float* base_d, float* out, void* stream);
int cvx_label_overlap_i64(const float* a, const float* b, int64_t n, int num_labels, int64_t* counts, void* stream);
int cvx_map_coordinates_linear_f64(const double* moving, const double* disp, int H, int W, int D, double* out, void* stream);

/* ---- Euclidean feature transform: the nearest-in-mask search of the masked feature path -------------
 *   convex_adam_MIND.py:44,49: scipy.ndimage.distance_transform_edt(mask == 0, return_indices=True)[1]
 *   cvx_feature_transform_i32  : obj [H][W][D] (non-zero = voxel that looks for its nearest ZERO voxel) ->
 *                                feat [3][H][W][D] int32 coordinates of that voxel, scipy's tie-breaking included
 *   cvx_feature_flat_index_i64 : the reference's index expression idx[0]*D//2*W//2 + idx[1]*D//2 + idx[2] (:45,50) with the
 *                                full-resolution W_full, D_full -> out [H][W][D] int64 (input of cvx_gather_f32) */
size_t cvx_feature_transform_workspace_bytes(int H, int W, int D);
int cvx_feature_transform_i32(const float* obj, int H, int W, int D, int* feat, void* workspace, size_t workspace_bytes,
                              void* stream);
int cvx_feature_flat_index_i64(const int* feat, int H, int W, int D, int W_full, int D_full, int64_t* out, void* stream);

/* ---- Hausdorff-95 building blocks: cupy_hd95, self_configuring/convexAdam_hyper_util.py:32-51 -------------
 *   per label: dist = edt(mask) + edt(1 - mask), surf = (edt(mask) == 1), hd95 = max(P95(dist_fix[surf_mov]), P95(dist_mov[surf_fix]))
 *   cvx_label_mask_f32       : (:33-34) inside/outside masks of one label on the nearest-upsampled grid [H*p][W*p][D*p], voxel count (count may be NULL)
 *   cvx_edt_sqdist_i32       : (:40,42) squared distance to the nearest zero voxel of obj from its feature transform (exact integers;
 *                              the reference's float32 distance is the correctly rounded square root)
 *   cvx_surface_hist_i64     : (:48) histogram over the squared distance a_in2 + a_out2 of the voxels with b_in2 == 1; *overflow is
 *                              set when a value does not fit nbins
 *   cvx_hist_order_stats_i64 : the k0-th / k1-th smallest value and the entry count -> out3
 *   cvx_hist_percentile_neighbours_i64 : the same for the two order statistics numpy.percentile(x, 100*quantile) interpolates for float32
 *                              data (virtual index float32(n-1)*quantile evaluated on the device), no host round trip for n
 *   cvx_edt_squared_i32      : (:40,42) exact squared Euclidean distance to the nearest zero voxel, distances only (Meijster's
 *                              lower-envelope passes) for `batch` independent volumes [batch][H][W][D]; 0 on zero voxels */
int cvx_label_mask_f32(const float* seg, int H, int W, int D, int label, int precision, float* inside, float* outside,
                       int64_t* count, void* stream);
/* the same for ANY scale factor of F.interpolate(.., scale_factor=s) in nearest mode (cupy_hd95 with a non-integer `precision`):
 * (Ho, Wo, Do) = (int)(extent * s) per axis, scale_inv = (float)(1.0 / s); per axis the source index is dst when the extent is
 * unchanged, dst >> 1 when it doubles, else min(floor(dst * scale_inv), in - 1) in float32 (ATen nearest_idx) */
int cvx_label_mask_scaled_f32(const float* seg, int H, int W, int D, int label, int Ho, int Wo, int Do, float scale_inv, float* inside,
                              float* outside, int64_t* count, void* stream);
int cvx_edt_sqdist_i32(const float* obj, const int* feat, int H, int W, int D, int* d2, void* stream);
int cvx_surface_hist_i64(const int* a_in2, const int* a_out2, const int* b_in2, int64_t n, int nbins, int64_t* hist, int* overflow,
                         void* stream);
int cvx_hist_order_stats_i64(const int64_t* hist, int nbins, int64_t k0, int64_t k1, int64_t* out3, void* stream);
int cvx_hist_percentile_neighbours_i64(const int64_t* hist, int nbins, float quantile, int64_t* out3, void* stream);
/* cvx_surface_hist_i64 for n_hist (a, b) combinations in one launch: volumes_dev = DEVICE table of 3 n_hist addresses (a_in2, a_out2, b_in2
 * per histogram), hist [n_hist][nbins], overflow [n_hist] */
int cvx_surface_hist_batch_i64(const void* const* volumes_dev, int n_hist, int64_t n, int nbins, int64_t* hist, int* overflow, void* stream);
/* the same for n_hist histograms [n_hist][nbins] in one launch -> out3 [n_hist][3] (all labels x both directions of one cupy_hd95 call) */
int cvx_hist_percentile_neighbours_batch_i64(const int64_t* hist, int nbins, int n_hist, float quantile, int64_t* out3, void* stream);
size_t cvx_edt_squared_workspace_bytes(int batch, int H, int W, int D);
int cvx_edt_squared_i32(const float* obj, int batch, int H, int W, int D, int* d2, void* workspace, size_t workspace_bytes,
                        void* stream);
/* the same for the 2 n volumes (mask, complement) of n labels of ONE label map, read straight from the map: d2 [n][2][H][W][D];
 * labels_host [n] (host array, n <= 64) -- the transforms cupy_hd95 needs at precision 1 (hyper_util:39-46) without materialising the masks */
int cvx_edt_squared_labels_i32(const float* seg, int H, int W, int D, const int* labels_host, int n_labels, int* d2, void* workspace,
                               size_t workspace_bytes, void* stream);
/* cupy_hd95 (hyper_util.py:32-51) without volume-sized transforms: the reference reads its four distance transforms per label on the
 * SURFACE of the other map only (dist1[surf2], dist2[surf1], :48), so the distances are computed at those voxels alone.
 *   cvx_label_bits_u64            : one bit per (label, voxel) of a label map: bits [num_labels][H*W][ceil(D/64)] (bit z%64 of word z/64 of
 *                                   row h*W+w of plane l-1 <=> seg[h][w][z] == l); cvx_label_bits_bytes = its size
 *   cvx_surface_distance_hist_i64 : every voxel of seg_b carrying an ACTIVE label l (bit l of the 256-bit host mask active4[4]) with an
 *                                   in-bounds 6-neighbour of another value (inside distance exactly 1, :41/:45) adds one count to
 *                                   hist[(l-1)*hist_stride + d2], d2 = exact squared distance to the nearest voxel of map a outside l
 *                                   (voxel inside l in a) or inside l (voxel outside) = (edt(a==l) + edt(a!=l))**2 there; overflow
 *                                   [(l-1)*overflow_stride] = 1 if map a holds no such voxel.  The cost of one voxel grows with the
 *                                   square of its distance (the transforms' cost does not depend on it): max_radius > 0 bounds the
 *                                   search to rows (h', w') within max_radius of the voxel's row; a voxel that needs more is not
 *                                   counted, overflow = 2 and the remaining far voxels of that label are skipped (the label's
 *                                   histogram is void: use the transforms then), 0 = unbounded.  hist / overflow are
 *                                   accumulated into: zero them first.  1 .. 255 labels; H, W <= 2047 */
size_t cvx_label_bits_bytes(int H, int W, int D, int num_labels);
int cvx_label_bits_u64(const float* seg, int H, int W, int D, int num_labels, uint64_t* bits, void* stream);
int cvx_surface_distance_hist_i64(const float* seg_b, const uint64_t* bits_a, int H, int W, int D, int num_labels, const uint64_t* active4,
                                  int nbins, int64_t* hist, int64_t hist_stride, int* overflow, int overflow_stride, int max_radius, void* stream);
/* the same histogram from the bit planes of BOTH maps (bits_b = cvx_label_bits_u64 of the map whose surface is walked): 64 voxels per
 * thread, squared distances up to 24 by word arithmetic on the planes, a ring search per voxel only beyond (csrc/surfdist.hip).  Counts
 * and flags as cvx_surface_distance_hist_i64 (flag 2 also when the internal work lists overflow) */
size_t cvx_surface_distance_hist_bits_workspace_bytes(int H, int W, int D, int num_labels);
int cvx_surface_distance_hist_bits_i64(const uint64_t* bits_b, const uint64_t* bits_a, int H, int W, int D, int num_labels,
                                       const uint64_t* active4, int nbins, int64_t* hist, int64_t hist_stride, int* overflow,
                                       int overflow_stride, int max_radius, void* workspace, size_t workspace_bytes, void* stream);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* CONVEXADAM_HIP_H */
