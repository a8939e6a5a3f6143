// Issue rate of packed fp32 VALU instructions on gfx950: the same number of float multiplies / adds as 8 scalar chains (v_mul_f32 /
// v_add_f32) and as 4 packed chains (v_pk_mul_f32 / v_pk_add_f32), at 1..6 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o tools/pk_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* t, int iters) {
    float a[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float c = 1.000001f;
    const f32x2 c2 = {c, c};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 4) {               // the broadcast form the warp kernel would need: both halves take the low word of src1
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(c2));
        } else {                              // dependent chain of one packed multiply (latency)
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(c2));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}
template <int MODE>
static void run(const char* name, int wg_per_cu, float* out, unsigned long long* t) {
    const int grid = 256 * wg_per_cu, iters = 20000;
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, out, t, iters);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, t, grid * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < grid; ++i) c += h[i];
    // 8 float operations per thread per iteration in modes 0-4, one wave per SIMD per workgroup
    printf("%-28s %d waves/SIMD: %.2f shader clocks per 64 float ops per SIMD\n", name, wg_per_cu, (c / grid) / (iters * 8.0 * wg_per_cu));
}
int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMalloc(&t, 256 * 8 * 8);
    for (int w : {1, 2, 4, 6}) {
        run<0>("v_mul_f32 x8", w, out, t);
        run<1>("v_pk_mul_f32 x4", w, out, t);
        run<2>("v_add_f32 x8", w, out, t);
        run<3>("v_pk_add_f32 x4", w, out, t);
        run<4>("v_pk_mul_f32 x4 (op_sel_hi)", w, out, t);
    }
    run<5>("v_pk_mul_f32 dependent (x4 per it; /8*4 = latency)", 1, out, t);
    return 0;
}
