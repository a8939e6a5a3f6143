// valu_rate.hip -- issue rate of scalar and packed fp32 VALU instructions on gfx950 (wave64), per SIMD, at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/experiments/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i + threadIdx.x, seed * 2 + i};
    f2 m = {1.0000001f, 0.9999999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(m));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* d, int wg_per_cu) {
    const int iters = 2000, nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves (a 256-thread workgroup = 4 waves = one per SIMD), each iters*64 instructions
    const double instr = (double)iters * 64 * wg_per_cu;
    printf("%-14s waves/SIMD %d : %.3f ms  -> %.2f clk/instr at 2.4 GHz\n", name, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / instr);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", d, w); run<1>("v_pk_mul_f32", d, w); run<2>("v_add_f32", d, w); run<3>("v_pk_add_f32", d, w);
        run<4>("v_fma_f32", d, w); run<5>("v_pk_fma_f32", d, w); run<6>("v_pk_mul bcast", d, w);
    }
    return 0;
}
