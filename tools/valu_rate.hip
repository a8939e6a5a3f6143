#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template<int MODE>
__global__ void k(float* out, int iters) {
    float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    f32x2 p0={a0,a1}, p1={a2,a3}, p2={a4,a5}, p3={a6,a7}, q={1.0f,2.0f};
    float c = 1.5f;
    for (int i=0;i<iters;++i) {
        if (MODE==0) { // 8 independent scalar adds
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        } else if (MODE==1) { // 4 independent pk adds (8 flops-adds)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
              : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3) : "v"(q));
        } else if (MODE==2) { // 1 dependent scalar chain x8
            asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
        } else if (MODE==3) { // dependent pk chain x4
            asm volatile("v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0" : "+v"(p0) : "v"(q));
        } else if (MODE==4) { // 8 v_mul_lo_u32
            int x=(int)a0; 
            asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(3)); a0=x;
        } else if (MODE==5) { // 8 independent fma
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        }
    }
    out[blockIdx.x*blockDim.x+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7+p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y;
}
template<int MODE> void run(const char* name, int wavesPerSimd, double opsPerIter) {
    float* d; hipMalloc(&d, 256*8*1024*4*8);
    int iters=20000; int blocks=256*wavesPerSimd; // 256 threads/block = 4 waves = 1 per SIMD
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks,256>>>(d,100); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks,256>>>(d,iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double instr_per_simd = (double)iters*opsPerIter*wavesPerSimd; // wave-instr per SIMD
    double cycles = ms*1e-3*2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr (at 2.4GHz)\n", name, wavesPerSimd, ms, cycles/instr_per_simd);
    hipFree(d);
}
int main(){
    for (int w : {1,2,4,8}) {
        run<0>("v_add_f32 x8 indep", w, 8); run<1>("v_pk_add_f32 x4 indep", w, 4); run<2>("v_add_f32 dep chain", w, 8);
        run<3>("v_pk_add_f32 dep (+nop)", w, 4); run<4>("v_mul_lo_u32 dep", w, 8); run<5>("v_fma_f32 x8 indep", w, 8);
    }
    return 0;
}
