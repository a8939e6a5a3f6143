#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(s) s s s s s s s s
template<int MODE>
__global__ void k(float* out, int iters) {
    float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    float c = 1.0001f, e = 0.5f;
    f32x2 p0={a0,a1}, p1={a2,a3}, p2={a4,a5}, p3={a6,a7}, q={1.0f,1.0001f};
    int i0 = threadIdx.x, i1 = 3;
    for (int i=0;i<iters;++i) {
        if (MODE==0) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==1) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==2) asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==3) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c), "v"(e));
        if (MODE==4) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "vcc");
        if (MODE==5) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==6) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3) : "v"(q));
        if (MODE==7) asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(7));
        if (MODE==8) asm volatile("v_add_f32 %0, %0, 1.0\n v_add_f32 %1, %1, 1.0\n v_add_f32 %2, %2, 1.0\n v_add_f32 %3, %3, 1.0\n v_add_f32 %4, %4, 1.0\n v_add_f32 %5, %5, 1.0\n v_add_f32 %6, %6, 1.0\n v_add_f32 %7, %7, 1.0"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
        if (MODE==9) asm volatile("v_mul_f32 %0, %0, 1.0\n v_mul_f32 %1, %1, 1.0\n v_mul_f32 %2, %2, 1.0\n v_mul_f32 %3, %3, 1.0\n v_mul_f32 %4, %4, 1.0\n v_mul_f32 %5, %5, 1.0\n v_mul_f32 %6, %6, 1.0\n v_mul_f32 %7, %7, 1.0"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));
        if (MODE==10) asm volatile("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==11) asm volatile("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==12) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c), "v"(e));
        if (MODE==13) asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
    }
    out[blockIdx.x*blockDim.x+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7+p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y+i0+i1;
}
template<int MODE> void run(const char* name, int wavesPerSimd, double opsPerIter) {
    float* d; hipMalloc(&d, 256*8*1024*4*8);
    int iters=20000; int blocks=256*wavesPerSimd;
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks,256>>>(d,100); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks,256>>>(d,iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double instr_per_simd = (double)iters*opsPerIter*wavesPerSimd;
    double cycles = ms*1e-3*2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr (at 2.4GHz)\n", name, wavesPerSimd, ms, cycles/instr_per_simd);
    hipFree(d);
}
int main(){
    for (int w : {4,8}) {
        run<0>("v_add_f32", w, 8); run<1>("v_mul_f32", w, 8); run<2>("v_sub_f32", w, 8); run<3>("v_fmac_f32", w, 8);
        run<4>("v_cndmask_b32", w, 8); run<5>("v_mov_b32", w, 8); run<6>("v_pk_mul_f32 x4", w, 4); run<7>("v_add_u32 (2 chains)", w, 8);
        run<8>("v_add_f32 const", w, 8); run<9>("v_mul_f32 const", w, 8); run<10>("v_add_f32_dpp wave_shr", w, 8); run<11>("v_add_f32_dpp row_shr", w, 8);
        run<12>("v_fma_f32 3 vgpr", w, 8); run<13>("mul/add alternating", w, 8);
    }
    return 0;
}
