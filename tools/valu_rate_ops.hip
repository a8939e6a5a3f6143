#include <hip/hip_runtime.h>
#include <stdio.h>
template<int MODE>
__global__ void k(float* out, int iters) {
    float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    float c = 1.0001f;
    for (int i=0;i<iters;++i) {
        if (MODE==0) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "vcc");
        if (MODE==1) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "s20","s21");
        if (MODE==2) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]\n v_cndmask_b32_e64 %1, 0, %1, s[20:21]\n v_cndmask_b32_e64 %2, 0, %2, s[20:21]\n v_cndmask_b32_e64 %3, 0, %3, s[20:21]\n v_cndmask_b32_e64 %4, 0, %4, s[20:21]\n v_cndmask_b32_e64 %5, 0, %5, s[20:21]\n v_cndmask_b32_e64 %6, 0, %6, s[20:21]\n v_cndmask_b32_e64 %7, 0, %7, s[20:21]" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "s20","s21");
        if (MODE==3) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "vcc");
        if (MODE==4) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[20:21], %1, %8\n v_cmp_lt_f32_e64 s[20:21], %2, %8\n v_cmp_lt_f32_e64 s[20:21], %3, %8\n v_cmp_lt_f32_e64 s[20:21], %4, %8\n v_cmp_lt_f32_e64 s[20:21], %5, %8\n v_cmp_lt_f32_e64 s[20:21], %6, %8\n v_cmp_lt_f32_e64 s[20:21], %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "s20","s21");
        if (MODE==5) asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==6) asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==7) asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==8) asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==9) asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==10) asm volatile("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==11) asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==12) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==13) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==14) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==15) asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==16) asm volatile("v_bfe_u32 %0, %0, 1, 8\n v_bfe_u32 %1, %1, 1, 8\n v_bfe_u32 %2, %2, 1, 8\n v_bfe_u32 %3, %3, 1, 8\n v_bfe_u32 %4, %4, 1, 8\n v_bfe_u32 %5, %5, 1, 8\n v_bfe_u32 %6, %6, 1, 8\n v_bfe_u32 %7, %7, 1, 8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==17) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==18) asm volatile("v_mul_f32 %0, -%0, %8\n v_mul_f32 %1, -%1, %8\n v_mul_f32 %2, -%2, %8\n v_mul_f32 %3, -%3, %8\n v_mul_f32 %4, -%4, %8\n v_mul_f32 %5, -%5, %8\n v_mul_f32 %6, -%6, %8\n v_mul_f32 %7, -%7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==19) asm volatile("v_add_f32_e64 %0, |%0|, %8\n v_add_f32_e64 %1, |%1|, %8\n v_add_f32_e64 %2, |%2|, %8\n v_add_f32_e64 %3, |%3|, %8\n v_add_f32_e64 %4, |%4|, %8\n v_add_f32_e64 %5, |%5|, %8\n v_add_f32_e64 %6, |%6|, %8\n v_add_f32_e64 %7, |%7|, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==20) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==21) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==22) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
        if (MODE==23) asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));
    }
    out[blockIdx.x*blockDim.x+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}
template<int MODE> void run(const char* name, int wavesPerSimd, double opsPerIter) {
    float* d; hipMalloc(&d, 256*8*1024*4*8);
    int iters=10000; int blocks=256*wavesPerSimd;
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks,256>>>(d,100); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks,256>>>(d,iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double instr_per_simd = (double)iters*opsPerIter*wavesPerSimd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr (at 2.4GHz)\n", name, wavesPerSimd, ms, ms*1e-3*2.4e9/instr_per_simd);
    hipFree(d);
}
int main(){
    for (int w : {8}) {
        run<0>("cndmask_vcc", w, 8);
        run<1>("cndmask_e64_sgpr", w, 8);
        run<2>("cndmask_const0", w, 8);
        run<3>("v_cmp_lt_f32 vcc", w, 8);
        run<4>("v_cmp_lt_f32 sgpr", w, 8);
        run<5>("v_max_f32", w, 8);
        run<6>("v_and_b32", w, 8);
        run<7>("v_lshlrev_b32", w, 8);
        run<8>("v_mad_u32_u24", w, 8);
        run<9>("v_lshl_add_u32", w, 8);
        run<10>("v_add3_u32", w, 8);
        run<11>("v_cvt_f32_i32", w, 8);
        run<12>("v_rcp_f32", w, 8);
        run<13>("v_sqrt_f32", w, 8);
        run<14>("v_mul_lo_u32", w, 8);
        run<15>("v_mul_u32_u24", w, 8);
        run<16>("v_bfe_u32", w, 8);
        run<17>("v_fma 2vgpr(a*c+a)", w, 8);
        run<18>("v_mul neg mod", w, 8);
        run<19>("v_add_f32 e64 abs", w, 8);
        run<20>("v_mov_dpp wave_shr", w, 8);
        run<21>("v_mov_dpp row_shr", w, 8);
        run<22>("v_mov_dpp quad_perm", w, 8);
        run<23>("ds_bpermute", w, 8);
    }
    return 0;
}
