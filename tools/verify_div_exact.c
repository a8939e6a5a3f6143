#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <omp.h>
// exhaustive check: q = x * r; e = fma(-D, q, x); q' = fma(e, r, q)  == x / D  for all floats x ?
static inline float u2f(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static inline uint32_t f2u(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
int main(int argc, char** argv){
  float Ds[] = {27.0f, 125.0f, 8.0f, 216.0f, 12.0f, 64.0f, 343.0f};
  for (int di=0; di<7; ++di){
    const float D = Ds[di]; const float r = 1.0f / D;
    long long bad=0; uint32_t firstbad=0; float minbad=1e38f, maxbad=0;
    #pragma omp parallel for reduction(+:bad) 
    for (long long i=0;i<(1LL<<32);++i){
      uint32_t u=(uint32_t)i; float x=u2f(u);
      if (x!=x || isinf(x)) continue;
      float q=x*r; float e=fmaf(-D,q,x); float q2=fmaf(e,r,q);
      float ref=x/D;
      if (f2u(q2)!=f2u(ref)) { bad++; 
        #pragma omp critical
        { float ax=fabsf(x); if(ax<minbad)minbad=ax; if(ax>maxbad)maxbad=ax; firstbad=u; }
      }
    }
    printf("D=%g: mismatches %lld  (|x| range of mismatches: %g .. %g) example %08x\n", D, bad, minbad, maxbad, firstbad);
  }
  return 0;
}
