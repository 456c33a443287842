// RN(x / 3.0) by multiply + two fused operations (csrc/ssf_math.hpp, div3_exact) against the division: gcc -O2 -mfma -ffp-contract=off tools/probe/div3_check.c -lm
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline double div3(double x){ const double c=0x1.5555555555555p-2; double q=x*c; double e=fma(-3.0,q,x); return fma(e,c,q);} 
static uint64_t s=88172645463325252ull; static inline uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
int main(){ long bad=0; 
  for(long i=0;i<1500000000L;i++){ uint64_t r=rnd(); uint64_t bits=(r&0x000FFFFFFFFFFFFFull)|((uint64_t)(1023-8+(r>>60))<<52); if(i&1){ /* near multiples of 3 of small mantissas: products 3*m +- few ulps */ uint64_t m=(r>>12)|1; double q=(double)(m& ((1ull<<53)-1)); double x=3.0*q; memcpy(&bits,&x,8); bits+= (int64_t)((r>>3)&7)-3; bits=(bits&0x000FFFFFFFFFFFFFull)|((uint64_t)(1023-2+(r&3))<<52);} double x; memcpy(&x,&bits,8); double a=x/3.0, b=div3(x); if(a!=b){ if(bad<5) printf("bad %a: %a vs %a\n",x,a,b); bad++; } }
  printf("bad=%ld\n",bad); return 0; }
