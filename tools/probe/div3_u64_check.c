// b / 3 from 32-bit pieces (csrc/ssf_math.hpp, div3_u64) against the 64-bit division: gcc -O2 (second program of this file's pair)
#include <stdio.h>
#include <stdint.h>
static uint64_t s=88172645463325252ull; static inline uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
static inline uint64_t div3_u64(uint64_t b){ uint32_t hi=(uint32_t)(b>>32), lo=(uint32_t)b; uint32_t qh=(uint32_t)(((uint64_t)hi*0xAAAAAAABull)>>33), t=(uint32_t)(((uint64_t)lo*0xAAAAAAABull)>>33); uint32_t qh2=qh+qh, t2=t+t; uint32_t r=hi-(qh2+qh), sm=lo-(t2+t); uint32_t rc=((0u-(r&1u))&0x55555555u)|((0u-(r>>1))&0xAAAAAAAAu); uint32_t ql=rc+t+((r+sm)>=3u?1u:0u); return ((uint64_t)qh<<32)|ql; }
int main(){ long bad=0; for(long i=0;i<2000000000L;i++){ uint64_t b=rnd(); if(i&1) b&=0x7FFFFFFFFFFFFFFFull; if((i&3)==3) b=(b&0xFFFFFFFF00000000ull)|(0xFFFFFFFFu-(uint32_t)(i&7)); if(div3_u64(b)!=b/3){ if(bad<5) printf("bad %llx\n",(unsigned long long)b); bad++; } } printf("bad=%ld\n",bad); return 0; }
