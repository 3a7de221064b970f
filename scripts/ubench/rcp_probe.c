#include <xmmintrin.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static float rcp(float x){ return _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x))); }
int main(){
  // mantissa sweep at exponent 0 (x in [1,2))
  for (int k=8;k<=16;++k){
    int ok=1; uint32_t block=1u<<(23-k);
    for (uint32_t m=0;m<(1u<<23) && ok;m+=block){
      uint32_t b=0x3f800000u|m; float x; memcpy(&x,&b,4); float r0=rcp(x);
      for(uint32_t j=1;j<block;++j){ uint32_t bb=b+j; float y; memcpy(&y,&bb,4); if(rcp(y)!=r0){ok=0;break;} }
    }
    printf("piecewise constant on top %d mantissa bits: %d\n",k,ok);
  }
  // exponent independence: rcp(x*2^e) == rcp(x)*2^-e ?
  int bad=0;
  for (uint32_t m=0;m<(1u<<23);m+=977){ uint32_t b=0x3f800000u|m; float x; memcpy(&x,&b,4);
    for(int e=-20;e<=20;e+=3){ float s=__builtin_ldexpf(1.0f,e); if(rcp(x*s)!=rcp(x)/s) ++bad; } }
  printf("exponent-dependent cases: %d\n",bad);
  // number of distinct low bits in outputs
  uint32_t orbits=0; for(uint32_t m=0;m<(1u<<23);m+=1){uint32_t b=0x3f800000u|m; float x; memcpy(&x,&b,4); float r=rcp(x); uint32_t rb; memcpy(&rb,&r,4); orbits|=rb;}
  printf("or of output bits: %08x\n",orbits);
  return 0; }
