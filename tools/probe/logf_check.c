// Probe: does the ARM-optimized-routines logf algorithm (glibc >= 2.27) reproduce this host's
// logf bit for bit on the 2^24 uniform grid (u and u/(1-u))?  Two contraction variants.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static const double T[16][2] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 },
  { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 },
  { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 },
  { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
  { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },
  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },
  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};
static const double Ln2 = 0x1.62e42fefa39efp-1;
static const double A[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float asfloat(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static float mylog(float x, int use_fma){
  uint32_t ix=asuint(x);
  if(ix==0x3f800000) return 0;
  if(ix-0x00800000>=0x7f800000-0x00800000){
    if(ix*2==0) return -INFINITY;
    if(ix==0x7f800000) return x;
    if((ix&0x80000000)||ix*2>=0xff000000) return NAN;
    ix=asuint(x*0x1p23f); ix-=23<<23;
  }
  uint32_t tmp=ix-0x3f330000;
  int i=(tmp>>19)%16;
  int k=(int32_t)tmp>>23;
  uint32_t iz=ix-(tmp&0x1ffu<<23);
  double invc=T[i][0], logc=T[i][1];
  double z=(double)asfloat(iz);
  double r,y0,r2,y;
  if(use_fma){
    r=fma(z,invc,-1.0);
    y0=fma((double)k,Ln2,logc);
    r2=r*r;
    y=fma(A[1],r,A[2]);
    y=fma(A[0],r2,y);
    y=fma(y,r2,y0+r);
  } else {
    r=z*invc-1;
    y0=logc+(double)k*Ln2;
    r2=r*r;
    y=A[1]*r+A[2];
    y=A[0]*r2+y;
    y=y*r2+(y0+r);
  }
  return (float)y;
}
int main(){
  for(int v=0;v<2;++v){
    long bad_u=0,bad_l=0;
    for(uint32_t k=0;k<(1u<<24);++k){
      float u=(float)k/16777216.0f;
      float a=logf(u), b=mylog(u,v);
      if(asuint(a)!=asuint(b)) { if(bad_u<3) printf("u k=%u host=%a mine=%a\n",k,a,b); bad_u++; }
      float q=u/(1.0f-u);
      a=logf(q); b=mylog(q,v);
      if(asuint(a)!=asuint(b)) { if(bad_l<3) printf("logit k=%u host=%a mine=%a\n",k,a,b); bad_l++; }
    }
    printf("variant fma=%d: mismatches ln(u)=%ld logit=%ld\n",v,bad_u,bad_l);
  }
  return 0;
}
