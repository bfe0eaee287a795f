// Probe: does the ARM-optimized-routines expf algorithm (glibc >= 2.27) reproduce this host's expf
// bit for bit?  T[i] = bits(2^(i/32)) - (i << 47) generated from a 60-digit evaluation.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static const uint64_t T[32] = {
  0x3ff0000000000000ULL,
  0x3fefd9b0d3158574ULL,
  0x3fefb5586cf9890fULL,
  0x3fef9301d0125b51ULL,
  0x3fef72b83c7d517bULL,
  0x3fef54873168b9aaULL,
  0x3fef387a6e756238ULL,
  0x3fef1e9df51fdee1ULL,
  0x3fef06fe0a31b715ULL,
  0x3feef1a7373aa9cbULL,
  0x3feedea64c123422ULL,
  0x3feece086061892dULL,
  0x3feebfdad5362a27ULL,
  0x3feeb42b569d4f82ULL,
  0x3feeab07dd485429ULL,
  0x3feea47eb03a5585ULL,
  0x3feea09e667f3bcdULL,
  0x3fee9f75e8ec5f74ULL,
  0x3feea11473eb0187ULL,
  0x3feea589994cce13ULL,
  0x3feeace5422aa0dbULL,
  0x3feeb737b0cdc5e5ULL,
  0x3feec49182a3f090ULL,
  0x3feed503b23e255dULL,
  0x3feee89f995ad3adULL,
  0x3feeff76f2fb5e47ULL,
  0x3fef199bdd85529cULL,
  0x3fef3720dcef9069ULL,
  0x3fef5818dcfba487ULL,
  0x3fef7c97337b9b5fULL,
  0x3fefa4afa2a490daULL,
  0x3fefd0765b6e4540ULL
};
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint64_t asuint64(double f){uint64_t u;memcpy(&u,&f,8);return u;}
static inline double asdouble(uint64_t u){double f;memcpy(&f,&u,8);return f;}
static float myexp(float x){
  const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5/32/32/32, C1 = 0x1.ebfce50fac4f3p-3/32/32, C2 = 0x1.62e42ff0c52d6p-1/32;
  double xd = (double)x;
  uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
  if (abstop >= (asuint(88.0f) >> 20)) {
    if (asuint(x) == asuint(-INFINITY)) return 0.0f;
    if (abstop >= (asuint(INFINITY) >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return INFINITY;
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  double z = InvLn2N * xd;
  double kd = z + SHIFT;
  uint64_t ki = asuint64(kd);
  kd -= SHIFT;
  double r = z - kd;
  uint64_t t = T[ki % 32];
  t += ki << (52 - 5);
  double s = asdouble(t);
  z = C0 * r + C1;
  double r2 = r * r;
  double y = C2 * r + 1;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}
int main(){
  long bad=0, n=0;
  // every float in [-104, 0] with a stride, plus a dense band near zero
  for (uint32_t b = 0x80000000u; b <= asuint(-104.0f); b += 7) { float x; memcpy(&x,&b,4); float a=expf(x), c=myexp(x); n++; if (asuint(a)!=asuint(c)) { if (bad<5) printf("x=%a host=%a mine=%a\n",x,a,c); bad++; } }
  printf("checked %ld inputs, mismatches %ld\n", n, bad);
  return 0;
}
