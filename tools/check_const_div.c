// Exhaustive check backing cn_div1000 / cn_div100 in csrc/crowdnav_device.h:
//   gcc -O2 -mfma -ffp-contract=off -fopenmp tools/check_const_div.c -o /tmp/divchk -lm && /tmp/divchk
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <omp.h>
// Is  q2 = fma(fma(-q,P,r), inv, q), q = r*inv, inv = RN(1/P)  equal to RN(r/P) for every integer |r| < 2^31 ?
int main(){
  const double Ps[2] = {1000.0, 100.0};
  for (int t=0;t<2;t++){
    const double P = Ps[t], inv = 1.0/P;
    long bad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (int64_t r = -2147483648LL; r <= 2147483647LL; ++r){
      double x = (double)r;
      double q = x*inv;
      double rem = fma(-q, P, x);
      double q2 = fma(rem, inv, q);
      double ref = x / P;
      if (q2 != ref) bad++;
    }
    printf("P=%g: mismatches over all int32 numerators: %ld\n", P, bad);
  }
  // also random non-integer numerators (not needed by the kernel, for information)
  long bad2=0; uint64_t s=88172645463325252ULL;
  for (long i=0;i<200000000L;i++){ s^=s<<13; s^=s>>7; s^=s<<17; double x = (double)(int64_t)(s>>11) * 1e-9 - 4e6; double q=x*(1.0/1000.0); double q2=fma(fma(-q,1000.0,x),1.0/1000.0,q); if(q2!=x/1000.0) bad2++; }
  printf("random real numerators / 1000: mismatches %ld of 2e8\n", bad2);
  return 0;
}
