// accuracy of the device atan2 restated on the host (same operations, fma where the device uses fma) against libm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static const double T[] = {
    3.33333333333333314830e-01, -1.99999999999955130336e-01, 1.42857142846651796741e-01, -1.11111110151467143425e-01,
    9.09090457366906051773e-02, -7.69218308734202632637e-02, 6.66450998981804459964e-02, -5.85813625190614653548e-02,
    5.08538348884283106233e-02, -3.92297445366122307653e-02, 1.91745435720213318331e-02,
    0.41421356237309503,            // tan(pi/8)
    7.85398163397448278999e-01, 3.06161699786838301793e-17,   // pi/4 hi, lo
    1.57079632679489655800e+00, 6.12323399573676603587e-17,   // pi/2
    3.14159265358979311600e+00, 1.22464679914735317723e-16 }; // pi
static double my_atan2(double y, double x)
{
    const double* t = T;
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const int hi = mn > t[11] * mx;
    const double num = hi ? mn - mx : mn, den = hi ? mn + mx : mx;
    double r = (mx == 0.0) ? 0.0 : num / den;
    const double z = r * r;
    double q = fma(z, t[10], t[9]);
    for (int k = 8; k >= 0; --k) q = fma(z, q, t[k]);
    double a = fma(-r, z * q, r);                   // atan(r) = r - r z Q(z)
    if (hi) a = t[12] + (a + t[13]);
    if (ay > ax) a = t[14] - (a - t[15]);
    if (signbit(x)) a = t[16] - (a - t[17]);
    return copysign(a, y);
}
static uint64_t rs = 88172645463325252ull;
static double u01(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
int main(void)
{
    double maxulp = 0, maxabs = 0; long n = 0;
    for (long i = 0; i < 40000000; ++i) {
        double x, y;
        switch (i & 3) {
        case 0: x = (u01() - 0.5) * 6; y = (u01() - 0.5) * 6; break;
        case 1: x = round((u01() - 0.5) * 6000) / 1000.0; y = round((u01() - 0.5) * 6000) / 1000.0; break;
        case 2: { double a = (u01() - 0.5) * 2 * M_PI, rr = exp((u01() - 0.5) * 20); x = rr * cos(a); y = rr * sin(a); } break;
        default: { double a = (u01() - 0.5) * 2 * M_PI; a = round(a * 8 / M_PI) * M_PI / 8 + (u01() - 0.5) * 1e-6; x = cos(a); y = sin(a); }
        }
        double g = my_atan2(y, x), ref = atan2(y, x);
        long double refl = atan2l((long double)y, (long double)x);
        double ulp = fabs(ref) > 0 ? nextafter(fabs(ref), INFINITY) - fabs(ref) : 5e-324;
        double e = fabs((double)((long double)g - refl)) / ulp;
        if (e > maxulp) maxulp = e;
        if (fabs(g - ref) > maxabs) maxabs = fabs(g - ref);
        ++n;
    }
    printf("%ld points: max error %.3f ulp vs atan2l, max |mine - libm atan2| %.3e\n", n, maxulp, maxabs);
    double sp[][2] = {{0,0},{0,-0.0},{-0.0,-0.0},{-0.0,0.0},{1,0},{-1,0},{0,1},{0,-1},{1,1},{-1,-1},{1e-300,1},{1,1e-300},{0.0,-1.0},{-0.0,-1.0}};
    for (unsigned k = 0; k < sizeof sp / sizeof sp[0]; ++k)
        printf("atan2(%g, %g): mine %.17g libm %.17g\n", sp[k][0], sp[k][1], my_atan2(sp[k][0], sp[k][1]), atan2(sp[k][0], sp[k][1]));
    return 0;
}
