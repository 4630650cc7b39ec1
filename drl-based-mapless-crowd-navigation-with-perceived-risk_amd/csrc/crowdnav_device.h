// crowdnav_device.h -- device-side arithmetic shared by the kernels of libcrowdnav.so (gfx950).
//
// Every function states the reference lines it implements (ENV = environment_stage_1_nobonus.py,
// UTL = utils.py under /root/reference/turtlebot3_rl_sim/src).  All arithmetic that feeds a
// rounding, a comparison or an index is float64 and written with explicit fma() (the file is
// compiled with -ffp-contract=off) so that it is reproducible operation-for-operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CN_PI 3.14159265358979323846

// ---- Python / numpy rounding -------------------------------------------------------------
// cn_config.py2_round, read IN PLACE from the kernel-argument block and only inside the (rare) exact-tie branch: the flag costs
// the common path nothing.  0: Python-3 round() (ties-to-even on the exact binary value); 1: Python-2.7 round() (floatobject.c
// _Py_double_round: correctly rounded, an exact tie goes away from zero).
typedef const __attribute__((address_space(4))) int32_t* cn_kflag;

// Python round(x, nd) scaled by p = 10^nd, i.e. its integral part: round(x, 3) == cn_div1000(cn_round_scaled(x, 1000, py2)).
// x*p = y + err exactly (fma); rint(y) can only be misled when y is exactly a half-integer.
__device__ __forceinline__ double cn_round_scaled(double x, double p, cn_kflag py2)
{
    double y = x * p;
    double r = rint(y);
    double d = y - r;
    // A decimal tie of the PRODUCT is rare: the common path pays ONE compare and ONE scalar branch on "any lane has a tie" (the
    // wave's vote) -- as a per-lane `if` it was a compare, an exec save, a branch and an exec restore at each of the ~40 roundings
    // of a step (round 5: the instruction stream is what a step costs, whatever the class: tools/micro/issue_cost.hip).
    const bool tie = fabs(d) == 0.5;
    unsigned long long any_tie = __builtin_amdgcn_ballot_w64(tie);
    asm("" : "+s"(any_tie));                          // (opaque: otherwise the compiler folds the vote back into a per-lane branch)
    if (__builtin_expect(any_tie != 0ull, 0)) {
        if (tie) {
            double err = fma(x, p, -y);
            if (err > 0.0) r = y + 0.5;
            else if (err < 0.0) r = y - 0.5;
            else if (*py2) r = y + copysign(0.5, y);     // exact tie under Python 2.7: half away from zero
        }
    }
    return r;
}
__device__ __forceinline__ double cn_py_round(double x, double p, cn_kflag py2) { return cn_round_scaled(x, p, py2) / p; }
// numpy around / round(np.float64, nd) under Python 3: multiply, rint, divide (ENV:255, ENV:1042)
__device__ __forceinline__ double cn_np_around(double x, double p) { return rint(x * p) / p; }

// ---- exact division by 1000 / 100 without the divide sequence ------------------------------------
// q = x*inv, rem = fma(-q, P, x), q' = fma(rem, inv, q) with inv = RN(1/P) is the correctly rounded
// x / P (Markstein); verified exhaustively for every integer |x| < 2^31 for P = 1000 and P = 100
// (tools/check_const_div.c).  All callers pass integral x (a rint() result) in that range.
#define CN_INV1000 (1.0 / 1000.0)
#define CN_INV100 (1.0 / 100.0)
__device__ __forceinline__ double cn_div1000(double r)
{
    double q = r * CN_INV1000;
    return fma(fma(-q, 1000.0, r), CN_INV1000, q);
}
__device__ __forceinline__ double cn_div100(double r)
{
    double q = r * CN_INV100;
    return fma(fma(-q, 100.0, r), CN_INV100, q);
}
__device__ __forceinline__ double cn_py_round3(double x, cn_kflag py2)
{
    double r = cn_round_scaled(x, 1000.0, py2);
    return (fabs(r) < 2147483648.0) ? cn_div1000(r) : r / 1000.0;
}
__device__ __forceinline__ double cn_py_round2(double x, cn_kflag py2)
{
    double r = cn_round_scaled(x, 100.0, py2);
    return (fabs(r) < 2147483648.0) ? cn_div100(r) : r / 100.0;
}
__device__ __forceinline__ double cn_np_around3(double x)
{
    double r = rint(x * 1000.0);
    return (fabs(r) < 2147483648.0) ? cn_div1000(r) : r / 1000.0;
}
// The same roundings for values a simulated run bounds far below 2^31 thousandths (coordinates, ranges, velocities of a room
// of a few metres): without the range test and the generic-divide branch behind it (compare + exec save/restore + two
// branches per use).  External data (odometry, scans) keeps the guarded forms.
template <bool SMALL> __device__ __forceinline__ double cn_py_round3_t(double x, cn_kflag py2)
{
    if constexpr (SMALL) return cn_div1000(cn_round_scaled(x, 1000.0, py2)); else return cn_py_round3(x, py2);
}
template <bool SMALL> __device__ __forceinline__ double cn_np_around3_t(double x)
{
    if constexpr (SMALL) return cn_div1000(rint(x * 1000.0)); else return cn_np_around3(x);
}
__device__ __forceinline__ double cn_np_around2(double x)
{
    double r = rint(x * 100.0);
    return (fabs(r) < 2147483648.0) ? cn_div100(r) : r / 100.0;
}
template <bool SMALL> __device__ __forceinline__ double cn_py_round2_t(double x, cn_kflag py2)
{
    if constexpr (SMALL) return cn_div100(cn_round_scaled(x, 100.0, py2)); else return cn_py_round2(x, py2);
}
template <bool SMALL> __device__ __forceinline__ double cn_np_around2_t(double x)
{
    if constexpr (SMALL) return cn_div100(rint(x * 100.0)); else return cn_np_around2(x);
}
// round(np.float64, 2) (ENV:255, ORIG:280, RW:209): numpy's multiply / rint / divide under Python 3; under Python 2.7 the builtin
// takes the value as a C double and rounds it like any float.  The two only differ when x * 100 lands exactly on a half-integer.
template <bool SMALL> __device__ __forceinline__ double cn_round_np64_2_t(double x, cn_kflag py2)
{
    const double y = x * 100.0;
    double r = rint(y);
    if (__builtin_expect(fabs(y - r) == 0.5, 0) && *py2) r = cn_round_scaled(x, 100.0, py2);
    if constexpr (SMALL) return cn_div100(r); else return (fabs(r) < 2147483648.0) ? cn_div100(r) : r / 100.0;
}

// ---- deterministic sin/cos ----------------------------------------------------------------
// Cody-Waite reduction by pi/2 + degree-13/14 minimax kernels; only + * fma rint, so host and
// device produce identical bits (the simulator's contract, DESIGN.md "physics").
__host__ __device__ inline void cn_det_sincos(double x, double* sn, double* cs)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00;
    const double p2 = 6.07710050630396597660e-11;
    const double p3 = 2.02226624879595063154e-21;
    double fn = rint(x * two_over_pi);
    double r = fma(-fn, p1, x);
    r = fma(-fn, p2, r);
    r = fma(-fn, p3, r);
    double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = fma(z, S6, S5);
    ps = fma(z, ps, S4);
    ps = fma(z, ps, S3);
    ps = fma(z, ps, S2);
    ps = fma(z, ps, S1);
    double s = fma(r * z, ps, r);
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = fma(z, C6, C5);
    pc = fma(z, pc, C4);
    pc = fma(z, pc, C3);
    pc = fma(z, pc, C2);
    pc = fma(z, pc, C1);
    double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    long long q = (long long)fn;
    switch (q & 3) {
    case 0: *sn = s;  *cs = c;  break;
    case 1: *sn = c;  *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

// Deterministic exp (ped_mode 2, the social-force repulsions): Cody-Waite reduction by ln 2, degree-13 Taylor polynomial on
// |r| <= ln2 / 2, 2^k through the exponent field -- only + * fma rint, the same sequence as the CPU oracle's, so the same bits.
// |x| <= 700 (callers pass arguments in [-12, ~2]).
__host__ __device__ inline double cn_det_exp(double x)
{
    const double LOG2E = 1.44269504088896338700e+00, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    if (x < -700.0) return 0.0;
    if (x > 700.0) return INFINITY;
    const double kf = rint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    union { unsigned long long u; double d; } sc;
    sc.u = (unsigned long long)((long long)kf + 1023) << 52;
    return q * sc.d;
}

// The same evaluation with its 16 constants read from a table in the constant address space (the kernel argument block).
// Written as literals, each Horner step costs three VALU instructions on gfx950: the compiler emits v_fmac_f64 with the
// 64-bit addend moved into the destination by two v_mov_b32 first.  From the table the addend is a scalar-register operand
// of one v_fma_f64 (s_load runs on the scalar unit, beside the vector pipe the kernel is bound by).
#define CN_TRIG_TABLE { 6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050630396597660e-11, 2.02226624879595063154e-21, \
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06, \
    -2.50507602534068634195e-08, 1.58969099521155010221e-10, \
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07, \
    2.08757232129817482790e-09, -1.13596475577881948265e-11, \
    /* [16..26] cn_atan2_t: Q(z) = (r - atan r) / r^3, z = r^2 <= tan^2(pi/8), degree 10 (tools/fit_atan.py: 7e-18 relative) */ \
    3.33333333333333314830e-01, -1.99999999999955130336e-01, 1.42857142846651796741e-01, -1.11111110151467143425e-01, \
    9.09090457366906051773e-02, -7.69218308734202632637e-02, 6.66450998981804459964e-02, -5.85813625190614653548e-02, \
    5.08538348884283106233e-02, -3.92297445366122307653e-02, 1.91745435720213318331e-02, \
    /* [27] tan(pi/8)  [28..33] pi/4, pi/2, pi as hi + lo */ \
    0.41421356237309503, 7.85398163397448278999e-01, 3.06161699786838301793e-17, \
    1.57079632679489655800e+00, 6.12323399573676603587e-17, 3.14159265358979311600e+00, 1.22464679914735317723e-16 }
#define CN_TRIG_COUNT 34
typedef const __attribute__((address_space(4))) double* cn_ktab;
// fma(a, b, c) with the addend c in scalar registers, spelled out: for a uniform addend the compiler would pick the
// two-address v_fmac_f64 and copy c into the destination with two v_mov_b32 first (three vector instructions per Horner step).
__device__ __forceinline__ double cn_fma_s(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// sqrt() for the squares of lengths between a micrometre and a few metres (and exactly 0): the compiler's expansion of the
// float64 square root -- v_rsq_f64 seed, one coupled Newton step on (g, h) ~ (sqrt x, 1 / (2 sqrt x)), two residual
// corrections; correctly rounded -- without its exponent scaling for arguments below 2^-767 (five of its 17 instructions).
// Same operations in the same order, so the same bits wherever the scaling was the identity; 0, +inf and NaN as sqrt().
__device__ __forceinline__ double cn_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;      // +-0, +inf: the seed is inf or 0 there
}

// a / b for a finite a and a normal, non-zero b whose quotient is neither huge nor denormal (every use states why): the
// compiler's float64 division -- v_rcp_f64 seed, two Newton steps on the reciprocal, one residual correction of the quotient;
// correctly rounded -- without v_div_scale x 2 / v_div_fmas' post-scale / v_div_fixup, which only act on extreme exponents,
// zeros, infinities and NaNs.  Same operations in the same order: the same bits wherever those were the identity (8
// instructions instead of 12).  b == 0 gives NaN here (not +-inf): callers guard it or discard the lane's result; a numerator
// of -0 over a positive b gives +0 where IEEE gives -0 (equal as numbers; tests/test_gpu_parity.py::test_device_math_*).
__device__ __forceinline__ double cn_div(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y = fma(y, fma(-b, y, 1.0), y);
    y = fma(y, fma(-b, y, 1.0), y);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// cn_div with IEEE results kept for a zero divisor (the flagged anomalies: a zero time step from external clocks)
__device__ __forceinline__ double cn_div_z(double a, double b) { return (b == 0.0) ? a / b : cn_div(a, b); }

// fmin / fmax as the bare instruction.  Through the builtin the compiler first canonicalises every operand it cannot prove
// free of signalling NaNs (v_max_f64 x, x, x: up to three instructions per min); the hardware instruction already returns
// the other operand for a quiet NaN, which is all fmin()/fmax() promise and all these values can be.
__device__ __forceinline__ double cn_vmin(double a, double b)
{
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double cn_vmax(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double cn_vmax_s(double a, double b)     // b: wave-uniform, held in scalar registers
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b));
    return d;
}
// base[i] with a wave-uniform base and an unsigned 32-bit index: the byte offset stays a 32-bit VGPR next to the scalar
// base (global_load/store ... v, s[base]) instead of a sign-extended 64-bit address built from four vector instructions.
template <typename T> __device__ __forceinline__ T cn_ldg(const T* base, unsigned i)
{
    return *(const T*)((const char*)base + (size_t)(i * (unsigned)sizeof(T)));
}
template <typename T> __device__ __forceinline__ void cn_stg(T* base, unsigned i, T v)
{
    *(T*)((char*)base + (size_t)(i * (unsigned)sizeof(T))) = v;
}
// x with its sign flipped where s is negative (x * sign(s) for s != 0)
__device__ __forceinline__ double cn_xorsign(double x, double s)
{
    return __hiloint2double(__double2hiint(x) ^ (__double2hiint(s) & (int)0x80000000), __double2loint(x));
}
__device__ __forceinline__ double cn_vclamp(double x, double lo, double hi) { return cn_vmin(cn_vmax(x, lo), hi); }
__device__ __forceinline__ void cn_det_sincos_t(cn_ktab t, double x, double* sn, double* cs)
{
    double fn = rint(x * t[0]);
    double r = fma(-fn, t[1], x);
    r = fma(-fn, t[2], r);
    r = fma(-fn, t[3], r);
    double z = r * r;
    double ps = fma(z, t[9], t[8]);
    ps = cn_fma_s(z, ps, t[7]);
    ps = cn_fma_s(z, ps, t[6]);
    ps = cn_fma_s(z, ps, t[5]);
    ps = cn_fma_s(z, ps, t[4]);
    double s = fma(r * z, ps, r);
    double pc = fma(z, t[15], t[14]);
    pc = cn_fma_s(z, pc, t[13]);
    pc = cn_fma_s(z, pc, t[12]);
    pc = cn_fma_s(z, pc, t[11]);
    pc = cn_fma_s(z, pc, t[10]);
    double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)(long long)fn & 3;
    const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;     // q: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s)
    *sn = (q & 2) ? -a : a;
    *cs = (q == 1 || q == 2) ? -b : b;
}

// atan2 for the heading to the goal (ENV:222-237, math.atan2), which is rounded to 2 decimals right after.  The device
// library's atan2 costs ~110 vector instructions here, half of them v_mov_b32 of polynomial constants (see cn_fma_s); this
// one is a divide, one Horner chain with scalar addends and the quadrant fix-ups:
//   atan(mn / mx), mn <= mx:  mn <= tan(pi/8) mx ?  A(mn / mx)  :  pi/4 + A((mn - mx) / (mn + mx)),   A(r) = r - r z Q(z)
// Within 2.5 ulp of the exact value on 4e7 points incl. the reference's 3-decimal coordinates, at most 1 ulp (4.4e-16)
// from glibc's (tools/check_atan2.c); a different last bit changes the heading only when heading * 100 lies that close
// to a half-integer.  Signed zeros as C99 (atan2(+-0, -0) = +-pi); infinite arguments do not occur (positions are finite).
__device__ __forceinline__ double cn_atan2_t(cn_ktab t, double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = cn_vmax(ax, ay), mn = cn_vmin(ax, ay);
    const bool hi = mn > t[27] * mx;
    const double num = hi ? mn - mx : mn, den = hi ? mn + mx : mx;
    const double r = (mx == 0.0) ? 0.0 : cn_div(num, den);      // den = mx or mn + mx >= the larger coordinate difference
    const double z = r * r;
    double q = fma(z, t[26], t[25]);
    q = cn_fma_s(z, q, t[24]);
    q = cn_fma_s(z, q, t[23]);
    q = cn_fma_s(z, q, t[22]);
    q = cn_fma_s(z, q, t[21]);
    q = cn_fma_s(z, q, t[20]);
    q = cn_fma_s(z, q, t[19]);
    q = cn_fma_s(z, q, t[18]);
    q = cn_fma_s(z, q, t[17]);
    q = cn_fma_s(z, q, t[16]);
    double a = fma(-r, z * q, r);
    if (hi) a = t[28] + (a + t[29]);
    if (ay > ax) a = t[30] - (a - t[31]);
    if (__double2hiint(x) < 0) a = t[32] - (a - t[33]);
    return copysign(a, y);
}

// hypot() for lengths of a few metres, BIT-EQUAL to the C library's the reference's math.hypot resolves to under Python 2.7 and the
// goldens were recorded on (ENV:754, UTL:283-284; DESIGN.md section 4 has the note): glibc 2.35's algorithm -- the square
// root of the plain sum of squares and one correction step -- restated operation by operation (the build has -ffp-contract=off;
// cn_sqrt / cn_div are correctly rounded).  Rounds 1-6 used sqrt(fma(a, a, b b)), which differs from it in the last bit on 13 % of
// the arguments; that was inside every float tolerance, but ENV:826 compares two speeds for EXACT equality (`relative_vel == 0`
// picks between two formulas a factor of two apart) and a robot driving straight past a static object makes them equal on
// paper -- tools/fuzz_parity.py found worlds whose top-K sets differed for it.  glibc's scaling branches for huge / tiny
// arguments never apply to coordinates in thousandths of a metre; hypot(x, 0) = |x| exactly, as in C.
__device__ __forceinline__ double cn_hypot(double x, double y)
{
    const double fx = fabs(x), fy = fabs(y);
    const double ax = cn_vmax(fx, fy), ay = cn_vmin(fx, fy);
    double h = cn_sqrt(ax * ax + ay * ay);
    const bool lo = h <= 2.0 * ay;
    const double delta = h - (lo ? ay : ax);
    const double t1 = lo ? ax * (2.0 * delta - ax) : 2.0 * delta * (ax - 2.0 * ay);
    const double t2 = lo ? (delta - 2.0 * (ax - ay)) * delta : (4.0 * delta - ay) * ay + delta * delta;
    h -= cn_div(t1 + t2, 2.0 * h);
    return (ay == 0.0) ? ax : h;
}

// ---- counter-based RNG (CROWD:101-102 random.uniform) ---------------------------------------
__host__ __device__ inline uint64_t cn_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline double cn_rng_u01(uint64_t seed, int64_t env, uint32_t stream, uint32_t a, uint32_t b)
{
    uint64_t h = cn_mix64(seed ^ cn_mix64((uint64_t)env));
    h = cn_mix64(h ^ (((uint64_t)stream << 32) | (uint64_t)a));
    h = cn_mix64(h ^ (uint64_t)b);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline double cn_clamp(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// ---- UTL:422-460 IoU of two axis-aligned squares, rounded to 3 decimals ---------------------
__device__ __forceinline__ double cn_iou3(double ax, double ay, double bx, double by, double half, cn_kflag py2)
{
    double axp = ax + half, axm = ax - half, ayp = ay + half, aym = ay - half;
    double bxp = bx + half, bxm = bx - half, byp = by + half, bym = by - half;
    double ix = cn_vmin(axp, bxp) - cn_vmax(axm, bxm);
    double iy = cn_vmin(ayp, byp) - cn_vmax(aym, bym);
    if (!(ix > 0.0 && iy > 0.0)) return 0.0;
    double inter = ix * iy;
    double area_a = (axp - axm) * (ayp - aym);
    double area_b = (bxp - bxm) * (byp - bym);
    double uni = area_a + area_b - inter;
    return cn_py_round3(inter / uni, py2);
}

// round(IoU, 3) > 0 without the divide in the common cases.  The boxes either do not overlap (IoU = 0) or overlap
// well: inter > 0.00075 * union means the quotient exceeds 0.00075 (1 - 2^-52), which rounds to >= 0.001.  Only the
// sliver in between takes the exact path, so the result equals cn_iou3(...) > 0.0 always.
__device__ __forceinline__ bool cn_iou3_positive(double ax, double ay, double bx, double by, double half, cn_kflag py2)
{
    double axp = ax + half, axm = ax - half, ayp = ay + half, aym = ay - half;
    double bxp = bx + half, bxm = bx - half, byp = by + half, bym = by - half;
    double ix = cn_vmin(axp, bxp) - cn_vmax(axm, bxm);
    double iy = cn_vmin(ayp, byp) - cn_vmax(aym, bym);
    if (!(ix > 0.0 && iy > 0.0)) return false;
    double inter = ix * iy;
    double area_a = (axp - axm) * (ayp - aym);
    double area_b = (bxp - bxm) * (byp - bym);
    double uni = area_a + area_b - inter;
    if (inter > 0.00075 * uni) return true;
    return cn_py_round3(inter / uni, py2) > 0.0;
}

// ---- wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ double cn_shfl_xor_d(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
// Wave-wide reductions on the DPP path (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// across the rows; the total lands in lane 63 and is read back as a scalar).  A step is two or three VALU instructions;
// the __shfl_xor butterfly is two ds_bpermute LDS round trips per step, ~1 k cycles of pure latency per reduction on a
// wavefront's critical path, six times per observation.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int cn_dpp_i(int ident, int v) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double cn_dpp_d(double ident, double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i - N (row_shr) / lane i + N (row_shl) of its own row of 16 lanes, 1 <= N <= 15; a lane whose source would lie
// outside the row keeps `ident`.  One VALU instruction where __shfl_up / __shfl_down are a ds_bpermute round trip through the LDS
// crossbar (an address, the permute, a wait: tools/profc counted 60 of them per env-step, a fifth of the LDS instructions, eighteen
// of them in dependent scan steps) -- for the lane = word scans, whose at most 16 words all sit in row 0.
template <int N> __device__ __forceinline__ int cn_row_shr_i(int ident, int v) { static_assert(N >= 1 && N <= 15, "row_shr"); return cn_dpp_i<0x110 + N, 0xf>(ident, v); }
template <int N> __device__ __forceinline__ int cn_row_shl_i(int ident, int v) { static_assert(N >= 1 && N <= 15, "row_shl"); return cn_dpp_i<0x100 + N, 0xf>(ident, v); }
// v with lane `l` (wave-uniform) replaced by the wave-uniform 64-bit value x: two v_writelane_b32 -- where `if (lane == l) v = x`
// compiles to a compare and two selects with the scalar pair moved into vector registers first
// (clang has no writelane builtin; the lane select goes through m0, which does not count against the one-scalar-operand limit.
// m0 is a reserved register the compiler only loads right in front of the few instructions that read it -- none in these kernels:
// no LDS instruction of gfx950 does -- hence the warning about clobbering it is switched off here.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ unsigned long long cn_writelane_u64(unsigned long long v, unsigned long long x, int l)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned xl = __builtin_amdgcn_readfirstlane((unsigned)x), xh = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    const int ls = __builtin_amdgcn_readfirstlane(l);
    asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(lo), "+v"(hi) : "s"(xl), "s"(xh), "s"(ls) : "m0");
    return ((unsigned long long)hi << 32) | lo;
}
#pragma clang diagnostic pop
// lane `l` (wave-uniform) of v as a wave-uniform 64-bit value: two v_readlane_b32
__device__ __forceinline__ unsigned long long cn_readlane_u64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
#define CN_DPP_REDUCE(T, DPP, v, ident, OP)                         \
    do {                                                            \
        v = OP(v, DPP<0x111, 0xf>(ident, v)); /* row_shr:1 */       \
        v = OP(v, DPP<0x112, 0xf>(ident, v)); /* row_shr:2 */       \
        v = OP(v, DPP<0x114, 0xf>(ident, v)); /* row_shr:4 */       \
        v = OP(v, DPP<0x118, 0xf>(ident, v)); /* row_shr:8 */       \
        v = OP(v, DPP<0x142, 0xa>(ident, v)); /* row_bcast:15 */    \
        v = OP(v, DPP<0x143, 0xc>(ident, v)); /* row_bcast:31 */    \
    } while (0)
__device__ __forceinline__ double cn_lane63_d(double v)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cn_wave_min_d(double v)
{
    const double id = INFINITY;
    CN_DPP_REDUCE(double, cn_dpp_d, v, id, cn_vmin);
    return cn_lane63_d(v);
}
__device__ __forceinline__ double cn_wave_max_d(double v)
{
    const double id = -INFINITY;
    CN_DPP_REDUCE(double, cn_dpp_d, v, id, cn_vmax);
    return cn_lane63_d(v);
}
__device__ __forceinline__ int cn_add_i(int a, int b) { return a + b; }
__device__ __forceinline__ int cn_wave_min_i(int v)
{
    const int id = 0x7fffffff;
    CN_DPP_REDUCE(int, cn_dpp_i, v, id, min);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int cn_wave_max_i(int v)
{
    const int id = (int)0x80000000;
    CN_DPP_REDUCE(int, cn_dpp_i, v, id, max);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int cn_wave_sum_i(int v)
{
    const int id = 0;
    CN_DPP_REDUCE(int, cn_dpp_i, v, id, cn_add_i);
    return __builtin_amdgcn_readlane(v, 63);
}
