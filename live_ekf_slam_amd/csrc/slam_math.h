// slam_math.h — deterministic fp64 elementary functions shared by the HIP kernels and the CPU oracle.
//
// Why this exists: the reference filter calls libm sin/cos/atan2/remainder/sqrt on fp64 values
// (ekf.cpp:48-59,115-129,146-165; sim_node.py:222-236).  glibc (host) and ROCm's OCML (device) agree only
// to ~1 ulp on sin/cos/atan2, and the EKF then truncates to float (ekf.cpp:115,129-131), so a 1-ulp
// disagreement can flip a float rounding and show up as a 1e-8 jump in the state.  To make "GPU == oracle"
// a BIT-EXACT statement we evaluate these three functions with one fixed sequence of IEEE-754 fp64
// operations (+,-,*,/ only; no FMA contraction — both sides are compiled with -ffp-contract=off) written
// once, here, and compiled by both gcc (oracle) and hipcc (device).  sqrt, division, remainder(x,2π) and
// double→float conversion are correctly rounded / exact on both sides and are used directly.
//
// The algorithms are the classical ones (Cody–Waite π/2 reduction; minimax odd/even polynomials on
// [-π/4, π/4] for sin/cos; 4-interval argument reduction + degree-11 odd polynomial for atan), with the
// published fdlibm/msun coefficient sets; error < 1 ulp.  The oracle can also be built against libm
// (LibmMath policy in oracle/slam_oracle.cpp) to show how little the choice matters (tests/test_oracle_math.py).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SLAM_HD __host__ __device__ __forceinline__
#else
#define SLAM_HD inline
#endif

namespace slam {
// |d| as the unknown-id association of ekf.cpp:91-92 sees it: the double overload of abs (as_int = 0), or - if the unqualified `abs`
// resolves to ::abs(int) - the value truncated to an int first (out-of-range values saturate, here and in the oracle alike).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline float assoc_abs(double d, int as_int) {
    if (!as_int) return (float)fabs(d);
    const double t = d >= 2147483647.0 ? 2147483647.0 : (d <= -2147483647.0 ? -2147483647.0 : (d != d ? 0.0 : d));
    const int i = (int)t;
    return (float)(i < 0 ? -i : i);
}

// filter.h:42  `#define pi 3.14159265358979323846` ; every wrap in the reference is remainder(x, 2*pi).
static constexpr double kPi = 3.14159265358979323846;
static constexpr double kTwoPi = 2 * 3.14159265358979323846;

// ---- kernels on |x| <= pi/4 (x + y is the reduced argument, y the tail) -------------------------------
SLAM_HD double k_sin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

SLAM_HD double k_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// sin and cos of x together.  Valid for |x| < ~1e6 (Cody–Waite reduction); the filters only ever pass wrapped
// headings plus a bearing, and the simulator's unwrapped true yaw stays within a few hundred radians.
SLAM_HD void det_sincos(double x, double* s, double* c) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double pio2_2 = 6.07710050630396597660e-11;   // next 33 bits
    const double pio2_2t = 2.02226624879595063154e-21;  // pi/2 - (pio2_1 + pio2_2)
    double ax = fabs(x);
    if (ax <= 0.78539816339744827900) {  // <= pi/4: no reduction
        *s = k_sin(x, 0.0);
        *c = k_cos(x, 0.0);
        return;
    }
    double fn = rint(x * invpio2);
    // two compensated subtraction rounds: 33 + 33 + 53 bits of pi/2 (abs. error ~ fn * 2^-119)
    double t = x - fn * pio2_1;  // exact: fn < 2^20 and pio2_1 has 33 significant bits
    double w = fn * pio2_2;
    double r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double ks = k_sin(y0, y1);
    double kc = k_cos(y0, y1);
    int q = ((int)(long long)fn) & 3;
    double ss = (q & 1) ? kc : ks;
    double cc = (q & 1) ? ks : kc;
    if (q == 1) cc = -cc;
    if (q == 2) { ss = -ss; cc = -cc; }
    if (q == 3) ss = -ss;
    *s = ss;
    *c = cc;
}

SLAM_HD double det_sin(double x) { double s, c; det_sincos(x, &s, &c); return s; }
SLAM_HD double det_cos(double x) { double s, c; det_sincos(x, &s, &c); return c; }

// atan for finite x (sign handled by the caller through |x|)
SLAM_HD double det_atan(double x) {
    const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
                 hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
    const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
                 lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
                 a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
                 a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
                 a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
                 a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
                 a10 = 1.62858201153657823623e-02;
    bool neg = x < 0.0;
    double ax = fabs(x);
    if (ax >= 73786976294838206464.0) {  // 2^66: atan = ±pi/2
        double z = hi3 + lo3;
        return neg ? -z : z;
    }
    int id;
    double hi = 0.0, lo = 0.0;
    if (ax < 0.4375) {
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); hi = hi0; lo = lo0; }
        else             { id = 1; ax = (ax - 1.0) / (ax + 1.0);       hi = hi1; lo = lo1; }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); hi = hi2; lo = lo2; }
        else             { id = 3; ax = -1.0 / ax;                     hi = hi3; lo = lo3; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    double res;
    if (id < 0) res = ax - ax * (s1 + s2);
    else        res = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -res : res;
}

// atan2(y, x) for finite arguments (the filters never feed infinities; NaN propagates).
SLAM_HD double det_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    const double pi_o_2 = 1.5707963267948965580E+00;
    if (x != x || y != y) return x + y;
    bool xneg = signbit(x), yneg = signbit(y);
    if (y == 0.0) {  // ±0 or ±pi
        if (!xneg) return y;
        return yneg ? -pi : pi;
    }
    if (x == 0.0) return yneg ? -pi_o_2 : pi_o_2;
    double q = fabs(y / x);
    double z = det_atan(q);
    if (!xneg) return yneg ? -z : z;
    return yneg ? (z - pi_lo) - pi : pi - (z - pi_lo);
}

// remainder(x, 2*pi): IEEE-754 remainder is exact, so libm (host) and OCML (device) must agree bit for bit;
// tests/test_parity_gpu.py::test_device_math_bit_exact checks that on the device.
SLAM_HD double wrap2pi(double x) { return remainder(x, kTwoPi); }

// The same value as remainder(x, 2*pi), bit for bit, in a dozen instructions for |x| <= 4*pi (every wrap of a heading in the
// filters): the library routine is an iterative reduction (~300 cycles of dependent instructions on the GPU).  IEEE
// remainder is x - n*y with n = x/y rounded to nearest, ties to even; for |x| <= 2y = 4*pi, n is in {0, +-1, +-2} and every
// subtraction below is exact (Sterbenz: a - b is exact when b/2 <= a <= 2b), so the result IS the IEEE remainder.  y/2 is
// exactly fl(pi) and 2y is exact.  Larger or non-finite arguments take the library path.  Checked against the host's libm
// on the device by tests/test_parity_gpu.py::test_device_math_bit_exact.
SLAM_HD double rem2pi(double x) {
    const double y = kTwoPi, hy = kPi;
    const double ax = fabs(x);
    if (!(ax <= 2.0 * y)) return remainder(x, y);
    double r = ax;                      // n = 0: |x| <= y/2 (the tie 0.5 rounds to the even 0)
    if (ax > hy) {
        const double d = ax - y;        // exact: y/2 < ax <= 2y
        r = d;                          // n = 1: |x|/y in (0.5, 1.5)
        if (d >= hy) r = d - y;         // n = 2: |x|/y in [1.5, 2] (the tie 1.5 rounds to the even 2); exact: y/2 <= d <= y
    }
    return signbit(x) ? -r : r;         // remainder(-x) = -remainder(x); a zero result carries the sign of x
}

}  // namespace slam
