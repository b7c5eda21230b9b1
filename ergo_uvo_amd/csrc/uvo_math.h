// uvo_math.h -- scalar helpers shared by host and device code of libuvo_hip.
// Everything is plain IEEE arithmetic (compiled with -ffp-contract=off): round-half-even
// conversions as OpenCV's cvRound, and sin/cos/acos/hypot built from +,-,*,/,sqrt only so the
// device needs no libm transcendental inside the pose solvers (hard part 7).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include <stdint.h>

#define UVO_HD __host__ __device__ __forceinline__

namespace uvo {

UVO_HD int cv_round_f(float v)  { return (int)rintf(v); }
UVO_HD int cv_round_d(double v) { return (int)rint(v); }
UVO_HD int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
UVO_HD int cv_ceil_d(double v)  { int i = (int)v; return i + (i < v); }

UVO_HD double det_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0.0) return 0.0;
    double t = b / a;
    return a * sqrt(1.0 + t * t);
}

// Cody-Waite reduction by pi/2, Taylor polynomials on [-pi/4, pi/4]
UVO_HD void det_sincos(double x, double* s, double* c)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    double kf = rint(x * TWO_OVER_PI);
    double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    double r2 = r * r;
    double ps = -1.0 / 121645100408832000.0;
    ps = ps * r2 + 1.0 / 355687428096000.0;
    ps = ps * r2 - 1.0 / 1307674368000.0;
    ps = ps * r2 + 1.0 / 6227020800.0;
    ps = ps * r2 - 1.0 / 39916800.0;
    ps = ps * r2 + 1.0 / 362880.0;
    ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0;
    ps = ps * r2 - 1.0 / 6.0;
    double sr = r + r * (r2 * ps);
    double pc = 1.0 / 6402373705728000.0;
    pc = pc * r2 - 1.0 / 20922789888000.0;
    pc = pc * r2 + 1.0 / 87178291200.0;
    pc = pc * r2 - 1.0 / 479001600.0;
    pc = pc * r2 + 1.0 / 3628800.0;
    pc = pc * r2 - 1.0 / 40320.0;
    pc = pc * r2 + 1.0 / 720.0;
    pc = pc * r2 - 1.0 / 24.0;
    pc = pc * r2 + 0.5;
    double cr = 1.0 - r2 * pc;
    long long k = (long long)kf;
    switch (k & 3) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default:*s = -cr; *c = sr;  break;
    }
}

// acos by 6 Newton steps on cos(theta) = c
UVO_HD double det_acos(double c)
{
    const double PI = 3.14159265358979323846;
    if (c >= 1.0) return 0.0;
    if (c <= -1.0) return PI;
    bool neg = c < 0.0;
    double a = neg ? -c : c;
    double th = sqrt(2.0 * (1.0 - a));
    for (int it = 0; it < 6; it++) {
        double s, cc;
        det_sincos(th, &s, &cc);
        th = th + (cc - a) / s;
    }
    return neg ? PI - th : th;
}

// cv::RNG multiply-with-carry step
UVO_HD uint32_t rng_next(uint64_t& state)
{
    state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

}  // namespace uvo
