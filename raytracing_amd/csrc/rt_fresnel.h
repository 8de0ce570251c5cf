// Fresnel integrals S(x) = int_0^x sin(pi t^2 / 2) dt and C(x) = int_0^x cos(pi t^2 / 2) dt, the pair the reference takes
// from scipy.special.fresnel for the UTD transition function (electromagnetism.py:355-363), and their auxiliary
// functions f, g (Abramowitz & Stegun 7.3.5-6):
//     C = 1/2 + f sin(pi x^2 / 2) - g cos(pi x^2 / 2),     S = 1/2 - f cos(pi x^2 / 2) - g sin(pi x^2 / 2).
// Below RT_FRESNEL_X0: C / x and S / x^3 as Chebyshev series in x^4.  Above: pi x f and pi^2 x^3 g as Chebyshev series in
// x^-4 on three intervals.  The coefficients (rt_fresnel_tables.h, made with 50-digit arithmetic by
// scripts/make_fresnel_tables.py) are cut below 1e-17 of the function: f and g come out within 3e-17, S and C within a
// few 1e-16 for x below 10 (beyond that the argument pi x^2 / 2 of sine and cosine carries its own rounding, as it does
// in SciPy); tests/test_fresnel.py holds a host build against SciPy and mpmath.  A fixed number of fused operations
// and no division in the series: a series run to convergence with a division per term cost ten times as much.
#pragma once
#include <math.h>

#include "rt_fresnel_tables.h"

#ifdef __HIPCC__
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

// Chebyshev series on [a, b] by Clenshaw's recurrence
template <int N>
RT_HD inline double rt_chebyshev(const double (&coef)[N], double a, double b, double u) {
    const double t2 = 2.0 * ((2.0 * u - (a + b)) / (b - a));
    double b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int k = N - 1; k >= 1; --k) {
        const double b0 = coef[k] + t2 * b1 - b2;
        b2 = b1;
        b1 = b0;
    }
    return coef[0] + 0.5 * t2 * b1 - b2;
}

// 0 <= x < RT_FRESNEL_X0
RT_HD inline void rt_fresnel_small(double x, double *S, double *C) {
    constexpr double pc[RT_FRESNEL_PC_N] = RT_FRESNEL_PC_COEF, ps[RT_FRESNEL_PS_N] = RT_FRESNEL_PS_COEF;
    const double x2 = x * x, u = x2 * x2;
    *C = x * rt_chebyshev(pc, RT_FRESNEL_PC_A, RT_FRESNEL_PC_B, u);
    *S = x * x2 * rt_chebyshev(ps, RT_FRESNEL_PS_A, RT_FRESNEL_PS_B, u);
}

// x >= RT_FRESNEL_X0
RT_HD inline void rt_fresnel_fg(double x, double *f, double *g) {
    constexpr double p0[RT_FRESNEL_PHI0_N] = RT_FRESNEL_PHI0_COEF, g0[RT_FRESNEL_GAM0_N] = RT_FRESNEL_GAM0_COEF;
    constexpr double p1[RT_FRESNEL_PHI1_N] = RT_FRESNEL_PHI1_COEF, g1[RT_FRESNEL_GAM1_N] = RT_FRESNEL_GAM1_COEF;
    constexpr double p2[RT_FRESNEL_PHI2_N] = RT_FRESNEL_PHI2_COEF, g2[RT_FRESNEL_GAM2_N] = RT_FRESNEL_GAM2_COEF;
    const double PI = 3.14159265358979323846;
    const double x2 = x * x, u = 1.0 / (x2 * x2);
    double phi, gam;
    if (x < RT_FRESNEL_X1) {
        phi = rt_chebyshev(p0, RT_FRESNEL_PHI0_A, RT_FRESNEL_PHI0_B, u);
        gam = rt_chebyshev(g0, RT_FRESNEL_GAM0_A, RT_FRESNEL_GAM0_B, u);
    } else if (x < RT_FRESNEL_X2) {
        phi = rt_chebyshev(p1, RT_FRESNEL_PHI1_A, RT_FRESNEL_PHI1_B, u);
        gam = rt_chebyshev(g1, RT_FRESNEL_GAM1_A, RT_FRESNEL_GAM1_B, u);
    } else {
        phi = rt_chebyshev(p2, RT_FRESNEL_PHI2_A, RT_FRESNEL_PHI2_B, u);
        gam = rt_chebyshev(g2, RT_FRESNEL_GAM2_A, RT_FRESNEL_GAM2_B, u);
    }
    *f = phi / (PI * x);
    *g = gam / (PI * PI * x * x2);
}

RT_HD inline void rt_fresnel(double x, double *S, double *C) {
    const double ax = fabs(x);
    double s, c;
    if (ax < RT_FRESNEL_X0) {
        rt_fresnel_small(ax, &s, &c);
    } else if (ax > 36974.0) {  // as SciPy: beyond, f and g are below the rounding of 1/2
        s = c = 0.5;
    } else {
        double f, g;
        rt_fresnel_fg(ax, &f, &g);
        const double z = 0.5 * 3.14159265358979323846 * ax * ax;
        const double sz = sin(z), cz = cos(z);
        c = 0.5 + f * sz - g * cz;
        s = 0.5 - f * cz - g * sz;
    }
    if (x < 0) {
        s = -s;
        c = -c;
    }
    *S = s;
    *C = c;
}
