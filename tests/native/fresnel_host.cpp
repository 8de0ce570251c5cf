// Host build of the library's Fresnel integrals (raytracing_amd/csrc/rt_fresnel.h), for the CPU test that holds them
// against SciPy and mpmath.  Test infrastructure: the product runs the same header on the GPU.
#include "../../raytracing_amd/csrc/rt_fresnel.h"

extern "C" void rt_fresnel_host(const double *x, long n, double *S, double *C) {
    for (long i = 0; i < n; ++i) rt_fresnel(x[i], S + i, C + i);
}
// f and g for x >= RT_FRESNEL_X0
extern "C" void rt_fresnel_fg_host(const double *x, long n, double *f, double *g) {
    for (long i = 0; i < n; ++i) rt_fresnel_fg(x[i], f + i, g + i);
}
extern "C" double rt_fresnel_x0() { return RT_FRESNEL_X0; }
