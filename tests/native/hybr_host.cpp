// Host build of the library's restatement of the reference's solver (raytracing_amd/csrc/rt_hybrd.h), for the CPU test
// that holds it against SciPy's hybr on the reference's own zero functions.  Test infrastructure: the product runs the
// same header on the GPU.
#include "../../raytracing_amd/csrc/rt_hybrd.h"

extern "C" int rt_hybr_host_solve(const double *origin, const double *destination, const double *planes, int n_refl,
                                  const double *edge, double *out_points, int *nfev) {
    return rt_hybr::solve_like_reference(origin, destination, planes, n_refl, edge, out_points, nfev) ? 1 : 0;
}

// the run-time-sized form alone (the entry point above takes the unrolled forms for the shapes that have one)
extern "C" int rt_hybr_host_solve_generic(const double *origin, const double *destination, const double *planes, int n_refl,
                                          const double *edge, double *out_points, int *nfev) {
    return rt_hybr::solve_like_reference_generic(origin, destination, planes, n_refl, edge, out_points, nfev) ? 1 : 0;
}

// the one-loop form (rt_hybr::hybr_flat) for the shapes that have one
extern "C" int rt_hybr_host_solve_flat(const double *origin, const double *destination, const double *planes, int n_refl,
                                       const double *edge, double *out_points, int *nfev) {
    return rt_hybr::solve_like_reference_flat(origin, destination, planes, n_refl, edge, out_points, nfev) ? 1 : 0;
}

// the one-loop form in pieces of `piece` trips, each piece resumed from the state the one before it left
extern "C" int rt_hybr_host_solve_in_pieces(const double *origin, const double *destination, const double *planes, int n_refl,
                                            const double *edge, double *out_points, int *nfev, int piece) {
    return rt_hybr::solve_like_reference_in_pieces(origin, destination, planes, n_refl, edge, out_points, nfev, piece) ? 1 : 0;
}
