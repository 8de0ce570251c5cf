// Radiomap of a result table (electromagnetism_fun/scenario_measures.py:31-88): the per-receiver power interpolated
// linearly over a regular grid (scipy.interpolate.griddata(method="linear"), :66) and the grid points inside the street
// polygon (shapely contains, :54).
//
// griddata's linear method is barycentric interpolation over the Delaunay triangulation of the receivers.  The
// triangulation itself is Qhull's and stays with the caller; the kernel takes it the way scipy holds it (vertex indices
// and, per triangle, the barycentric transform: the inverse of [a - c, b - c] and c) and does what
// LinearNDInterpolator does for every grid point: find a triangle whose three barycentric coordinates lie in
// [-eps, 1 + eps] with eps = 100 * DBL_EPSILON, then sum coordinate * value in vertex order, none of it fused, so that
// a grid point gets scipy's value bit for bit whenever it gets scipy's triangle (a point on a shared edge may get the
// neighbour: the two values then differ by rounding only).  Points in no triangle get the fill value.
//
// Work mapping: lane = grid point, loop = all triangles, staged through LDS 256 at a time (the records of a stage are
// read by every lane at the same address: a broadcast).  90 000 grid points x 20 000 triangles is a millisecond.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <string>

#include "../../include/rt_pathsolve.h"
#include "rt_common.h"

namespace {

struct TriRec {
    double t[6];  // rows of the inverse, then the offset (scipy.spatial.Delaunay.transform[i], row-major)
    double v[3];  // values at the triangle's vertices, in vertex order
};

__global__ void k_tri_records(const double *__restrict__ transform, const int32_t *__restrict__ tris,
                              const double *__restrict__ values, int64_t m, TriRec *__restrict__ recs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    TriRec r;
    for (int k = 0; k < 6; ++k) r.t[k] = transform[6 * i + k];
    for (int k = 0; k < 3; ++k) r.v[k] = values[tris[3 * i + k]];
    recs[i] = r;
}

#define RADIO_STAGE 256
__global__ __launch_bounds__(256) void k_interpolate(const TriRec *__restrict__ recs, int64_t m,
                                                      const double *__restrict__ query, int64_t q, double fill,
                                                      double *__restrict__ out) {
    __shared__ TriRec stage[RADIO_STAGE];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < q;
    const double x = active ? query[2 * i] : 0.0, y = active ? query[2 * i + 1] : 0.0;
    const double eps = 100.0 * DBL_EPSILON;
    bool found = !active;
    double value = fill;
    for (int64_t base = 0; base < m; base += RADIO_STAGE) {
        if (__syncthreads_and(found)) break;  // also: the previous stage has been read by everybody
        const int cnt = (int)(m - base < RADIO_STAGE ? m - base : RADIO_STAGE);
        {
            const double *src = (const double *)(recs + base);
            double *dst = (double *)stage;
            for (int k = threadIdx.x; k < cnt * 9; k += blockDim.x) dst[k] = src[k];
        }
        __syncthreads();
        if (found) continue;
        for (int k = 0; k < cnt; ++k) {
            const TriRec &r = stage[k];
            if (!(r.t[0] == r.t[0])) continue;  // degenerate triangle: scipy marks its transform with NaN
            const double dx = __dsub_rn(x, r.t[4]), dy = __dsub_rn(y, r.t[5]);
            const double c0 = __dadd_rn(__dadd_rn(0.0, __dmul_rn(r.t[0], dx)), __dmul_rn(r.t[1], dy));
            const double c1 = __dadd_rn(__dadd_rn(0.0, __dmul_rn(r.t[2], dx)), __dmul_rn(r.t[3], dy));
            const double c2 = __dsub_rn(__dsub_rn(1.0, c0), c1);
            if (c0 >= -eps && c0 <= 1.0 + eps && c1 >= -eps && c1 <= 1.0 + eps && c2 >= -eps && c2 <= 1.0 + eps) {
                value = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(c0, r.v[0])), __dmul_rn(c1, r.v[1])), __dmul_rn(c2, r.v[2]));
                found = true;
                break;
            }
        }
    }
    if (active) out[i] = value;
}

// shapely's Polygon.contains(Point) for a ring without holes: strictly inside (a point on the boundary is not contained)
__global__ void k_ring_contains(const double *__restrict__ ring, int nv, const double *__restrict__ query, int64_t q,
                                uint8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    const double px = query[2 * i], py = query[2 * i + 1];
    bool inside = false, boundary = false;
    double ax = ring[2 * (nv - 1)], ay = ring[2 * (nv - 1) + 1];
    for (int k = 0; k < nv; ++k) {
        const double bx = ring[2 * k], by = ring[2 * k + 1];
        const double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
        if (cr == 0.0 && fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by)) boundary = true;
        if ((ay > py) != (by > py)) {
            if ((cr > 0.0) == (by > ay)) inside = !inside;
        }
        ax = bx;
        ay = by;
    }
    out[i] = inside && !boundary ? 1 : 0;
}

struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};

}  // namespace

extern "C" int rt_radiomap(int device, int64_t n_points, const double *values, int64_t n_triangles,
                           const int32_t *triangles, const double *transform, int64_t n_query, const double *query,
                           double fill_value, double *out_values, int32_t n_ring, const double *ring, uint8_t *out_inside) {
    if (n_points < 0 || n_triangles < 0 || n_query < 0 || n_ring < 0) RT_FAIL("rt_radiomap: negative count");
    if (n_query == 0) return 0;
    if (!query) RT_FAIL("rt_radiomap: null query points");
    if (out_values && n_triangles > 0 && (!values || !triangles || !transform)) RT_FAIL("rt_radiomap: null triangulation");
    if (out_inside && (n_ring < 3 || !ring)) RT_FAIL("rt_radiomap: the ring needs at least 3 vertices");
    for (int64_t i = 0; out_values && i < 3 * n_triangles; ++i)
        if (triangles[i] < 0 || triangles[i] >= n_points) RT_FAIL("rt_radiomap: vertex index out of range");
    int ndev = 0;
    RT_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) RT_FAIL("rt_radiomap: no such device");
    RT_HIP(rt_use_device(device));
    Buf d_query, d_values, d_tris, d_transform, d_recs, d_out, d_ring, d_inside;
    if (d_query.alloc(sizeof(double) * 2 * n_query)) RT_FAIL("rt_radiomap: out of device memory");
    RT_HIP(hipMemcpy(d_query.p, query, sizeof(double) * 2 * n_query, hipMemcpyHostToDevice));
    const unsigned q_blocks = (unsigned)((n_query + 255) / 256);
    if (out_values) {
        if (d_out.alloc(sizeof(double) * n_query) || d_recs.alloc(sizeof(TriRec) * n_triangles) ||
            d_values.alloc(sizeof(double) * n_points) || d_tris.alloc(sizeof(int32_t) * 3 * n_triangles) ||
            d_transform.alloc(sizeof(double) * 6 * n_triangles))
            RT_FAIL("rt_radiomap: out of device memory");
        if (n_triangles > 0) {
            RT_HIP(hipMemcpy(d_values.p, values, sizeof(double) * n_points, hipMemcpyHostToDevice));
            RT_HIP(hipMemcpy(d_tris.p, triangles, sizeof(int32_t) * 3 * n_triangles, hipMemcpyHostToDevice));
            RT_HIP(hipMemcpy(d_transform.p, transform, sizeof(double) * 6 * n_triangles, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_tri_records, dim3((unsigned)((n_triangles + 255) / 256)), dim3(256), 0, 0,
                               (const double *)d_transform.p, (const int32_t *)d_tris.p, (const double *)d_values.p,
                               n_triangles, (TriRec *)d_recs.p);
        }
        hipLaunchKernelGGL(k_interpolate, dim3(q_blocks), dim3(256), 0, 0, (const TriRec *)d_recs.p, n_triangles,
                           (const double *)d_query.p, n_query, fill_value, (double *)d_out.p);
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpy(out_values, d_out.p, sizeof(double) * n_query, hipMemcpyDeviceToHost));
    }
    if (out_inside) {
        if (d_ring.alloc(sizeof(double) * 2 * n_ring) || d_inside.alloc(n_query)) RT_FAIL("rt_radiomap: out of device memory");
        RT_HIP(hipMemcpy(d_ring.p, ring, sizeof(double) * 2 * n_ring, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ring_contains, dim3(q_blocks), dim3(256), 0, 0, (const double *)d_ring.p, (int)n_ring,
                           (const double *)d_query.p, n_query, (uint8_t *)d_inside.p);
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpy(out_inside, d_inside.p, n_query, hipMemcpyDeviceToHost));
    }
    return 0;
}
