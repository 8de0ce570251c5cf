// Sharp (diffracting) edges: polygons_sharp_edges_iter of the reference (geometry.py:482-565).
//
// One thread per unordered polygon pair (i < j); the pair list is appended to with an atomic cursor and put in the
// reference's (i, j) order on the host (there are at most two edges per footprint vertex, so the list is small).
#include <algorithm>
#include <cmath>
#include <numeric>

#include "rt_common.h"

struct SharpEdge {
    int32_t i, j;
    double e[6];
};

__device__ inline bool lex_less(d3 a, d3 b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
}

// numpy.allclose(a, b) with default tolerances, element by element: |a - b| <= 1e-8 + 1e-5 |b|
__device__ inline bool close3(d3 a, d3 b) {
    return fabs(a.x - b.x) <= 1e-8 + 1e-5 * fabs(b.x) && fabs(a.y - b.y) <= 1e-8 + 1e-5 * fabs(b.y) &&
           fabs(a.z - b.z) <= 1e-8 + 1e-5 * fabs(b.z);
}

__global__ __launch_bounds__(256) void k_sharp_edges(SceneView sc, double max_cos, SharpEdge *__restrict__ out,
                                                      int64_t cap, unsigned long long *counter) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= sc.n || j <= i) return;
    const double *ba = sc.bbox + 6 * (size_t)i, *bb = sc.bbox + 6 * (size_t)j;
    // geometry.py:518-521: centres further apart than the sum of the extents on some axis cannot touch
    for (int k = 0; k < 3; ++k) {
        const double ca = (ba[k] + ba[3 + k]) / 2, cb = (bb[k] + bb[3 + k]) / 2;
        if (fabs(ca - cb) > (ba[3 + k] - ba[k]) + (bb[3 + k] - bb[k])) return;
    }
    const d3 na = ld3(sc.planes + 4 * (size_t)i), nb = ld3(sc.planes + 4 * (size_t)j);
    if (fabs(dot(na, nb)) >= max_cos) return;  // geometry.py:523
    // geometry.py:526-546: do the normals drawn from the centres meet behind the faces?
    const d3 ca = mk3((ba[0] + ba[3]) / 2, (ba[1] + ba[4]) / 2, (ba[2] + ba[5]) / 2);
    const d3 cb = mk3((bb[0] + bb[3]) / 2, (bb[1] + bb[4]) / 2, (bb[2] + bb[5]) / 2);
    d3 fx, fy, fz;
    frame_from_vector(cross(na, nb), fx, fy, fz);
    const d3 a1 = ca + na, b1 = cb + nb;
    const double h0x = dot(ca, fx), h0y = dot(ca, fy), h1x = dot(a1, fx), h1y = dot(a1, fy);
    const double h2x = dot(cb, fx), h2y = dot(cb, fy), h3x = dot(b1, fx), h3y = dot(b1, fy);
    const d3 l1 = cross(mk3(h0x, h0y, 1.0), mk3(h1x, h1y, 1.0));
    const d3 l2 = cross(mk3(h2x, h2y, 1.0), mk3(h3x, h3y, 1.0));
    const d3 p = cross(l1, l2);
    if (p.z == 0) return;
    const double px = p.x / p.z, py = p.y / p.z;
    if (isnan(px) || isnan(py)) return;
    const double vx = h1x - h0x, vy = h1y - h0y;
    double t;
    if (vx != 0)
        t = (px - h0x) / vx;
    else if (vy != 0)
        t = (py - h0y) / vy;
    else
        return;
    if (!(t < 0)) return;
    // geometry.py:547-565: first pair of coinciding edges, endpoints sorted lexicographically
    const int a0 = sc.vert_off[i], an = sc.vert_off[i + 1] - a0;
    const int b0 = sc.vert_off[j], bn = sc.vert_off[j + 1] - b0;
    for (int k = 0; k < an; ++k) {
        d3 e0 = ld3(sc.verts + 3 * (size_t)(a0 + (k == 0 ? an - 1 : k - 1)));
        d3 e1 = ld3(sc.verts + 3 * (size_t)(a0 + k));
        if (lex_less(e1, e0)) {
            d3 tmp = e0;
            e0 = e1;
            e1 = tmp;
        }
        for (int m = 0; m < bn; ++m) {
            d3 g0 = ld3(sc.verts + 3 * (size_t)(b0 + (m == 0 ? bn - 1 : m - 1)));
            d3 g1 = ld3(sc.verts + 3 * (size_t)(b0 + m));
            if (lex_less(g1, g0)) {
                d3 tmp = g0;
                g0 = g1;
                g1 = tmp;
            }
            if (close3(e0, g0) && close3(e1, g1)) {
                const int64_t slot = (int64_t)atomicAdd(counter, 1ull);
                if (slot < cap) {
                    out[slot].i = i;
                    out[slot].j = j;
                    st3(out[slot].e, e0);
                    st3(out[slot].e + 3, e1);
                }
                return;
            }
        }
    }
}

extern "C" int rt_sharp_edges(rt_scene *scene, double min_angle_deg, int32_t *pairs, double *edges, int64_t capacity,
                              int64_t *count) {
    if (!scene || !count) RT_FAIL("rt_sharp_edges: null argument");
    if (capacity < 0 || (capacity > 0 && (!pairs || !edges))) RT_FAIL("rt_sharp_edges: bad output arrays");
    RT_HIP(rt_use_device(scene->device));
    const int n = scene->view.n;
    if (n > 65535) RT_FAIL("rt_sharp_edges: more than 65535 polygons");
    int64_t cap = std::max<int64_t>(capacity, 2 * (int64_t)scene->view.vtot + 16);
    SharpEdge *d_out = nullptr;
    unsigned long long *d_cnt = nullptr;
    RT_HIP(hipMalloc(&d_out, sizeof(SharpEdge) * cap));
    if (hipMalloc(&d_cnt, sizeof(unsigned long long)) != hipSuccess) {
        (void)hipFree(d_out);
        RT_FAIL("rt_sharp_edges: out of device memory");
    }
    std::vector<SharpEdge> h;
    unsigned long long found = 0;
    hipError_t e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), scene->stream);
    if (e == hipSuccess) {
        const double max_cos = std::cos(min_angle_deg * (M_PI / 180.0));
        hipLaunchKernelGGL(k_sharp_edges, dim3((n + 255) / 256, n), dim3(256), 0, scene->stream, scene->view, max_cos, d_out,
                           cap, d_cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&found, d_cnt, sizeof(found), hipMemcpyDeviceToHost, scene->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(scene->stream);
    if (e == hipSuccess) {
        h.resize(std::min<int64_t>((int64_t)found, cap));
        if (!h.empty()) e = hipMemcpy(h.data(), d_out, sizeof(SharpEdge) * h.size(), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_out);
    (void)hipFree(d_cnt);
    if (e != hipSuccess) RT_FAIL(std::string("rt_sharp_edges: ") + hipGetErrorString(e));
    if ((int64_t)found > cap) RT_FAIL("rt_sharp_edges: internal buffer too small");
    std::sort(h.begin(), h.end(), [](const SharpEdge &a, const SharpEdge &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    *count = (int64_t)found;
    for (int64_t k = 0; k < std::min<int64_t>((int64_t)found, capacity); ++k) {
        pairs[2 * k] = h[k].i;
        pairs[2 * k + 1] = h[k].j;
        for (int q = 0; q < 6; ++q) edges[6 * k + q] = h[k].e[q];
    }
    return 0;
}
