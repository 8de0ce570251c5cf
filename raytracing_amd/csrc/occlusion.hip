// Segment-versus-scene obstruction (reference geometry.py:836-898) and point containment (geometry.py:1778-1803).
//
// Work mapping: one lane per segment; every lane of a wave walks the same polygon, so the polygon's plane and box
// are wave-uniform and come through the scalar data path (no per-lane memory traffic at all in the inner loop).
// Per (segment, polygon) the loop does a plane crossing and a padded box test; only a lane whose crossing point lies
// in the polygon's padded box runs the exact test of the reference (projection along the segment, extent overlap,
// boundary-inclusive point-in-polygon, 1e-8 < t < 1-1e-8).  The padded pre-test can only let too many pairs through,
// never too few, so the result is the exact test's.
#include "rt_common.h"

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);  // one Newton step: plenty for a pre-test with a 1e-6 pad
}

template <bool MASKED>
__global__ __launch_bounds__(256) void k_obstruct(SceneView sc, const double *__restrict__ segs, int64_t n_seg,
                                                   const uint8_t *__restrict__ mask, int poly_chunk,
                                                   uint8_t *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = s < n_seg;
    d3 A = mk3(0, 0, 0), B = mk3(1, 0, 0);
    if (active) {
        A = ld3(segs + 6 * s);
        B = ld3(segs + 6 * s + 3);
    }
    const d3 V = B - A;
    const double pad = 1e-5 * fmax(fabs(V.x), fmax(fabs(V.y), fabs(V.z))) + 1e-6;
    bool hit = false;
    const int p_begin = blockIdx.y * poly_chunk;
    const int p_end = p_begin + poly_chunk < sc.n ? p_begin + poly_chunk : sc.n;
    for (int p = p_begin; p < p_end; ++p) {
        if (MASKED) {
            if (!mask[p]) continue;  // wave-uniform
        }
        const double *pl = sc.planes + 4 * (size_t)p;
        const double *bb = sc.bbox + 6 * (size_t)p;
        const double nx = pl[0], ny = pl[1], nz = pl[2], d = pl[3];
        const double den = V.x * nx + V.y * ny + V.z * nz;
        const double num = d + (A.x * nx + A.y * ny + A.z * nz);
        // t = -num/den must lie in (1e-8, 1 - 1e-8); compared without dividing, with room for rounding (1e-11 is
        // far above the error of num for coordinates of a few km and far below 1e-8 |den| unless the segment grazes
        // the plane): a segment that merely starts or ends on the polygon is rejected here
        const double ad = fabs(den);
        const double sn = den < 0.0 ? num : -num;
        bool cand = active && !hit && sn > 1e-8 * ad - 1e-11 && sn < (1.0 - 1e-8) * ad + 1e-11;
        const double t = sn * fast_rcp(ad);
        const double hx = A.x + t * V.x, hy = A.y + t * V.y, hz = A.z + t * V.z;
        cand = cand && hx >= bb[0] - pad && hx <= bb[3] + pad && hy >= bb[1] - pad && hy <= bb[4] + pad &&
               hz >= bb[2] - pad && hz <= bb[5] + pad;
        if (cand) {
            if (polygon_cuts_segment(sc, p, A, B)) hit = true;
        }
        if ((p & 31) == 31) {
            if (__all(hit || !active)) break;
        }
    }
    if (active && hit) out[s] = 1;  // out starts at zero; several polygon chunks may set the same byte
}

int rt_launch_obstruct(rt_scene *scene, const double *d_segs, int64_t n_seg, const uint8_t *d_mask, uint8_t *d_out) {
    if (n_seg == 0) return 0;
    const int block = 256;
    const int64_t grid = (n_seg + block - 1) / block;
    if (grid > 0x7fffffffLL) RT_FAIL("rt_obstruct_segments: too many segments for one launch");
    // few segments: split the polygon list over grid.y so the launch still fills the chip (target >= ~2048 waves)
    const int64_t waves = grid * (block / RT_WAVE);
    int64_t ny = waves >= 2048 ? 1 : (2048 + waves - 1) / waves;
    const int64_t max_ny = (scene->view.n + 31) / 32;
    ny = ny > max_ny ? max_ny : ny;
    ny = ny < 1 ? 1 : ny;
    int poly_chunk = (int)((scene->view.n + ny - 1) / ny);
    poly_chunk = (poly_chunk + 31) & ~31;  // keeps the early-exit test aligned
    ny = (scene->view.n + poly_chunk - 1) / poly_chunk;
    RT_HIP(hipMemsetAsync(d_out, 0, (size_t)n_seg, scene->stream));
    if (d_mask)
        hipLaunchKernelGGL(k_obstruct<true>, dim3((unsigned)grid, (unsigned)ny), dim3(block), 0, scene->stream,
                           scene->view, d_segs, n_seg, d_mask, poly_chunk, d_out);
    else
        hipLaunchKernelGGL(k_obstruct<false>, dim3((unsigned)grid, (unsigned)ny), dim3(block), 0, scene->stream,
                           scene->view, d_segs, n_seg, d_mask, poly_chunk, d_out);
    RT_HIP(hipGetLastError());
    return 0;
}

extern "C" int rt_obstruct_segments(rt_scene *scene, const double *segs, int64_t n_seg, const uint8_t *mask,
                                    uint8_t *out, int on_device) {
    if (!scene) RT_FAIL("rt_obstruct_segments: null scene");
    if (n_seg < 0) RT_FAIL("rt_obstruct_segments: negative count");
    if (n_seg == 0) return 0;
    if (!segs || !out) RT_FAIL("rt_obstruct_segments: null array");
    RT_HIP(hipSetDevice(scene->device));
    if (on_device) {
        if (rt_launch_obstruct(scene, segs, n_seg, mask, out)) return -1;
        RT_HIP(hipStreamSynchronize(scene->stream));
        return 0;
    }
    double *d_segs = nullptr;
    uint8_t *d_out = nullptr, *d_mask = nullptr;
    int rc = -1;
    do {
        if (hipMalloc(&d_segs, sizeof(double) * 6 * n_seg) != hipSuccess) break;
        if (hipMalloc(&d_out, n_seg) != hipSuccess) break;
        if (mask && hipMalloc(&d_mask, scene->view.n) != hipSuccess) break;
        if (hipMemcpyAsync(d_segs, segs, sizeof(double) * 6 * n_seg, hipMemcpyHostToDevice, scene->stream) != hipSuccess)
            break;
        if (mask && hipMemcpyAsync(d_mask, mask, scene->view.n, hipMemcpyHostToDevice, scene->stream) != hipSuccess) break;
        if (rt_launch_obstruct(scene, d_segs, n_seg, d_mask, d_out)) break;
        if (hipMemcpyAsync(out, d_out, n_seg, hipMemcpyDeviceToHost, scene->stream) != hipSuccess) break;
        if (hipStreamSynchronize(scene->stream) != hipSuccess) break;
        rc = 0;
    } while (0);
    if (rc) rt_set_error(std::string("rt_obstruct_segments: ") + hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_segs);
    (void)hipFree(d_out);
    (void)hipFree(d_mask);
    return rc;
}

// one thread per polygon, exact test only: the return_polygons=True form
__global__ void k_segment_hits(SceneView sc, d3 A, d3 B, uint8_t *__restrict__ out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < sc.n) out[p] = polygon_cuts_segment(sc, p, A, B) ? 1 : 0;
}

extern "C" int rt_segment_hits(rt_scene *scene, const double *seg, uint8_t *out) {
    if (!scene || !seg || !out) RT_FAIL("rt_segment_hits: null argument");
    RT_HIP(hipSetDevice(scene->device));
    uint8_t *d_out = nullptr;
    RT_HIP(hipMalloc(&d_out, scene->view.n));
    hipLaunchKernelGGL(k_segment_hits, dim3((scene->view.n + 255) / 256), dim3(256), 0, scene->stream, scene->view,
                       ld3(seg), ld3(seg + 3), d_out);
    hipError_t e = hipMemcpyAsync(out, d_out, scene->view.n, hipMemcpyDeviceToHost, scene->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(scene->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) RT_FAIL(std::string("rt_segment_hits: ") + hipGetErrorString(e));
    return 0;
}

__global__ void k_contains(SceneView sc, int p, const double *__restrict__ pts, int64_t n, int check_in_plane,
                           double plane_tol, uint8_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = polygon_contains_point(sc, p, ld3(pts + 3 * i), check_in_plane != 0, plane_tol) ? 1 : 0;
}

extern "C" int rt_contains_points(rt_scene *scene, int32_t polygon, const double *points, int64_t n_points,
                                  int check_in_plane, double plane_tol, uint8_t *out, int on_device) {
    if (!scene) RT_FAIL("rt_contains_points: null scene");
    if (polygon < 0 || polygon >= scene->view.n) RT_FAIL("rt_contains_points: polygon index out of range");
    if (n_points < 0) RT_FAIL("rt_contains_points: negative count");
    if (n_points == 0) return 0;
    if (!points || !out) RT_FAIL("rt_contains_points: null array");
    RT_HIP(hipSetDevice(scene->device));
    const unsigned grid = (unsigned)((n_points + 255) / 256);
    if (on_device) {
        hipLaunchKernelGGL(k_contains, dim3(grid), dim3(256), 0, scene->stream, scene->view, polygon, points, n_points,
                           check_in_plane, plane_tol, out);
        RT_HIP(hipGetLastError());
        RT_HIP(hipStreamSynchronize(scene->stream));
        return 0;
    }
    double *d_pts = nullptr;
    uint8_t *d_out = nullptr;
    RT_HIP(hipMalloc(&d_pts, sizeof(double) * 3 * n_points));
    hipError_t e = hipMalloc(&d_out, n_points);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts, points, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice, scene->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_contains, dim3(grid), dim3(256), 0, scene->stream, scene->view, polygon, d_pts, n_points,
                           check_in_plane, plane_tol, d_out);
        e = hipMemcpyAsync(out, d_out, n_points, hipMemcpyDeviceToHost, scene->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(scene->stream);
    (void)hipFree(d_pts);
    (void)hipFree(d_out);
    if (e != hipSuccess) RT_FAIL(std::string("rt_contains_points: ") + hipGetErrorString(e));
    return 0;
}
