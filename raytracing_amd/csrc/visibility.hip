// Face-to-scene visibility (reference geometry.py:924-1048, non-strict branch; strict = the z filter and the
// normal filter only).
//
// For a face A every other polygon B is moved to A's frame (origin = centre of A's bounding box), dropped unless it
// reaches z > 1e-4, mapped vertex by vertex to (azimuth, polar angle, distance), and the survivors are painted front
// to back in the order (nearest vertex distance, length of the centre of the mapped bounding box, index): B passes
// when its (azimuth, polar) polygon is a valid ring and some part of it is not covered by the polygons painted before
// it.  Last, B must face A: some vertex p of A has normalize(p - centre(B))·n_B > 1e-4.
//
// The canvas a polygon is tested against holds the polygons that passed before it; a polygon hidden by the canvas
// lies inside it, so leaving it out changes nothing and keeps the canvas small.  Faces are independent: one workgroup
// (four or sixteen waves) paints one face, walking its candidates in sorted order.
//
// Coverage is decided exactly with witness points.  If part of B is uncovered, the uncovered region is bounded by
// pieces of B's edges (region on B's inner side) and pieces of the C's edges (region on that C's outer side).  Every
// edge is cut at its crossings with all other edges and at the feet of endpoints that lie on it; along one piece
// nothing changes, so one point next to the middle of each piece (DELTA off the edge, on the side named above)
// decides it: B is visible iff some such point is inside B and outside every C.
//
// Kernels: k_vis_prepare (one thread per (face, polygon): mapping, keys, ring validity), k_vis_sort (one workgroup per
// face: bitonic sort of its candidates by the painter's key) and k_vis_paint (persistent workgroups pulling faces from
// a counter; the face's canvas lives in LDS, a candidate is staged next to it, the candidate's edges and the canvas
// edges that enter it are dealt to the waves, and inside a wave the lanes take single tests: one ring edge against the
// edge being cut, one witness point against one polygon).
#include <algorithm>
#include <cstring>

#include "rt_common.h"

#define TOL_DZ 1e-4      /* geometry.py:947 */
#define TOL_DOT 1e-4     /* geometry.py:948 */
#define VIS_DELTA 2.5e-10    /* witness offset from its edge, radians */
#define VIS_MIN_PIECE 1e-9   /* shorter pieces of an edge are not examined */
#define VIS_ON_LINE 1e-9     /* an endpoint this close to an edge's line cuts the edge at its foot */
#define VIS_RING_MARGIN 1e-8 /* ring edges this near a polygon's box are listed for its decision (the examined parts reach 4 VIS_ON_LINE out of the box, a cutting end lies VIS_ON_LINE from them) */

// Work counters of the painter (candidates, listed polygons, edges examined, pieces examined, shortcut hits, polygons
// shown visible, edge_has_witness calls from canvas edges, cut gathers), summed over a launch; read by the profiling scripts.
__device__ unsigned long long g_vis_stats[8];
__device__ unsigned long long g_vis_phase[16];
#ifdef RT_VIS_STAMPS
// (profiling builds) start, duration (10 ns ticks), candidates and visible polygons of every work item of the last launch
#define VIS_FACE_LOG 16384
__device__ unsigned long long g_vis_face_log[4 * VIS_FACE_LOG];
extern "C" int rt_debug_visibility_face_log(unsigned long long *out, int n) {
    if (n > VIS_FACE_LOG) n = VIS_FACE_LOG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vis_face_log), sizeof(unsigned long long) * 4 * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
#ifdef RT_VIS_STAMPS
#define VIS_HIST_ADD(i, v) atomicAdd(&g_vis_hist[i], v)
__device__ unsigned long long g_vis_hist[128];
extern "C" int rt_debug_visibility_hist(unsigned long long *out128) {
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_vis_hist), sizeof(g_vis_hist)) != hipSuccess) return -1;  // 128 words
    unsigned long long z[128] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vis_hist), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int rt_debug_visibility_phases(unsigned long long *out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_vis_phase), sizeof(g_vis_phase)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    z[15] = ~0ull;  // slot of a minimum
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vis_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
// Compiled in only for profiling builds (-DRT_VIS_STATS, ~170 M updates on a 7 615-polygon scene); otherwise
// rt_debug_visibility_stats reads zeros, except [6] candidates held against the covered-cell bitmap and [7] those it
// settled, which every build counts (two updates per 256 candidates).
#ifdef RT_VIS_STATS
#define VIS_STAT(i, v) atomicAdd(&g_vis_stats[i], (unsigned long long)(v))
#else
#define VIS_STAT(i, v) ((void)0)
#endif

struct PolyMeta {
    double key1, key2;
    double xmin, xmax, ymin, ymax;
    int32_t state;  // 0: not a candidate, 1: invalid ring, 2: candidate
    int32_t sgn;    // +1 counter-clockwise in (azimuth, polar), -1 clockwise
};

struct FaceBatch {
    int32_t n_faces;
    const int32_t *a_off;   // [n_faces+1] offsets into a_verts
    const double *a_verts;  // vertices of the faces
    const double *a_normal; // [n_faces*3]
    const int32_t *a_self;  // [n_faces] scene index of the face or -1
};

__device__ inline double orient2(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
__device__ inline bool in_box(double ax, double ay, double bx, double by, double px, double py) {
    return fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by);
}
__device__ inline bool segments_meet(double ax, double ay, double bx, double by, double cx, double cy, double dx,
                                     double dy) {
    const double o1 = orient2(ax, ay, bx, by, cx, cy), o2 = orient2(ax, ay, bx, by, dx, dy);
    const double o3 = orient2(cx, cy, dx, dy, ax, ay), o4 = orient2(cx, cy, dx, dy, bx, by);
    if (((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0))) return true;
    if (o1 == 0 && in_box(ax, ay, bx, by, cx, cy)) return true;
    if (o2 == 0 && in_box(ax, ay, bx, by, dx, dy)) return true;
    if (o3 == 0 && in_box(cx, cy, dx, dy, ax, ay)) return true;
    if (o4 == 0 && in_box(cx, cy, dx, dy, bx, by)) return true;
    return false;
}

// Valid ring: finite, >= 3 distinct vertices after dropping repeats, non-zero area, edges meet only at shared ends.
// (Index arithmetic wraps with a comparison, not a remainder: an integer division costs tens of instructions and this
// runs for every (face, polygon) pair.)
template <typename Coords>
__device__ bool ring_is_valid(Coords x, Coords y, int n_in, int *sgn) {
    for (int i = 0; i < n_in; ++i)
        if (!isfinite(x[i]) || !isfinite(y[i])) return false;
    // a vertex equal to its predecessor is a repeat and is skipped (repeated points are accepted)
    auto kept = [&](int i) {
        const int p = i == 0 ? n_in - 1 : i - 1;
        return !(x[i] == x[p] && y[i] == y[p]);
    };
    int n = 0;
    for (int i = 0; i < n_in; ++i) n += kept(i) ? 1 : 0;
    if (n < 3) return false;
    double area2 = 0;
    for (int i = 0; i < n_in; ++i) {
        const int j = i + 1 == n_in ? 0 : i + 1;
        area2 += x[i] * y[j] - x[j] * y[i];
    }
    if (area2 == 0) return false;
    *sgn = area2 > 0 ? 1 : -1;
    auto next_kept = [&](int i) {
        int j = i + 1 == n_in ? 0 : i + 1;
        while (!kept(j)) j = j + 1 == n_in ? 0 : j + 1;
        return j;
    };
    for (int i = 0; i < n_in; ++i) {
        if (!kept(i)) continue;
        const int i2 = next_kept(i);
        for (int j = i + 1; j < n_in; ++j) {
            if (!kept(j)) continue;
            const int j2 = next_kept(j);
            if (j == i2 || j2 == i) {
                // consecutive edges: only a fold-back onto each other is an intersection
                int a, b, c;
                if (j == i2) { a = i; b = i2; c = j2; } else { a = j; b = j2; c = i2; }
                const double cr = orient2(x[a], y[a], x[b], y[b], x[c], y[c]);
                const double dt = (x[b] - x[a]) * (x[c] - x[b]) + (y[b] - y[a]) * (y[c] - y[b]);
                if (cr == 0 && dt < 0) return false;
                continue;
            }
            if (segments_meet(x[i], y[i], x[i2], y[i2], x[j], y[j], x[j2], y[j2])) return false;
        }
    }
    return true;
}

// The same predicate decided by one wave: lanes take vertices, then edge pairs (i uniform, lanes over j > i).  x, y may
// be in LDS or global memory; kept and nxt are per-wave scratch of n_in ints each.  Wave-uniform result.
__device__ bool ring_is_valid_wave(const double *x, const double *y, int n_in, int *sgn, int *kept, int *nxt) {
    const int lane = threadIdx.x & 63;
    bool bad = false;
    int n = 0;
    for (int base = 0; base < n_in; base += 64) {
        const int i = base + lane;
        bool k = false;
        if (i < n_in) {
            if (!isfinite(x[i]) || !isfinite(y[i])) bad = true;
            const int p = i == 0 ? n_in - 1 : i - 1;
            k = !(x[i] == x[p] && y[i] == y[p]);
            kept[i] = k ? 1 : 0;
        }
        n += __popcll(__ballot(k));
    }
    if (__any(bad)) return false;
    if (n < 3) return false;
    __builtin_amdgcn_wave_barrier();
    double area2 = 0;  // summed in ring order by every lane alike: its sign and its being zero are what matter
    for (int i = 0; i < n_in; ++i) {
        const int j = i + 1 == n_in ? 0 : i + 1;
        area2 += x[i] * y[j] - x[j] * y[i];
    }
    if (area2 == 0) return false;
    *sgn = area2 > 0 ? 1 : -1;
    for (int base = 0; base < n_in; base += 64) {
        const int i = base + lane;
        if (i < n_in) {
            int j = i + 1 == n_in ? 0 : i + 1;
            while (!kept[j]) j = j + 1 == n_in ? 0 : j + 1;
            nxt[i] = j;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < n_in; ++i) {
        if (!kept[i]) continue;  // uniform
        const int i2 = nxt[i];
        for (int base = i + 1; base < n_in; base += 64) {
            const int j = base + lane;
            bool meet = false;
            if (j < n_in && kept[j]) {
                const int j2 = nxt[j];
                if (j == i2 || j2 == i) {
                    int a, b, c;
                    if (j == i2) { a = i; b = i2; c = j2; } else { a = j; b = j2; c = i2; }
                    const double cr = orient2(x[a], y[a], x[b], y[b], x[c], y[c]);
                    const double dt = (x[b] - x[a]) * (x[c] - x[b]) + (y[b] - y[a]) * (y[c] - y[b]);
                    meet = cr == 0 && dt < 0;
                } else {
                    meet = segments_meet(x[i], y[i], x[i2], y[i2], x[j], y[j], x[j2], y[j2]);
                }
            }
            if (__any(meet)) return false;
        }
    }
    return true;
}

__device__ inline void face_frame(const FaceBatch &fb, int f, d3 &fx, d3 &fy, d3 &fz, d3 &c) {
    const int v0 = fb.a_off[f], nv = fb.a_off[f + 1] - v0;
    const double *p = fb.a_verts + 3 * (size_t)v0;
    d3 lo = ld3(p), hi = ld3(p);
    for (int v = 1; v < nv; ++v) {
        d3 q = ld3(p + 3 * v);
        lo = mk3(fmin(lo.x, q.x), fmin(lo.y, q.y), fmin(lo.z, q.z));
        hi = mk3(fmax(hi.x, q.x), fmax(hi.y, q.y), fmax(hi.z, q.z));
    }
    c = mk3((lo.x + hi.x) / 2, (lo.y + hi.y) / 2, (lo.z + hi.z) / 2);
    fz = ld3(fb.a_normal + 3 * (size_t)f);
    fx = ld3(p + 3) - ld3(p);
    fy = cross(fz, fx);
    fx = divs(fx, norm3(fx));
    fy = divs(fy, norm3(fy));
}

// Polygons of up to VIS_SMALL_POLY vertices (walls are quads: most of a scene) are prepared one per thread; the others
// (roofs, up to a hundred vertices and more) one per wave, in k_vis_prepare_big: in a wave of 64 polygons one such roof
// used to keep 63 quads waiting through its vertex loop and its ring check, which is quadratic in the vertices.
#define VIS_SMALL_POLY 4
#define VIS_BIG_LDS 256 /* vertices of a polygon whose mapped ring is staged in LDS for the wave's ring check */

// one thread per (face, polygon)
__global__ __launch_bounds__(256) void k_vis_prepare(SceneView sc, FaceBatch fb, double *__restrict__ sph,
                                                      PolyMeta *__restrict__ meta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (b >= sc.n) return;
    const int v0 = sc.vert_off[b], nv = sc.vert_off[b + 1] - v0;
    if (nv > VIS_SMALL_POLY) return;  // k_vis_prepare_big writes its record
    PolyMeta m;
    m.state = 0;
    m.sgn = 1;
    m.key1 = m.key2 = 0;
    m.xmin = m.xmax = m.ymin = m.ymax = 0;
    PolyMeta *out = meta + (size_t)f * sc.n + b;
    if (b == fb.a_self[f]) {  // geometry.py:964
        *out = m;
        return;
    }
    d3 fx, fy, fz, c;
    face_frame(fb, f, fx, fy, fz, c);
    double *az = sph + ((size_t)f * sc.vtot + v0) * 2;
    double *pol = az + nv;
    double zmax = -INFINITY, rmin = INFINITY;
    double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    // the z filter first (geometry.py:967-970): about half of a scene lies behind a face, and those polygons are not
    // mapped (nobody reads the mapped vertices of a polygon that is not a candidate)
    d3 q[VIS_SMALL_POLY];
#pragma unroll
    for (int v = 0; v < VIS_SMALL_POLY; ++v) {
        q[v] = ld3(sc.verts + 3 * (size_t)(v0 + (v < nv ? v : 0))) - c;
        if (v < nv) zmax = fmax(zmax, dot(q[v], fz));
    }
    if (!(zmax > TOL_DZ)) {
        *out = m;
        return;
    }
    double xs[VIS_SMALL_POLY], ys[VIS_SMALL_POLY];
#pragma unroll
    for (int v = 0; v < VIS_SMALL_POLY; ++v) {
        const double qx = dot(q[v], fx), qy = dot(q[v], fy), qz = dot(q[v], fz);
        const double r = sqrt(qx * qx + qy * qy + qz * qz);  // geometry.py:302-306
        const double a = atan2(qy, qx);
        const double p = acos(qz / r);
        xs[v] = a;
        ys[v] = p;
        if (v < nv) {
            az[v] = a;
            pol[v] = p;
            rmin = fmin(rmin, r);
            lo0 = fmin(lo0, a); hi0 = fmax(hi0, a);
            lo1 = fmin(lo1, p); hi1 = fmax(hi1, p);
            lo2 = fmin(lo2, r); hi2 = fmax(hi2, r);
        }
    }
    const double cx = (lo0 + hi0) / 2, cy = (lo1 + hi1) / 2, cz = (lo2 + hi2) / 2;
    m.key1 = rmin;  // geometry.py:980-984
    m.key2 = sqrt(cx * cx + cy * cy + cz * cz);
    m.xmin = lo0; m.xmax = hi0; m.ymin = lo1; m.ymax = hi1;
    int sgn = 1;
    m.state = ring_is_valid(xs, ys, nv, &sgn) ? 2 : 1;
    m.sgn = sgn;
    *out = m;
}

// Polygons of a few more vertices (up to VIS_MID_POLY: most roofs): one thread per (face, polygon) again, but with the
// polygon the same in all lanes of a wave and the faces along the lanes, so that the loops over its vertices run the
// same number of times in every lane and its vertices come through the scalar path.  The mapped ring of each lane sits
// in LDS, vertex-major (coordinate v of lane l at [v][l]: no bank conflicts), for the ring check.  A wave per polygon
// (k_vis_prepare_big) spends most of its time reducing over lanes and waiting, with a dozen lanes busy.
#define VIS_MID_POLY 32
struct LaneColumn {  // coordinate i of this lane in a [.][64] array
    const double *p;
    __device__ double operator[](int i) const { return p[64 * i]; }
};
template <int NVM>
__global__ __launch_bounds__(64) void k_vis_prepare_mid(SceneView sc, FaceBatch fb, const int32_t *__restrict__ mid, int n_mid,
                                                         double *__restrict__ sph, PolyMeta *__restrict__ meta) {
    __shared__ double s_x[NVM][64], s_y[NVM][64];
    const int lane = threadIdx.x;
    const int f = blockIdx.x * 64 + lane;
    const int b = mid[blockIdx.y];  // the same polygon in every lane
    if (f >= fb.n_faces) return;
    const int v0 = sc.vert_off[b], nv = sc.vert_off[b + 1] - v0;
    PolyMeta m;
    m.state = 0;
    m.sgn = 1;
    m.key1 = m.key2 = 0;
    m.xmin = m.xmax = m.ymin = m.ymax = 0;
    PolyMeta *out = meta + (size_t)f * sc.n + b;
    if (b == fb.a_self[f]) {
        *out = m;
        return;
    }
    d3 fx, fy, fz, c;
    face_frame(fb, f, fx, fy, fz, c);
    double zmax = -INFINITY;
    for (int v = 0; v < nv; ++v) zmax = fmax(zmax, dot(ld3(sc.verts + 3 * (size_t)(v0 + v)) - c, fz));
    if (!(zmax > TOL_DZ)) {
        *out = m;
        return;
    }
    double *az = sph + ((size_t)f * sc.vtot + v0) * 2;
    double *pol = az + nv;
    double rmin = INFINITY;
    double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    for (int v = 0; v < nv; ++v) {
        const d3 q = ld3(sc.verts + 3 * (size_t)(v0 + v)) - c;
        const double qx = dot(q, fx), qy = dot(q, fy), qz = dot(q, fz);
        const double r = sqrt(qx * qx + qy * qy + qz * qz);
        const double a = atan2(qy, qx);
        const double p = acos(qz / r);
        az[v] = a;
        pol[v] = p;
        s_x[v][lane] = a;
        s_y[v][lane] = p;
        rmin = fmin(rmin, r);
        lo0 = fmin(lo0, a); hi0 = fmax(hi0, a);
        lo1 = fmin(lo1, p); hi1 = fmax(hi1, p);
        lo2 = fmin(lo2, r); hi2 = fmax(hi2, r);
    }
    const double cx = (lo0 + hi0) / 2, cy = (lo1 + hi1) / 2, cz = (lo2 + hi2) / 2;
    m.key1 = rmin;
    m.key2 = sqrt(cx * cx + cy * cy + cz * cz);
    m.xmin = lo0; m.xmax = hi0; m.ymin = lo1; m.ymax = hi1;
    int sgn = 1;
    m.state = ring_is_valid(LaneColumn{&s_x[0][lane]}, LaneColumn{&s_y[0][lane]}, nv, &sgn) ? 2 : 1;
    m.sgn = sgn;
    *out = m;
}

// one wave per (face, polygon of more than VIS_SMALL_POLY vertices): big[] lists those polygons
__global__ __launch_bounds__(256) void k_vis_prepare_big(SceneView sc, FaceBatch fb, const int32_t *__restrict__ big, int n_big,
                                                          double *__restrict__ sph, PolyMeta *__restrict__ meta) {
    __shared__ double s_xy[4][2 * VIS_BIG_LDS];
    __shared__ int s_kept[4][VIS_BIG_LDS], s_nxt[4][VIS_BIG_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wave;
    const int f = blockIdx.y;
    if (k >= n_big) return;
    const int b = big[k];
    PolyMeta m;
    m.state = 0;
    m.sgn = 1;
    m.key1 = m.key2 = 0;
    m.xmin = m.xmax = m.ymin = m.ymax = 0;
    PolyMeta *out = meta + (size_t)f * sc.n + b;
    if (b == fb.a_self[f]) {
        if (lane == 0) *out = m;
        return;
    }
    d3 fx, fy, fz, c;
    face_frame(fb, f, fx, fy, fz, c);
    const int v0 = sc.vert_off[b], nv = sc.vert_off[b + 1] - v0;
    double zmax = -INFINITY;
    for (int base = 0; base < nv; base += 64) {
        const int v = base + lane;
        if (v < nv) zmax = fmax(zmax, dot(ld3(sc.verts + 3 * (size_t)(v0 + v)) - c, fz));
    }
    for (int off = 32; off > 0; off >>= 1) zmax = fmax(zmax, __shfl_xor(zmax, off));
    if (!(zmax > TOL_DZ)) {
        if (lane == 0) *out = m;
        return;
    }
    double *az = sph + ((size_t)f * sc.vtot + v0) * 2;
    double *pol = az + nv;
    const bool staged = nv <= VIS_BIG_LDS;
    double *xs = staged ? s_xy[wave] : az, *ys = staged ? s_xy[wave] + nv : pol;
    double rmin = INFINITY;
    double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    for (int base = 0; base < nv; base += 64) {
        const int v = base + lane;
        if (v < nv) {
            const d3 q = ld3(sc.verts + 3 * (size_t)(v0 + v)) - c;
            const double qx = dot(q, fx), qy = dot(q, fy), qz = dot(q, fz);
            const double r = sqrt(qx * qx + qy * qy + qz * qz);
            const double a = atan2(qy, qx);
            const double p = acos(qz / r);
            az[v] = a;
            pol[v] = p;
            if (staged) {
                xs[v] = a;
                ys[v] = p;
            }
            rmin = fmin(rmin, r);
            lo0 = fmin(lo0, a); hi0 = fmax(hi0, a);
            lo1 = fmin(lo1, p); hi1 = fmax(hi1, p);
            lo2 = fmin(lo2, r); hi2 = fmax(hi2, r);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        rmin = fmin(rmin, __shfl_xor(rmin, off));
        lo0 = fmin(lo0, __shfl_xor(lo0, off)); hi0 = fmax(hi0, __shfl_xor(hi0, off));
        lo1 = fmin(lo1, __shfl_xor(lo1, off)); hi1 = fmax(hi1, __shfl_xor(hi1, off));
        lo2 = fmin(lo2, __shfl_xor(lo2, off)); hi2 = fmax(hi2, __shfl_xor(hi2, off));
    }
    const double cx = (lo0 + hi0) / 2, cy = (lo1 + hi1) / 2, cz = (lo2 + hi2) / 2;
    m.key1 = rmin;
    m.key2 = sqrt(cx * cx + cy * cy + cz * cz);
    m.xmin = lo0; m.xmax = hi0; m.ymin = lo1; m.ymax = hi1;
    int sgn = 1;
    bool valid;
    if (staged) {
        __builtin_amdgcn_wave_barrier();
        valid = ring_is_valid_wave(xs, ys, nv, &sgn, s_kept[wave], s_nxt[wave]);
    } else {
        // a ring too long for the wave's LDS scratch: one lane walks it, reading the mapped vertices back
        __threadfence_block();
        int ok = 0;
        if (lane == 0) ok = ring_is_valid(az, pol, nv, &sgn) ? 1 : 0;
        valid = __shfl(ok, 0) != 0;
        sgn = __shfl(sgn, 0);
    }
    m.state = valid ? 2 : 1;
    m.sgn = sgn;
    if (lane == 0) *out = m;
}

__device__ inline bool painted_before(const PolyMeta &c, int ci, const PolyMeta &b, int bi) {
    if (c.key1 != b.key1) return c.key1 < b.key1;
    if (c.key2 != b.key2) return c.key2 < b.key2;
    return ci < bi;
}

// even-odd point-in-ring
template <typename P>
__device__ inline bool pip_ring(P x, P y, int nv, double px, double py) {
    bool in = false;
    double ax = x[nv - 1], ay = y[nv - 1];
    for (int k = 0; k < nv; ++k) {
        const double bx = x[k], by = y[k];
        if ((ay > py) != (by > py)) {
            const double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
            if ((cr > 0.0) == (by > ay)) in = !in;
        }
        ax = bx;
        ay = by;
    }
    return in;
}

struct PaintScratch {
    int32_t *list;    // per wave [n]: canvas polygons whose box meets the current polygon's
    int32_t *canvas;  // per wave [n]: polygons painted so far
    double *cbox;     // per wave [n*4]: their boxes (xmin, xmax, ymin, ymax)
    double *splits;   // per wave [split_cap]
    int64_t split_cap;
    int32_t *work;    // per wave [3 * vtot]: (entry, vertex) of canvas edges that enter the current polygon, then the ring list
    int32_t *tlist;   // per workgroup [n]: candidates of the tile being painted
    uint32_t *grid;   // per workgroup [VIS_GRID_WORDS]: the covered-cell bitmap of the face being painted
    int32_t *chunk;   // per workgroup [1024]: the candidates of the current stretch that are left to paint
    int grid_in_tiles;  // the bitmap also inside the tiles of a face painted in pieces (four-wave workgroups)
};

// The polygons one coverage decision works on: entry 0 = the polygon B being decided, entries 1..count-1 = the canvas
// polygons whose box meets B's.  Two views with the same interface: straight from global memory, or a copy staged in
// LDS (a single wave chasing list -> record -> offsets -> coordinates through global memory is latency-bound; staged,
// the same loops run out of LDS).
struct GlobalSet {
    static constexpr bool flat = false;  // no running vertex count: loops go entry by entry
    __device__ int coord_offset(int) const { return 0; }
    __device__ const double *coords() const { return nullptr; }
    __device__ int total_vertices() const { return 0; }
    __device__ void locate(int, int &i, int &v) const { i = v = 0; }
    const SceneView *sc;
    const double *sph_f;
    const PolyMeta *meta_f;
    const int32_t *list;
    int B, n;
    __device__ int count() const { return n; }
    __device__ int poly(int i) const { return i == 0 ? B : list[i - 1]; }
    __device__ int nv(int i) const {
        const int P = poly(i);
        return sc->vert_off[P + 1] - sc->vert_off[P];
    }
    __device__ const double *x(int i) const { return sph_f + (size_t)sc->vert_off[poly(i)] * 2; }
    __device__ const double *y(int i) const { return x(i) + nv(i); }
    __device__ void box(int i, double &x0, double &x1, double &y0, double &y1) const {
        const PolyMeta &m = meta_f[poly(i)];
        x0 = m.xmin; x1 = m.xmax; y0 = m.ymin; y1 = m.ymax;
    }
    __device__ int sgn(int i) const { return meta_f[poly(i)].sgn; }
    __device__ uint32_t twin_mask(int) const { return 0u; }
};

#define LS_MAX_POLY 96
#define LS_MAX_COORD 1280  /* doubles: 640 vertices */
struct LocalStore {
    int nvs[LS_MAX_POLY], offs[LS_MAX_POLY], sgns[LS_MAX_POLY];
    double boxes[LS_MAX_POLY * 4];
    double xy[LS_MAX_COORD];
};
struct LocalSet {
    static constexpr bool flat = false;
    __device__ int coord_offset(int) const { return 0; }
    __device__ const double *coords() const { return nullptr; }
    __device__ int total_vertices() const { return 0; }
    __device__ void locate(int, int &i, int &v) const { i = v = 0; }
    LocalStore *st;
    int n;
    __device__ int count() const { return n; }
    __device__ int nv(int i) const { return st->nvs[i]; }
    __device__ const double *x(int i) const { return st->xy + st->offs[i]; }
    __device__ const double *y(int i) const { return st->xy + st->offs[i] + st->nvs[i]; }
    __device__ void box(int i, double &x0, double &x1, double &y0, double &y1) const {
        x0 = st->boxes[4 * i]; x1 = st->boxes[4 * i + 1]; y0 = st->boxes[4 * i + 2]; y1 = st->boxes[4 * i + 3];
    }
    __device__ int sgn(int i) const { return st->sgns[i]; }
    __device__ uint32_t twin_mask(int) const { return 0u; }
};

// The canvas of the face being painted, resident in LDS: every polygon found visible so far with its box and mapped
// vertices, so that a candidate only brings its own vertices in from global memory (the listed polygons of every
// candidate used to be gathered again and again).  The candidate is staged in the next free slot; when it turns out
// visible the slot is simply kept.  A face whose canvas outgrows the store continues on the staged / global views.
// Two sizes: a small store when many faces are painted at once (three workgroups per CU), a large one for a handful
// of faces that see a lot (the emitter's screens), one workgroup per CU.
template <int CS_MAX_POLY, int CS_MAX_COORD>
struct CanvasStore {
    static constexpr int max_poly = CS_MAX_POLY, max_coord = CS_MAX_COORD;  // coordinates in doubles
    // (narrow types: every byte of LDS counts here, three workgroups share a compute unit's)
    uint16_t nvs[CS_MAX_POLY], offs[CS_MAX_POLY];
    int8_t sgns[CS_MAX_POLY];
    // a list of any length, for the candidate that is decided on its own (the waves' own lists are short):
    uint16_t list[CS_MAX_POLY];      // slots listed for the candidate
    uint16_t lpre[CS_MAX_POLY + 2];  // vertices before entry i of the set (entry 0 = the candidate, then the listed)
    double boxes[CS_MAX_POLY * 4];
    double xy[CS_MAX_COORD];
    // bit v of twin[slot]: edge v (from vertex v - 1 to v) of the canvas polygon in that slot is also an edge of another
    // canvas polygon that covers the strip along its outer side (mark_twins); such an edge is not examined
    uint32_t twin[CS_MAX_POLY];
};
using SmallCanvasStore = CanvasStore<320, 3072>;
using LargeCanvasStore = CanvasStore<768, 6144>;
template <typename Store>
struct CanvasSet {
    // the vertices of all entries are numbered through lpre, so that a loop over every ring edge of every entry can
    // give each lane one edge (a roof has tens of vertices: one lane walking its ring is the slow way)
    static constexpr bool flat = true;
    __device__ int total_vertices() const { return lpre[n]; }
    __device__ void locate(int e, int &i, int &v) const {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (lpre[mid] <= e) lo = mid; else hi = mid - 1;
        }
        i = lo;
        v = e - lpre[lo];
    }
    const Store *st;
    const uint16_t *list, *lpre;  // the listed slots and the running vertex count (n + 1 values)
    int b_slot, n;
    __device__ int slot(int i) const { return i == 0 ? b_slot : list[i - 1]; }
    __device__ int count() const { return n; }
    __device__ int nv(int i) const { return st->nvs[slot(i)]; }
    __device__ const double *x(int i) const { return st->xy + st->offs[slot(i)]; }
    __device__ const double *y(int i) const {
        const int k = slot(i);
        return st->xy + st->offs[k] + st->nvs[k];
    }
    __device__ void box(int i, double &x0, double &x1, double &y0, double &y1) const {
        const int k = slot(i);
        x0 = st->boxes[4 * k]; x1 = st->boxes[4 * k + 1]; y0 = st->boxes[4 * k + 2]; y1 = st->boxes[4 * k + 3];
    }
    __device__ int sgn(int i) const { return st->sgns[slot(i)]; }
    __device__ uint32_t twin_mask(int i) const { return i == 0 ? 0u : st->twin[slot(i)]; }
    // where entry i's coordinates begin in the store (x first, then y), and the store's coordinates
    __device__ int coord_offset(int i) const { return st->offs[slot(i)]; }
    __device__ const double *coords() const { return st->xy; }
};

// A face with very many candidates is painted in tiles: rectangles of the (azimuth, polar) plane ("sectors" below: they
// began as azimuth intervals), each by its own workgroup, with the
// candidates whose box reaches into the sector and a canvas of the polygons found visible INSIDE the sector.  A polygon
// is visible iff part of it shows in some sector.  Inside a sector the uncovered part of B is bounded by pieces of B's
// edges, of canvas edges and of the sector's borders, so: every edge is also cut where it crosses a border, a
// witness only counts inside the sector, and the borders themselves are examined like edges (witness on the sector's
// inner side, inside B, outside every canvas polygon).  An unbounded span is the whole face.
// The part of a face's view that one work item paints: a rectangle of the (azimuth, polar) plane, open-ended where it
// is the outermost in a direction; a face painted whole has the unbounded one.
struct TileSpan {
    double x0, x1, y0, y1;
    __device__ bool bounded() const { return x0 > -INFINITY || x1 < INFINITY || y0 > -INFINITY || y1 < INFINITY; }
    __device__ bool holds(double x, double y) const { return x > x0 && x < x1 && y > y0 && y < y1; }
};

// Is the witness next to the middle of piece (t0, t1) of edge a->b, on side `side` (+1 = left of a->b), inside the
// tile, inside B and outside every listed polygon?
template <typename Set>
__device__ inline bool witness_uncovered(const Set &ps, double ax, double ay, double bx, double by, double t0, double t1,
                                         double side, bool need_in_b, const TileSpan &tile) {
    const double ex = bx - ax, ey = by - ay;
    const double len = sqrt(ex * ex + ey * ey);
    const double tm = 0.5 * (t0 + t1);
    const double wx = ax + tm * ex - side * VIS_DELTA * (ey / len);
    const double wy = ay + tm * ey + side * VIS_DELTA * (ex / len);
    if (!tile.holds(wx, wy)) return false;
    if (need_in_b) {
        if (!pip_ring(ps.x(0), ps.y(0), ps.nv(0), wx, wy)) return false;
    }
    const int n = ps.count();
    for (int i = 1; i < n; ++i) {
        double x0, x1, y0, y1;
        ps.box(i, x0, x1, y0, y1);
        if (wx < x0 || wx > x1 || wy < y0 || wy > y1) continue;
        if (pip_ring(ps.x(i), ps.y(i), ps.nv(i), wx, wy)) return false;
    }
    return true;
}

// Cuts of edge a->b by edge c->d: appended (as parameters along a->b, strictly inside (0,1)) to out; returns count.
__device__ inline int edge_cuts(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy,
                                double *out) {
    const double ex = bx - ax, ey = by - ay;
    const double len2 = ex * ex + ey * ey;
    const double len = sqrt(len2);
    const double sc_ = ex * (cy - ay) - ey * (cx - ax);  // signed area: c relative to line a->b
    const double sd_ = ex * (dy - ay) - ey * (dx - ax);
    int k = 0;
    if ((sc_ > 0 && sd_ < 0) || (sc_ < 0 && sd_ > 0)) {
        const double u = sc_ / (sc_ - sd_);
        const double qx = cx + u * (dx - cx), qy = cy + u * (dy - cy);
        const double t = ((qx - ax) * ex + (qy - ay) * ey) / len2;
        if (t > 0.0 && t < 1.0) out[k++] = t;
    }
    if (fabs(sc_) <= VIS_ON_LINE * len) {
        const double t = ((cx - ax) * ex + (cy - ay) * ey) / len2;
        if (t > 0.0 && t < 1.0) out[k++] = t;
    }
    if (fabs(sd_) <= VIS_ON_LINE * len) {
        const double t = ((dx - ax) * ex + (dy - ay) * ey) / len2;
        if (t > 0.0 && t < 1.0) out[k++] = t;
    }
    return k;
}

// a / d for small non-negative a (below 2^21) and a divisor that is the same for every lane: the hardware has no
// integer division, the compiler's expansion is some forty instructions, and these loops are bound by how many
// instructions one wave can issue.  (a + 0.5) / d is never within 0.5 / d of an integer, far more than the rounding.
struct SmallDivisor {
    int d;
    float inv;
    __device__ explicit SmallDivisor(int d_) : d(d_), inv(1.0f / (float)d_) {}
    __device__ int quotient(int a) const { return (int)(((float)a + 0.5f) * inv); }
    __device__ void split(int a, int &q, int &r) const {
        q = quotient(a);
        r = a - q * d;
    }
};

// Cut parameters of the edge being examined: the first SPLIT_LDS live in LDS, the rare overflow in global memory.
#define SPLIT_LDS 192
#define PIECE_LDS 64 /* pieces (or sample points) whose witnesses are tested with lanes = (piece, entry) pairs */
struct SplitBuf {
    double *lds;     // [SPLIT_LDS]
    double *glob;    // [cap]
    int64_t cap;
    double *px, *py; // [PIECE_LDS] witness points of the pieces of the edge being examined
    int *pstate;     // [PIECE_LDS] 1 while the witness is still a proof of visibility
    int32_t *ring;   // [2 vtot], one per workgroup: the ring edges near the polygon being decided, two words each (see polygon_uncovered, edge_has_witness)
    int32_t *work_shared;  // [2 vtot], one per workgroup: (entry, vertex) of the edges that enter the polygon being decided
    __device__ void put(int i, double t) const {
        if (i < SPLIT_LDS) lds[i] = t;
        else if (i < cap) glob[i] = t;
    }
    __device__ double get(int i) const { return i < SPLIT_LDS ? lds[i] : glob[i]; }
};

// Examine every piece of edge a->b of entry `owner` (witness side `side`): returns true (wave-uniform) when a witness
// proves B visible.
template <typename Set>
__device__ __forceinline__ bool edge_has_witness(const Set &ps, int owner, double ax, double ay, double bx, double by, double side,
                                 bool need_in_b, const SplitBuf &splits, int *lds_count, const TileSpan &tile,
                                 double t_lo = 0.0, double t_hi = 1.0, int n_ring = -1) {
    // (t_lo, t_hi): the part of the edge that is examined.  An edge of a canvas polygon is often many times as long
    // as B is wide, and a witness has to be inside B: only the part inside B's box is cut into pieces, and only the
    // entries that reach that part are looked at.
    const int lane = threadIdx.x & 63;
    const double ex = bx - ax, ey = by - ay;
    const double len = sqrt(ex * ex + ey * ey);
    if (!(len > VIS_MIN_PIECE)) return false;
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
    unsigned long long t_st = clock64();
    const bool ex_on = ((t_st >> 5) & 63ull) == 0ull;  // one examination in 64: every one of them would queue on five words
#define EX_STAMP(i)                                                          \
    do {                                                                     \
        const unsigned long long now_ = clock64();                           \
        if (lane == 0 && ex_on) atomicAdd(&g_vis_hist[86 + (i)], now_ - t_st); \
        t_st = now_;                                                         \
    } while (0)
#else
#define EX_STAMP(i)
#endif
    // ---- gather the cuts: lanes = entries, each walks its own ring
    if (lane == 0) {
        int k = 1;
        splits.put(0, t_lo);
        // the tile's borders cut the edge too (a piece must not straddle a border: its witness stands for all of it)
        if (ex != 0.0) {
            const double ta = (tile.x0 - ax) / ex, tb = (tile.x1 - ax) / ex;
            if (ta > t_lo && ta < t_hi) splits.put(k++, ta);
            if (tb > t_lo && tb < t_hi) splits.put(k++, tb);
        }
        if (ey != 0.0) {
            const double ta = (tile.y0 - ay) / ey, tb = (tile.y1 - ay) / ey;
            if (ta > t_lo && ta < t_hi) splits.put(k++, ta);
            if (tb > t_lo && tb < t_hi) splits.put(k++, tb);
        }
        *lds_count = k;
    }
    __builtin_amdgcn_wave_barrier();
    const int n = ps.count();
    // the examined part's own ends and box
    const double pax = ax + t_lo * ex, pay = ay + t_lo * ey, pbx = ax + t_hi * ex, pby = ay + t_hi * ey;
    if (Set::flat) {
        // lanes = the ring edges that the caller has listed (the ones that come near the polygon being decided, where
        // the examined part lies; no other can cut it): two words each, (entry, vertex) and (where the entry's
        // coordinates begin, how many vertices it has), so that an edge's ends are two reads away
        const double *xy = ps.coords();
        const double px0 = fmin(pax, pbx) - 2 * VIS_ON_LINE, px1 = fmax(pax, pbx) + 2 * VIS_ON_LINE;
        const double py0 = fmin(pay, pby) - 2 * VIS_ON_LINE, py1 = fmax(pay, pby) + 2 * VIS_ON_LINE;
        for (int base = 0; base < n_ring; base += 64) {
            const int e = base + lane;
            if (e >= n_ring) continue;
            const int2 rec = ((const int2 *)splits.ring)[e];
            const int li = rec.x >> 16, v = rec.x & 0xffff;
            if (li == owner) continue;
            const int nv = rec.y & 0xffff;
            const double *x = xy + (int)((unsigned)rec.y >> 16), *y = x + nv;
            const int u = v == 0 ? nv - 1 : v - 1;
            // (a ring edge cuts the part where they cross or where one of its ends lies within VIS_ON_LINE of it: its
            // own box then comes that near the part's)
            if (fmax(x[u], x[v]) < px0 || fmin(x[u], x[v]) > px1 || fmax(y[u], y[v]) < py0 || fmin(y[u], y[v]) > py1) continue;
            double t[3];
            int k = edge_cuts(ax, ay, bx, by, x[u], y[u], x[v], y[v], t);
            int kept = 0;
            for (int q = 0; q < k; ++q)
                if (t[q] > t_lo && t[q] < t_hi) t[kept++] = t[q];
            k = kept;
            if (k > 0) {
                const int slot = atomicAdd(lds_count, k);
                for (int q = 0; q < k; ++q) splits.put(slot + q, t[q]);
            }
        }
    } else
    for (int base = 0; base < n; base += 64) {
        const int li = base + lane;
        if (li < n && li != owner) {
            double x0, x1, y0, y1;
            ps.box(li, x0, x1, y0, y1);
            // the other ring can only cut this edge where their boxes meet
            const bool near = !(x1 < fmin(pax, pbx) - VIS_ON_LINE || x0 > fmax(pax, pbx) + VIS_ON_LINE ||
                                y1 < fmin(pay, pby) - VIS_ON_LINE || y0 > fmax(pay, pby) + VIS_ON_LINE);
            if (near) {
                const int nv = ps.nv(li);
                const double *x = ps.x(li), *y = ps.y(li);
                double cx = x[nv - 1], cy = y[nv - 1];
                for (int v = 0; v < nv; ++v) {
                    const double dx = x[v], dy = y[v];
                    double t[3];
                    int k = edge_cuts(ax, ay, bx, by, cx, cy, dx, dy, t);
                    int kept = 0;
                    for (int q = 0; q < k; ++q)
                        if (t[q] > t_lo && t[q] < t_hi) t[kept++] = t[q];
                    k = kept;
                    if (k > 0) {
                        const int slot = atomicAdd(lds_count, k);
                        for (int q = 0; q < k; ++q) splits.put(slot + q, t[q]);
                    }
                    cx = dx;
                    cy = dy;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    EX_STAMP(0);
    int n_splits = *lds_count;
    n_splits = __builtin_amdgcn_readfirstlane(n_splits);
    if (n_splits > SPLIT_LDS) __threadfence_block();
    if (n_splits > splits.cap) n_splits = (int)splits.cap;  // cannot happen: capacity is 3 cuts per scene edge + 1
    if (lane == 0) {
        VIS_STAT(7, 1);
    }
    // ---- every piece (t_j, next larger cut or 1)
    if (n_splits <= PIECE_LDS) {
        // lanes = pieces: the witness point of each; then lanes = (piece, entry) pairs: a witness outside B (when it has
        // to be inside) or inside another entry proves nothing.  Same tests as witness_uncovered, spread over the lanes
        // instead of one lane walking every entry.
        // (the cuts are few enough for a lane each: the next larger one is found from the lanes' registers, every lane
        // taking part, instead of every lane reading every cut from LDS)
        const double t0 = lane < n_splits ? splits.lds[lane] : INFINITY;
        double t1 = t_hi;
        {
            const long long bits = __double_as_longlong(t0);
            const int lo = (int)bits, hi = (int)(bits >> 32);
            for (int i = 0; i < n_splits; ++i) {
                const unsigned long long b = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(lo, i) |
                                             ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, i) << 32);
                const double ti = __longlong_as_double((long long)b);
                if (ti > t0 && ti < t1) t1 = ti;
            }
        }
        if (lane < n_splits) {
            const double tm = 0.5 * (t0 + t1);
            const double wx = ax + tm * ex - side * VIS_DELTA * (ey / len);
            const double wy = ay + tm * ey + side * VIS_DELTA * (ex / len);
            splits.px[lane] = wx;
            splits.py[lane] = wy;
            splits.pstate[lane] = ((t1 - t0) * len > VIS_MIN_PIECE && tile.holds(wx, wy)) ? 1 : 0;
        }
        EX_STAMP(1);
        // only an entry whose box reaches the edge's can hold a witness (they stand VIS_DELTA beside the edge): those
        // are listed first, in the part of the cut buffer that this few cuts leave free (B itself, when the witnesses
        // have to be inside it, is always listed)
        const int first = need_in_b ? 0 : 1;
        int *near_list = (int *)(splits.lds + PIECE_LDS);
        const int near_cap = (SPLIT_LDS - PIECE_LDS) * 2;
        int n_near = 0;
        {
            const double e_x0 = fmin(pax, pbx) - 2 * VIS_DELTA, e_x1 = fmax(pax, pbx) + 2 * VIS_DELTA;
            const double e_y0 = fmin(pay, pby) - 2 * VIS_DELTA, e_y1 = fmax(pay, pby) + 2 * VIS_DELTA;
            for (int base = first; base < n; base += 64) {
                const int li = base + lane;
                bool reaches = false;
                if (li < n) {
                    double x0, x1, y0, y1;
                    ps.box(li, x0, x1, y0, y1);
                    reaches = li == 0 || !(x1 < e_x0 || x0 > e_x1 || y1 < e_y0 || y0 > e_y1);
                }
                const unsigned long long mk = __ballot(reaches);
                if (reaches) {
                    const int slot = n_near + __popcll(mk & ((1ull << lane) - 1));
                    if (slot < near_cap) near_list[slot] = li;
                }
                n_near += __popcll(mk);
            }
        }
        const bool listed = n_near <= near_cap;
        __builtin_amdgcn_wave_barrier();
        EX_STAMP(2);
        const int per_piece = listed ? n_near : n - first;
        // Entry by entry, the pieces that are still uncovered at once: most pieces lie inside one of the first few
        // entries, and a piece found covered is not tested again -- the pieces left are counted again before every
        // trip, and a trip holds them against as many entries as the lanes allow (64 / pieces left); when none is
        // left the edge has no witness.
        int *alive_list = (int *)splits.lds;  // (the cuts themselves are no longer needed: the witness points are made)
        for (int at0 = 0; at0 < per_piece;) {
            const bool alive = lane < n_splits && splits.pstate[lane] != 0;
            const unsigned long long am = __ballot(alive);
            const int na = __popcll(am);
            if (na == 0) break;
            if (alive) alive_list[__popcll(am & ((1ull << lane) - 1))] = lane;
            __builtin_amdgcn_wave_barrier();
            const SmallDivisor by_alive(na);
            int per_trip = by_alive.quotient(64);
            if (per_trip > per_piece - at0) per_trip = per_piece - at0;
            if (lane < per_trip * na) {
                int eo, r;
                by_alive.split(lane, eo, r);
                const int j = alive_list[r];
                const int i = listed ? near_list[at0 + eo] : first + at0 + eo;
                const double wx = splits.px[j], wy = splits.py[j];
                if (i == 0) {
                    if (!pip_ring(ps.x(0), ps.y(0), ps.nv(0), wx, wy)) splits.pstate[j] = 0;
                } else {
                    double x0, x1, y0, y1;
                    ps.box(i, x0, x1, y0, y1);
                    if (!(wx < x0 || wx > x1 || wy < y0 || wy > y1) && pip_ring(ps.x(i), ps.y(i), ps.nv(i), wx, wy))
                        splits.pstate[j] = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
            at0 += per_trip;
        }
        __builtin_amdgcn_wave_barrier();
        EX_STAMP(3);
        return __any(lane < n_splits && splits.pstate[lane] != 0);
    }
    bool found = false;
    for (int base = 0; base < n_splits && !found; base += 64) {
        const int j = base + lane;
        bool mine = false;
        if (j < n_splits) {
            const double t0 = splits.get(j);
            double t1 = t_hi;
            for (int i = 0; i < n_splits; ++i) {
                const double ti = splits.get(i);
                if (ti > t0 && ti < t1) t1 = ti;
            }
            if ((t1 - t0) * len > VIS_MIN_PIECE) mine = witness_uncovered(ps, ax, ay, bx, by, t0, t1, side, need_in_b, tile);
        }
        found = __any(mine);
    }
    EX_STAMP(4);
    return found;
}

// Is entry 0 of `ps` (B) inside ONE other entry: all of B's vertices strictly inside it and no edge of it meeting B's
// ring?  Then B is hidden.  One wave; lanes = entries for the box test; then, for each entry whose box holds B's box,
// lanes = the single tests: (vertex of B, one of five points around it) inside the entry, and (edge of the entry, edge
// of B) apart.  Wave-uniform result.
template <typename Set>
__device__ inline bool inside_one_entry(const Set &ps) {
    const int lane = threadIdx.x & 63;
    const int n = ps.count();
    const int nv = ps.nv(0);
    const double *x = ps.x(0), *y = ps.y(0);
    double bx0, bx1, by0, by1;
    ps.box(0, bx0, bx1, by0, by1);
    bool hidden = false;
    for (int base = 1; base < n && !hidden; base += 64) {
        const int li = base + lane;
        bool holds = false;
        if (li < n) {
            double x0, x1, y0, y1;
            ps.box(li, x0, x1, y0, y1);
            holds = x0 < bx0 && x1 > bx1 && y0 < by0 && y1 > by1;
        }
        unsigned long long todo = __ballot(holds);
        while (todo != 0 && !hidden) {
            const int e = base + (__ffsll((long long)todo) - 1);  // wave-uniform
            todo &= todo - 1;
            const int cn = ps.nv(e);
            const double *cx = ps.x(e), *cy = ps.y(e);
            const int n_pips = nv * 5, n_meets = cn * nv;
            const SmallDivisor by_nv(nv);
            bool ok = true;
            for (int w0 = 0; w0 < n_pips + n_meets && ok; w0 += 64) {
                const int w = w0 + lane;
                bool mine = true;
                if (w < n_pips) {
                    // the vertex and four points 2e-9 around it must all be inside: strictly inside with margin
                    const int v = w / 5, k = w % 5;
                    const double ox = k == 1 ? 8 * VIS_DELTA : (k == 2 ? -8 * VIS_DELTA : 0.0);
                    const double oy = k == 3 ? 8 * VIS_DELTA : (k == 4 ? -8 * VIS_DELTA : 0.0);
                    mine = pip_ring(cx, cy, cn, x[v] + ox, y[v] + oy);
                } else if (w < n_pips + n_meets) {
                    const int m = w - n_pips;
                    int q, v;
                    by_nv.split(m, q, v);
                    const int p = q == 0 ? cn - 1 : q - 1, u = v == 0 ? nv - 1 : v - 1;
                    mine = !segments_meet(cx[p], cy[p], cx[q], cy[q], x[u], y[u], x[v], y[v]);
                }
                ok = __all(mine);
            }
            hidden = ok;
        }
    }
    return hidden;
}

// Is entry 0 of `ps` (B, convex) inside the union of TWO other convex entries A and C?  For convex A, C and B it is
// enough that B's ring lies in A u C: a point of B outside both would have every chord through it end once in A only
// and once in C only, and turning the chord half round swaps the ends, so some chord ends in both.  Each edge of B is
// clipped to A shrunk by VIS_TWO_MARGIN; what sticks out must lie in C shrunk likewise, which for a convex C is a
// test of the end points of those parts.  Two walls of a building share their corner edge exactly (the same mapped
// vertices, bit for bit): along such an edge nothing is shrunk - A ends where C starts - and a point where B's edge
// leaves A through it is not asked to be on C's side of it.  One wave, lanes = pairs (A, C) among the first
// VIS_TWO_ENTRIES listed; only rings of at most VIS_TWO_NV vertices take part.  A proof that fails says nothing.
#define VIS_TWO_ENTRIES 11
#define VIS_TWO_NV 8
#define VIS_TWO_MARGIN 1e-9
template <typename P>
__device__ inline bool ring_turns_one_way(P x, P y, int nv, double sgn) {
    for (int v = 0; v < nv; ++v) {
        const int u = v == 0 ? nv - 1 : v - 1, w = v + 1 == nv ? 0 : v + 1;
        if (sgn * ((x[v] - x[u]) * (y[w] - y[v]) - (y[v] - y[u]) * (x[w] - x[v])) < 0.0) return false;
    }
    return true;
}
// inside every edge of the convex ring by the margin; edge `skip` (its end vertex) is not asked, edge `closed` only
// has to hold without margin
template <typename P>
__device__ inline bool inside_shrunk(P x, P y, int nv, double sgn, double px, double py, int skip, int closed) {
    for (int k = 0; k < nv; ++k) {
        if (k == skip) continue;
        const int u = k == 0 ? nv - 1 : k - 1;
        const double ex = x[k] - x[u], ey = y[k] - y[u];
        const double cr = sgn * (ex * (py - y[u]) - ey * (px - x[u]));
        if (cr < 0.0) return false;
        if (k != closed && cr * cr < VIS_TWO_MARGIN * VIS_TWO_MARGIN * (ex * ex + ey * ey)) return false;
    }
    return true;
}
// One edge (u -> v) of B against the pair: clipped to A (shrunk, except along the shared edge `ea`), and what sticks
// out asked of C (shrunk; along the shared edge `ec` only closed, and not asked at all of a point where B's edge
// leaves A through the shared edge).
template <typename Set>
__device__ inline bool edge_in_pair(const Set &ps, int ia, int ic, int ea, int ec, int v) {
    const int nv = ps.nv(0), an = ps.nv(ia), cn = ps.nv(ic);
    const double *x = ps.x(0), *y = ps.y(0);
    const double *ax = ps.x(ia), *ay = ps.y(ia), *cx = ps.x(ic), *cy = ps.y(ic);
    const double sa = (double)ps.sgn(ia), sc = (double)ps.sgn(ic);
    const int u = v == 0 ? nv - 1 : v - 1;
    const double px = x[u], py = y[u], qx = x[v], qy = y[v];
    double t0 = 0.0, t1 = 1.0;
    int by0 = -1, by1 = -1;  // the edges of A that bound the part inside
    for (int k = 0; k < an; ++k) {
        const int w = k == 0 ? an - 1 : k - 1;
        const double ex = ax[k] - ax[w], ey = ay[k] - ay[w];
        const double back = k == ea ? 0.0 : VIS_TWO_MARGIN * sqrt(ex * ex + ey * ey);
        const double fp = sa * (ex * (py - ay[w]) - ey * (px - ax[w])) - back;
        const double fq = sa * (ex * (qy - ay[w]) - ey * (qx - ax[w])) - back;
        if (fp < 0.0 && fq < 0.0) {
            t0 = 2.0;
        } else if (fp < 0.0) {
            const double t = fp / (fp - fq);
            if (t > t0) { t0 = t; by0 = k; }
        } else if (fq < 0.0) {
            const double t = fp / (fp - fq);
            if (t < t1) { t1 = t; by1 = k; }
        }
    }
    const bool none = t0 > t1;  // nothing of this edge in A: both ends in C
    bool ok = true;
    if (none || t0 > 0.0) ok = inside_shrunk(cx, cy, cn, sc, px, py, -1, ec);
    if (ok && (none || t1 < 1.0)) ok = inside_shrunk(cx, cy, cn, sc, qx, qy, -1, ec);
    if (ok && !none && t0 > 0.0)
        ok = inside_shrunk(cx, cy, cn, sc, px + t0 * (qx - px), py + t0 * (qy - py), by0 == ea ? ec : -1, ec);
    if (ok && !none && t1 < 1.0)
        ok = inside_shrunk(cx, cy, cn, sc, px + t1 * (qx - px), py + t1 * (qy - py), by1 == ea ? ec : -1, ec);
    return ok;
}
// Wave-uniform.  Steps, each with its own meaning of a lane: (entry, vertex of B) - which vertices of B each entry
// holds; entry - is it a small convex ring; pair - do the two hold all of B's vertices between them (a necessary
// condition that leaves a pair or two); then per such pair: (edge of A, edge of C) for the shared edge and edge of B
// for the clipping.
template <typename Set>
__device__ inline bool inside_two_entries(const Set &ps) {
    const int lane = threadIdx.x & 63;
    const int n = ps.count(), nv = ps.nv(0);
    if (n < 3 || nv > VIS_TWO_NV) return false;
    if (!ring_turns_one_way(ps.x(0), ps.y(0), nv, (double)ps.sgn(0))) return false;
    int m = n - 1 < VIS_TWO_ENTRIES ? n - 1 : VIS_TWO_ENTRIES;
    if (m * nv > 64) m = 64 / nv;
    const double *x = ps.x(0), *y = ps.y(0);
    bool holds = false;
    if (lane < m * nv) {
        int e, v;
        SmallDivisor(nv).split(lane, e, v);
        e += 1;
        holds = inside_shrunk(ps.x(e), ps.y(e), ps.nv(e), (double)ps.sgn(e), x[v], y[v], -1, -1);
    }
    const unsigned long long held = __ballot(holds);
    bool usable = false;
    if (lane < m) {
        const int e = 1 + lane, cn = ps.nv(e);
        usable = cn <= VIS_TWO_NV && ring_turns_one_way(ps.x(e), ps.y(e), cn, (double)ps.sgn(e));
    }
    const unsigned long long convex = __ballot(usable);
    const unsigned long long all_b = (1ull << nv) - 1;
    bool wanted = false;
    int i = 1, j = 2;
    if (lane < m * (m - 1) / 2) {
        int rest = lane;
        while (rest >= m - i) {
            rest -= m - i;
            ++i;
        }
        j = i + 1 + rest;
        const unsigned long long hi = (held >> ((i - 1) * nv)) & all_b, hj = (held >> ((j - 1) * nv)) & all_b;
        wanted = ((convex >> (i - 1)) & (convex >> (j - 1)) & 1) != 0 && (hi | hj) == all_b;
    }
    unsigned long long todo = __ballot(wanted);
    while (todo != 0) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int ia = __shfl(i, src), ic = __shfl(j, src);
        const int an = ps.nv(ia), cn = ps.nv(ic);
        // the edge A and C share, if any (end vertex of the edge in each ring): the same two mapped vertices, bit for
        // bit, and the two interiors on different sides of it
        bool shared = false;
        if (lane < an * cn) {
            const double *ax = ps.x(ia), *ay = ps.y(ia), *cx = ps.x(ic), *cy = ps.y(ic);
            int k, q;
            SmallDivisor(cn).split(lane, k, q);
            const int u = k == 0 ? an - 1 : k - 1, w = q == 0 ? cn - 1 : q - 1;
            const bool same = ax[u] == cx[w] && ay[u] == cy[w] && ax[k] == cx[q] && ay[k] == cy[q];
            const bool turned = ax[u] == cx[q] && ay[u] == cy[q] && ax[k] == cx[w] && ay[k] == cy[w];
            const bool apart = ps.sgn(ia) != ps.sgn(ic);
            shared = (same && apart) || (turned && !apart);
        }
        const unsigned long long sh = __ballot(shared);
        int ea = -1, ec = -1;
        if (sh != 0) {
            const int at = __ffsll((long long)sh) - 1;
            ea = at / cn;
            ec = at % cn;
        }
        const bool ok = lane >= nv || edge_in_pair(ps, ia, ic, ea, ec, lane);
        if (__all(ok)) return true;
    }
    return false;
}

// Decide whether entry 0 of `ps` has a part outside the union of the other entries (wave-uniform result).
// Called by all waves of the workgroup with identical arguments (the result is uniform over the workgroup): the cheap
// shortcuts are computed by every wave, the edge examinations are dealt out to the waves and joined through `flag`.
template <typename Set>
__device__ __forceinline__ bool polygon_uncovered(const Set &ps, const SplitBuf &splits, int *lds_count, int32_t *work, int *flag,
                                  const TileSpan &tile, bool inside_one_done) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int n = ps.count();
    const int nv = ps.nv(0);
    const double *x = ps.x(0), *y = ps.y(0);
    double bx0, bx1, by0, by1;
    ps.box(0, bx0, bx1, by0, by1);
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
    unsigned long long t_prev = wall_clock64();
#define PHASE(i)                                                                                 \
    do {                                                                                         \
        const unsigned long long now_ = wall_clock64();                                          \
        if (threadIdx.x == 0) {                                                                  \
            atomicAdd(&g_vis_phase[2 * (i)], now_ - t_prev);                                     \
            atomicAdd(&g_vis_phase[2 * (i) + 1], 1ull);                                          \
        }                                                                                        \
        t_prev = now_;                                                                           \
    } while (0)
#else
#define PHASE(i)
#endif
    bool visible = false;
    // ---- shortcut: B inside ONE other entry is hidden.  This settles nine candidates in ten, so it comes first (the
    // caller may already have had one wave do it); the shortcut after it, which can only find B visible, never
    // contradicts it (a point of B is then a point of that entry).
    if (!inside_one_done && inside_one_entry(ps)) return false;
    PHASE(0);
    // ---- the ring edges that come near B's box, listed once per decision (the waves share the work; the list is
    // complete behind the barrier that ends the samples): everything that is examined further down lies in that box
    // (B's own edges; of another entry's edge the part inside the box and 4 VIS_ON_LINE more), a ring edge cuts an
    // examined part where they cross or where one of its ends lies within VIS_ON_LINE of the part, and the
    // examinations used to walk every ring edge of every entry, a dozen times per decision, to find those.
    if (Set::flat) {
        const int tv = ps.total_vertices();
        for (int base = 64 * wave; base < tv; base += 64 * n_waves) {
            const int e = base + lane;
            bool take = false;
            int li = 0, v = 0;
            if (e < tv) {
                ps.locate(e, li, v);
                const int cn = ps.nv(li);
                const double *cx = ps.x(li), *cy = ps.y(li);
                const int u = v == 0 ? cn - 1 : v - 1;
                const double ax = cx[u], ay = cy[u], ex = cx[v], ey = cy[v];
                take = !(fmax(ax, ex) < bx0 - VIS_RING_MARGIN || fmin(ax, ex) > bx1 + VIS_RING_MARGIN ||
                         fmax(ay, ey) < by0 - VIS_RING_MARGIN || fmin(ay, ey) > by1 + VIS_RING_MARGIN);
            }
            const unsigned long long mk = __ballot(take);
            if (mk != 0ull) {
                int at = 0;
                if (lane == 0) at = atomicAdd(flag + 1, __popcll(mk));
                at = __builtin_amdgcn_readfirstlane(at);
                if (take)
                    ((int2 *)splits.ring)[at + __popcll(mk & ((1ull << lane) - 1))] =
                        make_int2((li << 16) | v, (int)(((unsigned)ps.coord_offset(li) << 16) | (unsigned)ps.nv(li)));
            }
        }
        __threadfence_block();
    }
    // ---- shortcut: a few points inside B (vertex average and the points halfway to some vertices); one of them
    // outside every other entry proves B visible.  Lanes = entries.
    {
        double gx = 0, gy = 0;
        for (int v = 0; v < nv; ++v) {
            gx += x[v];
            gy += y[v];
        }
        gx /= nv;
        gy /= nv;
        const int n_samples = nv < 8 ? nv + 1 : 9;
        // lanes = samples: the point, and whether it is inside B; then lanes = (sample, entry) pairs
        if (lane < n_samples) {
            double px = lane == 0 ? gx : 0.5 * (gx + x[(lane - 1) * nv / (n_samples - 1)]);
            double py = lane == 0 ? gy : 0.5 * (gy + y[(lane - 1) * nv / (n_samples - 1)]);
            // a sample outside the tile proves nothing there: it is moved onto the middle lines of the part of the tile
            // that B's box reaches (and then has to be inside B like any other)
            if (!(px > tile.x0 && px < tile.x1)) px = 0.5 * (fmax(tile.x0, bx0) + fmin(tile.x1, bx1));
            if (!(py > tile.y0 && py < tile.y1)) py = 0.5 * (fmax(tile.y0, by0) + fmin(tile.y1, by1));
            splits.px[lane] = px;
            splits.py[lane] = py;
            splits.pstate[lane] = (tile.holds(px, py) && pip_ring(x, y, nv, px, py)) ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        // the samples are dealt to the waves (wave w holds samples w, w + n_waves, ... against every other entry): all
        // waves used to make all the tests, and a barrier costs less than three quarters of them
        const int per_sample = n - 1;
        const int my_samples = wave < n_samples ? (n_samples - wave + n_waves - 1) / n_waves : 0;
        const int n_pairs = my_samples * per_sample;
        // (entry by entry, the wave's samples at once: when the last of them has been found covered the rest of the
        // entries need not be looked at)
        const SmallDivisor by_entry(my_samples > 0 ? my_samples : 1);
        for (int base = 0; base < n_pairs; base += 64) {
            if (!__any(lane < my_samples && splits.pstate[wave + lane * n_waves] != 0)) break;
            const int pair = base + lane;
            if (pair < n_pairs) {
                int sidx, li;
                by_entry.split(pair, li, sidx);
                sidx = wave + sidx * n_waves;
                li += 1;
                if (splits.pstate[sidx]) {
                    const double px = splits.px[sidx], py = splits.py[sidx];
                    double x0, x1, y0, y1;
                    ps.box(li, x0, x1, y0, y1);
                    if (!(px < x0 || px > x1 || py < y0 || py > y1)) {
                        const int cn = ps.nv(li);
                        const double *cx = ps.x(li), *cy = ps.y(li);
                        // closed test: a point on another entry's boundary proves nothing
                        if (pip_ring(cx, cy, cn, px, py) || pip_ring(cx, cy, cn, px + VIS_DELTA, py + VIS_DELTA) ||
                            pip_ring(cx, cy, cn, px - VIS_DELTA, py - VIS_DELTA))
                            splits.pstate[sidx] = 0;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // (a sample without another entry to be held against, per_sample = 0, stands as it is: wave 0 reports those)
        const bool mine = lane < n_samples && (per_sample == 0 ? wave == 0 : lane % n_waves == wave) && splits.pstate[lane] != 0;
        // The flag says 2 for a sample, 1 for a witness on an edge: a wave that is quick to leave the barrier may find
        // a witness on one of B's own edges before a slow one has read the flag, and only a 2 may send a wave out of
        // here (every wave has to take the same way through the barriers).
        if (__any(mine) && lane == 0) {
            VIS_STAT(4, 1);
            *flag = 2;
        }
        __syncthreads();
        visible = *(volatile int *)flag == 2;
    }
    PHASE(1);
    if (visible) return true;
    const int n_ring = Set::flat ? *(volatile int *)(flag + 1) : -1;  // (listed before the samples, see there)
    // ---- B's own edges, witness on B's inner side
    const double sgn_b = (double)ps.sgn(0);
    for (int v = wave; v < nv; v += n_waves) {
        if (*(volatile int *)flag) break;
        const int u = v == 0 ? nv - 1 : v - 1;
        if (edge_has_witness(ps, 0, x[u], y[u], x[v], y[v], sgn_b, false, splits, lds_count, tile, 0.0, 1.0, n_ring) && lane == 0)
            *flag = 1;
    }
    // ---- the edges of the other entries that enter B (an endpoint inside B or a crossing with B's ring), for the step
    // after this one: every wave collects its share of the ring list into one list of the workgroup, which the barrier
    // below completes (for nothing when a witness has just been found; the order of the list changes no result)
    if (Set::flat && !*(volatile int *)flag) {
        for (int base = 64 * wave; base < n_ring; base += 64 * n_waves) {
            const int e = base + lane;
            bool rel = false;
            int li = 0, v = 0;
            if (e < n_ring) {
                const int packed = splits.ring[2 * e];
                li = packed >> 16;
                v = packed & 0xffff;
            }
            if (li != 0) {  // (entry 0 is B itself)
                const int cn = ps.nv(li);
                const double *cx = ps.x(li), *cy = ps.y(li);
                const int u = v == 0 ? cn - 1 : v - 1;
                const double ax = cx[u], ay = cy[u], ex = cx[v], ey = cy[v];
                // (an edge that a neighbouring canvas polygon shares and covers from outside has no witness: every one
                // would stand inside that neighbour, which is in the list whenever the edge reaches into B's box)
                const bool buried = v < 32 && ((ps.twin_mask(li) >> v) & 1u) != 0u;
                if (!buried && !(fmax(ax, ex) < bx0 || fmin(ax, ex) > bx1 || fmax(ay, ey) < by0 || fmin(ay, ey) > by1)) {
                    rel = pip_ring(x, y, nv, ax, ay) || pip_ring(x, y, nv, ex, ey);
                    for (int q = 0; q < nv && !rel; ++q) {
                        const int p = q == 0 ? nv - 1 : q - 1;
                        rel = segments_meet(ax, ay, ex, ey, x[p], y[p], x[q], y[q]);
                    }
                }
            }
            const unsigned long long mk = __ballot(rel);
            if (mk != 0ull) {
                int at = 0;
                if (lane == 0) at = atomicAdd(flag + 2, __popcll(mk));
                at = __builtin_amdgcn_readfirstlane(at);
                if (rel) {
                    const int slot = at + __popcll(mk & ((1ull << lane) - 1));
                    splits.work_shared[2 * slot] = li;
                    splits.work_shared[2 * slot + 1] = v;
                }
            }
        }
        __threadfence_block();
    }
    __syncthreads();
    PHASE(2);
    // (only a witness on one of B's own edges, a 1, sends the waves out here: a wave that is quick to leave the barrier
    // can have found a witness on another entry's edge, a 4, before a slow one reads the flag, and a wave that left here
    // on that would miss the barrier below while the quick one waits in it.  This was so from the first version of this
    // routine and went unseen because few decisions end with such a witness; cut into small tiles, whose borders are
    // examined like edges and often hold one, a painting in a dozen came out wrong or did not end.)
    if (*(volatile int *)flag == 1) return true;
    // ---- edges of the other entries that enter B (an endpoint inside B or a crossing with B's ring): collected with
    // lanes = entries, then examined one by one, witness on their outer side
    int n_work = 0;
    const int32_t *collected = work;
    if (Set::flat) {
        n_work = *(volatile int *)(flag + 2);  // (collected ahead of the barrier above, the waves sharing the work)
        collected = splits.work_shared;
    } else
    for (int base = 1; base < n; base += 64) {
        const int li = base + lane;
        const int cn = li < n ? ps.nv(li) : 0;
        const double *cx = li < n ? ps.x(li) : x, *cy = li < n ? ps.y(li) : y;
        for (int v = 0;; ++v) {
            const bool act = v < cn;
            if (!__any(act)) break;
            bool rel = false;
            if (act) {
                const int u = v == 0 ? cn - 1 : v - 1;
                const double ax = cx[u], ay = cy[u], ex = cx[v], ey = cy[v];
                if (!(fmax(ax, ex) < bx0 || fmin(ax, ex) > bx1 || fmax(ay, ey) < by0 || fmin(ay, ey) > by1)) {
                    rel = pip_ring(x, y, nv, ax, ay) || pip_ring(x, y, nv, ex, ey);
                    for (int q = 0; q < nv && !rel; ++q) {
                        const int p = q == 0 ? nv - 1 : q - 1;
                        rel = segments_meet(ax, ay, ex, ey, x[p], y[p], x[q], y[q]);
                    }
                }
            }
            const unsigned long long mk = __ballot(rel);
            if (rel) {
                const int slot = n_work + __popcll(mk & ((1ull << lane) - 1));
                work[2 * slot] = li;
                work[2 * slot + 1] = v;
            }
            n_work += __popcll(mk);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    PHASE(3);
    if (lane == 0 && wave == 0) VIS_STAT(2, n_work);
    // (the tile's four borders, where they pass through B's box, are examined with them: witness on the tile's inner
    // side, inside B like the others; the vertical ones a->b upwards, so side +1 is the side of smaller azimuth; the
    // horizontal ones a->b towards larger azimuth, so side +1 is the side of the larger polar angle)
    const int n_border = tile.bounded() ? 4 : 0;
    for (int w = wave; w < n_work + n_border; w += n_waves) {
        if (*(volatile int *)flag) break;
        int li = -1;
        double ax, ay, ex_, ey_, side;
        if (w < n_work) {
            li = collected[2 * w];
            const int v = collected[2 * w + 1];
            const int cn = ps.nv(li);
            const double *cx = ps.x(li), *cy = ps.y(li);
            const int u = v == 0 ? cn - 1 : v - 1;
            ax = cx[u]; ay = cy[u]; ex_ = cx[v]; ey_ = cy[v];
            side = -(double)ps.sgn(li);
        } else if (w < n_work + 2) {
            const double t = w == n_work ? tile.x0 : tile.x1;
            if (!(t > bx0 && t < bx1)) continue;
            ax = t; ay = by0 - 1e-6; ex_ = t; ey_ = by1 + 1e-6;
            side = w == n_work ? -1.0 : 1.0;
        } else {
            const double t = w == n_work + 2 ? tile.y0 : tile.y1;
            if (!(t > by0 && t < by1)) continue;
            ax = bx0 - 1e-6; ay = t; ex_ = bx1 + 1e-6; ey_ = t;
            side = w == n_work + 2 ? 1.0 : -1.0;
        }
        if (lane == 0) VIS_STAT(6, 1);
        // the part of the edge inside B's box (a little more), by clipping against the box's four sides
        double t_lo = 0.0, t_hi = 1.0;
        {
            const double dx = ex_ - ax, dy = ey_ - ay, pad = 4 * VIS_ON_LINE;
            const double lo_x = bx0 - pad, hi_x = bx1 + pad, lo_y = by0 - pad, hi_y = by1 + pad;
            if (dx != 0.0) {
                const double ta = (lo_x - ax) / dx, tb = (hi_x - ax) / dx;
                t_lo = fmax(t_lo, fmin(ta, tb));
                t_hi = fmin(t_hi, fmax(ta, tb));
            } else if (ax < lo_x || ax > hi_x) {
                t_hi = -1.0;
            }
            if (dy != 0.0) {
                const double ta = (lo_y - ay) / dy, tb = (hi_y - ay) / dy;
                t_lo = fmax(t_lo, fmin(ta, tb));
                t_hi = fmin(t_hi, fmax(ta, tb));
            } else if (ay < lo_y || ay > hi_y) {
                t_hi = -1.0;
            }
        }
        if (!(t_hi > t_lo)) continue;  // it passes B's box by
#ifdef RT_VIS_STAMPS
        const unsigned long long t_ex0 = wall_clock64();
#endif
        const bool w_found = edge_has_witness(ps, li, ax, ay, ex_, ey_, side, true, splits, lds_count, tile, t_lo, t_hi, n_ring);
#ifdef RT_VIS_STAMPS
        if (lane == 0) {
            const unsigned long long dt = wall_clock64() - t_ex0;
            const int ns = *lds_count;  // cuts of the examined part
            int b = 0;
            while (b < 7 && (1 << b) < ns) ++b;  // 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, more
            int c = 0;
            while (c < 3 && (8 << c) < n) ++c;  // entries: up to 8, 16, 32, more
            atomicAdd(&g_vis_hist[8 * c + b], dt);
            atomicAdd(&g_vis_hist[32 + 8 * c + b], 1ull);
        }
#endif
        if (w_found && lane == 0) *flag = 4;
    }
    __syncthreads();
    PHASE(4);
    return *(volatile int *)flag != 0;
}

// painter's order (geometry.py:980-984; Python's sort is stable, ties keep the scene order); -1 sorts last
__device__ inline bool order_less(const PolyMeta *meta_f, int a, int b) {
    if (a < 0) return false;
    if (b < 0) return true;
    return painted_before(meta_f[a], a, meta_f[b], b);
}

// one workgroup per face: candidates (valid rings in front of the face) sorted by the painter's key.  Up to
// VIS_SORT_LDS candidates the keys are brought into LDS once and the bitonic network runs there (a compare used to read
// two 64-byte records from global memory); a face with more candidates sorts through global memory as before.
#define VIS_SORT_LDS 8192
#define VIS_SORT_THREADS 1024
__global__ __launch_bounds__(VIS_SORT_THREADS) void k_vis_sort(int n, const PolyMeta *__restrict__ meta,
                                                               int32_t *__restrict__ order, int m_stride,
                                                               int32_t *__restrict__ counts) {
    __shared__ int cnt;
    __shared__ double s_k1[VIS_SORT_LDS];  // the second key is read from the records, on a tie of the first only
    __shared__ int32_t s_id[VIS_SORT_LDS];
    const int f = blockIdx.x, tid = threadIdx.x;
    const PolyMeta *meta_f = meta + (size_t)f * n;
    int32_t *ord = order + (size_t)f * m_stride;
    // the candidates, in scene order: a wave's lanes count themselves with a ballot and the wave takes its places with
    // one addition (one addition per candidate on one LDS word took longer than the sort)
    __shared__ int s_wave_cnt[VIS_SORT_THREADS / 64];
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int base = 0; base < n; base += VIS_SORT_THREADS) {
        const int b = base + tid;
        const bool take = b < n && meta_f[b].state == 2;
        const unsigned long long mk = __ballot(take);
        if ((tid & 63) == 0) s_wave_cnt[tid >> 6] = __popcll(mk);
        __syncthreads();
        int at = cnt + __popcll(mk & ((1ull << (tid & 63)) - 1));
        for (int w = 0; w < (tid >> 6); ++w) at += s_wave_cnt[w];
        if (take) ord[at] = b;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < VIS_SORT_THREADS / 64; ++w) tot += s_wave_cnt[w];
            cnt += tot;
        }
        __syncthreads();
    }
    const int m = cnt;
    int m_pad = 1;
    while (m_pad < m) m_pad <<= 1;
    if (m_pad <= VIS_SORT_LDS) {
        // -1 (padding) sorts last: key1 = +infinity, and among equals the larger index loses
        for (int i = tid; i < m_pad; i += VIS_SORT_THREADS) {
            const int b = i < m ? ord[i] : -1;
            s_id[i] = b;
            s_k1[i] = b >= 0 ? meta_f[b].key1 : INFINITY;
        }
        __syncthreads();
        auto less = [&](int i, int l) {  // entry i before entry l?
            const int a = s_id[i], b = s_id[l];
            if (a < 0) return false;
            if (b < 0) return true;
            if (s_k1[i] != s_k1[l]) return s_k1[i] < s_k1[l];
            const double a2 = meta_f[a].key2, b2 = meta_f[b].key2;
            if (a2 != b2) return a2 < b2;
            return a < b;
        };
        for (int k = 2; k <= m_pad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                // a thread per pair (i, i + j): i = the pair's number with a zero put in at bit j
                for (int p = tid; p < (m_pad >> 1); p += VIS_SORT_THREADS) {
                    const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    const int l = i | j;
                    const bool up = (i & k) == 0;
                    if (up ? less(l, i) : less(i, l)) {
                        const int ti = s_id[i];
                        const double t1 = s_k1[i];
                        s_id[i] = s_id[l]; s_k1[i] = s_k1[l];
                        s_id[l] = ti; s_k1[l] = t1;
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < m_pad; i += VIS_SORT_THREADS) ord[i] = s_id[i];
        if (tid == 0) counts[f] = m;
        return;
    }
    for (int i = m + tid; i < m_pad; i += VIS_SORT_THREADS) ord[i] = -1;
    __syncthreads();
    for (int k = 2; k <= m_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m_pad; i += VIS_SORT_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const int a = ord[i], b = ord[l];
                    const bool up = (i & k) == 0;
                    if (up ? order_less(meta_f, b, a) : order_less(meta_f, a, b)) {
                        ord[i] = b;
                        ord[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) counts[f] = m;
}

__device__ inline bool faces_viewer(const SceneView &sc, const FaceBatch &fb, int f, int B) {
    // geometry.py:1004-1014
    const double *bb = sc.bbox + 6 * (size_t)B;
    const d3 cb = mk3((bb[0] + bb[3]) / 2, (bb[1] + bb[4]) / 2, (bb[2] + bb[5]) / 2);
    const d3 nb = ld3(sc.planes + 4 * (size_t)B);
    const int a0 = fb.a_off[f], an = fb.a_off[f + 1] - a0;
    for (int v = 0; v < an; ++v) {
        d3 dvec = ld3(fb.a_verts + 3 * (size_t)(a0 + v)) - cb;
        dvec = divs(dvec, norm3(dvec));
        if (dot(dvec, nb) > TOL_DOT) return true;
    }
    return false;
}

// The order in which the persistent workgroups take the faces: the ones with the most candidates first (1 024 classes of
// the candidate count, most first; inside a class in whatever order the atomics give, which changes no result: faces are
// independent).  A face with thousands of candidates lasts a hundred times as long as a wall with a few dozen, and taken
// last it would keep one compute unit busy long after the others have run out of faces.
// (A face in tiles keeps its place: placed by the candidates of one tile instead, the faces painted whole start earlier
// and the tiles later, and a share of lln's rows takes longer, an eighth 85-100 ms instead of 78-82.)
__global__ __launch_bounds__(1024) void k_face_queue(const int32_t *__restrict__ counts, int n_faces, int n_poly,
                                                      int32_t *__restrict__ queue) {
    __shared__ int hist[1024];
    __shared__ int start[1024];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const long long scale = n_poly > 0 ? n_poly : 1;
    auto size_of = [&](int f) { return counts[f]; };
    for (int f = threadIdx.x; f < n_faces; f += 1024) {
        const int cls = (int)((long long)size_of(f) * 1023 / scale);
        atomicAdd(&hist[1023 - (cls < 0 ? 0 : (cls > 1023 ? 1023 : cls))], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < 1024; ++b) {
            start[b] = acc;
            acc += hist[b];
        }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < n_faces; f += 1024) {
        const int cls = (int)((long long)size_of(f) * 1023 / scale);
        queue[atomicAdd(&start[1023 - (cls < 0 ? 0 : (cls > 1023 ? 1023 : cls))], 1)] = f;
    }
}

#define VIS_MAX_TILES 128
#define VIS_TILE_BINS 1024
#define VIS_TILE_THREADS 1024

// Tiles of every face: a face with fewer than min_candidates candidates is one unbounded tile; a larger one gets up to
// ceil(m / per_tile) rectangles of the (azimuth, polar) plane (at most VIS_MAX_TILES) that hold about the same number of
// candidate centres each: the tile with the most centres is cut at their median (a 1 024-bin histogram over their
// range) along azimuth or along the polar angle, whichever cut has fewer of the tile's boxes lying across it, until no
// tile holds more than per_tile.  Azimuth alone does
// not do: a wall looks along the ground, every far building lies on the horizon line, and that line has ONE azimuth
// from the wall's centre (sectors of a side-looking screen of lln's emitter held 1 000 to 1 300 candidates each where
// 48 were asked for).  bounds[4 (f VIS_MAX_TILES + t)] .. + 3 = x0, x1, y0, y1 of tile t; the outermost tiles are
// open-ended.  One workgroup per face.
__global__ __launch_bounds__(VIS_TILE_THREADS) void k_vis_tiles(int n, const PolyMeta *__restrict__ meta, const int32_t *__restrict__ order,
                                                    int m_stride, const int32_t *__restrict__ counts, int min_candidates,
                                                    int per_tile, const int32_t *__restrict__ queue, int top_faces,
                                                    int top_per_tile, double *__restrict__ bounds,
                                                    int32_t *__restrict__ n_tiles) {
    __shared__ int hist[2 * VIS_TILE_BINS];  // along azimuth, then along the polar angle
    __shared__ double tb[VIS_MAX_TILES][4];
    __shared__ int tcnt[VIS_MAX_TILES];  // centres in the tile; negative once it cannot be cut further
    __shared__ double red[VIS_TILE_THREADS / 64][4];
    __shared__ double s_lo[2], s_scale[2], s_range[2], s_cut[2];
    __shared__ int s_sel, s_n, s_bin[2], s_cum[2], s_straddle[2];
    __shared__ int s_wsum[VIS_TILE_THREADS / 64];
    static_assert(VIS_TILE_BINS == VIS_TILE_THREADS, "one bin of the histogram a thread");
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = counts[f];
    double *b = bounds + 4 * (size_t)f * VIS_MAX_TILES;
    int T = 1, per = per_tile;
    if (m >= min_candidates && per_tile > 0) T = min(VIS_MAX_TILES, (m + per_tile - 1) / per_tile);
    // the first top_faces faces of the queue (those with the most candidates) go in tiles of top_per_tile whatever
    // the rule above says
    if (T <= 1 && top_per_tile > 0 && m >= 2 * top_per_tile) {
        bool top = false;
        for (int i = 0; i < top_faces; ++i) top = top || queue[i] == f;
        if (top) {
            T = min(VIS_MAX_TILES, (m + top_per_tile - 1) / top_per_tile);
            per = top_per_tile;
        }
    }
    if (T <= 1) {
        if (tid == 0) {
            b[0] = -INFINITY;
            b[1] = INFINITY;
            b[2] = -INFINITY;
            b[3] = INFINITY;
            n_tiles[f] = 1;
        }
        return;
    }
    const PolyMeta *meta_f = meta + (size_t)f * n;
    const int32_t *ord = order + (size_t)f * m_stride;
    // the centres and half widths of this thread's candidates' boxes: in registers when eight a thread are enough (every
    // cut reads them three times)
    const bool cached = m <= 8 * VIS_TILE_THREADS;
    double ccx[8], ccy[8];
    float chx[8], chy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = tid + j * VIS_TILE_THREADS;
        ccx[j] = ccy[j] = NAN;  // (no tile holds it)
        chx[j] = chy[j] = 0.0f;
        if (cached && k < m) {
            const PolyMeta &mb = meta_f[ord[k]];
            ccx[j] = 0.5 * (mb.xmin + mb.xmax);
            ccy[j] = 0.5 * (mb.ymin + mb.ymax);
            chx[j] = (float)(0.5 * (mb.xmax - mb.xmin));
            chy[j] = (float)(0.5 * (mb.ymax - mb.ymin));
        }
    }
    // fn(centre x, centre y, half width x, half width y) for each of this thread's candidates
    auto each = [&](auto fn) {
        if (cached) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fn(ccx[j], ccy[j], (double)chx[j], (double)chy[j]);
        } else {
            for (int k = tid; k < m; k += VIS_TILE_THREADS) {
                const PolyMeta &mb = meta_f[ord[k]];
                fn(0.5 * (mb.xmin + mb.xmax), 0.5 * (mb.ymin + mb.ymax), 0.5 * (mb.xmax - mb.xmin), 0.5 * (mb.ymax - mb.ymin));
            }
        }
    };
    if (tid == 0) {
        tb[0][0] = -INFINITY;
        tb[0][1] = INFINITY;
        tb[0][2] = -INFINITY;
        tb[0][3] = INFINITY;
        tcnt[0] = m;
        s_n = 1;
    }
    __syncthreads();
    for (int iter = 0; iter < 4 * VIS_MAX_TILES; ++iter) {
        if (tid == 0) {
            int best = -1, most = per;  // only a tile with more than `per` centres is cut
            for (int i = 0; i < s_n; ++i)
                if (tcnt[i] > most) {
                    most = tcnt[i];
                    best = i;
                }
            s_sel = s_n < T ? best : -1;
        }
        __syncthreads();
        const int sel = s_sel;
        if (sel < 0) break;
        const double x0 = tb[sel][0], x1 = tb[sel][1], y0 = tb[sel][2], y1 = tb[sel][3];
        // (a centre belongs to the tile whose half-open rectangle (x0, x1] x (y0, y1] holds it)
        double lo_x = INFINITY, hi_x = -INFINITY, lo_y = INFINITY, hi_y = -INFINITY;
        each([&](double cx, double cy, double, double) {
            if (cx > x0 && cx <= x1 && cy > y0 && cy <= y1) {
                lo_x = fmin(lo_x, cx);
                hi_x = fmax(hi_x, cx);
                lo_y = fmin(lo_y, cy);
                hi_y = fmax(hi_y, cy);
            }
        });
        for (int off = 32; off > 0; off >>= 1) {
            lo_x = fmin(lo_x, __shfl_xor(lo_x, off));
            hi_x = fmax(hi_x, __shfl_xor(hi_x, off));
            lo_y = fmin(lo_y, __shfl_xor(lo_y, off));
            hi_y = fmax(hi_y, __shfl_xor(hi_y, off));
        }
        if (lane == 0) {
            red[wave][0] = lo_x;
            red[wave][1] = hi_x;
            red[wave][2] = lo_y;
            red[wave][3] = hi_y;
        }
        for (int i = tid; i < 2 * VIS_TILE_BINS; i += VIS_TILE_THREADS) hist[i] = 0;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < VIS_TILE_THREADS / 64; ++w) {
                red[0][0] = fmin(red[0][0], red[w][0]);
                red[0][1] = fmax(red[0][1], red[w][1]);
                red[0][2] = fmin(red[0][2], red[w][2]);
                red[0][3] = fmax(red[0][3], red[w][3]);
            }
            // a cut is looked for along both directions; one along which the centres sit on one spot is no cut
            for (int a = 0; a < 2; ++a) {
                const double range = red[0][2 * a + 1] - red[0][2 * a];
                s_range[a] = range;
                s_lo[a] = red[0][2 * a];
                s_scale[a] = range > 1e-5 ? VIS_TILE_BINS / range : 0.0;
                s_straddle[a] = 0;
            }
        }
        __syncthreads();
        const double lo_a[2] = {s_lo[0], s_lo[1]}, scale_a[2] = {s_scale[0], s_scale[1]};
        each([&](double cx, double cy, double, double) {
            if (cx > x0 && cx <= x1 && cy > y0 && cy <= y1) {
                for (int a = 0; a < 2; ++a) {
                    if (scale_a[a] == 0.0) continue;
                    int bin = (int)(((a == 0 ? cx : cy) - lo_a[a]) * scale_a[a]);
                    bin = bin < 0 ? 0 : (bin >= VIS_TILE_BINS ? VIS_TILE_BINS - 1 : bin);
                    atomicAdd(&hist[a * VIS_TILE_BINS + bin], 1);
                }
            }
        });
        __syncthreads();
        // along each direction, the bin at whose upper end half of the tile's centres are reached: a scan of the
        // histogram over the workgroup (one bin a thread), the first such bin by an atomic minimum
        const int cnt = tcnt[sel];
        for (int a = 0; a < 2; ++a) {
            int cum = hist[a * VIS_TILE_BINS + tid];
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(cum, off);
                if (lane >= off) cum += y;
            }
            if (lane == 63) s_wsum[wave] = cum;
            if (tid == 0) s_bin[a] = VIS_TILE_BINS - 2;  // (the last bin is never the cut: its upper end is the range's)
            __syncthreads();
            for (int w = 0; w < wave; ++w) cum += s_wsum[w];
            if (2 * cum >= cnt && tid < VIS_TILE_BINS - 2) atomicMin(&s_bin[a], tid);
            __syncthreads();
            if (tid == s_bin[a]) {
                s_cum[a] = cum;
                // the border at the upper end of that bin, nudged off the round value so that it does not run along a
                // polygon edge that sits exactly on a bin's end
                s_cut[a] = scale_a[a] > 0.0 ? lo_a[a] + (tid + 1) / scale_a[a] + 1.234567e-7 : 0.0;
            }
            __syncthreads();
        }
        // Of the two cuts the one that fewer boxes lie across is taken: a box across the cut is a candidate of both
        // tiles.  (The centres of a wall's far candidates spread widest in azimuth although nearly all of them sit at
        // the horizon's one azimuth: a median there has most boxes across it, and four tiles of a face of 6 200
        // candidates held 6 200, 4 800, 3 100 and 2 000 of them.)
        {
            const double cut_x = s_cut[0], cut_y = s_cut[1];
            int across_x = 0, across_y = 0;
            each([&](double cx, double cy, double hx, double hy) {
                if (cx > x0 && cx <= x1 && cy > y0 && cy <= y1) {
                    across_x += (cx - hx < cut_x && cx + hx > cut_x) ? 1 : 0;
                    across_y += (cy - hy < cut_y && cy + hy > cut_y) ? 1 : 0;
                }
            });
            for (int off = 32; off > 0; off >>= 1) {
                across_x += __shfl_xor(across_x, off);
                across_y += __shfl_xor(across_y, off);
            }
            if (lane == 0) {
                atomicAdd(&s_straddle[0], across_x);
                atomicAdd(&s_straddle[1], across_y);
            }
        }
        __syncthreads();
        if (tid == 0) {
            bool ok[2];
            for (int a = 0; a < 2; ++a) ok[a] = s_scale[a] > 0.0 && s_cum[a] > 0 && s_cum[a] < cnt;
            int axis = -1;
            if (ok[0] && ok[1])
                axis = s_straddle[0] != s_straddle[1] ? (s_straddle[0] < s_straddle[1] ? 0 : 1) : (s_range[0] >= s_range[1] ? 0 : 1);
            else if (ok[0] || ok[1])
                axis = ok[0] ? 0 : 1;
            if (axis < 0) {
                tcnt[sel] = -cnt;  // its centres sit on one spot, or all to one side of every cut
            } else {
                const int nw = s_n, left = s_cum[axis];
                const double cut = s_cut[axis];
                for (int q = 0; q < 4; ++q) tb[nw][q] = tb[sel][q];
                tb[sel][axis == 0 ? 1 : 3] = cut;
                tb[nw][axis == 0 ? 0 : 2] = cut;
                tcnt[sel] = left;
                tcnt[nw] = cnt - left;
                s_n = nw + 1;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const int n_out = s_n;
    for (int i = tid; i < 4 * n_out; i += VIS_TILE_THREADS) b[i] = tb[i >> 2][i & 3];
    if (tid == 0) n_tiles[f] = n_out;
}

// item_off[q] = tiles of the faces before place q of the queue (queue null: face order); item_off[n_faces] = all
__global__ __launch_bounds__(1024) void k_vis_item_offsets(const int32_t *__restrict__ n_tiles, const int32_t *__restrict__ queue,
                                                            int n_faces, int32_t *__restrict__ item_off) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_faces; base += 1024) {
        const int q = base + threadIdx.x;
        const int v = q < n_faces ? n_tiles[queue ? queue[q] : q] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (q < n_faces) item_off[q] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) item_off[n_faces] = carry;
}

// ---- pieces of the painter's resident path, each run by one wave
#define VIS_WAVE_LIST 96 /* listed canvas polygons a wave's own list holds; a longer list is decided with the store's */

// running vertex count over a set: the candidate (nvb vertices) first, then the listed slots
template <typename Store>
__device__ inline void canvas_prefix(const Store *cs, const uint16_t *list, int n_list, int nvb, uint16_t *lpre) {
    const int lane = threadIdx.x & 63;
    int total = nvb;
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < n_list; base += 64) {
        const int li = base + lane;
        const int cnt = li < n_list ? cs->nvs[list[li]] : 0;
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (li < n_list) lpre[li + 1] = (uint16_t)(total + incl - cnt);
        total += __shfl(incl, 63);
    }
    if (lane == 0) {
        lpre[0] = 0;
        lpre[n_list + 1] = (uint16_t)total;
    }
}

// the ring staged in slot `from` becomes slot `to` with its vertices at `to_off`: never further up than it is, so the
// copy goes upwards in steps of a wave, each step read before it is written
template <typename Store>
__device__ inline void move_staged(Store *cs, int from, int to, int to_off) {
    const int lane = threadIdx.x & 63;
    const int from_off = cs->offs[from], n2 = 2 * cs->nvs[from];
    if (from == to && from_off == to_off) return;
    if (from_off != to_off)
        for (int base = 0; base < n2; base += 64) {
            const int q = base + lane;
            const double v = q < n2 ? cs->xy[from_off + q] : 0.0;
            __builtin_amdgcn_wave_barrier();
            if (q < n2) cs->xy[to_off + q] = v;
            __builtin_amdgcn_wave_barrier();
        }
    const int nv = cs->nvs[from], sg = cs->sgns[from];
    const double bx = lane < 4 ? cs->boxes[4 * from + lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (lane < 4) cs->boxes[4 * to + lane] = bx;
    if (lane == 0) {
        cs->nvs[to] = nv;
        cs->offs[to] = to_off;
        cs->sgns[to] = sg;
    }
    __builtin_amdgcn_wave_barrier();
}

// one wave tries the shortcuts on the candidate staged in slot b_slot against the listed slots: 0 = hidden, 1 = shows,
// VIS_UNSETTLED = neither shortcut says
#define VIS_UNSETTLED 2
template <typename Store>
__device__ __forceinline__ int shortcuts_alone(const Store *cs, const uint16_t *list, const uint16_t *lpre, int b_slot,
                                               int n_list, const TileSpan &span) {
    // nothing of the canvas near it: it shows (a valid ring has area); in a tile it still has to be shown that it
    // reaches into the sector at all (its box does, its ring may not)
    if (n_list == 0) return span.bounded() ? VIS_UNSETTLED : 1;
    const CanvasSet<Store> set{cs, list, lpre, b_slot, n_list + 1};
    // inside one canvas polygon settles nine candidates in ten; inside two convex ones most of the rest (two walls
    // that meet in a corner of a building)
    if (inside_one_entry(set)) return 0;
    if (inside_two_entries(set)) {
        if ((threadIdx.x & 63) == 0) VIS_STAT(3, 1);
        return 0;
    }
    return VIS_UNSETTLED;
}

// ---- Shared edges.  Two walls of a building share the edge where they meet, a wall and its roof the wall's top: in the
// canvas such an edge appears twice, once per polygon, and the strip along the outer side of either copy lies inside the
// other polygon.  A third of the canvas edges that the full decisions examine are of that kind (lln: 3.6 of 10.9
// million), and examining one can never find a witness: a witness stands VIS_DELTA outside its edge, next to the middle
// of a piece at least VIS_MIN_PIECE long, i.e. at least 2 VIS_DELTA from the edge's ends, and that point is inside the
// neighbour when the neighbour is convex (or has four vertices at most), has the edge bit for bit (in either direction), and leaves it at both ends at
// an interior angle of 30 degrees or more towards vertices that lie at least VIS_TWIN_REACH beyond the edge's line on
// the witness side (the hull of the edge and those two vertices then holds the point, and a convex polygon holds the
// hull of its vertices).  The polygon that has just been put into canvas slot s is held against every older canvas
// polygon, both ways round; the flags only ever grow, like the canvas.  Threads first, first + stride, ... share the
// older polygons.
#define VIS_TWIN_REACH 1e-7
#define VIS_TWIN_COS 0.86602540378443865 /* 30 degrees: the point stands 2 VIS_DELTA from the end, cot 30 = 1.73 < 2 */
template <typename Store>
__device__ inline bool twin_covers(const Store *cs, int cover, int jc, bool reversed, double ax, double ay, double bx, double by,
                                   double outer) {
    // does polygon `cover`, whose edge jc (from its vertex jc - 1 to jc) is a->b (or b->a when reversed), hold the strip
    // along the side `outer` (+1 left of a->b, -1 right) of a->b?
    const int nv = cs->nvs[cover];
    const double *x = cs->xy + cs->offs[cover], *y = x + nv;
    // (a ring of three or four vertices needs no convexity: with both other vertices on the witness side of the edge's
    // line, VIS_TWIN_REACH away, nothing of its boundary but the two edges that leave the shared one comes near the strip)
    if (nv > 4 && !ring_turns_one_way(x, y, nv, (double)cs->sgns[cover])) return false;
    const int j0 = jc == 0 ? nv - 1 : jc - 1;
    const int before = j0 == 0 ? nv - 1 : j0 - 1, after = jc + 1 == nv ? 0 : jc + 1;
    // the cover's vertices next to a and to b
    const int ia = reversed ? after : before, ib = reversed ? before : after;
    const double ex = bx - ax, ey = by - ay;
    const double len = sqrt(ex * ex + ey * ey);
    if (!(len > VIS_MIN_PIECE)) return false;
    const double qax = x[ia] - ax, qay = y[ia] - ay, qbx = x[ib] - bx, qby = y[ib] - by;
    const double ha = outer * (ex * qay - ey * qax), hb = outer * (ex * qby - ey * qbx);  // heights over the line, times len
    if (!(ha >= VIS_TWIN_REACH * len && hb >= VIS_TWIN_REACH * len)) return false;
    const double la = sqrt(qax * qax + qay * qay), lb = sqrt(qbx * qbx + qby * qby);
    // interior angles at a (between b - a and qa) and at b (between a - b and qb) of 30 degrees or more
    return (ex * qax + ey * qay) <= VIS_TWIN_COS * len * la && -(ex * qbx + ey * qby) <= VIS_TWIN_COS * len * lb;
}
template <typename Store>
__device__ inline void mark_twins(Store *cs, int s, int first, int stride) {
    const int nvp = cs->nvs[s];
    if (nvp > 32) return;
    const double *px = cs->xy + cs->offs[s], *py = px + nvp;
    const double outer_p = -(double)cs->sgns[s];  // a canvas edge's witness stands on the side -sgn of its direction
    for (int c = first; c < s; c += stride) {
        const int nvc = cs->nvs[c];
        if (nvc > 32) continue;
        const double *bp = cs->boxes + 4 * s, *bc = cs->boxes + 4 * c;
        if (bc[1] < bp[0] || bc[0] > bp[1] || bc[3] < bp[2] || bc[2] > bp[3]) continue;
        const double *cx = cs->xy + cs->offs[c], *cy = cx + nvc;
        const double outer_c = -(double)cs->sgns[c];
        for (int k = 0; k < nvc; ++k) {
            const int k0 = k == 0 ? nvc - 1 : k - 1;
            const double ax = cx[k0], ay = cy[k0], bx = cx[k], by = cy[k];
            // (looked for through the vertex of the new polygon that is b: nearly every pair of polygons has no vertex
            // in common, and this way that costs one comparison per pair of vertices)
            for (int m = 0; m < nvp; ++m) {
                if (px[m] != bx || py[m] != by) continue;
                const int m0 = m == 0 ? nvp - 1 : m - 1, m1 = m + 1 == nvp ? 0 : m + 1;
                for (int way = 0; way < 2; ++way) {
                    // the new polygon's edge j, from j0 to j: a->b itself (it ends in m), or b->a (it starts in m)
                    const bool reversed = way == 1;
                    const int j = reversed ? m1 : m, j0 = reversed ? m : m0;
                    const int at_a = reversed ? m1 : m0;
                    if (px[at_a] != ax || py[at_a] != ay) continue;
                    // the new polygon over the outer side of the older one's edge k (a->b) ...
                    if (twin_covers(cs, s, j, reversed, ax, ay, bx, by, outer_c)) atomicOr(&cs->twin[c], 1u << k);
                    // ... and the older one over the outer side of the new polygon's edge j (its own direction)
                    if (twin_covers(cs, c, k, reversed, px[j0], py[j0], px[j], py[j], outer_p)) atomicOr(&cs->twin[s], 1u << j);
                }
            }
        }
    }
}

// ---- The covered-cell bitmap.  Nine candidates in ten are hidden behind ONE nearer polygon, and finding that out took
// most of the painter's time although the answer is usually obvious: the candidate is a far wall a few cells of view
// wide, the nearer polygon a wall that fills a tenth of the view.  So the (azimuth, polar) box of a face's candidates is
// cut into VIS_GRID x VIS_GRID cells with one bit each: set when the cell lies wholly inside some CONVEX polygon of the
// canvas, shrunk by VIS_GRID_MARGIN.  A candidate whose box (widened by VIS_GRID_EPS) lies in set cells lies inside the
// union of canvas polygons with a margin four hundred times the witness offset: every witness the full decision could
// place is covered, so "hidden" is the answer the full decision gives.  A bit that is not set proves nothing and the
// candidate goes the usual way.  Bits are only ever added while a face is painted (the canvas only grows), so a test
// against an older state of the bitmap errs on the side of the usual way.  The bitmap lives in global memory (LDS is
// taken by the canvas), written with atomics and read past the first-level cache.
#ifndef VIS_GRID
#define VIS_GRID 128 /* lln's matrix: 64 cells a side paint in 132 ms, 128 in 117, 256 in 130, 512 in 188 (setting the bits costs) */
#endif
#define VIS_GRID_WORDS (VIS_GRID * VIS_GRID / 32)
#define VIS_GRID_MARGIN 1e-7
#define VIS_GRID_EPS 1e-9
#define VIS_STRETCH 256             /* candidates held against the bitmap at a time (one per thread of a four-wave workgroup) */
#define VIS_GRID_MIN_CANDIDATES 256 /* faces with fewer candidates are painted without the bitmap */
#ifndef VIS_GRID_MAX_CELLS
#define VIS_GRID_MAX_CELLS 1024     /* a candidate over more cells goes the usual way (64: the same for a whole matrix, a sixth more for a rank's share of one, whose longest tiles hold the wide polygons round the pole of the view) */
#endif
#define VIS_GRID_MAX_NV 8           /* canvas polygons with more vertices set no bits */
struct GridDomain {
    double x0, y0, cw, ch, inv_cw, inv_ch;
};

__device__ inline bool grid_hides(const uint32_t *grid, const GridDomain &g, double xmin, double xmax, double ymin, double ymax) {
    int ix0 = (int)floor((xmin - VIS_GRID_EPS - g.x0) * g.inv_cw), ix1 = (int)floor((xmax + VIS_GRID_EPS - g.x0) * g.inv_cw);
    int iy0 = (int)floor((ymin - VIS_GRID_EPS - g.y0) * g.inv_ch), iy1 = (int)floor((ymax + VIS_GRID_EPS - g.y0) * g.inv_ch);
    // (the domain is the box of all candidates: what sticks out of it is the widening alone)
    ix0 = ix0 < 0 ? 0 : ix0;
    iy0 = iy0 < 0 ? 0 : iy0;
    ix1 = ix1 >= VIS_GRID ? VIS_GRID - 1 : ix1;
    iy1 = iy1 >= VIS_GRID ? VIS_GRID - 1 : iy1;
    if (ix1 < ix0 || iy1 < iy0 || (ix1 - ix0 + 1) * (iy1 - iy0 + 1) > VIS_GRID_MAX_CELLS) return false;
    for (int iy = iy0; iy <= iy1; ++iy)
        for (int w = ix0 >> 5; w <= (ix1 >> 5); ++w) {
            const int lo = w == (ix0 >> 5) ? (ix0 & 31) : 0, hi = w == (ix1 >> 5) ? (ix1 & 31) : 31;
            const uint32_t need = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            const uint32_t have = __hip_atomic_load(grid + iy * (VIS_GRID / 32) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((have & need) != need) return false;
        }
    return true;
}

// the canvas polygon in `slot` has just been found visible: the cells wholly inside it get their bit (all threads)
// (also called for a candidate that the usual way has just found HIDDEN: all of it lies behind the canvas, so the cells
// inside it do too, although no single canvas polygon holds them — the seam between two walls, for one)
// Threads first, first + stride, ... of the caller share the cells: a whole workgroup (0, blockDim.x) or one wave (lane, 64).
template <typename Store>
__device__ inline void grid_mark(const Store *cs, int slot, uint32_t *grid, const GridDomain &g, int first, int stride) {
    const int nv = cs->nvs[slot];
    if (nv > VIS_GRID_MAX_NV) return;
    const double *x = cs->xy + cs->offs[slot], *y = x + nv;
    const double sgn = (double)cs->sgns[slot];
    if (!ring_turns_one_way(x, y, nv, sgn)) return;
    const double *bb = cs->boxes + 4 * slot;
    int ix0 = (int)ceil((bb[0] - g.x0) * g.inv_cw), ix1 = (int)floor((bb[1] - g.x0) * g.inv_cw);  // cells [ix0, ix1) lie in the box
    int iy0 = (int)ceil((bb[2] - g.y0) * g.inv_ch), iy1 = (int)floor((bb[3] - g.y0) * g.inv_ch);
    ix0 = ix0 < 0 ? 0 : ix0;
    iy0 = iy0 < 0 ? 0 : iy0;
    ix1 = ix1 > VIS_GRID ? VIS_GRID : ix1;
    iy1 = iy1 > VIS_GRID ? VIS_GRID : iy1;
    const int ncx = ix1 - ix0, ncy = iy1 - iy0;
    if (ncx <= 0 || ncy <= 0) return;
    for (int c = first; c < ncx * ncy; c += stride) {
        const int ix = ix0 + c % ncx, iy = iy0 + c / ncx;
        const double xs[2] = {g.x0 + ix * g.cw, g.x0 + (ix + 1) * g.cw}, ys[2] = {g.y0 + iy * g.ch, g.y0 + (iy + 1) * g.ch};
        bool inside = true;
        for (int k = 0; k < nv && inside; ++k) {
            const int u = k == 0 ? nv - 1 : k - 1;
            const double ex = x[k] - x[u], ey = y[k] - y[u];
            const double far2 = VIS_GRID_MARGIN * VIS_GRID_MARGIN * (ex * ex + ey * ey);
            for (int q = 0; q < 4; ++q) {
                const double cr = sgn * (ex * (ys[q >> 1] - y[u]) - ey * (xs[q & 1] - x[u]));
                if (!(cr > 0.0 && cr * cr >= far2)) inside = false;
            }
        }
        if (inside) atomicOr(grid + iy * (VIS_GRID / 32) + (ix >> 5), 1u << (ix & 31));
    }
}

// One workgroup (NW waves: four when many faces are painted at once, sixteen for a handful) paints one face at a time.  Per candidate polygon, in painter's order: wave 0 lists the canvas
// polygons whose box meets it and stages them in LDS; all waves then decide coverage together.
template <typename Store, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 1) void k_vis_paint(SceneView sc, FaceBatch fb, const double *__restrict__ sph,
                                                    const PolyMeta *__restrict__ meta, const int32_t *__restrict__ order,
                                                    int m_stride, const int32_t *__restrict__ counts, PaintScratch ps,
                                                    int strict, unsigned long long *work_counter,
                                                    const int32_t *__restrict__ face_queue,
                                                    const double *__restrict__ tile_bounds,
                                                    const int32_t *__restrict__ item_off, uint8_t *__restrict__ out) {
    __shared__ int lds_counts[NW];
    __shared__ int s_wcnt[NW];
    __shared__ int s_tile, s_mt;
    __shared__ TileSpan s_span;
    __shared__ union {
        LocalStore staged;   // used once the face's canvas no longer fits the resident store
        Store canvas;        // used until then
    } stores;
    __shared__ uint16_t s_wlist[NW][VIS_WAVE_LIST], s_wlpre[NW][VIS_WAVE_LIST + 2];  // each wave's own list
    __shared__ int s_gnv[NW], s_gB[NW], s_verdict[NW], s_again[NW];  // the group: coordinates staged, polygon, verdict
    __shared__ double lds_splits[NW][SPLIT_LDS];
    __shared__ double lds_px[NW][PIECE_LDS], lds_py[NW][PIECE_LDS];
    __shared__ int lds_pstate[NW][PIECE_LDS];
    __shared__ int s_face, s_nlist, s_fits, s_flag[3];  // s_flag: the full decision's exit flag, the lengths of its ring list and of its collected edges
    // (every byte of LDS counts here: three workgroups share a compute unit, and 1.2 KB more per workgroup made it two)
    __shared__ int s_mc, s_grid_on, s_own;
    __shared__ GridDomain s_gdom;
    double (*s_red)[PIECE_LDS] = lds_px;  // free between two decisions: scratch of the reduction over the candidates' boxes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *grid = ps.grid + (int64_t)blockIdx.x * VIS_GRID_WORDS;
    // the candidates of the stretch being painted that the bitmap did not settle (global memory: read ahead like the order)
    int32_t *s_chunk = ps.chunk + (int64_t)blockIdx.x * (64 * NW);
    LocalStore *store = &stores.staged;
    Store *cs = &stores.canvas;
    int32_t *list = ps.list + (int64_t)blockIdx.x * sc.n;
    int32_t *tlist = ps.tlist + (int64_t)blockIdx.x * sc.n;  // the candidates of the tile being painted, in painter's order
    int32_t *canvas = ps.canvas + (int64_t)blockIdx.x * sc.n;
    double *cbox = ps.cbox + (int64_t)blockIdx.x * sc.n * 4;
    const int64_t wave_id = (int64_t)blockIdx.x * NW + wave;
    // per wave 3 vtot words: 2 vtot for the edges a decision collects (the decisions on the resident canvas collect into
    // the first wave's, for the workgroup); the workgroup's ring list, 2 vtot words from an even offset, begins in its
    // first wave's last third and runs into the second wave's words, which only the other decisions use
    int32_t *work = ps.work + wave_id * 3 * (int64_t)sc.vtot;
    const SplitBuf splits{lds_splits[wave], ps.splits + wave_id * ps.split_cap, ps.split_cap,
                          lds_px[wave],     lds_py[wave],                        lds_pstate[wave],
                          ps.work + ((((int64_t)blockIdx.x * NW * 3 + 2) * (int64_t)sc.vtot + 1) & ~(int64_t)1),
                          ps.work + (int64_t)blockIdx.x * NW * 3 * (int64_t)sc.vtot};
    int *lds_count = &lds_counts[wave];
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long fq = atomicAdd(work_counter, 1ull);
            if (item_off) {
                // work item = (face, tile): place q of the queue holds items item_off[q] .. item_off[q + 1]
                const unsigned long long n_items = (unsigned long long)item_off[fb.n_faces];
                if (fq < n_items) {
                    int lo = 0, hi = fb.n_faces - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if ((unsigned long long)item_off[mid] <= fq) lo = mid; else hi = mid - 1;
                    }
                    s_face = face_queue ? face_queue[lo] : lo;
                    s_tile = (int)fq - item_off[lo];
                } else {
                    s_face = -1;
                }
            } else {
                s_face = fq < (unsigned long long)fb.n_faces ? (face_queue ? face_queue[fq] : (int)fq) : -1;
                s_tile = 0;
            }
        }
        __syncthreads();
        const int f = __builtin_amdgcn_readfirstlane(s_face);
        if (f < 0) break;
        // the sector lives in LDS and is read where it is needed: held in registers through the whole decision it
        // pushed the kernel from 138 to 219 vector registers and into scratch memory
        if (threadIdx.x == 0) {
            const double *tb = tile_bounds ? tile_bounds + 4 * ((size_t)f * VIS_MAX_TILES + s_tile) : nullptr;
            s_span.x0 = tb ? tb[0] : -INFINITY;
            s_span.x1 = tb ? tb[1] : INFINITY;
            s_span.y0 = tb ? tb[2] : -INFINITY;
            s_span.y1 = tb ? tb[3] : INFINITY;
        }
        __syncthreads();
        const TileSpan &span = s_span;
        const PolyMeta *meta_f = meta + (size_t)f * sc.n;
        const double *sph_f = sph + (size_t)f * sc.vtot * 2;
        uint8_t *out_f = out + (size_t)f * sc.n;
        if (strict) {
            // z filter and normal filter only (geometry.py:974 skipped)
            for (int B = threadIdx.x; B < sc.n; B += blockDim.x)
                if (meta_f[B].state != 0 && faces_viewer(sc, fb, f, B)) out_f[B] = 1;
            continue;
        }
        const int32_t *ord = order + (size_t)f * m_stride;
        int m = counts[f];
        if (span.bounded()) {
            // the tile's candidates: those whose box reaches into the sector, kept in painter's order
            if (threadIdx.x == 0) s_mt = 0;
            __syncthreads();
            for (int base = 0; base < m; base += 64 * NW) {
                const int k = base + (int)threadIdx.x;
                int B = -1;
                bool take = false;
                if (k < m) {
                    B = ord[k];
                    take = meta_f[B].xmax > span.x0 && meta_f[B].xmin < span.x1 && meta_f[B].ymax > span.y0 &&
                           meta_f[B].ymin < span.y1;
                }
                const unsigned long long mk = __ballot(take);
                if (lane == 0) s_wcnt[wave] = __popcll(mk);
                __syncthreads();
                int at = s_mt + __popcll(mk & ((1ull << lane) - 1));
                for (int w = 0; w < wave; ++w) at += s_wcnt[w];
                if (take) tlist[at] = B;
                __syncthreads();
                if (threadIdx.x == 0) {
                    int tot = 0;
                    for (int w = 0; w < NW; ++w) tot += s_wcnt[w];
                    s_mt += tot;
                }
                __syncthreads();
            }
            __threadfence_block();
            m = s_mt;
            ord = tlist;
        }
        m = __builtin_amdgcn_readfirstlane(m);
        int n_canvas = 0;
        int n_cs = 0, used = 0;
        // ---- the covered-cell bitmap of this face (whole faces with enough candidates to pay for it): its domain is
        // the box of the candidates' boxes
        // (a sector of a four-wave workgroup has its own: what of a candidate can show there lies in its box cut to the
        // sector, and so does every witness of the sector's decision; the sixteen-wave workgroups that paint a handful
        // of faces in sectors do without: lln's six screens 44 -> 47 ms with it, the bits cost what they settle)
        const bool want_grid = ((NW == 4 && ps.grid_in_tiles) || !span.bounded()) && m >= VIS_GRID_MIN_CANDIDATES;
        if (want_grid) {
            double bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
            for (int q = threadIdx.x; q < m; q += blockDim.x) {
                const PolyMeta &mq = meta_f[ord[q]];
                bx0 = fmin(bx0, mq.xmin);
                bx1 = fmax(bx1, mq.xmax);
                by0 = fmin(by0, mq.ymin);
                by1 = fmax(by1, mq.ymax);
            }
            for (int off = 32; off > 0; off >>= 1) {
                bx0 = fmin(bx0, __shfl_xor(bx0, off));
                bx1 = fmax(bx1, __shfl_xor(bx1, off));
                by0 = fmin(by0, __shfl_xor(by0, off));
                by1 = fmax(by1, __shfl_xor(by1, off));
            }
            if (lane == 0) {
                s_red[wave][0] = bx0;
                s_red[wave][1] = bx1;
                s_red[wave][2] = by0;
                s_red[wave][3] = by1;
            }
            for (int q = threadIdx.x; q < VIS_GRID_WORDS; q += blockDim.x) grid[q] = 0u;
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < NW; ++w) {
                    s_red[0][0] = fmin(s_red[0][0], s_red[w][0]);
                    s_red[0][1] = fmax(s_red[0][1], s_red[w][1]);
                    s_red[0][2] = fmin(s_red[0][2], s_red[w][2]);
                    s_red[0][3] = fmax(s_red[0][3], s_red[w][3]);
                }
                s_red[0][0] = fmax(s_red[0][0], s_span.x0);
                s_red[0][1] = fmin(s_red[0][1], s_span.x1);
                s_red[0][2] = fmax(s_red[0][2], s_span.y0);
                s_red[0][3] = fmin(s_red[0][3], s_span.y1);
                const double wx = s_red[0][1] - s_red[0][0], wy = s_red[0][3] - s_red[0][2];
                const bool ok = wx > 1e-6 && wy > 1e-6 && wx < 1e3 && wy < 1e3;
                s_grid_on = ok ? 1 : 0;
                if (ok) s_gdom = GridDomain{s_red[0][0], s_red[0][2], wx / VIS_GRID, wy / VIS_GRID, VIS_GRID / wx, VIS_GRID / wy};
            }
            __syncthreads();
        }
        const bool grid_on = want_grid && __builtin_amdgcn_readfirstlane(s_grid_on) != 0;
#ifdef RT_VIS_STAMPS
        const unsigned long long t_item = wall_clock64();
#endif
        // ---- While the canvas fits the resident store: NW candidates at a time, one per wave.  Each wave decides its
        // candidate alone against the canvas as it stood BEFORE the group.  That canvas is part of the canvas the
        // candidate really has to be held against (which also has the group's earlier members that show), and hidden
        // behind a part is hidden behind the whole: a "hidden" stands, and so does a "shows" when no earlier member of
        // the group showed.  Only a "shows" after another "shows" in the same group is decided again (one candidate
        // in sixty shows, so that is rare).  A workgroup used to decide one candidate at a time, and for nine in ten
        // one wave did all of it while the others waited: a wave issues one instruction every four or five cycles at
        // best, and that was the painter's speed.
        //
        // A candidate's record is three dependent reads away (its index in the order, its box and vertex range, its
        // mapped vertices).  The reads run ahead, in vector registers (vector loads stay in flight across LDS
        // traffic; scalar loads share their counter with it): the order in batches of 64 (lane = place), the record
        // of the wave's candidate two groups on (lanes 0-13 its words, lanes 14/15 its vertex range) and the vertices
        // of its candidate of the next group (lane = coordinate; longer rings are read when they are staged).
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // The candidates are taken in stretches of one per thread.  Each thread first holds its candidate's box against
        // the bitmap (as it stands: a test against an older canvas can only send a candidate the usual way); the
        // candidates that are left are put together, in order, and painted in groups as described below.
        int k = 0;                    // place in the face's order where the candidates not yet looked at begin
        int left_k = 0, left_m = 0;   // candidates of the last stretch the resident path did not get to: s_chunk[left_k, left_m)
        bool resident = true;
        while (k < m && resident) {
            const int chunk_n = m - k < VIS_STRETCH ? m - k : VIS_STRETCH;
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
            unsigned long long t_g = clock64();
#define G_STAMP(i, on)                                                                 \
    do {                                                                               \
        const unsigned long long now_ = clock64();                                     \
        if (lane == 0 && (on)) atomicAdd(&g_vis_hist[96 + (i)], now_ - t_g);           \
        t_g = now_;                                                                    \
    } while (0)
#else
#define G_STAMP(i, on)
#endif
            {
                int B = -1;
                bool keep = false;
                if ((int)threadIdx.x < chunk_n) {
                    B = ord[k + threadIdx.x];
                    keep = true;
                    if (grid_on) {
                        const PolyMeta &mq = meta_f[B];
                        keep = !grid_hides(grid, s_gdom, fmax(mq.xmin, span.x0), fmin(mq.xmax, span.x1), fmax(mq.ymin, span.y0),
                                           fmin(mq.ymax, span.y1));
                    }
                }
                const unsigned long long mk = __ballot(keep);
                if (lane == 0) s_wcnt[wave] = __popcll(mk);
                __syncthreads();
                int at = __popcll(mk & ((1ull << lane) - 1));
                for (int w = 0; w < wave; ++w) at += s_wcnt[w];
                if (keep) s_chunk[at] = B;
                if (threadIdx.x == 0) {
                    int tot = 0;
                    for (int w = 0; w < NW; ++w) tot += s_wcnt[w];
                    s_mc = tot;
                }
                __threadfence_block();
                __syncthreads();
            }
            const int m_c = __builtin_amdgcn_readfirstlane(s_mc);
            G_STAMP(6, wave_u == 0);
            if (threadIdx.x == 0 && grid_on) {  // (two updates per stretch of 256 candidates: kept in every build)
                atomicAdd(&g_vis_stats[6], (unsigned long long)chunk_n);
                atomicAdd(&g_vis_stats[7], (unsigned long long)(chunk_n - m_c));
            }
            const int32_t *ord_c = s_chunk;
            int kc = 0;
            {
                int ord_cur = lane < m_c ? ord_c[lane] : 0, ord_next = 64 + lane < m_c ? ord_c[64 + lane] : 0;
                auto place = [&](int idx, int from) -> int {  // candidate at place idx (uniform), seen from group `from`
                    return (idx >> 6) == (from >> 6) ? __builtin_amdgcn_readlane(ord_cur, idx & 63)
                                                     : __builtin_amdgcn_readlane(ord_next, idx & 63);
                };
                auto record = [&](int Bx) -> int {
                    return lane < 14 ? ((const int32_t *)(meta_f + Bx))[lane] : (lane < 16 ? sc.vert_off[Bx + lane - 14] : 0);
                };
                int rec0 = 0, rec1 = 0;
                double xy0 = 0.0;
                if (wave_u < m_c) {
                    rec0 = record(place(wave_u, 0));
                    const int v0x = __builtin_amdgcn_readlane(rec0, 14), n2 = 2 * (__builtin_amdgcn_readlane(rec0, 15) - v0x);
                    if (lane < n2) xy0 = sph_f[(size_t)v0x * 2 + lane];
                }
                if (NW + wave_u < m_c) rec1 = record(place(NW + wave_u, 0));
                uint16_t *my_list = s_wlist[wave_u], *my_lpre = s_wlpre[wave_u];
                while (kc < m_c) {
                    if (kc != 0 && (kc & 63) == 0) {
                        ord_cur = ord_next;
                        ord_next = kc + 64 + lane < m_c ? ord_c[kc + 64 + lane] : 0;
                    }
                    const int mine = kc + wave_u;
                    const bool have = mine < m_c;
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
                    const bool g_on = ((kc / NW) & 7) == 0;  // (profiling builds: one group in eight is timed)
                    G_STAMP(7, false);
                    if (lane == 0 && g_on && wave_u == 0) VIS_HIST_ADD(127, 1ull);
#endif
                    const int B = have ? place(mine, kc) : 0;
                    PolyMeta mb;
                    {
                        auto word = [&](int i) { return (unsigned long long)(unsigned)__builtin_amdgcn_readlane(rec0, i); };
                        mb.xmin = __longlong_as_double((long long)(word(4) | (word(5) << 32)));
                        mb.xmax = __longlong_as_double((long long)(word(6) | (word(7) << 32)));
                        mb.ymin = __longlong_as_double((long long)(word(8) | (word(9) << 32)));
                        mb.ymax = __longlong_as_double((long long)(word(10) | (word(11) << 32)));
                        mb.sgn = __builtin_amdgcn_readlane(rec0, 13);
                    }
                    const int b_v0 = __builtin_amdgcn_readlane(rec0, 14);
                    const int nv2 = have ? 2 * (__builtin_amdgcn_readlane(rec0, 15) - b_v0) : 0;
                    const double b_xy = xy0;
                    // the reads that run ahead
                    int rec2 = 0;
                    if (mine + 2 * NW < m_c) rec2 = record(place(mine + 2 * NW, kc));
                    double xy1 = 0.0;
                    if (mine + NW < m_c) {
                        const int v0x = __builtin_amdgcn_readlane(rec1, 14), n2 = 2 * (__builtin_amdgcn_readlane(rec1, 15) - v0x);
                        if (lane < n2) xy1 = sph_f[(size_t)v0x * 2 + lane];
                    }
                    if (lane == 0) {
                        s_gnv[wave_u] = nv2;
                        s_gB[wave_u] = B;
                    }
                    __syncthreads();
                    G_STAMP(0, g_on);
                    int before = 0, total = 0;  // coordinates the group's earlier members / all its members stage
                    for (int w = 0; w < NW; ++w) {
                        const int c = s_gnv[w];
                        if (w < wave_u) before += c;
                        total += c;
                    }
                    if (n_cs + NW > Store::max_poly || used + total > Store::max_coord) {
                        resident = false;
                        break;  // the same in every wave: the rest of the face goes the slow way
                    }
                    const int slot_b = n_cs + wave_u, at_b = used + before;
                    int verdict = 0;  // 0 hidden, 1 shows, or unsettled
                    int n_list_mine = VIS_WAVE_LIST + 1;  // how many canvas polygons this wave listed for its candidate
                    if (have) {
                        // what of B can show in this tile lies in B's box cut to the sector: only canvas polygons reaching
                        // there matter
                        const double lxmin = fmax(mb.xmin, span.x0), lxmax = fmin(mb.xmax, span.x1);
                        const double lymin = fmax(mb.ymin, span.y0), lymax = fmin(mb.ymax, span.y1);
                        int n_list = 0;
                        for (int base = 0; base < n_cs; base += 64) {
                            const int ci = base + lane;
                            bool take = false;
                            if (ci < n_cs) {
                                const double *cb = cs->boxes + 4 * ci;
                                take = !(cb[1] < lxmin || cb[0] > lxmax || cb[3] < lymin || cb[2] > lymax);
                            }
                            const unsigned long long mk = __ballot(take);
                            const int at = n_list + __popcll(mk & ((1ull << lane) - 1));
                            if (take && at < VIS_WAVE_LIST) my_list[at] = (uint16_t)ci;
                            n_list += __popcll(mk);
                        }
                        n_list_mine = n_list;
                        const double *src = sph_f + (size_t)b_v0 * 2;
                        if (lane < nv2) cs->xy[at_b + lane] = b_xy;
                        for (int q = 64 + lane; q < nv2; q += 64) cs->xy[at_b + q] = src[q];
                        if (lane == 0) {
                            cs->nvs[slot_b] = nv2 / 2;
                            cs->offs[slot_b] = at_b;
                            cs->sgns[slot_b] = mb.sgn;
                            cs->boxes[4 * slot_b] = mb.xmin;
                            cs->boxes[4 * slot_b + 1] = mb.xmax;
                            cs->boxes[4 * slot_b + 2] = mb.ymin;
                            cs->boxes[4 * slot_b + 3] = mb.ymax;
                            VIS_STAT(0, 1);
                            VIS_STAT(1, n_list);
                        }
                        G_STAMP(1, g_on);
                        if (n_list > VIS_WAVE_LIST) {
                            verdict = VIS_UNSETTLED;
                        } else {
                            canvas_prefix(cs, my_list, n_list, nv2 / 2, my_lpre);
                            __builtin_amdgcn_wave_barrier();
                            verdict = shortcuts_alone(cs, my_list, my_lpre, slot_b, n_list, span);
                            G_STAMP(2, g_on);
                            // hidden behind the canvas as it stood, and the bitmap had not said so: the cells inside it
                            // are covered too (by this wave, now: once the verdicts are gone through, a member that
                            // shows is moved over the staged rings of the members before it)
                            if (verdict == 0 && grid_on) grid_mark(cs, slot_b, grid, s_gdom, lane, 64);
                            G_STAMP(3, g_on);
                        }
                    }
                    if (lane == 0) s_verdict[wave_u] = verdict;
                    __syncthreads();
                    G_STAMP(4, g_on);
                    // ---- the group's verdicts in order
                    int shown = 0;
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
                    const unsigned long long t_keep = t_g;  // (the finer stamps below move t_g; the loop's own stamp counts from here)
#endif
                    for (int j = 0; j < NW && kc + j < m_c; ++j) {
                        int v = s_verdict[j];
                        const int nv2_j = s_gnv[j];
                        bool in_place = false;
                        if (v == VIS_UNSETTLED || (v == 1 && shown > 0)) {
                            // the full decision, by all waves together, against the canvas as it is now: its wave moves it
                            // to the canvas's next slot and lists again (any length; the shortcuts again only if the
                            // canvas has grown since they were tried)
                            if (wave_u == j) {
                                move_staged(cs, slot_b, n_cs, used);
                                // (no member of the group has shown yet and the wave's own list holds: the canvas is
                                // what it was when the wave listed it, and the shortcuts have been tried against it)
                                const bool own = shown == 0 && v == VIS_UNSETTLED && n_list_mine <= VIS_WAVE_LIST;
                                int n_list = n_list_mine;
                                int again = VIS_UNSETTLED;
                                if (!own) {
                                    n_list = 0;
                                    const double lxmin = fmax(mb.xmin, span.x0), lxmax = fmin(mb.xmax, span.x1);
                                    const double lymin = fmax(mb.ymin, span.y0), lymax = fmin(mb.ymax, span.y1);
                                    for (int base = 0; base < n_cs; base += 64) {
                                        const int ci = base + lane;
                                        bool take = false;
                                        if (ci < n_cs) {
                                            const double *cb = cs->boxes + 4 * ci;
                                            take = !(cb[1] < lxmin || cb[0] > lxmax || cb[3] < lymin || cb[2] > lymax);
                                        }
                                        const unsigned long long mk = __ballot(take);
                                        if (take) cs->list[n_list + __popcll(mk & ((1ull << lane) - 1))] = (uint16_t)ci;
                                        n_list += __popcll(mk);
                                    }
                                    canvas_prefix(cs, cs->list, n_list, nv2 / 2, cs->lpre);
                                    __builtin_amdgcn_wave_barrier();
                                    if (shown > 0 || v != VIS_UNSETTLED || n_list > VIS_WAVE_LIST)
                                        again = shortcuts_alone(cs, cs->list, cs->lpre, n_cs, n_list, span);
                                }
                                if (lane == 0) {
                                    // (slots of their own: a wave that is late may still be reading the first verdict,
                                    // and every wave has to take the same way through the barriers)
                                    s_again[j] = again;
                                    s_nlist = n_list;
                                    s_own = own ? 1 : 0;
                                    s_flag[0] = 0;
                                    s_flag[1] = 0;
                                    s_flag[2] = 0;
                                }
                            }
                            __syncthreads();
                            G_STAMP(8, g_on && wave_u == 0);
                            v = s_again[j];
                            if (v == VIS_UNSETTLED) {
                                const bool own = s_own != 0;
                                const CanvasSet<Store> cv{cs, own ? s_wlist[j] : cs->list, own ? s_wlpre[j] : cs->lpre, n_cs, s_nlist + 1};
#ifdef RT_VIS_STAMPS
                                const unsigned long long t_dec = wall_clock64();
#endif
                                v = polygon_uncovered(cv, splits, lds_count, work, s_flag, span, true) ? 1 : 0;
#ifdef RT_VIS_STAMPS
                                if (threadIdx.x == 0) {
                                    // (measurement) by the width of the candidate's box: up to 0.25, 0.5, 1, 2, 4 rad, more
                                    const double wbox = cs->boxes[4 * n_cs + 1] - cs->boxes[4 * n_cs];
                                    int c = 0;
                                    while (c < 5 && wbox > 0.25 * (1 << c)) ++c;
                                    atomicAdd(&g_vis_hist[64 + c], wall_clock64() - t_dec);
                                    atomicAdd(&g_vis_hist[72 + c], 1ull);
                                    atomicAdd(&g_vis_hist[80 + c], (unsigned long long)(v ? 1 : 0));
                                }
#endif
                                __syncthreads();  // the flag and the list are the next unsettled candidate's after this
                                G_STAMP(9, g_on && wave_u == 0);
                            }
                            in_place = true;
                        }
                        if (v == 1) {
                            if (wave_u == j && !in_place) move_staged(cs, slot_b, n_cs, used);
                            if (threadIdx.x == 0) cs->twin[n_cs] = 0u;
                            __syncthreads();  // the moved ring is part of the canvas for what comes after
                            if (threadIdx.x == 0) {
                                const int Bj = s_gB[j];
                                VIS_STAT(5, 1);
                                canvas[n_canvas] = Bj;  // (kept in global memory too, for a face that outgrows the store)
                                double *cb = cbox + 4 * (size_t)n_canvas;
                                const double *bb = cs->boxes + 4 * n_cs;
                                cb[0] = bb[0]; cb[1] = bb[1]; cb[2] = bb[2]; cb[3] = bb[3];
                                if (faces_viewer(sc, fb, f, Bj)) out_f[Bj] = 1;
                                __threadfence_block();
                            }
                            ++n_cs;
                            used += nv2_j;
                            ++n_canvas;
                            ++shown;
                            // the edges it shares with older canvas polygons (read by the next full decision, which
                            // begins behind a barrier)
                            G_STAMP(10, g_on && wave_u == 0);
                            mark_twins(cs, n_cs - 1, (int)threadIdx.x, (int)blockDim.x);
                            G_STAMP(11, g_on && wave_u == 0);
                            if (grid_on) {
                                // (no fence: a test that does not see a bit yet only sends its candidate the usual way)
                                grid_mark(cs, n_cs - 1, grid, s_gdom, (int)threadIdx.x, (int)blockDim.x);
                            }
                            G_STAMP(12, g_on && wave_u == 0);
                        } else if (grid_on && in_place && nv2_j > 0) {
                            // hidden by the full decision, and the bitmap had not said so: the cells inside it are
                            // covered too (its ring is in the slot it was moved to, which the next member that shows
                            // will take: hence the barrier)
                            grid_mark(cs, n_cs, grid, s_gdom, (int)threadIdx.x, (int)blockDim.x);
                            __syncthreads();
                        }
                    }
                    rec0 = rec1;
                    rec1 = rec2;
                    xy0 = xy1;
                    kc += NW;
                    __syncthreads();  // the staging slots and the verdicts are rewritten by the next group
#if defined(RT_VIS_STAMPS) && !defined(RT_VIS_ITEMS_ONLY)
                    t_g = t_keep;
#endif
                    G_STAMP(5, g_on);
                }
            }
            k += chunk_n;
            if (kc < m_c) {
                left_k = kc;
                left_m = m_c;
            }
            __syncthreads();  // the list of the stretch is rewritten by the next one
        }
        __threadfence_block();
        // ---- what is left once the canvas has outgrown the resident store: one candidate at a time, all waves
        // together, the canvas read from global memory (staged in LDS per candidate when it fits)
        const int n_left = left_m - left_k;
        for (int q_rest = 0; q_rest < n_left + (m - k); ++q_rest) {
            // first what the resident path left of its last stretch, then the rest of the face's order
            const int B = q_rest < n_left ? s_chunk[left_k + q_rest] : ord[k + q_rest - n_left];  // uniform over the workgroup
            const PolyMeta mb = meta_f[B];
            const double lxmin = fmax(mb.xmin, span.x0), lxmax = fmin(mb.xmax, span.x1);
            const double lymin = fmax(mb.ymin, span.y0), lymax = fmin(mb.ymax, span.y1);
            // (the bitmap goes on serving here: the same test as in the stretches above.  Every thread makes it and the
            // candidate is passed by only when all of them find it hidden: a bit that another wave has set a moment ago
            // may not have arrived for every reader -- a barrier does not wait for writes to global memory -- and the
            // waves must take the same way through the barriers below; a reader that misses a bit only sends the
            // candidate the usual way)
            if (grid_on && __syncthreads_and(grid_hides(grid, s_gdom, lxmin, lxmax, lymin, lymax) ? 1 : 0)) continue;
            if (wave == 0) {
                // ---- canvas polygons whose box meets B's box
                int n_list = 0;
                for (int base = 0; base < n_canvas; base += 64) {
                    const int ci = base + lane;
                    bool take = false;
                    int C = -1;
                    if (ci < n_canvas) {
                        const double *cb = cbox + 4 * (size_t)ci;
                        take = !(cb[1] < lxmin || cb[0] > lxmax || cb[3] < lymin || cb[2] > lymax);
                        C = canvas[ci];
                    }
                    const unsigned long long mk = __ballot(take);
                    if (take) list[n_list + __popcll(mk & ((1ull << lane) - 1))] = C;
                    n_list += __popcll(mk);
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                if (lane == 0) {
                    VIS_STAT(0, 1);
                    VIS_STAT(1, n_list);
                }
                // ---- stage B and the listed polygons in LDS when they fit
                const int n_entries = n_list + 1;
                int total = 0;
                bool fits = (n_list > 0 || span.bounded()) && n_entries <= LS_MAX_POLY;
                if (fits) {
                    for (int base = 0; base < n_entries; base += 64) {
                        const int li = base + lane;
                        int nv2 = 0;
                        if (li < n_entries) {
                            const int P = li == 0 ? B : list[li - 1];
                            nv2 = 2 * (sc.vert_off[P + 1] - sc.vert_off[P]);
                        }
                        int incl = nv2;  // inclusive prefix over the wave
                        for (int off = 1; off < 64; off <<= 1) {
                            const int t = __shfl_up(incl, off);
                            if (lane >= off) incl += t;
                        }
                        if (li < n_entries && total + incl <= LS_MAX_COORD) {
                            store->offs[li] = total + incl - nv2;
                            store->nvs[li] = nv2 / 2;
                        }
                        total += __shfl(incl, 63);
                    }
                    fits = total <= LS_MAX_COORD;
                }
                if (fits) {
                    __builtin_amdgcn_wave_barrier();
                    for (int base = 0; base < n_entries; base += 64) {
                        const int li = base + lane;
                        if (li < n_entries) {
                            const int P = li == 0 ? B : list[li - 1];
                            const PolyMeta &mp = meta_f[P];
                            store->sgns[li] = mp.sgn;
                            store->boxes[4 * li] = mp.xmin;
                            store->boxes[4 * li + 1] = mp.xmax;
                            store->boxes[4 * li + 2] = mp.ymin;
                            store->boxes[4 * li + 3] = mp.ymax;
                            const double *src = sph_f + (size_t)sc.vert_off[P] * 2;
                            double *dst = store->xy + store->offs[li];
                            const int cnt = 2 * store->nvs[li];
                            for (int q = 0; q < cnt; ++q) dst[q] = src[q];
                        }
                    }
                }
                if (lane == 0) {
                    s_nlist = n_list;
                    s_fits = fits ? 1 : 0;
                    s_flag[0] = 0;
                    s_flag[1] = 0;
                    s_flag[2] = 0;
                }
            }
            __syncthreads();
            const int n_list = s_nlist;
            // nothing of the canvas near B: B shows (a valid ring has area); in a tile it still has to be shown that B
            // reaches into the sector at all (its box does, its ring may not)
            bool visible = n_list == 0 && !span.bounded();
            if (!visible) {
                if (s_fits) {
                    LocalSet ls{store, n_list + 1};
                    visible = polygon_uncovered(ls, splits, lds_count, work, s_flag, span, false);
                } else {
                    GlobalSet gs{&sc, sph_f, meta_f, list, B, n_list + 1};
                    visible = polygon_uncovered(gs, splits, lds_count, work, s_flag, span, false);
                }
            }
            if (visible) {
                if (threadIdx.x == 0) {
                    VIS_STAT(5, 1);
                    canvas[n_canvas] = B;
                    double *cb = cbox + 4 * (size_t)n_canvas;
                    cb[0] = mb.xmin; cb[1] = mb.xmax; cb[2] = mb.ymin; cb[3] = mb.ymax;
                    if (faces_viewer(sc, fb, f, B)) out_f[B] = 1;
                    __threadfence_block();
                }
                ++n_canvas;
            }
            // (shown or hidden, the cells inside B are behind the canvas now; its ring is entry 0 of the staged store)
            if (grid_on && s_fits && n_list > 0) grid_mark(store, 0, grid, s_gdom, (int)threadIdx.x, (int)blockDim.x);
            __syncthreads();  // canvas / LDS store are rewritten for the next candidate
        }
#ifdef RT_VIS_STAMPS
        if (threadIdx.x == 0) {
            const unsigned long long t_end = wall_clock64();
            atomicAdd(&g_vis_phase[12], t_end - t_item);
            atomicMax(&g_vis_phase[13], t_end - t_item);
            atomicMax(&g_vis_phase[14], t_end);
            atomicMin(&g_vis_phase[15], t_item);
            atomicAdd(&g_vis_stats[3], (unsigned long long)m);
            atomicAdd(&g_vis_stats[4], (unsigned long long)n_canvas);
            // (a batch of few faces logs its (face, sector) items instead)
            const int slot = fb.n_faces * VIS_MAX_TILES <= VIS_FACE_LOG ? f * VIS_MAX_TILES + s_tile : f;
            if (slot < VIS_FACE_LOG) {
                g_vis_face_log[4 * slot] = t_item;
                g_vis_face_log[4 * slot + 1] = t_end - t_item;
                g_vis_face_log[4 * slot + 2] = (unsigned long long)m | ((unsigned long long)s_tile << 32) |
                                               ((unsigned long long)(span.bounded() ? 1 : 0) << 48);
                g_vis_face_log[4 * slot + 3] = (unsigned long long)n_canvas;
            }
        }
#endif
    }
}

__global__ void k_symmetrise(const uint8_t *__restrict__ rows, int n, uint8_t *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) out[(size_t)i * n + j] = (rows[(size_t)i * n + j] | rows[(size_t)j * n + i]) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};

// Device buffers of one batch of faces: they live until the batch's kernels have run.
struct RowsJob {
    // The batch's buffers are parts of ONE allocation (a dozen gigabytes for a town's matrix in some twenty pieces:
    // allocated and freed piece by piece that was 3 ms a call, and a call that asks for other sizes than the one before
    // it could wait a hundred milliseconds and more for the driver to give memory back and map it again).  A part is
    // asked for with alloc(), which only notes its size; commit() allocates and hands out the addresses.
    struct Part {
        void *p = nullptr;
        size_t bytes = 0;
        std::vector<Part *> *asked = nullptr;
        int alloc(size_t b) {
            bytes = b;
            asked->push_back(this);
            return 0;
        }
    };
    std::vector<Part *> asked;
    DevBuf arena;
    Part b_off, b_verts, b_normal, b_self, b_sph, b_meta, b_list, b_canvas, b_cbox, b_splits, b_counter, b_order, b_counts, b_work,
        b_queue, b_tlist, b_bounds, b_ntiles, b_itemoff, b_grid, b_chunk;
    DevBuf b_big;  // (its size is known later than the others')
    std::vector<int32_t> h_big;
    RowsJob() {
        for (Part *q : {&b_off, &b_verts, &b_normal, &b_self, &b_sph, &b_meta, &b_list, &b_canvas, &b_cbox, &b_splits, &b_counter,
                        &b_order, &b_counts, &b_work, &b_queue, &b_tlist, &b_bounds, &b_ntiles, &b_itemoff, &b_grid, &b_chunk})
            q->asked = &asked;
    }
    RowsJob(const RowsJob &) = delete;
    RowsJob &operator=(const RowsJob &) = delete;
    int commit() {
        size_t total = 0;
        for (const Part *q : asked) total += (q->bytes + 255) & ~(size_t)255;
        if (arena.alloc(total)) return -1;
        char *at = (char *)arena.p;
        for (Part *q : asked) {
            q->p = at;
            at += (q->bytes + 255) & ~(size_t)255;
        }
        return 0;
    }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start, mapped, sorted, painted
    ~RowsJob() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    // after the stream has been waited for: ms[0..3] += mapping, sorting, painting, all
    void add_times(double *ms) const {
        if (!ev[0] || !ev[3]) return;
        float t = 0;
        if (hipEventElapsedTime(&t, ev[0], ev[1]) == hipSuccess) ms[0] += t;
        if (hipEventElapsedTime(&t, ev[1], ev[2]) == hipSuccess) ms[1] += t;
        if (hipEventElapsedTime(&t, ev[2], ev[3]) == hipSuccess) ms[2] += t;
        if (hipEventElapsedTime(&t, ev[0], ev[3]) == hipSuccess) ms[3] += t;
    }
};

// Faces f0 .. f0 + n_faces - 1 of the scene itself as a batch: offsets from the first of them, unit normals, own indices
// (their vertices are the scene's, in place).
__global__ void k_scene_faces(SceneView sc, int f0, int n_faces, int32_t *off, double *normal, int32_t *self) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > n_faces) return;
    off[f] = sc.vert_off[f0 + f] - sc.vert_off[f0];
    if (f == n_faces) return;
    for (int k = 0; k < 3; ++k) normal[3 * f + k] = sc.planes[4 * (size_t)(f0 + f) + k];
    self[f] = f0 + f;
}

// Queues, on stream st, the kernels that write the visibility rows of a batch of faces into d_out[n_faces*n] (zeroed
// here).  scene_f0 >= 0: the batch is the scene's own faces scene_f0 .. scene_f0 + scene_count - 1, taken from the
// device's copy of the scene (the host arrays are not looked at: nothing of the geometry crosses the bus again).  Nothing is waited for: the caller synchronises st before it lets go of job and of the host arrays.
static int visibility_rows_enqueue(rt_scene *scene, hipStream_t st, const std::vector<int32_t> &a_off,
                                   const std::vector<double> &a_verts, const std::vector<double> &a_normal,
                                   const std::vector<int32_t> &a_self, int strict, uint8_t *d_out, RowsJob &job,
                                   int scene_f0 = -1, int scene_count = 0, int max_blocks = 0) {
    const SceneView &sc = scene->view;
    const bool resident = scene_f0 >= 0;
    const int n_faces = resident ? scene_count : (int)a_self.size();
    if (n_faces == 0) return 0;
    RowsJob::Part &b_off = job.b_off, &b_verts = job.b_verts, &b_normal = job.b_normal, &b_self = job.b_self, &b_sph = job.b_sph,
                  &b_meta = job.b_meta, &b_list = job.b_list, &b_canvas = job.b_canvas, &b_cbox = job.b_cbox,
                  &b_splits = job.b_splits, &b_counter = job.b_counter, &b_order = job.b_order, &b_counts = job.b_counts,
                  &b_work = job.b_work;
    // persistent workgroups of 4 waves, one face per workgroup at a time
    // (a face may be painted as up to VIS_MAX_TILES work items)
    const int64_t max_items = strict ? n_faces : (int64_t)n_faces * VIS_MAX_TILES;
    int n_blocks = (int)std::min<int64_t>(n_faces > 256 ? 1024 : 256, max_items);
    if (max_blocks > 0) n_blocks = std::min(n_blocks, max_blocks);
    const int64_t split_cap = 3 * (int64_t)sc.vtot + 8;
    int m_stride = 1;
    while (m_stride < sc.n) m_stride <<= 1;
    // few faces: sixteen waves per face (the canvas edges of a candidate are dealt to them), otherwise four
    const int waves_per_face = n_faces <= 256 ? 16 : 4;
    const size_t n_waves = (size_t)n_blocks * waves_per_face;
    if (b_off.alloc(sizeof(int32_t) * ((size_t)n_faces + 1)) || (!resident && b_verts.alloc(sizeof(double) * a_verts.size())) ||
        b_normal.alloc(sizeof(double) * 3 * (size_t)n_faces) || b_self.alloc(sizeof(int32_t) * (size_t)n_faces) ||
        b_sph.alloc(sizeof(double) * 2 * (size_t)sc.vtot * n_faces) || b_meta.alloc(sizeof(PolyMeta) * (size_t)sc.n * n_faces) ||
        b_list.alloc(sizeof(int32_t) * (size_t)sc.n * n_blocks) || b_canvas.alloc(sizeof(int32_t) * (size_t)sc.n * n_blocks) ||
        b_cbox.alloc(sizeof(double) * 4 * (size_t)sc.n * n_blocks) || b_splits.alloc(sizeof(double) * (size_t)split_cap * n_waves) ||
        b_counter.alloc(sizeof(unsigned long long)) || b_order.alloc(sizeof(int32_t) * (size_t)m_stride * n_faces) ||
        b_counts.alloc(sizeof(int32_t) * (size_t)n_faces) || job.b_queue.alloc(sizeof(int32_t) * (size_t)n_faces) ||
        b_work.alloc(sizeof(int32_t) * 3 * (size_t)sc.vtot * n_waves) || job.b_tlist.alloc(sizeof(int32_t) * (size_t)sc.n * n_blocks) ||
        job.b_grid.alloc(sizeof(uint32_t) * VIS_GRID_WORDS * (size_t)n_blocks) ||
        job.b_chunk.alloc(sizeof(int32_t) * 1024 * (size_t)n_blocks) ||
        job.b_bounds.alloc(sizeof(double) * 4 * VIS_MAX_TILES * (size_t)n_faces) ||
        job.b_ntiles.alloc(sizeof(int32_t) * (size_t)n_faces) || job.b_itemoff.alloc(sizeof(int32_t) * ((size_t)n_faces + 1)) ||
        job.commit())
        RT_FAIL("visibility: out of device memory");
    const double *d_face_verts;
    if (resident) {
        hipLaunchKernelGGL(k_scene_faces, dim3((n_faces + 256) / 256), dim3(256), 0, st, sc, scene_f0, n_faces,
                           (int32_t *)b_off.p, (double *)b_normal.p, (int32_t *)b_self.p);
        d_face_verts = sc.verts + 3 * (size_t)scene->h_vert_off[scene_f0];
    } else {
        if (a_off.size() != (size_t)n_faces + 1 || a_normal.size() != 3 * (size_t)n_faces)
            RT_FAIL("visibility: face arrays of different lengths");
        RT_HIP(hipMemcpyAsync(b_off.p, a_off.data(), sizeof(int32_t) * a_off.size(), hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(b_verts.p, a_verts.data(), sizeof(double) * a_verts.size(), hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(b_normal.p, a_normal.data(), sizeof(double) * a_normal.size(), hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(b_self.p, a_self.data(), sizeof(int32_t) * a_self.size(), hipMemcpyHostToDevice, st));
        d_face_verts = (const double *)b_verts.p;
    }
    // the polygons that are prepared one per wave
    // the polygons that are not prepared one per thread by k_vis_prepare, by size: up to 8, 16 and VIS_MID_POLY vertices
    // (a thread per face with the polygon shared by the wave), then the rest (a wave per polygon)
    std::vector<int32_t> &big = job.h_big;  // lives until the stream has been waited for
    big.clear();
    size_t class_end[4] = {0, 0, 0, 0};
    for (int cls = 0; cls < 4; ++cls) {
        const int lo = cls == 0 ? VIS_SMALL_POLY : (cls == 1 ? 8 : (cls == 2 ? 16 : VIS_MID_POLY));
        const int hi = cls == 0 ? 8 : (cls == 1 ? 16 : (cls == 2 ? VIS_MID_POLY : 0x7fffffff));
        for (int i = 0; i < sc.n; ++i) {
            const int nv = scene->h_vert_off[i + 1] - scene->h_vert_off[i];
            if (nv > lo && nv <= hi) big.push_back(i);
        }
        class_end[cls] = big.size();
    }
    if (job.b_big.alloc(sizeof(int32_t) * big.size())) RT_FAIL("visibility: out of device memory");
    if (!big.empty()) RT_HIP(hipMemcpyAsync(job.b_big.p, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemsetAsync(b_counter.p, 0, sizeof(unsigned long long), st));
    RT_HIP(hipMemsetAsync(d_out, 0, (size_t)sc.n * n_faces, st));
    FaceBatch fb{n_faces, (const int32_t *)b_off.p, d_face_verts, (const double *)b_normal.p, (const int32_t *)b_self.p};
    if (scene->timing) {
        for (hipEvent_t &e : job.ev) RT_HIP(hipEventCreate(&e));
        RT_HIP(hipEventRecord(job.ev[0], st));
    }
    hipLaunchKernelGGL(k_vis_prepare, dim3((sc.n + 255) / 256, n_faces), dim3(256), 0, st, sc, fb, (double *)b_sph.p,
                       (PolyMeta *)b_meta.p);
    {
        const int32_t *lists = (const int32_t *)job.b_big.p;
        const unsigned face_blocks = ((unsigned)n_faces + 63) / 64;
        const int n8 = (int)class_end[0], n16 = (int)(class_end[1] - class_end[0]), n32 = (int)(class_end[2] - class_end[1]),
                  n_rest = (int)(class_end[3] - class_end[2]);
        // (a grid's second dimension ends at 65 535)
        for (int at = 0; at < n8; at += 65535)
            hipLaunchKernelGGL(k_vis_prepare_mid<8>, dim3(face_blocks, (unsigned)std::min(65535, n8 - at)), dim3(64), 0, st, sc,
                               fb, lists + at, n8, (double *)b_sph.p, (PolyMeta *)b_meta.p);
        for (int at = 0; at < n16; at += 65535)
            hipLaunchKernelGGL(k_vis_prepare_mid<16>, dim3(face_blocks, (unsigned)std::min(65535, n16 - at)), dim3(64), 0, st,
                               sc, fb, lists + class_end[0] + at, n16, (double *)b_sph.p, (PolyMeta *)b_meta.p);
        for (int at = 0; at < n32; at += 65535)
            hipLaunchKernelGGL(k_vis_prepare_mid<VIS_MID_POLY>, dim3(face_blocks, (unsigned)std::min(65535, n32 - at)), dim3(64),
                               0, st, sc, fb, lists + class_end[1] + at, n32, (double *)b_sph.p, (PolyMeta *)b_meta.p);
        if (n_rest)
            hipLaunchKernelGGL(k_vis_prepare_big, dim3(((unsigned)n_rest + 3) / 4, n_faces), dim3(256), 0, st, sc, fb,
                               lists + class_end[2], n_rest, (double *)b_sph.p, (PolyMeta *)b_meta.p);
    }
    if (job.ev[1]) RT_HIP(hipEventRecord(job.ev[1], st));
    if (!strict)
        hipLaunchKernelGGL(k_vis_sort, dim3(n_faces), dim3(VIS_SORT_THREADS), 0, st, sc.n, (const PolyMeta *)b_meta.p,
                           (int32_t *)b_order.p, m_stride, (int32_t *)b_counts.p);
    const int32_t *face_queue = nullptr, *item_off = nullptr;
    const double *tile_bounds = nullptr;
    if (!strict) {
        // faces in order of their candidate count, most first; each cut into tiles ("sectors") when it has very many
        // candidates; the persistent workgroups take (face, sector) items from a counter
        hipLaunchKernelGGL(k_face_queue, dim3(1), dim3(1024), 0, st, (const int32_t *)b_counts.p, n_faces, sc.n,
                           (int32_t *)job.b_queue.p);
        face_queue = (const int32_t *)job.b_queue.p;
        // Sectors pay when there are too few faces to fill the chip (the emitter's six screens: one workgroup each
        // would paint for as long as the whole matrix takes everybody else).  With hundreds of faces in flight they
        // only add work: a polygon is decided again in every sector it reaches into (lln's matrix: 0.64 s whole,
        // 3.1 s in sectors of 192).
        int tile_min = scene->vis_tile_min, tile_size = scene->vis_tile_size;
        if (tile_min < 0) {
            tile_min = n_faces <= 256 ? 512 : 0;
            tile_size = n_faces <= 256 ? 48 : 0;
            // A sixth of a large scene's faces or less (a rank's share of a sharded matrix on six GPUs or more): the
            // batch lasts as long as its longest faces while most workgroups have long run out of faces, so the
            // faces that see half the scene go in large sectors (lln, an eighth of the rows: 115 -> 85 ms; with a
            // quarter or more of the rows the sectors' extra work costs more than the idle workgroups were worth).
            // (tiles of 2 048 since they are rectangles with the bitmap inside: rank 0's eighth 74 -> 60 ms, rank 5's 84 -> 76)
            if (n_faces > 256 && (int64_t)n_faces * 6 <= (int64_t)sc.n) {
                tile_min = 4096;
                tile_size = 2048;
            }
        }
        // A half or a quarter of a large scene's faces (a rank's share on two to five GPUs): the 128 faces with the most
        // candidates go in tiles of 2 048 all the same (64 faces in tiles of 1 024: a half 117-124 ms, a quarter 84-88).
        // Painted whole the share lasts as long as its longest face (lln:
        // 100-130 ms whatever the share), in tiles a quarter of the rows takes 80 instead of 126 ms and a half 111
        // instead of 132; tiles for every face of 4 096 candidates and more, as for smaller shares, would double the
        // work (a half: 167 ms).
        // The whole scene on one GPU is bound by its work but for the one or two faces that see all of it (lln's
        // ground polygon: 7 605 candidates, 121 ms whole while every other face is done after 100): the four faces
        // with the most candidates go in tiles of 1 024 (lln's matrix 117 -> 94 ms; 1 to 64 faces in tiles of 512 to
        // 2 048 all give 94-96).
        const bool share = n_faces * 2 <= sc.n + 1;
        const int top_faces = share ? 128 : 4, top_per_tile = share ? 2048 : 1024;
        const bool top_sectors = scene->vis_tile_min < 0 && n_faces > 256 && tile_size == 0;
        hipLaunchKernelGGL(k_vis_tiles, dim3(n_faces), dim3(VIS_TILE_THREADS), 0, st, sc.n, (const PolyMeta *)b_meta.p,
                           (const int32_t *)b_order.p, m_stride, (const int32_t *)b_counts.p, tile_min, tile_size, face_queue,
                           top_sectors ? std::min(top_faces, n_faces) : 0, top_sectors ? top_per_tile : 0,
                           (double *)job.b_bounds.p, (int32_t *)job.b_ntiles.p);
        hipLaunchKernelGGL(k_vis_item_offsets, dim3(1), dim3(1024), 0, st, (const int32_t *)job.b_ntiles.p, face_queue, n_faces,
                           (int32_t *)job.b_itemoff.p);
        tile_bounds = (const double *)job.b_bounds.p;
        item_off = (const int32_t *)job.b_itemoff.p;
    }
    if (job.ev[2]) RT_HIP(hipEventRecord(job.ev[2], st));
    // The bitmap inside the tiles of four-wave workgroups (an eighth of lln's rows 91-94 -> 75-80 ms).  It was off for a
    // while: with it, small forced tiles once gave a wrong painting and one that did not end; the cause was the flag of
    // the full decision (polygon_uncovered), not the bitmap.
    const int grid_in_tiles = 1;
    PaintScratch ps{(int32_t *)b_list.p, (int32_t *)b_canvas.p, (double *)b_cbox.p, (double *)b_splits.p, split_cap,
                    (int32_t *)b_work.p, (int32_t *)job.b_tlist.p, (uint32_t *)job.b_grid.p,
                    (int32_t *)job.b_chunk.p, grid_in_tiles};
    // at most one face per CU: the large canvas store (one workgroup per CU) costs no parallelism
    if (n_faces <= 256)
        hipLaunchKernelGGL((k_vis_paint<LargeCanvasStore, 16>), dim3(n_blocks), dim3(1024), 0, st, sc, fb,
                           (const double *)b_sph.p, (const PolyMeta *)b_meta.p, (const int32_t *)b_order.p, m_stride,
                           (const int32_t *)b_counts.p, ps, strict, (unsigned long long *)b_counter.p, face_queue, tile_bounds,
                           item_off, d_out);
    else
        hipLaunchKernelGGL((k_vis_paint<SmallCanvasStore, 4>), dim3(n_blocks), dim3(256), 0, st, sc, fb,
                           (const double *)b_sph.p, (const PolyMeta *)b_meta.p, (const int32_t *)b_order.p, m_stride,
                           (const int32_t *)b_counts.p, ps, strict, (unsigned long long *)b_counter.p, face_queue, tile_bounds,
                           item_off, d_out);
    if (job.ev[3]) RT_HIP(hipEventRecord(job.ev[3], st));
    RT_HIP(hipGetLastError());
    return 0;
}

// visibility rows of a batch of faces into d_out[n_faces*n], on the scene's stream, waited for; the batch's device
// times are added to ms4 (mapping, sorting, painting, all)
static int visibility_rows(rt_scene *scene, const std::vector<int32_t> &a_off, const std::vector<double> &a_verts,
                           const std::vector<double> &a_normal, const std::vector<int32_t> &a_self, int strict,
                           uint8_t *d_out, double *ms4, int scene_f0 = -1, int scene_count = 0) {
    RowsJob job;
    if (visibility_rows_enqueue(scene, scene->stream, a_off, a_verts, a_normal, a_self, strict, d_out, job, scene_f0,
                                scene_count)) {
        (void)hipStreamSynchronize(scene->stream);
        return -1;
    }
    RT_HIP(hipStreamSynchronize(scene->stream));
    job.add_times(ms4);
    return 0;
}

extern "C" int rt_faces_visibility(rt_scene *scene, int32_t n_faces, const int32_t *face_off, const double *face_verts,
                                   const double *face_normals, const int32_t *self_index, int strict, uint8_t *out) {
    if (!scene || !face_off || !face_verts || !face_normals || !out) RT_FAIL("rt_faces_visibility: null argument");
    if (n_faces < 0) RT_FAIL("rt_faces_visibility: negative count");
    if (n_faces == 0) return 0;
    if (n_faces > 65535) RT_FAIL("rt_faces_visibility: at most 65535 faces per call");
    if (face_off[0] != 0) RT_FAIL("rt_faces_visibility: face_off must start at 0");
    for (int f = 0; f < n_faces; ++f) {
        if (face_off[f + 1] - face_off[f] < 3) RT_FAIL("rt_faces_visibility: a face needs at least 3 vertices");
        if (self_index && (self_index[f] < -1 || self_index[f] >= scene->view.n))
            RT_FAIL("rt_faces_visibility: self_index out of range");
    }
    RT_HIP(rt_use_device(scene->device));
    std::vector<int32_t> a_off(face_off, face_off + n_faces + 1), a_self(n_faces, -1);
    if (self_index) a_self.assign(self_index, self_index + n_faces);
    std::vector<double> a_verts(face_verts, face_verts + 3 * (size_t)face_off[n_faces]);
    std::vector<double> a_normal(face_normals, face_normals + 3 * (size_t)n_faces);
    DevBuf d_out;
    if (d_out.alloc((size_t)scene->view.n * n_faces)) RT_FAIL("rt_faces_visibility: out of device memory");
    for (int k = 4; k < 8; ++k) scene->last_vis_ms[k] = 0;
    if (visibility_rows(scene, a_off, a_verts, a_normal, a_self, strict, (uint8_t *)d_out.p, scene->last_vis_ms + 4)) return -1;
    RT_HIP(hipMemcpy(out, d_out.p, (size_t)scene->view.n * n_faces, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rt_face_visibility(rt_scene *scene, const double *face_verts, int32_t n_face_verts,
                                  const double *face_normal, int32_t self_index, int strict, uint8_t *out) {
    const int32_t off[2] = {0, n_face_verts};
    return rt_faces_visibility(scene, 1, off, face_verts, face_normal, &self_index, strict, out);
}

// Rows of the visibility matrix before symmetrisation: row k of `out` = what polygon row_ids[k] sees
// (geometry.py:1036-1046 restricted to some i).  The full matrix is the OR of all rows with their transpose; a node
// shares the rows out over its GPUs and all-gathers them.
static int visibility_listed_rows(rt_scene *scene, int32_t n_rows, const int32_t *row_ids, int strict, uint8_t *out,
                                  int on_device) {
    if (!scene || (n_rows > 0 && (!row_ids || !out))) RT_FAIL("rt_visibility_rows: null argument");
    if (n_rows < 0) RT_FAIL("rt_visibility_rows: negative count");
    if (n_rows == 0) return 0;
    const int n = scene->view.n;
    RT_HIP(rt_use_device(scene->device));
    for (int k = 0; k < n_rows; ++k)
        if (row_ids[k] < 0 || row_ids[k] >= n) RT_FAIL("rt_visibility_rows: row index out of range");
    DevBuf d_own;
    if (!on_device && d_own.alloc((size_t)n * n_rows)) RT_FAIL("rt_visibility_rows: out of device memory");
    uint8_t *d_dst = on_device ? out : (uint8_t *)d_own.p;  // a destination on the device is painted into directly
    const size_t per_face = sizeof(double) * 2 * (size_t)scene->view.vtot + (sizeof(PolyMeta) + 8) * (size_t)n;
    int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_rows, ((size_t)24 << 30) / per_face));
    batch = std::min(batch, 65535);
    for (int k = 0; k < 4; ++k) scene->last_vis_ms[k] = 0;
    for (int k0 = 0; k0 < n_rows; k0 += batch) {
        const int k1 = std::min(n_rows, k0 + batch);
        std::vector<int32_t> a_off(1, 0), a_self;
        std::vector<double> a_verts, a_normal;
        for (int k = k0; k < k1; ++k) {
            const int i = row_ids[k];
            const int v0 = scene->h_vert_off[i], v1 = scene->h_vert_off[i + 1];
            a_verts.insert(a_verts.end(), scene->h_verts.begin() + 3 * (size_t)v0, scene->h_verts.begin() + 3 * (size_t)v1);
            a_off.push_back(a_off.back() + (v1 - v0));
            a_normal.insert(a_normal.end(), scene->h_planes.begin() + 4 * (size_t)i, scene->h_planes.begin() + 4 * (size_t)i + 3);
            a_self.push_back(i);
        }
        if (visibility_rows(scene, a_off, a_verts, a_normal, a_self, strict, d_dst + (size_t)k0 * n, scene->last_vis_ms))
            return -1;
    }
    RT_HIP(hipStreamSynchronize(scene->stream));
    if (!on_device) RT_HIP(hipMemcpy(out, d_dst, (size_t)n * n_rows, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rt_visibility_rows(rt_scene *scene, int32_t n_rows, const int32_t *row_ids, int strict, uint8_t *out) {
    return visibility_listed_rows(scene, n_rows, row_ids, strict, out, 0);
}

extern "C" int rt_visibility_matrix(rt_scene *scene, int strict, uint8_t *out, int on_device) {
    if (!scene || !out) RT_FAIL("rt_visibility_matrix: null argument");
    RT_HIP(rt_use_device(scene->device));
    const SceneView &sc = scene->view;
    const int n = sc.n;
    DevBuf d_rows, d_sym;
    if (d_rows.alloc((size_t)n * n)) RT_FAIL("rt_visibility_matrix: out of device memory");
    // batches of faces sized so the per-face scratch (mapped vertices + per-polygon records) stays under ~24 GiB: one
    // batch for every scene of the reference (a batch ends when its slowest face ends, so fewer batches is faster)
    const size_t per_face = sizeof(double) * 2 * (size_t)sc.vtot + (sizeof(PolyMeta) + 8) * (size_t)n;
    int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)24 << 30) / per_face));
    batch = std::min(batch, 65535);
    for (int k = 0; k < 4; ++k) scene->last_vis_ms[k] = 0;
    for (int f0 = 0; f0 < n; f0 += batch) {
        const int f1 = std::min(n, f0 + batch);
        // the faces are the scene's own polygons: read from the device's copy of the scene
        const std::vector<int32_t> no_ints;
        const std::vector<double> no_doubles;
        if (visibility_rows(scene, no_ints, no_doubles, no_doubles, no_ints, strict, (uint8_t *)d_rows.p + (size_t)f0 * n,
                            scene->last_vis_ms, f0, f1 - f0))
            return -1;
    }
    uint8_t *d_dst = out;
    if (!on_device) {
        if (d_sym.alloc((size_t)n * n)) RT_FAIL("rt_visibility_matrix: out of device memory");
        d_dst = (uint8_t *)d_sym.p;
    }
    hipLaunchKernelGGL(k_symmetrise, dim3((n + 255) / 256, n), dim3(256), 0, scene->stream, (const uint8_t *)d_rows.p, n, d_dst);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(scene->stream));
    if (!on_device) RT_HIP(hipMemcpy(out, d_dst, (size_t)n * n, hipMemcpyDeviceToHost));
    return 0;
}

// The visibility matrix (or some of its rows) and the visibility vectors of extra faces (the emitter's screens) in one
// call: what RayTracingProblem.precompute asks for one after the other (ray_tracing.py:208 then :210-228).  A few
// screens occupy a few compute units for as long as the whole matrix takes the rest of the chip, so the screens are
// painted on a stream of their own while the matrix is painted on the scene's.
// row_ids null: rows_out[N*N] = the symmetrised matrix (rt_visibility_matrix); otherwise rows_out[n_rows*N] = the
// listed rows as rt_visibility_rows gives them.  outputs_on_device: bit 0 -- rows_out, bit 1 -- faces_out is device
// memory of the scene's GPU and that result never visits the host: copying lln's 58 MB into pageable memory takes
// 33 ms, a sixth of what painting it takes; rt_enumerate can read the matrix where it lies, and a rank of a sharded
// run has its rows and screens painted straight into the buffer that the all-gather sends.
extern "C" int rt_visibility_with_faces(rt_scene *scene, int32_t n_rows, const int32_t *row_ids, int strict,
                                        uint8_t *rows_out, int32_t n_faces, const int32_t *face_off,
                                        const double *face_verts, const double *face_normals, const int32_t *self_index,
                                        int faces_strict, uint8_t *faces_out, int outputs_on_device) {
    if (!scene || !rows_out) RT_FAIL("rt_visibility_with_faces: null argument");
    if (n_faces < 0 || (row_ids && n_rows < 0)) RT_FAIL("rt_visibility_with_faces: negative count");
    if (outputs_on_device < 0 || outputs_on_device > 3) RT_FAIL("rt_visibility_with_faces: outputs_on_device must be 0..3");
    const int rows_on_device = outputs_on_device & 1, faces_on_device = (outputs_on_device >> 1) & 1;
    if (n_faces == 0) return row_ids ? visibility_listed_rows(scene, n_rows, row_ids, strict, rows_out, rows_on_device)
                                     : rt_visibility_matrix(scene, strict, rows_out, rows_on_device);
    if (!face_off || !face_verts || !face_normals || !faces_out) RT_FAIL("rt_visibility_with_faces: null argument");
    if (n_faces > 65535) RT_FAIL("rt_visibility_with_faces: at most 65535 faces per call");
    if (face_off[0] != 0) RT_FAIL("rt_visibility_with_faces: face_off must start at 0");
    for (int f = 0; f < n_faces; ++f) {
        if (face_off[f + 1] - face_off[f] < 3) RT_FAIL("rt_visibility_with_faces: a face needs at least 3 vertices");
        if (self_index && (self_index[f] < -1 || self_index[f] >= scene->view.n))
            RT_FAIL("rt_visibility_with_faces: self_index out of range");
    }
    RT_HIP(rt_use_device(scene->device));
    // a second stream of the default priority (streams of different priorities are not used anywhere in this library)
    if (!scene->side_stream) RT_HIP(rt_take_stream(scene->device, &scene->side_stream));
    const size_t n = (size_t)scene->view.n;
    std::vector<int32_t> a_off(face_off, face_off + n_faces + 1), a_self(n_faces, -1);
    if (self_index) a_self.assign(self_index, self_index + n_faces);
    std::vector<double> a_verts(face_verts, face_verts + 3 * (size_t)face_off[n_faces]);
    std::vector<double> a_normal(face_normals, face_normals + 3 * (size_t)n_faces);
    DevBuf d_faces;
    if (!faces_on_device && d_faces.alloc(n * n_faces)) RT_FAIL("rt_visibility_with_faces: out of device memory");
    uint8_t *d_faces_dst = faces_on_device ? faces_out : (uint8_t *)d_faces.p;
    RowsJob job;  // after d_faces and the host arrays: the stream is waited for before any of them goes
    // Beside many rows the screens get 96 workgroups instead of one per compute unit: a workgroup of the screens'
    // painter fills its unit's registers, so with 256 of them the rows' first kernels wait until the first screens'
    // workgroups run out of tiles (lln: 10 ms in) (lln, matrix and screens: 169 -> 161 ms when the screens were
    // painted in azimuth sectors, 155 ms since their tiles are rectangles; a call that paints few rows keeps one
    // workgroup per unit, which the screens alone need: 33 ms).
    const int beside_rows = (row_ids ? n_rows : scene->view.n) > 256 ? 96 : 0;
    int rc = visibility_rows_enqueue(scene, scene->side_stream, a_off, a_verts, a_normal, a_self, faces_strict, d_faces_dst,
                                     job, -1, 0, beside_rows);
    if (rc == 0)
        rc = row_ids ? visibility_listed_rows(scene, n_rows, row_ids, strict, rows_out, rows_on_device)
                     : rt_visibility_matrix(scene, strict, rows_out, rows_on_device);
    const hipError_t e = hipStreamSynchronize(scene->side_stream);
    for (int k = 4; k < 8; ++k) scene->last_vis_ms[k] = 0;
    if (rc == 0 && e == hipSuccess) job.add_times(scene->last_vis_ms + 4);
    if (rc) return -1;
    if (e != hipSuccess) RT_FAIL(std::string("rt_visibility_with_faces: ") + hipGetErrorString(e));
    if (!faces_on_device) RT_HIP(hipMemcpy(faces_out, d_faces_dst, n * n_faces, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rt_debug_set_visibility_tiling(rt_scene *scene, int32_t min_candidates, int32_t per_tile) {
    if (!scene || min_candidates < -1 || per_tile < 0) RT_FAIL("rt_debug_set_visibility_tiling: bad argument");
    scene->vis_tile_min = min_candidates;
    scene->vis_tile_size = per_tile;
    return 0;
}

extern "C" int rt_visibility_last_timing(rt_scene *scene, double *ms8) {
    if (!scene || !ms8) RT_FAIL("rt_visibility_last_timing: null argument");
    for (int k = 0; k < 8; ++k) ms8[k] = scene->last_vis_ms[k];
    return 0;
}

// Painter work counters accumulated since the last call (profiling aid; zeroes them).
extern "C" int rt_debug_visibility_stats(unsigned long long *out8) {
    if (!out8) RT_FAIL("rt_debug_visibility_stats: null argument");
    RT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_vis_stats), sizeof(unsigned long long) * 8));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_vis_stats), zero, sizeof(zero)));
    return 0;
}
