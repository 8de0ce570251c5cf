// Shared declarations of the path-solve library: scene layout in HBM, error handling, small fp64 vector helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rt_pathsolve.h"

#define RT_WAVE 64

// ---------------------------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------------------------
void rt_set_error(const std::string &msg);

#define RT_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            rt_set_error(std::string(#call) + ": " + hipGetErrorString(e_) + " at " + __FILE__ + ":" +    \
                         std::to_string(__LINE__));                                                       \
            return -1;                                                                                    \
        }                                                                                                 \
    } while (0)

// Makes `device` current and forgets an error that another user of the HIP runtime in this process left behind:
// hipGetLastError reports the last failure of any call on the thread (PyTorch's included) until somebody reads it, and a
// check after one of this library's launches must speak of that launch only.
// Streams of the default priority, kept for the life of the process: a scene takes its streams from here and hands them
// back when it is destroyed (making one costs 2 to 3 ms, a tenth of what some scenes take to set up).  The device must
// be current.  Defined in scene.hip.
hipError_t rt_take_stream(int device, hipStream_t *out);
void rt_give_stream(int device, hipStream_t stream);

static inline hipError_t rt_use_device(int device) {
    const hipError_t e = hipSetDevice(device);
    (void)hipGetLastError();
    return e;
}

#define RT_HIP_NULL(call)                                                                                 \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            rt_set_error(std::string(#call) + ": " + hipGetErrorString(e_) + " at " + __FILE__ + ":" +    \
                         std::to_string(__LINE__));                                                       \
            return nullptr;                                                                               \
        }                                                                                                 \
    } while (0)

#define RT_FAIL(msg)           \
    do {                       \
        rt_set_error(msg);     \
        return -1;             \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// scene: structure-of-arrays in device memory (all read-only after creation)
// ---------------------------------------------------------------------------------------------------------------
// One polygon as the obstruction pre-test wants it, 64 bytes: plane in fp64, box in fp32 relative to the grid origin
// and rounded outwards (the pre-test only has to be conservative).
// rows of the material table of the field step (materials_properties.py:47-109); a scene's material ids index it
#define RT_N_MATERIALS 7

struct OccRec {
    double nx, ny, nz, d;
    float lo[3], hi[3];
    int32_t poly;
    int32_t pad_;
};

// One grid cell: its records are cell_recs[begin, end); zmax = top of the highest polygon registered in it, relative to
// the grid origin and rounded up (a segment passing above it skips the cell).
struct OccCell {
    int32_t begin, end;
    float zmax;
    int32_t pad_;
};

struct SceneView {
    int32_t n;               // polygons
    int32_t vtot;            // vertices
    int32_t vmax;            // most vertices in one polygon
    const int32_t *vert_off; // [n+1]
    const double *verts;     // [vtot*3] world coordinates
    const double *planes;    // [n*4] nx ny nz d (as the polygon reports them)
    const double *frames;    // [n*9] rows: x axis, y axis, z axis (= reported normal) of the polygon's own frame
    const double *bbox;      // [n*6] min xyz, max xyz
    const double *uv;        // [vtot*2] vertices in the polygon's own frame (first two coordinates)
    const double *uv_recs;   // [n*16] x axis (3), y axis (3) of the own frame, then min u, min v, max u, max v of the ring
    const int32_t *material; // [n]
    const int32_t *building; // [n]
    // the polygons of each building, for the shortcut of the diffraction kernel (a ray that leaves an edge into its own
    // building is tested against that building's faces at once): building b of the polygons that were given an id owns
    // bld_polys[bld_off[b] .. bld_off[b + 1]); bld_of[p] = b, or -1 for a polygon without one.  n_bld = 0 without ids.
    int32_t n_bld;
    const int32_t *bld_off;   // [n_bld + 1]
    const int32_t *bld_polys; // [n]
    const int32_t *bld_of;    // [n]
    // Uniform grid of square cells over the horizontal extent of the scene (buildings are extrusions, so a 2-D grid
    // separates them well).  Cell c = row * gnx + column lists every polygon whose footprint comes within RT_GRID_PAD of
    // it; the extra cell gnx*gny lists the polygons too large to be worth registering cell by cell (the ground), which
    // every segment tests.
    double gx0, gy0, gz0;    // grid origin (minimum corner of the scene)
    double gside, ginv;      // cell side and its inverse
    int32_t gnx, gny;
    const OccCell *cells;    // [gnx*gny + 1]
    const OccRec *cell_recs;
    const OccRec *poly_recs; // [n] the same records in polygon order (every-polygon walk)
};

#define RT_GRID_PAD 1e-5

struct rt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t caller_event = nullptr;  // rt_scene_wait_for_stream: marks the caller's stream, created on first use
    hipStream_t side_stream = nullptr;  // created on first use: work that runs beside the main stream's (emitter screens,
                                        // the receiver-independent head of a large solve)
    hipEvent_t side_event = nullptr;    // end of the side stream's part of a solve, which the main stream waits for
    hipEvent_t rejudge_event = nullptr; // a point of the main stream that the side stream's next piece of work waits for
                                        // (the start of a solve; the start of the re-judging step)
    SceneView view{};
    std::vector<void *> allocations;
    // host copies used by host-side set-up code
    std::vector<int32_t> h_vert_off;
    std::vector<double> h_verts, h_planes, h_bbox;
    // growable workspace owned by the solve
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // solve state that outlives a call, so that a call in a loop allocates nothing: device counters, their pinned host
    // mirror, timing events, and the sizes the last call needed (the next call's launch and queue sizes start there)
    unsigned long long *d_counters = nullptr;  // [16]
    unsigned long long *h_counters = nullptr;  // [16] pinned
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    bool timing = true;  // per-kernel-family events around the launches of a solve
    int reflect_mode = 0;  // 0 by batch size, 1 receivers one by one, 2 receivers in spatial tiles
    int wall_hints = 1;    // 1: the diffraction kernel of a large batch uses wall hints around the receivers (default), 0: never
    bool counters_dirty = true;  // d_counters[0..3] may be non-zero (first solve, or a solve that failed half-way)
    int64_t hint_paths = 0, hint_segments = 0, hint_clear = 0;
    int32_t *late_items = nullptr;  // items handed from the first to the second obstruction pass (grows on demand)
    size_t late_cap = 0;
    // a reflection level whose chains would take more bytes than this is not stored: its chains are formed from the
    // level before when a kernel asks for them
    int64_t stored_chain_bytes_limit = 256ll << 20;
    int64_t queue_capacity_override = 0;  // > 0: first size of the path queue (tests of the grow-and-redo route)
    int occlusion_mode = 0;  // 0 = by scene size, 1 = every-polygon walk, 2 = grid walk
    bool reference_solver = false;  // re-judge every found path with the reference's own solver (reference_solver.hip)
    // painter: a face with at least vis_tile_min candidates is painted in tiles (rectangles of its view) of about vis_tile_size
    // candidates each, by different workgroups (vis_tile_size 0: never)
    // vis_tile_min -1 (the default): chosen per call, sectors only when a call paints few faces
    int vis_tile_min = -1, vis_tile_size = 0;
    // timing of the last solve
    double last_ms[4] = {0, 0, 0, 0};
    int64_t last_launches[4] = {0, 0, 0, 0};
    int64_t last_segments = 0;
    // the reference-compatible re-judging of the last solve: device time of its kernels (zero with the timing off),
    // paths that went in, paths kept
    double last_rejudge_ms = 0;
    int64_t last_rejudge_in = 0, last_rejudge_kept = 0;
    // device time of the last rt_fields call: [0] k_fields, [1] everything the call queued
    double last_fields_ms[2] = {0, 0};
    // rt_scene_set_deferred_sync: the device-pointer forms of the path description, the field step and the receiver
    // records then only queue their work; rt_scene_synchronize waits for it (and reads the field step's events)
    bool defer_sync = false;
    bool fields_events_pending = false;
    bool fields_events_with_kernels = false;
    // device time of the last visibility call: rows / matrix [0] mapping, [1] sorting, [2] painting, [3] all of it;
    // [4..7] the same for the extra faces of rt_visibility_with_faces / rt_faces_visibility
    double last_vis_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Device memory a call needs only while it runs (candidate set-up, receiver tiles, staging of host arrays) is taken
    // from this pool and handed back when the call ends, so that a call in a loop stops allocating after the first.
    struct TempBlock {
        void *p;
        size_t bytes;
        bool busy;
        int idle;  // calls in a row that did not use it
    };
    std::vector<TempBlock> temp_pool;
};

// a block of at least `bytes` from the scene's pool (nullptr when the device is out of memory); valid until
// rt_scene_temp_release_all, which every entry point calls before it returns
void *rt_scene_temp(rt_scene *scene, size_t bytes);
void rt_scene_temp_release_all(rt_scene *scene);

int rt_scene_workspace(rt_scene *scene, size_t bytes, void **out);
// looks up, once per device and with that device current, the address of the fixed scratch array that
// rt_obstruct_segments_noalloc works in (occlusion.hip)
int rt_noalloc_setup(int device);

// A path waiting for its obstruction test: its interaction points and where its segments are in the segment queue.
// A path is named by (group, index of the candidate inside its group).  Both travel as 32-bit integers, in the path
// records and in the arrays of rt_solve_paths / rt_describe_paths.  A reflection level that is not stored can hold more
// than 2^31 - 1 candidates (order 4 on a town of some thousand polygons); the index is then split: its low 31 bits are
// the "candidate" word, the bits above go into bits 8 and up of the "group" word, whose low 8 bits are the group.  For
// every index below 2^31 the two words are what they always were; a line of sight keeps group -1.
#define RT_GROUP_BITS 8
#define RT_MAX_WIDE_CANDIDATES (1ll << 53)
__host__ __device__ inline int32_t rt_group_word(int group, int64_t cand) {
    return group < 0 ? group : (int32_t)((int64_t)group | ((cand >> 31) << RT_GROUP_BITS));
}
__host__ __device__ inline int32_t rt_cand_word(int64_t cand) { return (int32_t)(cand & 0x7fffffffll); }
__host__ __device__ inline int rt_group_of(int32_t group_word) { return group_word < 0 ? group_word : (group_word & ((1 << RT_GROUP_BITS) - 1)); }
__host__ __device__ inline int64_t rt_cand_of(int32_t group_word, int32_t cand_word) {
    return group_word < 0 ? (int64_t)cand_word : (((int64_t)(group_word >> RT_GROUP_BITS)) << 31) | (int64_t)cand_word;
}

struct PathRec {
    int32_t rx, group, cand, nseg;  // group, cand: the two words above; nseg: segments still to be tested, counted from the receiver's end
    int64_t seg0;                   // segment k of those is segs[seg0 + k * seg_stride] when the segment queue is kept
    int32_t seg_stride, npts;       // npts interaction points: emitter, pts[0..npts), receiver is the whole path
    double pts[RT_MAX_ORDER * 3];
};

// ---------------------------------------------------------------------------------------------------------------
// fp64 3-vectors
// ---------------------------------------------------------------------------------------------------------------
struct d3 {
    double x, y, z;
};

__host__ __device__ inline d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
__host__ __device__ inline d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
__host__ __device__ inline void st3(double *p, d3 a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}
__host__ __device__ inline d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline d3 operator*(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
__host__ __device__ inline double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline d3 cross(d3 a, d3 b) {
    return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__host__ __device__ inline double norm3(d3 a) { return sqrt(dot(a, a)); }
__host__ __device__ inline d3 divs(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }

// Frame whose z axis is along v: y = ones - (ones·z) z, normalised; x = y × z (reference geometry.py:414-435).
__host__ __device__ inline void frame_from_vector(d3 v, d3 &fx, d3 &fy, d3 &fz) {
    fz = divs(v, norm3(v));
    double s = fz.x + fz.y + fz.z;
    fy = d3{1.0 - s * fz.x, 1.0 - s * fz.y, 1.0 - s * fz.z};
    fy = divs(fy, norm3(fy));
    fx = cross(fy, fz);
}

// Crossing-number point-in-polygon on a ring given by a coordinate functor; boundary points are inside
// (the predicate Shapely's Polygon.intersects(Point) answers in reference geometry.py:877, 1803).
template <typename GetXY>
__host__ __device__ inline bool pip_inclusive(int nv, GetXY get, double px, double py) {
    bool inside = false;
    double ax, ay;
    get(nv - 1, ax, ay);
    for (int k = 0; k < nv; ++k) {
        double bx, by;
        get(k, bx, by);
        double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
        if (cr == 0.0 && fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by))
            return true;
        if ((ay > py) != (by > py)) {
            if ((cr > 0.0) == (by > ay)) inside = !inside;
        }
        ax = bx;
        ay = by;
    }
    return inside;
}

// Exact form of the reference's per-polygon obstruction test (geometry.py:857-884) for one polygon and one
// segment A->B.
__device__ inline bool polygon_cuts_segment(const SceneView &sc, int p, d3 A, d3 B) {
    const double tol = 1e-8;
    d3 V = B - A;
    d3 fx, fy, fz;
    frame_from_vector(V, fx, fy, fz);
    const double z_min = dot(A, fz), z_max = dot(B, fz);
    const int v0 = sc.vert_off[p], nv = sc.vert_off[p + 1] - v0;
    // extent of the polygon along the segment (geometry.py:866-871): some vertex beyond the start and some before the
    // end.  The polygon's box bounds every vertex, so where the box settles the matter (with a margin far above the
    // rounding of the sums) the loop over the vertices is not needed; a roof can have tens of them.
    {
        const double *bb = sc.bbox + 6 * (size_t)p;
        const double cx = 0.5 * (bb[0] + bb[3]), cy = 0.5 * (bb[1] + bb[4]), cz = 0.5 * (bb[2] + bb[5]);
        const double zc = cx * fz.x + cy * fz.y + cz * fz.z;
        const double r = 0.5 * ((bb[3] - bb[0]) * fabs(fz.x) + (bb[4] - bb[1]) * fabs(fz.y) + (bb[5] - bb[2]) * fabs(fz.z));
        const double eps = 1e-9 * (fabs(zc) + r + fabs(z_min) + fabs(z_max) + 1.0);
        if (zc + r + eps <= z_min || zc - r - eps >= z_max) return false;
        if (!(zc - r - eps > z_min && zc + r + eps < z_max)) {
            double zlo = INFINITY, zhi = -INFINITY;
            for (int v = 0; v < nv; ++v) {
                double z = dot(ld3(sc.verts + 3 * (size_t)(v0 + v)), fz);
                zlo = fmin(zlo, z);
                zhi = fmax(zhi, z);
            }
            if (!(zhi > z_min && zlo < z_max)) return false;
        }
    }
    const double px = dot(A, fx), py = dot(A, fy);
    const double *vb = sc.verts + 3 * (size_t)v0;
    bool in = pip_inclusive(
        nv,
        [&](int k, double &x, double &y) {
            d3 P = ld3(vb + 3 * k);
            x = dot(P, fx);
            y = dot(P, fy);
        },
        px, py);
    if (!in) return false;
    d3 n = ld3(sc.planes + 4 * (size_t)p);
    double d = sc.planes[4 * (size_t)p + 3];
    double t = -(d + dot(A, n)) / dot(V, n);
    return tol < t && t < 1 - tol;
}

// The same test arranged for launches that are bound by latency rather than throughput: one pass over the ring with
// the vertex loads issued four at a time and no early exit (more arithmetic, a quarter of the dependent loads).  Same
// result as polygon_cuts_segment.
__device__ inline bool polygon_cuts_segment_batched(const SceneView &sc, int p, d3 A, d3 B) {
    const double tol = 1e-8;
    d3 V = B - A;
    d3 fx, fy, fz;
    frame_from_vector(V, fx, fy, fz);
    const double z_min = dot(A, fz), z_max = dot(B, fz);
    const int v0 = sc.vert_off[p], nv = sc.vert_off[p + 1] - v0;
    // extent of the polygon along the segment (geometry.py:866-871): some vertex beyond the start and some before the
    // end.  The polygon's box bounds every vertex, so where the box says no (with a margin far above the rounding of
    // the sums) the vertices need not be looked at.
    {
        const double *bb = sc.bbox + 6 * (size_t)p;
        const double cx = 0.5 * (bb[0] + bb[3]), cy = 0.5 * (bb[1] + bb[4]), cz = 0.5 * (bb[2] + bb[5]);
        const double zc = cx * fz.x + cy * fz.y + cz * fz.z;
        const double r = 0.5 * ((bb[3] - bb[0]) * fabs(fz.x) + (bb[4] - bb[1]) * fabs(fz.y) + (bb[5] - bb[2]) * fabs(fz.z));
        const double eps = 1e-9 * (fabs(zc) + r + fabs(z_min) + fabs(z_max) + 1.0);
        if (zc + r + eps <= z_min || zc - r - eps >= z_max) return false;
    }
    // One pass over the ring for both the extent and the crossing-number test of the segment's start in the plane
    // across the segment (pip_inclusive's arithmetic), four vertices at a time with their loads issued together: the
    // test is a chain of dependent loads, and a roof has tens of vertices.
    const double px = dot(A, fx), py = dot(A, fy);
    const double *vb = sc.verts + 3 * (size_t)v0;
    double zlo = INFINITY, zhi = -INFINITY;
    bool inside = false, boundary = false;
    double ax, ay;
    {
        const d3 P = ld3(vb + 3 * (nv - 1));
        ax = dot(P, fx);
        ay = dot(P, fy);
    }
    for (int k0 = 0; k0 < nv; k0 += 4) {
        d3 P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] = ld3(vb + 3 * (k0 + j < nv ? k0 + j : nv - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < nv) {
                const double z = dot(P[j], fz);
                zlo = fmin(zlo, z);
                zhi = fmax(zhi, z);
                const double bx = dot(P[j], fx), by = dot(P[j], fy);
                const double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
                if (cr == 0.0 && fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by))
                    boundary = true;
                if ((ay > py) != (by > py)) {
                    if ((cr > 0.0) == (by > ay)) inside = !inside;
                }
                ax = bx;
                ay = by;
            }
        }
    }
    if (!(zhi > z_min && zlo < z_max)) return false;
    if (!(boundary || inside)) return false;
    d3 n = ld3(sc.planes + 4 * (size_t)p);
    double d = sc.planes[4 * (size_t)p + 3];
    double t = -(d + dot(A, n)) / dot(V, n);
    return tol < t && t < 1 - tol;
}

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);  // one Newton step: plenty for a pre-test with a 1e-6 pad
}

// Padded, division-free form of the obstruction test for one pre-test record: false only when the exact test
// (polygon_cuts_segment) is certainly false.  pad = 1e-5 max|V| + 1e-6.
__device__ __forceinline__ bool occ_pretest(const OccRec *__restrict__ r, const SceneView &sc, d3 A, d3 V, double pad) {
    const double nx = r->nx, ny = r->ny, nz = r->nz, d = r->d;
    const double den = V.x * nx + V.y * ny + V.z * nz;
    const double num = d + (A.x * nx + A.y * ny + A.z * nz);
    const double ad = fabs(den);
    const double sn = den < 0.0 ? num : -num;
    bool cand = sn > 1e-8 * ad - 1e-11 && sn < (1.0 - 1e-8) * ad + 1e-11;
    const double t = sn * fast_rcp(ad);
    const double hx = A.x + t * V.x - sc.gx0, hy = A.y + t * V.y - sc.gy0, hz = A.z + t * V.z - sc.gz0;
    return cand && hx >= (double)r->lo[0] - pad && hx <= (double)r->hi[0] + pad && hy >= (double)r->lo[1] - pad &&
           hy <= (double)r->hi[1] + pad && hz >= (double)r->lo[2] - pad && hz <= (double)r->hi[2] + pad;
}

// OrientedPolygon.contains_point with check_in_plane (reference geometry.py:1792-1803): |n·p + d| within
// atol + 1e-5|d| (numpy.allclose), then point-in-polygon in the polygon's own frame.
__device__ inline bool polygon_contains_point(const SceneView &sc, int p, d3 X, bool check_in_plane, double plane_tol) {
    if (check_in_plane) {
        d3 n = ld3(sc.planes + 4 * (size_t)p);
        double d = sc.planes[4 * (size_t)p + 3];
        if (!(fabs(dot(n, X) + d) <= plane_tol + 1e-5 * fabs(d))) return false;
    }
    const double *f = sc.frames + 9 * (size_t)p;
    double px = dot(X, ld3(f)), py = dot(X, ld3(f + 3));
    const int v0 = sc.vert_off[p], nv = sc.vert_off[p + 1] - v0;
    const double *uv = sc.uv + 2 * (size_t)v0;
    // pip_inclusive's arithmetic, four ring vertices at a time with their loads issued together and no early exit
    bool inside = false, boundary = false;
    double ax = uv[2 * (nv - 1)], ay = uv[2 * (nv - 1) + 1];
    for (int k0 = 0; k0 < nv; k0 += 4) {
        double bxs[4], bys[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j < nv ? k0 + j : nv - 1;
            bxs[j] = uv[2 * k];
            bys[j] = uv[2 * k + 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < nv) {
                const double bx = bxs[j], by = bys[j];
                const double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
                if (cr == 0.0 && fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by))
                    boundary = true;
                if ((ay > py) != (by > py)) {
                    if ((cr > 0.0) == (by > ay)) inside = !inside;
                }
                ax = bx;
                ay = by;
            }
        }
    }
    return boundary || inside;
}
