// Face-to-scene visibility (reference geometry.py:924-1048, non-strict branch; strict = the z filter and the
// normal filter only).
//
// For a face A every other polygon B is moved to A's frame (origin = centre of A's bounding box), dropped unless it
// reaches z > 1e-4, mapped vertex by vertex to (azimuth, polar angle, distance), and the survivors are painted front
// to back in the order (nearest vertex distance, length of the centre of the mapped bounding box, index): B passes
// when its (azimuth, polar) polygon is a valid ring and some part of it is not covered by the polygons painted before
// it.  Last, B must face A: some vertex p of A has normalize(p - centre(B))·n_B > 1e-4.
//
// "Painted before" is a pairwise relation on the sort keys, so no sort and no sequential canvas are needed: every
// (A, B) pair is decided independently against the set {C valid, C before B, box(C) meets box(B)}.  A polygon hidden
// by the canvas lies inside it, so adding it to the canvas changes nothing: the set may contain hidden polygons.
//
// Coverage is decided exactly with witness points.  If part of B is uncovered, the uncovered region is bounded by
// pieces of B's edges (region on B's inner side) and pieces of the C's edges (region on that C's outer side).  Every
// edge is cut at its crossings with all other edges and at the feet of endpoints that lie on it; along one piece
// nothing changes, so one point next to the middle of each piece (DELTA off the edge, on the side named above)
// decides it: B is visible iff some such point is inside B and outside every C.
//
// Kernels: k_vis_prepare (one thread per (face, polygon): mapping, keys, ring validity) and k_vis_paint (persistent
// waves pulling (face, polygon) pairs from a counter; within a pair, lanes work on different polygons / pieces while
// the polygon being tested against is wave-uniform and comes through scalar loads).
#include <algorithm>
#include <cstring>

#include "rt_common.h"

#define TOL_DZ 1e-4      /* geometry.py:947 */
#define TOL_DOT 1e-4     /* geometry.py:948 */
#define VIS_DELTA 2.5e-10    /* witness offset from its edge, radians */
#define VIS_MIN_PIECE 1e-9   /* shorter pieces of an edge are not examined */
#define VIS_ON_LINE 1e-9     /* an endpoint this close to an edge's line cuts the edge at its foot */

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
__device__ bool ring_is_valid(const double *x, const double *y, int n_in, int *sgn) {
    for (int i = 0; i < n_in; ++i)
        if (!isfinite(x[i]) || !isfinite(y[i])) return false;
    // a vertex equal to its predecessor is a repeat and is skipped (repeated points are accepted)
    auto kept = [&](int i) {
        const int p = (i + n_in - 1) % n_in;
        return !(x[i] == x[p] && y[i] == y[p]);
    };
    int n = 0;
    for (int i = 0; i < n_in; ++i) n += kept(i) ? 1 : 0;
    if (n < 3) return false;
    double area2 = 0;
    for (int i = 0; i < n_in; ++i) {
        const int j = (i + 1) % n_in;
        area2 += x[i] * y[j] - x[j] * y[i];
    }
    if (area2 == 0) return false;
    *sgn = area2 > 0 ? 1 : -1;
    auto next_kept = [&](int i) {
        int j = (i + 1) % n_in;
        while (!kept(j)) j = (j + 1) % n_in;
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

// one thread per (face, polygon)
__global__ __launch_bounds__(256) void k_vis_prepare(SceneView sc, FaceBatch fb, double *__restrict__ sph,
                                                      PolyMeta *__restrict__ meta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (b >= sc.n) return;
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
    const int v0 = sc.vert_off[b], nv = sc.vert_off[b + 1] - v0;
    double *az = sph + ((size_t)f * sc.vtot + v0) * 2;
    double *pol = az + nv;
    double zmax = -INFINITY, rmin = INFINITY;
    double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    for (int v = 0; v < nv; ++v) {
        const d3 q = ld3(sc.verts + 3 * (size_t)(v0 + v)) - c;
        const double qx = dot(q, fx), qy = dot(q, fy), qz = dot(q, fz);
        zmax = fmax(zmax, qz);
        const double r = sqrt(qx * qx + qy * qy + qz * qz);  // geometry.py:302-306
        const double a = atan2(qy, qx);
        const double p = acos(qz / r);
        az[v] = a;
        pol[v] = p;
        rmin = fmin(rmin, r);
        lo0 = fmin(lo0, a); hi0 = fmax(hi0, a);
        lo1 = fmin(lo1, p); hi1 = fmax(hi1, p);
        lo2 = fmin(lo2, r); hi2 = fmax(hi2, r);
    }
    if (!(zmax > TOL_DZ)) {  // geometry.py:967-970
        *out = m;
        return;
    }
    const double cx = (lo0 + hi0) / 2, cy = (lo1 + hi1) / 2, cz = (lo2 + hi2) / 2;
    m.key1 = rmin;  // geometry.py:980-984
    m.key2 = sqrt(cx * cx + cy * cy + cz * cz);
    m.xmin = lo0; m.xmax = hi0; m.ymin = lo1; m.ymax = hi1;
    int sgn = 1;
    m.state = ring_is_valid(az, pol, nv, &sgn) ? 2 : 1;
    m.sgn = sgn;
    *out = m;
}

__device__ inline bool painted_before(const PolyMeta &c, int ci, const PolyMeta &b, int bi) {
    if (c.key1 != b.key1) return c.key1 < b.key1;
    if (c.key2 != b.key2) return c.key2 < b.key2;
    return ci < bi;
}

// even-odd point-in-ring, ring data wave-uniform
__device__ inline bool pip_ring(const double *__restrict__ x, const double *__restrict__ y, int nv, double px, double py) {
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
    int32_t *list;    // per wave [n]
    double *splits;   // per wave [split_cap]
    int64_t split_cap;
};

// Is the witness next to the middle of piece (t0, t1) of edge a->b, on side `side` (+1 = left of a->b), inside B and
// outside every listed polygon?
__device__ inline bool witness_uncovered(const SceneView &sc, const double *sph_f, const PolyMeta *meta_f, int B,
                                         const int32_t *list, int n_list, double ax, double ay, double bx, double by,
                                         double t0, double t1, double side, bool need_in_b) {
    const double ex = bx - ax, ey = by - ay;
    const double len = sqrt(ex * ex + ey * ey);
    const double tm = 0.5 * (t0 + t1);
    const double wx = ax + tm * ex - side * VIS_DELTA * (ey / len);
    const double wy = ay + tm * ey + side * VIS_DELTA * (ex / len);
    if (need_in_b) {
        const int v0 = sc.vert_off[B], nv = sc.vert_off[B + 1] - v0;
        const double *x = sph_f + (size_t)v0 * 2;
        if (!pip_ring(x, x + nv, nv, wx, wy)) return false;
    }
    for (int i = 0; i < n_list; ++i) {
        const int C = list[i];
        const PolyMeta &mc = meta_f[C];
        if (wx < mc.xmin || wx > mc.xmax || wy < mc.ymin || wy > mc.ymax) continue;
        const int v0 = sc.vert_off[C], nv = sc.vert_off[C + 1] - v0;
        const double *x = sph_f + (size_t)v0 * 2;
        if (pip_ring(x, x + nv, nv, wx, wy)) return false;
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

// Examine every piece of edge a->b (owner side `side`): returns true (wave-uniform) when a witness proves B visible.
__device__ bool edge_has_witness(const SceneView &sc, const double *sph_f, const PolyMeta *meta_f, int B,
                                 const int32_t *list, int n_list, int owner, double ax, double ay, double bx, double by,
                                 double side, bool need_in_b, double *splits, int64_t split_cap, int *lds_count) {
    const int lane = threadIdx.x & 63;
    const double ex = bx - ax, ey = by - ay;
    const double len = sqrt(ex * ex + ey * ey);
    if (!(len > VIS_MIN_PIECE)) return false;
    // ---- gather the cuts: lanes = polygons of {B} ∪ list, each walks its own ring
    if (lane == 0) {
        *lds_count = 1;
        splits[0] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int base = -1; base < n_list; base += 64) {
        const int li = base + lane;  // -1 stands for B itself
        if (li < n_list) {
            const int P = li < 0 ? B : list[li];
            if (P != owner) {
                const PolyMeta &mp = meta_f[P];
                // the other ring can only cut this edge where their boxes meet
                const bool near = !(mp.xmax < fmin(ax, bx) - VIS_ON_LINE || mp.xmin > fmax(ax, bx) + VIS_ON_LINE ||
                                    mp.ymax < fmin(ay, by) - VIS_ON_LINE || mp.ymin > fmax(ay, by) + VIS_ON_LINE);
                if (near) {
                    const int v0 = sc.vert_off[P], nv = sc.vert_off[P + 1] - v0;
                    const double *x = sph_f + (size_t)v0 * 2, *y = x + nv;
                    double cx = x[nv - 1], cy = y[nv - 1];
                    for (int v = 0; v < nv; ++v) {
                        const double dx = x[v], dy = y[v];
                        double t[3];
                        const int k = edge_cuts(ax, ay, bx, by, cx, cy, dx, dy, t);
                        if (k > 0) {
                            const int slot = atomicAdd(lds_count, k);
                            for (int q = 0; q < k; ++q)
                                if (slot + q < split_cap) splits[slot + q] = t[q];
                        }
                        cx = dx;
                        cy = dy;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int n_splits = *lds_count;
    n_splits = __builtin_amdgcn_readfirstlane(n_splits);
    if (n_splits > split_cap) n_splits = (int)split_cap;  // cannot happen: capacity is 3 cuts per scene edge + 1
    // ---- every piece (t_j, next larger cut or 1)
    bool found = false;
    for (int base = 0; base < n_splits && !found; base += 64) {
        const int j = base + lane;
        bool mine = false;
        if (j < n_splits) {
            const double t0 = splits[j];
            double t1 = 1.0;
            for (int i = 0; i < n_splits; ++i) {
                const double ti = splits[i];
                if (ti > t0 && ti < t1) t1 = ti;
            }
            if ((t1 - t0) * len > VIS_MIN_PIECE)
                mine = witness_uncovered(sc, sph_f, meta_f, B, list, n_list, ax, ay, bx, by, t0, t1, side, need_in_b);
        }
        found = __any(mine);
    }
    return found;
}

__global__ __launch_bounds__(256) void k_vis_paint(SceneView sc, FaceBatch fb, const double *__restrict__ sph,
                                                    const PolyMeta *__restrict__ meta, PaintScratch ps, int strict,
                                                    unsigned long long *work_counter, uint8_t *__restrict__ out) {
    __shared__ int lds_counts[4];
    const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave_in_block;
    int32_t *list = ps.list + wave_id * sc.n;
    double *splits = ps.splits + wave_id * ps.split_cap;
    int *lds_count = &lds_counts[wave_in_block];
    const int64_t total = (int64_t)fb.n_faces * sc.n;
    const int CHUNK = 8;
    for (;;) {
        unsigned long long start = 0;
        if (lane == 0) start = atomicAdd(work_counter, (unsigned long long)CHUNK);
        start = __shfl(start, 0);
        if ((int64_t)start >= total) break;
        for (int w = 0; w < CHUNK; ++w) {
            const int64_t item = (int64_t)start + w;
            if (item >= total) break;
            const int f = (int)(item / sc.n), B = (int)(item % sc.n);
            const PolyMeta *meta_f = meta + (size_t)f * sc.n;
            const double *sph_f = sph + (size_t)f * sc.vtot * 2;
            const PolyMeta mb = meta_f[B];
            if (strict ? mb.state == 0 : mb.state != 2) continue;  // wave-uniform
            bool visible = true;
            if (!strict) {
                // ---- polygons painted before B whose box meets B's box
                int n_list = 0;
                for (int base = 0; base < sc.n; base += 64) {
                    const int C = base + lane;
                    bool take = false;
                    if (C < sc.n && C != B) {
                        const PolyMeta &mc = meta_f[C];
                        take = mc.state == 2 && painted_before(mc, C, mb, B) &&
                               !(mc.xmax < mb.xmin || mc.xmin > mb.xmax || mc.ymax < mb.ymin || mc.ymin > mb.ymax);
                    }
                    const unsigned long long m = __ballot(take);
                    if (take) list[n_list + __popcll(m & ((1ull << lane) - 1))] = C;
                    n_list += __popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                visible = n_list == 0;
                if (!visible) {
                    // ---- B's own edges, witness on B's inner side
                    const int v0 = sc.vert_off[B], nv = sc.vert_off[B + 1] - v0;
                    const double *x = sph_f + (size_t)v0 * 2, *y = x + nv;
                    for (int v = 0; v < nv && !visible; ++v) {
                        const int u = v == 0 ? nv - 1 : v - 1;
                        visible = edge_has_witness(sc, sph_f, meta_f, B, list, n_list, B, x[u], y[u], x[v], y[v],
                                                   (double)mb.sgn, false, splits, ps.split_cap, lds_count);
                    }
                    // ---- edges of the painted polygons that reach into B's box, witness on their outer side
                    for (int i = 0; i < n_list && !visible; ++i) {
                        const int C = list[i];
                        const PolyMeta &mc = meta_f[C];
                        const int c0 = sc.vert_off[C], cn = sc.vert_off[C + 1] - c0;
                        const double *cx = sph_f + (size_t)c0 * 2, *cy = cx + cn;
                        for (int v = 0; v < cn && !visible; ++v) {
                            const int u = v == 0 ? cn - 1 : v - 1;
                            const double ax = cx[u], ay = cy[u], bx = cx[v], by = cy[v];
                            if (fmax(ax, bx) < mb.xmin || fmin(ax, bx) > mb.xmax || fmax(ay, by) < mb.ymin ||
                                fmin(ay, by) > mb.ymax)
                                continue;
                            visible = edge_has_witness(sc, sph_f, meta_f, B, list, n_list, C, ax, ay, bx, by,
                                                       -(double)mc.sgn, true, splits, ps.split_cap, lds_count);
                        }
                    }
                }
            }
            if (visible && lane == 0) {
                // ---- B must face A (geometry.py:1004-1014)
                const double *bb = sc.bbox + 6 * (size_t)B;
                const d3 cb = mk3((bb[0] + bb[3]) / 2, (bb[1] + bb[4]) / 2, (bb[2] + bb[5]) / 2);
                const d3 nb = ld3(sc.planes + 4 * (size_t)B);
                const int a0 = fb.a_off[f], an = fb.a_off[f + 1] - a0;
                bool facing = false;
                for (int v = 0; v < an && !facing; ++v) {
                    d3 dvec = ld3(fb.a_verts + 3 * (size_t)(a0 + v)) - cb;
                    dvec = divs(dvec, norm3(dvec));
                    facing = dot(dvec, nb) > TOL_DOT;
                }
                if (facing) out[(size_t)f * sc.n + B] = 1;
            }
        }
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

// visibility rows of a batch of faces into d_out[n_faces*n] (zeroed here)
static int visibility_rows(rt_scene *scene, const std::vector<int32_t> &a_off, const std::vector<double> &a_verts,
                           const std::vector<double> &a_normal, const std::vector<int32_t> &a_self, int strict,
                           uint8_t *d_out) {
    const SceneView &sc = scene->view;
    const int n_faces = (int)a_self.size();
    if (n_faces == 0) return 0;
    hipStream_t st = scene->stream;
    DevBuf b_off, b_verts, b_normal, b_self, b_sph, b_meta, b_list, b_splits, b_counter;
    const int n_blocks = 512;  // persistent workgroups of 4 waves
    const int64_t split_cap = 3 * (int64_t)sc.vtot + 8;
    if (b_off.alloc(sizeof(int32_t) * a_off.size()) || b_verts.alloc(sizeof(double) * a_verts.size()) ||
        b_normal.alloc(sizeof(double) * a_normal.size()) || b_self.alloc(sizeof(int32_t) * a_self.size()) ||
        b_sph.alloc(sizeof(double) * 2 * (size_t)sc.vtot * n_faces) || b_meta.alloc(sizeof(PolyMeta) * (size_t)sc.n * n_faces) ||
        b_list.alloc(sizeof(int32_t) * (size_t)sc.n * n_blocks * 4) ||
        b_splits.alloc(sizeof(double) * (size_t)split_cap * n_blocks * 4) || b_counter.alloc(sizeof(unsigned long long)))
        RT_FAIL("visibility: out of device memory");
    RT_HIP(hipMemcpyAsync(b_off.p, a_off.data(), sizeof(int32_t) * a_off.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(b_verts.p, a_verts.data(), sizeof(double) * a_verts.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(b_normal.p, a_normal.data(), sizeof(double) * a_normal.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(b_self.p, a_self.data(), sizeof(int32_t) * a_self.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemsetAsync(b_counter.p, 0, sizeof(unsigned long long), st));
    RT_HIP(hipMemsetAsync(d_out, 0, (size_t)sc.n * n_faces, st));
    FaceBatch fb{n_faces, (const int32_t *)b_off.p, (const double *)b_verts.p, (const double *)b_normal.p,
                 (const int32_t *)b_self.p};
    hipLaunchKernelGGL(k_vis_prepare, dim3((sc.n + 255) / 256, n_faces), dim3(256), 0, st, sc, fb, (double *)b_sph.p,
                       (PolyMeta *)b_meta.p);
    PaintScratch ps{(int32_t *)b_list.p, (double *)b_splits.p, split_cap};
    hipLaunchKernelGGL(k_vis_paint, dim3(n_blocks), dim3(256), 0, st, sc, fb, (const double *)b_sph.p,
                       (const PolyMeta *)b_meta.p, ps, strict, (unsigned long long *)b_counter.p, d_out);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int rt_face_visibility(rt_scene *scene, const double *face_verts, int32_t n_face_verts,
                                  const double *face_normal, int32_t self_index, int strict, uint8_t *out) {
    if (!scene || !face_verts || !face_normal || !out) RT_FAIL("rt_face_visibility: null argument");
    if (n_face_verts < 3) RT_FAIL("rt_face_visibility: a face needs at least 3 vertices");
    if (self_index < -1 || self_index >= scene->view.n) RT_FAIL("rt_face_visibility: self_index out of range");
    RT_HIP(hipSetDevice(scene->device));
    std::vector<int32_t> a_off{0, n_face_verts}, a_self{self_index};
    std::vector<double> a_verts(face_verts, face_verts + 3 * (size_t)n_face_verts);
    std::vector<double> a_normal(face_normal, face_normal + 3);
    DevBuf d_out;
    if (d_out.alloc(scene->view.n)) RT_FAIL("rt_face_visibility: out of device memory");
    if (visibility_rows(scene, a_off, a_verts, a_normal, a_self, strict, (uint8_t *)d_out.p)) return -1;
    RT_HIP(hipMemcpy(out, d_out.p, scene->view.n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rt_visibility_matrix(rt_scene *scene, int strict, uint8_t *out, int on_device) {
    if (!scene || !out) RT_FAIL("rt_visibility_matrix: null argument");
    RT_HIP(hipSetDevice(scene->device));
    const SceneView &sc = scene->view;
    const int n = sc.n;
    DevBuf d_rows, d_sym;
    if (d_rows.alloc((size_t)n * n)) RT_FAIL("rt_visibility_matrix: out of device memory");
    // batches of faces sized so the per-face scratch (mapped vertices + per-polygon records) stays under ~1 GiB
    const size_t per_face = sizeof(double) * 2 * (size_t)sc.vtot + sizeof(PolyMeta) * (size_t)n;
    int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)1 << 30) / per_face));
    batch = std::min(batch, 65535);
    for (int f0 = 0; f0 < n; f0 += batch) {
        const int f1 = std::min(n, f0 + batch);
        std::vector<int32_t> a_off, a_self;
        std::vector<double> a_verts, a_normal;
        a_off.push_back(0);
        for (int f = f0; f < f1; ++f) {
            const int v0 = scene->h_vert_off[f], v1 = scene->h_vert_off[f + 1];
            a_verts.insert(a_verts.end(), scene->h_verts.begin() + 3 * (size_t)v0, scene->h_verts.begin() + 3 * (size_t)v1);
            a_off.push_back((int32_t)(a_verts.size() / 3));
            a_normal.insert(a_normal.end(), scene->h_planes.begin() + 4 * (size_t)f, scene->h_planes.begin() + 4 * (size_t)f + 3);
            a_self.push_back(f);
        }
        if (visibility_rows(scene, a_off, a_verts, a_normal, a_self, strict, (uint8_t *)d_rows.p + (size_t)f0 * n)) return -1;
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
