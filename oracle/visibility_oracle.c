/*
 * CPU ORACLE — test infrastructure only.  Never linked into or called by the product path.
 *
 * Restates polygon_visibility_vector of the reference (raytracing/geometry.py:924-1019, non-strict branch)
 * in plain C.  The reference asks Shapely/GEOS three questions about the polygons after they have been mapped to
 * (azimuth, polar angle) as seen from face A:
 *     is_valid                              (geometry.py:992)
 *     B.difference(canvas).is_empty         (geometry.py:998-999)
 *     canvas = canvas.union(B)              (geometry.py:1000)
 * GEOS is not available, so they are answered here from first principles:
 *     valid   = finite coordinates, at least 3 distinct vertices, non-zero area, no two edges of the ring meet
 *               except consecutive ones at their shared vertex;
 *     covered = the part of B outside the union of the canvas polygons has no area, decided EXACTLY by a slab
 *               decomposition: between two consecutive abscissae taken from {vertices, pairwise edge
 *               intersections} nothing changes topologically, so one vertical line per slab decides it.
 *               A slab narrower than EPS_X or an uncovered stretch shorter than EPS_Y is ignored.  GEOS has no
 *               such thresholds (any difference with area counts), so they are kept as small as the coordinates
 *               allow; they are also the two lengths the HIP painter works with (a piece of an edge shorter than
 *               1e-9 is not examined, a witness stands 2.5e-10 beside its edge), so that both call the same slivers
 *               nothing.  With EPS_Y at 1e-9 one pair of lln's 58 million was decided differently: two walls 640 m
 *               apart, 3.25e-10 rad of one showing through a gap 6.7e-8 rad wide.
 * No reference fixture pins these two predicates: parity unpinned.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TOL_DZ 1e-4   /* geometry.py:947 */
#define TOL_DOT 1e-4  /* geometry.py:948 */
#define EPS_X 1e-9
#define EPS_Y 2.5e-10

typedef struct {
    int index;      /* polygon index in the scene */
    int nv;
    double *az;     /* nv */
    double *pol;    /* nv */
    double key1, key2;
    double xmin, xmax, ymin, ymax;
    int valid;
} SphPoly;

static double orient(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}

static int on_segment_box(double ax, double ay, double bx, double by, double px, double py) {
    return fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by);
}

/* closed segments ab and cd share at least one point */
static int segments_meet(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
    double o1 = orient(ax, ay, bx, by, cx, cy);
    double o2 = orient(ax, ay, bx, by, dx, dy);
    double o3 = orient(cx, cy, dx, dy, ax, ay);
    double o4 = orient(cx, cy, dx, dy, bx, by);
    if (((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0))) return 1;
    if (o1 == 0 && on_segment_box(ax, ay, bx, by, cx, cy)) return 1;
    if (o2 == 0 && on_segment_box(ax, ay, bx, by, dx, dy)) return 1;
    if (o3 == 0 && on_segment_box(cx, cy, dx, dy, ax, ay)) return 1;
    if (o4 == 0 && on_segment_box(cx, cy, dx, dy, bx, by)) return 1;
    return 0;
}

static int ring_is_valid(const double *x, const double *y, int n_in) {
    /* drop consecutive duplicates (GEOS accepts repeated points) */
    double *px = (double *)malloc(sizeof(double) * (size_t)n_in * 2);
    double *py = px + n_in;
    int n = 0;
    for (int i = 0; i < n_in; ++i) {
        if (!isfinite(x[i]) || !isfinite(y[i])) { free(px); return 0; }
        if (n > 0 && px[n - 1] == x[i] && py[n - 1] == y[i]) continue;
        px[n] = x[i]; py[n] = y[i]; ++n;
    }
    if (n > 1 && px[0] == px[n - 1] && py[0] == py[n - 1]) --n;
    if (n < 3) { free(px); return 0; }
    double area2 = 0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1) % n;
        area2 += px[i] * py[j] - px[j] * py[i];
    }
    if (area2 == 0) { free(px); return 0; }
    int ok = 1;
    for (int i = 0; i < n && ok; ++i) {
        int i2 = (i + 1) % n;
        for (int j = i + 1; j < n && ok; ++j) {
            int j2 = (j + 1) % n;
            if (j == i2 || j2 == i) {
                /* consecutive edges: only a fold-back onto each other is an intersection */
                int a, b, c;
                if (j == i2) { a = i; b = i2; c = j2; } else { a = j; b = j2; c = i2; }
                double cr = orient(px[a], py[a], px[b], py[b], px[c], py[c]);
                double dt = (px[b] - px[a]) * (px[c] - px[b]) + (py[b] - py[a]) * (py[c] - py[b]);
                if (cr == 0 && dt < 0) ok = 0;
                if (n == 3) continue;
            } else if (segments_meet(px[i], py[i], px[i2], py[i2], px[j], py[j], px[j2], py[j2])) {
                ok = 0;
            }
        }
    }
    free(px);
    return ok;
}

/* ---- slab coverage ---------------------------------------------------------------------------- */

typedef struct { double ax, ay, bx, by; int owner; } Edge;  /* owner: -1 = B, else canvas slot */

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

typedef struct { double lo, hi; } Interval;
static int cmp_interval(const void *a, const void *b) {
    double x = ((const Interval *)a)->lo, y = ((const Interval *)b)->lo;
    return (x > y) - (x < y);
}

/* ys where the vertical line x = xm crosses the ring of one polygon, sorted; returns count */
static int ring_crossings(const double *x, const double *y, int n, double xm, double *out) {
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1) % n;
        if ((x[i] > xm) != (x[j] > xm)) out[k++] = y[i] + (y[j] - y[i]) * (xm - x[i]) / (x[j] - x[i]);
    }
    qsort(out, (size_t)k, sizeof(double), cmp_double);
    return k;
}

/* 1 when some stretch of B on the line x = xm, longer than EPS_Y, is outside every canvas polygon */
static int uncovered_on_line(const SphPoly *B, SphPoly *const *canvas, int nc, double xm, double *scratch,
                             Interval *ivals) {
    double *yb = scratch;
    int kb = ring_crossings(B->az, B->pol, B->nv, xm, yb);
    if (kb < 2) return 0;
    double *yc = scratch + B->nv + 2;
    int ni = 0;
    for (int c = 0; c < nc; ++c) {
        const SphPoly *C = canvas[c];
        if (C->xmin > xm || C->xmax < xm) continue;
        int kc = ring_crossings(C->az, C->pol, C->nv, xm, yc);
        for (int q = 0; q + 1 < kc; q += 2) { ivals[ni].lo = yc[q]; ivals[ni].hi = yc[q + 1]; ++ni; }
    }
    qsort(ivals, (size_t)ni, sizeof(Interval), cmp_interval);
    for (int q = 0; q + 1 < kb; q += 2) {
        double cur = yb[q], end = yb[q + 1];
        for (int i = 0; i < ni && cur < end; ++i) {
            if (ivals[i].hi <= cur) continue;
            if (ivals[i].lo > cur) {
                if (fmin(ivals[i].lo, end) - cur > EPS_Y) return 1;
            }
            if (ivals[i].hi > cur) cur = ivals[i].hi;
        }
        if (end - cur > EPS_Y) return 1;
    }
    return 0;
}

static int covered_by_canvas(const SphPoly *B, SphPoly **canvas_all, int n_all) {
    /* canvas polygons whose box meets B's box */
    SphPoly **canvas = (SphPoly **)malloc(sizeof(SphPoly *) * (size_t)(n_all + 1));
    int nc = 0, max_nv = B->nv, tot_edges = B->nv;
    for (int c = 0; c < n_all; ++c) {
        SphPoly *C = canvas_all[c];
        if (C->xmax < B->xmin || C->xmin > B->xmax || C->ymax < B->ymin || C->ymin > B->ymax) continue;
        canvas[nc++] = C;
        if (C->nv > max_nv) max_nv = C->nv;
        tot_edges += C->nv;
    }
    if (nc == 0) { free(canvas); return 0; }
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(B->nv + max_nv + 8));
    Interval *ivals = (Interval *)malloc(sizeof(Interval) * (size_t)(tot_edges + 4));
    int result = 1;

    /* quick pass: any line showing an uncovered stretch proves visibility */
    double *xs = (double *)malloc(sizeof(double) * (size_t)(B->nv + 1));
    memcpy(xs, B->az, sizeof(double) * (size_t)B->nv);
    qsort(xs, (size_t)B->nv, sizeof(double), cmp_double);
    for (int i = 0; i + 1 < B->nv && result; ++i) {
        if (xs[i + 1] - xs[i] <= EPS_X) continue;
        if (uncovered_on_line(B, canvas, nc, 0.5 * (xs[i] + xs[i + 1]), scratch, ivals)) result = 0;
    }
    free(xs);

    if (result) {
        /* exact pass: every abscissa where the arrangement can change inside B's box */
        Edge *edges = (Edge *)malloc(sizeof(Edge) * (size_t)tot_edges);
        int ne = 0;
        for (int i = 0; i < B->nv; ++i) {
            int j = (i + 1) % B->nv;
            edges[ne].ax = B->az[i]; edges[ne].ay = B->pol[i]; edges[ne].bx = B->az[j]; edges[ne].by = B->pol[j];
            edges[ne].owner = -1; ++ne;
        }
        for (int c = 0; c < nc; ++c) {
            const SphPoly *C = canvas[c];
            for (int i = 0; i < C->nv; ++i) {
                int j = (i + 1) % C->nv;
                double ax = C->az[i], ay = C->pol[i], bx = C->az[j], by = C->pol[j];
                if (fmax(ax, bx) < B->xmin || fmin(ax, bx) > B->xmax || fmax(ay, by) < B->ymin || fmin(ay, by) > B->ymax)
                    continue;
                edges[ne].ax = ax; edges[ne].ay = ay; edges[ne].bx = bx; edges[ne].by = by; edges[ne].owner = c; ++ne;
            }
        }
        size_t cap = (size_t)ne * 2 + 16, nx = 0;
        double *ev = (double *)malloc(sizeof(double) * cap);
        for (int e = 0; e < ne; ++e) { ev[nx++] = edges[e].ax; ev[nx++] = edges[e].bx; }
        for (int e = 0; e < ne; ++e) {
            double ex = edges[e].bx - edges[e].ax, ey = edges[e].by - edges[e].ay;
            for (int f = e + 1; f < ne; ++f) {
                if (edges[e].owner == edges[f].owner) continue;  /* rings are simple */
                double fx = edges[f].bx - edges[f].ax, fy = edges[f].by - edges[f].ay;
                double den = ex * fy - ey * fx;
                if (den == 0) continue;  /* parallel: their endpoints are events already */
                double wx = edges[f].ax - edges[e].ax, wy = edges[f].ay - edges[e].ay;
                double s = (wx * fy - wy * fx) / den;
                double t = (wx * ey - wy * ex) / den;
                if (s < 0 || s > 1 || t < 0 || t > 1) continue;
                double x = edges[e].ax + s * ex;
                if (x <= B->xmin || x >= B->xmax) continue;
                if (nx == cap) { cap *= 2; ev = (double *)realloc(ev, sizeof(double) * cap); }
                ev[nx++] = x;
            }
        }
        qsort(ev, nx, sizeof(double), cmp_double);
        for (size_t i = 0; i + 1 < nx && result; ++i) {
            double x0 = ev[i], x1 = ev[i + 1];
            if (x1 <= B->xmin) continue;
            if (x0 >= B->xmax) break;
            if (x1 - x0 <= EPS_X) continue;
            if (uncovered_on_line(B, canvas, nc, 0.5 * (x0 + x1), scratch, ivals)) result = 0;
        }
        free(ev);
        free(edges);
    }
    free(ivals);
    free(scratch);
    free(canvas);
    return result;
}

static int cmp_sph(const void *a, const void *b) {
    const SphPoly *p = *(SphPoly *const *)a, *q = *(SphPoly *const *)b;
    if (p->key1 != q->key1) return (p->key1 > q->key1) - (p->key1 < q->key1);
    if (p->key2 != q->key2) return (p->key2 > q->key2) - (p->key2 < q->key2);
    return (p->index > q->index) - (p->index < q->index);  /* Python's sort is stable */
}

/*
 * a_pts: vertices of face A (na x 3); a_normal: the normal A reports (a top/bottom face may report the opposite of
 * what its vertex order gives, geometry.py:2360-2363).  Scene in CSR form.  self_index: position of A in the scene or
 * -1.  out: N bytes, only ever set to 1 (geometry.py:1016-1017).  stage (may be NULL): N bytes, 0 = self or failed
 * the z filter, 1 = invalid ring, 2 = hidden by the canvas, 3 = back-facing, 4 = visible.
 */
int oracle_face_visibility(const double *a_pts, int na, const double *a_normal, int n, const int *vert_off,
                           const double *verts, const double *planes, int self_index, unsigned char *out,
                           unsigned char *stage) {
    /* frame of A: geometry.py:1839-1851, origin at the bounding-box centre: geometry.py:960, 1462-1463 */
    double ax[3], ay[3], az_[3], c[3];
    for (int k = 0; k < 3; ++k) {
        double lo = a_pts[k], hi = a_pts[k];
        for (int v = 1; v < na; ++v) { lo = fmin(lo, a_pts[3 * v + k]); hi = fmax(hi, a_pts[3 * v + k]); }
        c[k] = (lo + hi) / 2;
        ax[k] = a_pts[3 + k] - a_pts[k];
        az_[k] = a_normal[k];
    }
    ay[0] = az_[1] * ax[2] - az_[2] * ax[1];
    ay[1] = az_[2] * ax[0] - az_[0] * ax[2];
    ay[2] = az_[0] * ax[1] - az_[1] * ax[0];
    double nx_ = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double ny_ = sqrt(ay[0] * ay[0] + ay[1] * ay[1] + ay[2] * ay[2]);
    for (int k = 0; k < 3; ++k) { ax[k] /= nx_; ay[k] /= ny_; }

    SphPoly *polys = (SphPoly *)calloc((size_t)n, sizeof(SphPoly));
    SphPoly **order = (SphPoly **)malloc(sizeof(SphPoly *) * (size_t)n);
    int m = 0;
    if (stage) memset(stage, 0, (size_t)n);
    for (int b = 0; b < n; ++b) {
        if (b == self_index) continue;  /* geometry.py:964 */
        int v0 = vert_off[b], nv = vert_off[b + 1] - v0;
        SphPoly *P = &polys[b];
        P->index = b; P->nv = nv;
        P->az = (double *)malloc(sizeof(double) * (size_t)nv * 2);
        P->pol = P->az + nv;
        double zmax = -INFINITY, rmin = INFINITY;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int v = 0; v < nv; ++v) {
            const double *p = verts + 3 * (size_t)(v0 + v);
            double d0 = p[0] - c[0], d1 = p[1] - c[1], d2 = p[2] - c[2];
            double qx = d0 * ax[0] + d1 * ax[1] + d2 * ax[2];
            double qy = d0 * ay[0] + d1 * ay[1] + d2 * ay[2];
            double qz = d0 * az_[0] + d1 * az_[1] + d2 * az_[2];
            if (qz > zmax) zmax = qz;
            /* geometry.py:302-306 */
            double r = sqrt(qx * qx + qy * qy + qz * qz);
            double azv = atan2(qy, qx);
            double polv = acos(qz / r);
            P->az[v] = azv; P->pol[v] = polv;
            if (r < rmin) rmin = r;
            double s[3] = {azv, polv, r};
            for (int k = 0; k < 3; ++k) { if (s[k] < lo[k]) lo[k] = s[k]; if (s[k] > hi[k]) hi[k] = s[k]; }
        }
        if (!(zmax > TOL_DZ)) { free(P->az); P->az = NULL; continue; }  /* geometry.py:967-970 */
        double cx = (lo[0] + hi[0]) / 2, cy = (lo[1] + hi[1]) / 2, cz = (lo[2] + hi[2]) / 2;
        P->key1 = rmin;                                 /* geometry.py:981-982 */
        P->key2 = sqrt(cx * cx + cy * cy + cz * cz);
        P->xmin = lo[0]; P->xmax = hi[0]; P->ymin = lo[1]; P->ymax = hi[1];
        P->valid = ring_is_valid(P->az, P->pol, nv);
        if (stage) stage[b] = 1;
        order[m++] = P;
    }
    qsort(order, (size_t)m, sizeof(SphPoly *), cmp_sph);

    /* painter: geometry.py:986-1001 */
    SphPoly **canvas = (SphPoly **)malloc(sizeof(SphPoly *) * (size_t)(m + 1));
    int nc = 0;
    for (int k = 0; k < m; ++k) {
        SphPoly *B = order[k];
        if (!B->valid) continue;
        if (stage) stage[B->index] = 2;
        if (covered_by_canvas(B, canvas, nc)) continue;
        canvas[nc++] = B;
        if (stage) stage[B->index] = 3;
        /* normal filter: geometry.py:1004-1014 */
        int v0 = vert_off[B->index], nv = B->nv;
        double lo[3], hi[3], cb[3];
        for (int q = 0; q < 3; ++q) { lo[q] = INFINITY; hi[q] = -INFINITY; }
        for (int v = 0; v < nv; ++v)
            for (int q = 0; q < 3; ++q) {
                double x = verts[3 * (size_t)(v0 + v) + q];
                if (x < lo[q]) lo[q] = x;
                if (x > hi[q]) hi[q] = x;
            }
        for (int q = 0; q < 3; ++q) cb[q] = (lo[q] + hi[q]) / 2;
        const double *nb = planes + 4 * (size_t)B->index;
        int facing = 0;
        for (int v = 0; v < na && !facing; ++v) {
            double d0 = a_pts[3 * v] - cb[0], d1 = a_pts[3 * v + 1] - cb[1], d2 = a_pts[3 * v + 2] - cb[2];
            double len = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            double dot = (d0 / len) * nb[0] + (d1 / len) * nb[1] + (d2 / len) * nb[2];
            if (dot > TOL_DOT) facing = 1;
        }
        if (facing) { out[B->index] = 1; if (stage) stage[B->index] = 4; }
    }
    for (int b = 0; b < n; ++b) if (polys[b].az) free(polys[b].az);
    free(canvas);
    free(order);
    free(polys);
    return 0;
}
