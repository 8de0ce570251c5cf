// Scene handle: uploads the polygon list once and derives frames, bounding boxes and in-plane coordinates.
#include <algorithm>
#include <mutex>
#include <utility>
#include <cmath>
#include <cstring>

#include "rt_common.h"

static thread_local std::string g_last_error;

void rt_set_error(const std::string &msg) { g_last_error = msg; }

extern "C" const char *rt_last_error(void) { return g_last_error.c_str(); }

extern "C" int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The scene's arrays live in ONE device allocation: a kernel of a small solve step touches most of them, and every
// separate allocation is another page the cold TLBs of 256 CUs have to walk for.  upload() only notes what has to go
// where; commit_uploads() allocates the arena, copies everything and sets the device pointers.
struct PendingUpload {
    const void *host;
    size_t bytes;
    const void **device_field;
    size_t offset;
};
static thread_local std::vector<PendingUpload> g_pending;
static thread_local std::vector<std::vector<unsigned char>> g_pending_copies;

template <typename T>
static int upload(rt_scene *, const std::vector<T> &h, const T **dptr) {
    // the caller's vector may be a local: keep a copy until the commit
    g_pending_copies.emplace_back((const unsigned char *)h.data(), (const unsigned char *)h.data() + sizeof(T) * h.size());
    g_pending.push_back({nullptr, sizeof(T) * h.size(), (const void **)dptr, 0});
    return 0;
}

static int commit_uploads(rt_scene *s) {
    size_t total = 0;
    for (PendingUpload &u : g_pending) {
        u.offset = total;
        total += (u.bytes + 255) & ~(size_t)255;
    }
    total += 256;  // the solve's counters, below
    void *arena = nullptr;
    hipError_t e = hipMalloc(&arena, total);
    if (e != hipSuccess) {
        g_pending.clear();
        g_pending_copies.clear();
        rt_set_error(std::string("rt_scene_create: hipMalloc: ") + hipGetErrorString(e));
        return -1;
    }
    s->allocations.push_back(arena);
    std::vector<unsigned char> staging(total, 0);
    for (size_t i = 0; i < g_pending.size(); ++i) {
        if (g_pending[i].bytes) memcpy(staging.data() + g_pending[i].offset, g_pending_copies[i].data(), g_pending[i].bytes);
        *g_pending[i].device_field = (const char *)arena + g_pending[i].offset;
    }
    s->d_counters = (unsigned long long *)((char *)arena + total - 256);
    g_pending.clear();
    g_pending_copies.clear();
    RT_HIP(hipMemcpy(arena, staging.data(), total, hipMemcpyHostToDevice));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// obstruction grid
// ---------------------------------------------------------------------------------------------------------------
static float round_down(double v) {
    float f = (float)v;
    return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
static float round_up(double v) {
    float f = (float)v;
    return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}

// does the 2-D segment P->Q come within the box [x0,x1]x[y0,y1]?  (slab clipping; errs on the side of yes)
static bool segment_meets_box(double px, double py, double qx, double qy, double x0, double y0, double x1, double y1) {
    double t0 = 0.0, t1 = 1.0;
    const double d[2] = {qx - px, qy - py}, p[2] = {px, py}, lo[2] = {x0, y0}, hi[2] = {x1, y1};
    for (int a = 0; a < 2; ++a) {
        if (std::fabs(d[a]) < 1e-300) {
            if (p[a] < lo[a] || p[a] > hi[a]) return false;
        } else {
            double ta = (lo[a] - p[a]) / d[a], tb = (hi[a] - p[a]) / d[a];
            if (ta > tb) std::swap(ta, tb);
            t0 = std::max(t0, ta);
            t1 = std::min(t1, tb);
        }
    }
    return t0 <= t1 + 1e-12;
}

static int build_grid(rt_scene *s, int n_poly, const int32_t *vert_off, const double *verts, const double *planes) {
    const double *bb = s->h_bbox.data();
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n_poly; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], bb[6 * (size_t)i + k]);
            hi[k] = std::fmax(hi[k], bb[6 * (size_t)i + 3 + k]);
        }
    const double pad = RT_GRID_PAD;
    const double w = std::fmax(hi[0] - lo[0], 1e-3), h = std::fmax(hi[1] - lo[1], 1e-3);
    // about one cell per polygon: a handful of records per occupied cell
    double side = std::sqrt(w * h / std::fmax(16.0, (double)n_poly));
    side = std::fmax(side, std::fmax(w, h) / 1024.0);
    const int gnx = std::max(1, (int)std::ceil(w / side)), gny = std::max(1, (int)std::ceil(h / side));
    const int n_cells = gnx * gny;
    const size_t big_limit = std::max<size_t>(64, (size_t)n_cells / 8);
    std::vector<std::vector<int32_t>> cells((size_t)n_cells + 1);
    std::vector<int32_t> mine;
    for (int i = 0; i < n_poly; ++i) {
        const double *b = bb + 6 * (size_t)i;
        const int c0 = std::min(gnx - 1, std::max(0, (int)std::floor((b[0] - pad - lo[0]) / side)));
        const int c1 = std::min(gnx - 1, std::max(0, (int)std::floor((b[3] + pad - lo[0]) / side)));
        const int r0 = std::min(gny - 1, std::max(0, (int)std::floor((b[1] - pad - lo[1]) / side)));
        const int r1 = std::min(gny - 1, std::max(0, (int)std::floor((b[4] + pad - lo[1]) / side)));
        const int v0 = vert_off[i], nv = vert_off[i + 1] - v0;
        const double *pv = verts + 3 * (size_t)v0;
        mine.clear();
        for (int r = r0; r <= r1; ++r)
            for (int c = c0; c <= c1; ++c) {
                const double x0 = lo[0] + c * side - 2 * pad, x1 = lo[0] + (c + 1) * side + 2 * pad;
                const double y0 = lo[1] + r * side - 2 * pad, y1 = lo[1] + (r + 1) * side + 2 * pad;
                // footprint test: a side of the polygon crosses the cell, or the cell's centre is inside the footprint
                bool touches = false;
                for (int v = 0; v < nv && !touches; ++v) {
                    const double *P = pv + 3 * v, *Q = pv + 3 * ((v + 1) % nv);
                    touches = segment_meets_box(P[0], P[1], Q[0], Q[1], x0, y0, x1, y1);
                }
                if (!touches)
                    touches = pip_inclusive(
                        nv, [&](int k, double &x, double &y) { x = pv[3 * k]; y = pv[3 * k + 1]; }, 0.5 * (x0 + x1),
                        0.5 * (y0 + y1));
                if (touches) mine.push_back(r * gnx + c);
            }
        if (mine.size() > big_limit) {
            cells[n_cells].push_back(i);
        } else {
            for (int32_t c : mine) cells[c].push_back(i);
        }
    }
    std::vector<OccRec> poly_recs((size_t)n_poly);
    for (int i = 0; i < n_poly; ++i) {
        OccRec &r = poly_recs[i];
        const double *pl = planes + 4 * (size_t)i, *b = bb + 6 * (size_t)i;
        r.nx = pl[0];
        r.ny = pl[1];
        r.nz = pl[2];
        r.d = pl[3];
        for (int a = 0; a < 3; ++a) {
            r.lo[a] = round_down(b[a] - lo[a]);
            r.hi[a] = round_up(b[3 + a] - lo[a]);
        }
        r.poly = i;
        r.pad_ = 0;
    }
    std::vector<OccCell> cell_table((size_t)n_cells + 1);
    std::vector<OccRec> recs;
    for (int c = 0; c <= n_cells; ++c) {
        OccCell &cell = cell_table[c];
        cell.begin = (int32_t)recs.size();
        cell.zmax = -INFINITY;
        cell.pad_ = 0;
        for (int32_t i : cells[c]) {
            cell.zmax = std::fmax(cell.zmax, poly_recs[i].hi[2]);
            recs.push_back(poly_recs[i]);
        }
        cell.end = (int32_t)recs.size();
    }
    s->view.gx0 = lo[0];
    s->view.gy0 = lo[1];
    s->view.gz0 = lo[2];
    s->view.gside = side;
    s->view.ginv = 1.0 / side;
    s->view.gnx = gnx;
    s->view.gny = gny;
    if (upload(s, cell_table, &s->view.cells)) return -1;
    if (upload(s, recs, &s->view.cell_recs)) return -1;
    if (upload(s, poly_recs, &s->view.poly_recs)) return -1;
    return 0;
}

extern "C" int rt_debug_set_path_queue_capacity(rt_scene *s, int64_t n_paths) {
    if (!s || n_paths < 0) RT_FAIL("rt_debug_set_path_queue_capacity: bad argument");
    s->queue_capacity_override = n_paths;
    return 0;
}

extern "C" int rt_debug_set_occlusion_mode(rt_scene *s, int mode) {
    if (!s || mode < 0 || mode > 2) RT_FAIL("rt_debug_set_occlusion_mode: bad argument");
    s->occlusion_mode = mode;
    return 0;
}

extern "C" int rt_debug_set_reflect_mode(rt_scene *s, int mode) {
    if (!s || mode < 0 || mode > 2) RT_FAIL("rt_debug_set_reflect_mode: bad argument");
    s->reflect_mode = mode;
    return 0;
}

extern "C" int rt_debug_set_wall_hints(rt_scene *s, int on) {
    if (!s || on < 0 || on > 1) RT_FAIL("rt_debug_set_wall_hints: bad argument");
    s->wall_hints = on;
    return 0;
}

extern "C" int rt_debug_set_stored_chain_limit(rt_scene *s, int64_t bytes) {
    if (!s || bytes < 0) RT_FAIL("rt_debug_set_stored_chain_limit: bad argument");
    s->stored_chain_bytes_limit = bytes;
    return 0;
}

extern "C" int rt_scene_set_timing(rt_scene *s, int on) {
    if (!s) RT_FAIL("rt_scene_set_timing: null scene");
    s->timing = on != 0;
    return 0;
}

extern "C" int rt_scene_wait_for_stream(rt_scene *s, void *stream) {
    if (!s) RT_FAIL("rt_scene_wait_for_stream: null scene");
    RT_HIP(rt_use_device(s->device));
    if (!s->caller_event) RT_HIP(hipEventCreateWithFlags(&s->caller_event, hipEventDisableTiming));
    RT_HIP(hipEventRecord(s->caller_event, static_cast<hipStream_t>(stream)));
    RT_HIP(hipStreamWaitEvent(s->stream, s->caller_event, 0));
    return 0;
}

extern "C" int rt_scene_set_deferred_sync(rt_scene *s, int on) {
    if (!s) RT_FAIL("rt_scene_set_deferred_sync: null scene");
    s->defer_sync = on != 0;
    return 0;
}

extern "C" int rt_scene_synchronize(rt_scene *s) {
    if (!s) RT_FAIL("rt_scene_synchronize: null scene");
    RT_HIP(rt_use_device(s->device));
    RT_HIP(hipStreamSynchronize(s->stream));
    if (s->fields_events_pending && s->event_pool.size() >= 4) {
        float t = 0;
        if (s->fields_events_with_kernels && hipEventElapsedTime(&t, s->event_pool[1], s->event_pool[2]) == hipSuccess)
            s->last_fields_ms[0] = t;
        if (hipEventElapsedTime(&t, s->event_pool[0], s->event_pool[3]) == hipSuccess) s->last_fields_ms[1] = t;
    }
    s->fields_events_pending = false;
    return 0;
}

static std::mutex g_stream_mutex;
static std::vector<std::pair<int, hipStream_t>> g_idle_streams;  // (device, stream)

hipError_t rt_take_stream(int device, hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (size_t i = 0; i < g_idle_streams.size(); ++i)
            if (g_idle_streams[i].first == device) {
                *out = g_idle_streams[i].second;
                g_idle_streams.erase(g_idle_streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void rt_give_stream(int device, hipStream_t stream) {
    if (!stream) return;
    (void)hipStreamSynchronize(stream);
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_idle_streams.emplace_back(device, stream);
}

extern "C" rt_scene *rt_scene_create(int device, int32_t n_poly, const int32_t *vert_off, const double *verts,
                                     const double *planes, const int32_t *material_id, const int32_t *building_id) {
    if (n_poly <= 0 || !vert_off || !verts || !planes) {
        rt_set_error("rt_scene_create: empty scene or null array");
        return nullptr;
    }
    int ndev = 0;
    RT_HIP_NULL(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        rt_set_error("rt_scene_create: no such device " + std::to_string(device));
        return nullptr;
    }
    if (vert_off[0] != 0) {
        rt_set_error("rt_scene_create: vertex offsets must start at 0");
        return nullptr;
    }
    for (int i = 0; material_id && i < n_poly; ++i) {
        if (material_id[i] < 0 || material_id[i] >= RT_N_MATERIALS) {  // the field kernel indexes its table with it
            rt_set_error("rt_scene_create: polygon " + std::to_string(i) + " has material id " +
                         std::to_string(material_id[i]) + ", outside the table of " + std::to_string(RT_N_MATERIALS));
            return nullptr;
        }
    }
    for (int i = 0; i < n_poly; ++i) {
        if (vert_off[i + 1] - vert_off[i] < 3) {
            rt_set_error("rt_scene_create: polygon " + std::to_string(i) + " has fewer than 3 vertices");
            return nullptr;
        }
    }
    RT_HIP_NULL(rt_use_device(device));
    g_pending.clear();
    g_pending_copies.clear();
    rt_scene *s = new rt_scene();
    s->device = device;
    const int vtot = vert_off[n_poly];
    s->h_vert_off.assign(vert_off, vert_off + n_poly + 1);
    s->h_verts.assign(verts, verts + 3 * (size_t)vtot);
    s->h_planes.assign(planes, planes + 4 * (size_t)n_poly);
    std::vector<double> frames(9 * (size_t)n_poly), uv(2 * (size_t)vtot), uv_recs(16 * (size_t)n_poly, 0.0);
    s->h_bbox.resize(6 * (size_t)n_poly);
    std::vector<int32_t> mat(n_poly, 1), bld(n_poly, 0);
    if (material_id) mat.assign(material_id, material_id + n_poly);
    if (building_id) bld.assign(building_id, building_id + n_poly);
    int vmax = 0;
    for (int i = 0; i < n_poly; ++i) {
        const int v0 = vert_off[i], nv = vert_off[i + 1] - v0;
        vmax = nv > vmax ? nv : vmax;
        const double *p = verts + 3 * (size_t)v0;
        // own frame (reference geometry.py:1839-1851): x along the first side, y = n x side, z = n
        d3 n = ld3(planes + 4 * (size_t)i);
        d3 fx = ld3(p + 3) - ld3(p);
        d3 fy = cross(n, fx);
        fx = divs(fx, norm3(fx));
        fy = divs(fy, norm3(fy));
        double *f = frames.data() + 9 * (size_t)i;
        st3(f, fx);
        st3(f + 3, fy);
        st3(f + 6, n);
        double *bb = s->h_bbox.data() + 6 * (size_t)i;
        for (int k = 0; k < 3; ++k) {
            bb[k] = INFINITY;
            bb[3 + k] = -INFINITY;
        }
        double *ur = uv_recs.data() + 16 * (size_t)i;
        st3(ur, fx);
        st3(ur + 3, fy);
        ur[6] = ur[7] = INFINITY;
        ur[8] = ur[9] = -INFINITY;
        for (int v = 0; v < nv; ++v) {
            d3 P = ld3(p + 3 * v);
            uv[2 * (size_t)(v0 + v)] = dot(P, fx);
            uv[2 * (size_t)(v0 + v) + 1] = dot(P, fy);
            ur[6] = std::fmin(ur[6], uv[2 * (size_t)(v0 + v)]);
            ur[7] = std::fmin(ur[7], uv[2 * (size_t)(v0 + v) + 1]);
            ur[8] = std::fmax(ur[8], uv[2 * (size_t)(v0 + v)]);
            ur[9] = std::fmax(ur[9], uv[2 * (size_t)(v0 + v) + 1]);
            for (int k = 0; k < 3; ++k) {
                bb[k] = std::fmin(bb[k], p[3 * v + k]);
                bb[3 + k] = std::fmax(bb[3 + k], p[3 * v + k]);
            }
        }
    }
    // polygons grouped by building id (any integers; without ids there are no groups)
    std::vector<int32_t> bld_off(1, 0), bld_polys, bld_of(n_poly, -1);
    if (building_id) {
        std::vector<int32_t> order(n_poly);
        for (int i = 0; i < n_poly; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return building_id[a] < building_id[b]; });
        for (int k = 0; k < n_poly; ++k) {
            if (k > 0 && building_id[order[k]] != building_id[order[k - 1]]) bld_off.push_back((int32_t)bld_polys.size());
            bld_of[order[k]] = (int32_t)bld_off.size() - 1;
            bld_polys.push_back(order[k]);
        }
        bld_off.push_back((int32_t)bld_polys.size());
    }
    if (bld_polys.empty()) bld_polys.push_back(0);
    s->view.n_bld = building_id ? (int32_t)bld_off.size() - 1 : 0;
    bool ok = upload(s, bld_off, &s->view.bld_off) == 0 && upload(s, bld_polys, &s->view.bld_polys) == 0 &&
              upload(s, bld_of, &s->view.bld_of) == 0 &&
              upload(s, s->h_vert_off, &s->view.vert_off) == 0 && upload(s, s->h_verts, &s->view.verts) == 0 &&
              upload(s, s->h_planes, &s->view.planes) == 0 && upload(s, frames, &s->view.frames) == 0 &&
              upload(s, s->h_bbox, &s->view.bbox) == 0 && upload(s, uv, &s->view.uv) == 0 &&
              upload(s, uv_recs, &s->view.uv_recs) == 0 &&
              upload(s, mat, &s->view.material) == 0 && upload(s, bld, &s->view.building) == 0 &&
              build_grid(s, n_poly, vert_off, verts, planes) == 0 && commit_uploads(s) == 0;
    if (ok && hipHostMalloc((void **)&s->h_counters, 16 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        rt_set_error("rt_scene_create: pinned host allocation failed");
        s->h_counters = nullptr;
        ok = false;
    }
    if (ok && rt_noalloc_setup(device) != 0) ok = false;
    // (both streams now: the second used to be made on its first use, in the middle of the first painting or solve)
    if (ok && (rt_take_stream(device, &s->stream) != hipSuccess || rt_take_stream(device, &s->side_stream) != hipSuccess)) {
        rt_set_error("rt_scene_create: hipStreamCreate failed");
        ok = false;
    }
    if (!ok) {
        rt_scene_destroy(s);
        return nullptr;
    }
    s->view.n = n_poly;
    s->view.vtot = vtot;
    s->view.vmax = vmax;
    return s;
}

extern "C" void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    (void)rt_use_device(s->device);
    if (s->caller_event) (void)hipEventDestroy(s->caller_event);
    if (s->side_event) (void)hipEventDestroy(s->side_event);
    if (s->rejudge_event) (void)hipEventDestroy(s->rejudge_event);
    rt_give_stream(s->device, s->side_stream);
    rt_give_stream(s->device, s->stream);
    for (hipEvent_t e : s->event_pool) (void)hipEventDestroy(e);
    for (void *p : s->allocations) (void)hipFree(p);
    if (s->ws) (void)hipFree(s->ws);
    if (s->late_items) (void)hipFree(s->late_items);
    for (const rt_scene::TempBlock &b : s->temp_pool) (void)hipFree(b.p);
    if (s->h_counters) (void)hipHostFree(s->h_counters);
    delete s;
}

extern "C" int32_t rt_scene_polygon_count(const rt_scene *s) { return s ? s->view.n : -1; }

void *rt_scene_temp(rt_scene *s, size_t bytes) {
    bytes = bytes ? bytes : 1;
    int best = -1;
    for (size_t i = 0; i < s->temp_pool.size(); ++i) {
        const rt_scene::TempBlock &b = s->temp_pool[i];
        if (b.busy || b.bytes < bytes || b.bytes > 4 * bytes + (1 << 20)) continue;
        if (best < 0 || b.bytes < s->temp_pool[best].bytes) best = (int)i;
    }
    if (best >= 0) {
        s->temp_pool[best].busy = true;
        s->temp_pool[best].idle = 0;
        return s->temp_pool[best].p;
    }
    void *d = nullptr;
    const size_t want = bytes + bytes / 8;  // a little room, so that a slightly larger next call still fits
    if (hipMalloc(&d, want) != hipSuccess) return nullptr;
    s->temp_pool.push_back({d, want, true, 0});
    return d;
}

void rt_scene_temp_release_all(rt_scene *s) {
    for (size_t i = 0; i < s->temp_pool.size();) {
        rt_scene::TempBlock &b = s->temp_pool[i];
        if (!b.busy && ++b.idle > 8) {  // sizes have changed: let the block go
            (void)hipFree(b.p);
            s->temp_pool.erase(s->temp_pool.begin() + i);
            continue;
        }
        b.busy = false;
        ++i;
    }
}

int rt_scene_workspace(rt_scene *s, size_t bytes, void **out) {
    if (bytes > s->ws_bytes) {
        if (s->ws) {
            RT_HIP(hipStreamSynchronize(s->stream));
            RT_HIP(hipFree(s->ws));
            s->ws = nullptr;
            s->ws_bytes = 0;
        }
        size_t want = bytes + bytes / 4 + (1 << 20);
        RT_HIP(hipMalloc(&s->ws, want));
        s->ws_bytes = want;
    }
    *out = s->ws;
    return 0;
}
