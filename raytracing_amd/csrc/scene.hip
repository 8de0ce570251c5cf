// Scene handle: uploads the polygon list once and derives frames, bounding boxes and in-plane coordinates.
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

template <typename T>
static int upload(rt_scene *s, const std::vector<T> &h, const T **dptr) {
    void *d = nullptr;
    size_t bytes = sizeof(T) * (h.empty() ? 1 : h.size());
    RT_HIP(hipMalloc(&d, bytes));
    s->allocations.push_back(d);
    if (!h.empty()) RT_HIP(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    *dptr = (const T *)d;
    return 0;
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
    for (int i = 0; i < n_poly; ++i) {
        if (vert_off[i + 1] - vert_off[i] < 3) {
            rt_set_error("rt_scene_create: polygon " + std::to_string(i) + " has fewer than 3 vertices");
            return nullptr;
        }
    }
    RT_HIP_NULL(hipSetDevice(device));
    rt_scene *s = new rt_scene();
    s->device = device;
    const int vtot = vert_off[n_poly];
    s->h_vert_off.assign(vert_off, vert_off + n_poly + 1);
    s->h_verts.assign(verts, verts + 3 * (size_t)vtot);
    s->h_planes.assign(planes, planes + 4 * (size_t)n_poly);
    std::vector<double> frames(9 * (size_t)n_poly), uv(2 * (size_t)vtot);
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
        for (int v = 0; v < nv; ++v) {
            d3 P = ld3(p + 3 * v);
            uv[2 * (size_t)(v0 + v)] = dot(P, fx);
            uv[2 * (size_t)(v0 + v) + 1] = dot(P, fy);
            for (int k = 0; k < 3; ++k) {
                bb[k] = std::fmin(bb[k], p[3 * v + k]);
                bb[3 + k] = std::fmax(bb[3 + k], p[3 * v + k]);
            }
        }
    }
    bool ok = upload(s, s->h_vert_off, &s->view.vert_off) == 0 && upload(s, s->h_verts, &s->view.verts) == 0 &&
              upload(s, s->h_planes, &s->view.planes) == 0 && upload(s, frames, &s->view.frames) == 0 &&
              upload(s, s->h_bbox, &s->view.bbox) == 0 && upload(s, uv, &s->view.uv) == 0 &&
              upload(s, mat, &s->view.material) == 0 && upload(s, bld, &s->view.building) == 0;
    if (ok && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
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
    (void)hipSetDevice(s->device);
    if (s->stream) {
        (void)hipStreamSynchronize(s->stream);
        (void)hipStreamDestroy(s->stream);
    }
    for (void *p : s->allocations) (void)hipFree(p);
    if (s->ws) (void)hipFree(s->ws);
    delete s;
}

extern "C" int32_t rt_scene_polygon_count(const rt_scene *s) { return s ? s->view.n : -1; }

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
