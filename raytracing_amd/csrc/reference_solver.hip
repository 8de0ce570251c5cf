// Reference-compatible mode of the solve: every path the closed-form kernels found is put through the reference's own
// solver (rt_hybrd.h: MINPACK hybr on the reference's zero functions, started where the reference starts it), and kept
// only when that solver reports success and its points pass the reference's checks (ray_tracing.py:307-314, 341-350,
// 369-376).  The reference loses valid paths to its solver (a stall on the common line of two planes reported as
// success, a root reached but reported as failure with the hand-written derivative of geometry.py:710-713); this mode
// loses the same ones.  The paths are few (the survivors of the obstruction tests), so one thread per path is enough.
#include "rt_candidates.h"
#include "rt_common.h"

// (after the other headers: it switches fused multiply-adds off for the rest of this file, which the restatement of
// the solver needs and nothing else here minds)
#include "rt_hybrd.h"

#define REJ_MAX_GROUPS (2 * RT_MAX_ORDER)
struct RejudgeGroups {
    int n_groups;
    int n_refl[REJ_MAX_GROUPS], has_edge[REJ_MAX_GROUPS];
    ChainSource chains[REJ_MAX_GROUPS];
    const int32_t *edge_ids[REJ_MAX_GROUPS];
};

// every polygon against one segment, as the reference's check does (geometry.py:857-884)
__device__ static bool any_polygon_cuts(const SceneView &sc, d3 A, d3 B) {
    for (int p = 0; p < sc.n; ++p)
        if (polygon_cuts_segment(sc, p, A, B)) return true;
    return false;
}

// check_reflections / check_reflections_and_diffraction (ray_tracing.py:243-279) on the solver's own points
__device__ static bool passes_reference_checks(const SceneView &sc, d3 tx, d3 R, const d3 *X, const int32_t *poly, int n_refl,
                                               bool has_edge, const double *edge) {
    const double min_distance_tol = 0.005;
    for (int i = 0; i < n_refl; ++i)
        if (!polygon_contains_point(sc, poly[i], X[i], true, 1e-8)) return false;
    // lines of check_reflections: the emitter, the reflection points, then the receiver or (with an edge) the
    // diffraction point
    d3 L[RT_MAX_ORDER + 2];
    L[0] = tx;
    for (int i = 0; i < n_refl; ++i) L[i + 1] = X[i];
    L[n_refl + 1] = has_edge ? X[n_refl] : R;
    for (int i = 0; i <= n_refl; ++i)
        if (any_polygon_cuts(sc, L[i], L[i + 1])) return false;
    for (int i = 0; i <= n_refl; ++i)
        if (norm3(L[i] - L[i + 1]) < min_distance_tol) return false;
    if (has_edge) {
        // point_on_edge_ (geometry.py:456-479)
        const d3 P0 = ld3(edge), P1 = ld3(edge + 3), D = X[n_refl];
        d3 fx, fy, fz;
        frame_from_vector(P1 - P0, fx, fy, fz);
        const double tol = 1e-8;
        const double ax = dot(P0, fx), ay = dot(P0, fy), az = dot(P0, fz), bz = dot(P1, fz);
        const double px = dot(D, fx), py = dot(D, fy), pz = dot(D, fz);
        if (!(ax - tol <= px && px <= ax + tol && ay - tol <= py && py <= ay + tol && az - tol <= pz && pz <= bz + tol))
            return false;
        if (any_polygon_cuts(sc, D, R)) return false;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_rejudge(SceneView sc, RejudgeGroups G, const double *__restrict__ edges, d3 tx,
                                                 const double *__restrict__ rx, const int32_t *__restrict__ p_rx,
                                                 const int32_t *__restrict__ p_group, const int32_t *__restrict__ p_cand,
                                                 double *__restrict__ p_pts, int64_t n, uint8_t *__restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = p_group[i];
    if (g < 0 || g >= G.n_groups) {  // a line of sight: no solver was involved
        keep[i] = 1;
        return;
    }
    const int n_refl = G.n_refl[g];
    const bool has_edge = G.has_edge[g] != 0;
    const int64_t c = p_cand[i];
    int32_t poly[RT_MAX_ORDER];
    double planes[4 * RT_MAX_ORDER];
    if (n_refl > 0) chain_load(G.chains[g], n_refl, c, poly);
    for (int k = 0; k < n_refl; ++k) {
        for (int q = 0; q < 4; ++q) planes[4 * k + q] = sc.planes[4 * (size_t)poly[k] + q];
    }
    double edge[6];
    if (has_edge) {
        const double *e = edges + 6 * (size_t)G.edge_ids[g][c];
        for (int q = 0; q < 6; ++q) edge[q] = e[q];
    }
    const double origin[3] = {tx.x, tx.y, tx.z};
    const double *dest = rx + 3 * (size_t)p_rx[i];
    double pts[3 * RT_MAX_ORDER];
    const int n_pts = n_refl + (has_edge ? 1 : 0);
    const bool ok = rt_hybr::solve_like_reference(origin, dest, planes, n_refl, has_edge ? edge : nullptr, pts, nullptr);
    if (!ok) {
        keep[i] = 0;
        return;
    }
    // The reference checks ITS solver's points (ray_tracing.py:310-313): they can sit a few 1e-7 m from the root the
    // closed form gives (the solver stops at xtol), which decides differently when the point is that close to the rim
    // of its polygon or a segment passes that close to a corner.  So the checks are redone on them, every polygon
    // against every segment as the reference does.
    double *mine = p_pts + (size_t)i * RT_MAX_ORDER * 3;
    d3 X[RT_MAX_ORDER];
    for (int q = 0; q < n_pts; ++q) X[q] = ld3(pts + 3 * q);
    const bool good = passes_reference_checks(sc, tx, ld3(dest), X, poly, n_refl, has_edge, edge);
    if (good)
        for (int q = 0; q < 3 * n_pts; ++q) mine[q] = pts[q];  // the reference stores its solver's points
    keep[i] = good ? 1 : 0;
}

__global__ void k_rejudge_scatter(const uint8_t *__restrict__ keep, int64_t n, const int32_t *__restrict__ a_rx,
                                  const int32_t *__restrict__ a_group, const int32_t *__restrict__ a_cand,
                                  const double *__restrict__ a_pts, int32_t *__restrict__ b_rx, int32_t *__restrict__ b_group,
                                  int32_t *__restrict__ b_cand, double *__restrict__ b_pts, unsigned long long *count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const unsigned long long at = atomicAdd(count, 1ull);
    b_rx[at] = a_rx[i];
    b_group[at] = a_group[i];
    b_cand[at] = a_cand[i];
    for (int q = 0; q < 3 * RT_MAX_ORDER; ++q) b_pts[at * 3 * RT_MAX_ORDER + q] = a_pts[(size_t)i * 3 * RT_MAX_ORDER + q];
}

__global__ void k_rejudge_zero(unsigned long long *count) { *count = 0ull; }

// Re-judges the n paths in the output arrays (device memory) and packs the kept ones to the front; *n_kept = how many.
// Queued on the scene's stream and waited for.
int rt_rejudge_paths(rt_scene *scene, const rt_candidates *cand, const double *tx, const double *d_rx, int32_t *d_out_rx,
                     int32_t *d_out_group, int32_t *d_out_cand, double *d_out_pts, int64_t n, int64_t *n_kept) {
    *n_kept = n;
    if (n <= 0) return 0;
    if ((int)cand->groups.size() > REJ_MAX_GROUPS) RT_FAIL("rt_rejudge_paths: too many candidate groups");
    hipStream_t st = scene->stream;
    RejudgeGroups G{};
    G.n_groups = (int)cand->groups.size();
    for (int g = 0; g < G.n_groups; ++g) {
        G.n_refl[g] = cand->groups[g].n_refl;
        G.has_edge[g] = cand->groups[g].has_edge;
        G.chains[g] = cand->groups[g].source;
        G.edge_ids[g] = cand->groups[g].edge_ids;
    }
    const size_t np = (size_t)n;
    uint8_t *keep = (uint8_t *)rt_scene_temp(scene, np);
    int32_t *b_rx = (int32_t *)rt_scene_temp(scene, 4 * np), *b_group = (int32_t *)rt_scene_temp(scene, 4 * np),
            *b_cand = (int32_t *)rt_scene_temp(scene, 4 * np);
    double *b_pts = (double *)rt_scene_temp(scene, sizeof(double) * 3 * RT_MAX_ORDER * np);
    unsigned long long *count = scene->d_counters + 12;
    if (!keep || !b_rx || !b_group || !b_cand || !b_pts) RT_FAIL("rt_rejudge_paths: out of device memory");
    hipLaunchKernelGGL(k_rejudge_zero, dim3(1), dim3(1), 0, st, count);
    hipLaunchKernelGGL(k_rejudge, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, scene->view, G, cand->d_edges, ld3(tx), d_rx,
                       d_out_rx, d_out_group, d_out_cand, d_out_pts, n, keep);
    hipLaunchKernelGGL(k_rejudge_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keep, n, d_out_rx, d_out_group,
                       d_out_cand, d_out_pts, b_rx, b_group, b_cand, b_pts, count);
    RT_HIP(hipGetLastError());
    unsigned long long kept = 0;
    RT_HIP(hipMemcpyAsync(&kept, count, sizeof(kept), hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    if (kept > 0) {
        RT_HIP(hipMemcpyAsync(d_out_rx, b_rx, 4 * kept, hipMemcpyDeviceToDevice, st));
        RT_HIP(hipMemcpyAsync(d_out_group, b_group, 4 * kept, hipMemcpyDeviceToDevice, st));
        RT_HIP(hipMemcpyAsync(d_out_cand, b_cand, 4 * kept, hipMemcpyDeviceToDevice, st));
        RT_HIP(hipMemcpyAsync(d_out_pts, b_pts, sizeof(double) * 3 * RT_MAX_ORDER * kept, hipMemcpyDeviceToDevice, st));
        RT_HIP(hipStreamSynchronize(st));
    }
    *n_kept = (int64_t)kept;
    return 0;
}

extern "C" int rt_scene_set_reference_solver(rt_scene *scene, int on) {
    if (!scene) RT_FAIL("rt_scene_set_reference_solver: null scene");
    scene->reference_solver = on != 0;
    return 0;
}
