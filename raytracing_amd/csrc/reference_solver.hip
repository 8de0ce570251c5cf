// Reference-compatible mode of the solve: every path the closed-form kernels found is put through the reference's own
// solver (rt_hybrd.h: MINPACK hybr on the reference's zero functions, started where the reference starts it), and kept
// only when that solver reports success and its points pass the reference's checks (ray_tracing.py:307-314, 341-350,
// 369-376).  The reference loses valid paths to its solver (a stall on the common line of two planes reported as
// success, a root reached but reported as failure with the hand-written derivative of geometry.py:710-713); this mode
// loses the same ones.  The paths are few (the survivors of the obstruction tests), so one thread per path is enough.
#include <algorithm>
#include <cstdlib>

#include "rt_candidates.h"
#include "rt_common.h"

// (after the other headers: it switches fused multiply-adds off for the rest of this file, which the restatement of
// the solver needs and nothing else here minds)
#include "rt_hybrd.h"

// (the run-time-sized form of the solver carries the unroll requests of the unrolled forms, which the optimiser cannot
// honour there and says so once per loop)
#pragma clang diagnostic ignored "-Wpass-failed"

#define REJ_MAX_GROUPS (2 * RT_MAX_ORDER)
// evaluations after which the first pass hands a run to the second, which goes on from the state the first left (on
// the lln map 97 in 100 runs of reflection + diffraction systems end before 96; a trip of the first pass costs about
// twice what the same evaluation costs a run that has a wave to itself, and the second pass holds one or two
// thousand runs at a time; lln step with the hand-over at 64 / 96 / 160 / 200 / 240 / 300 / 420 evaluations:
// re-judging 9.0 / 8.9 / 7.9 / 7.6 / 8.1 / 8.5 / 9.0 ms)
#define REJ_FIRST_PASS_NFEV 200

int rt_launch_obstruct(rt_scene *scene, const double *d_segs, int64_t n_seg, const uint8_t *d_mask, int64_t n_masked,
                       uint8_t *d_out, const unsigned long long *d_count, int64_t count_add, int64_t n_hint,
                       const PathRec *d_paths, int round, bool clear_flags, const int32_t *d_index, const double *tx,
                       const double *d_rx);

// The reference's checks that look at the path alone (ray_tracing.py:243-279 without the obstruction tests, which the
// obstruction kernels run afterwards on the solver's points): every reflection point in its polygon, consecutive
// points at least 5 mm apart, the diffraction point on its edge.
__device__ static bool passes_point_checks(const SceneView &sc, d3 tx, d3 R, const d3 *X, const int32_t *poly, int n_refl,
                                           bool has_edge, const double *edge) {
    const double min_distance_tol = 0.005;
    // (loops over the path's points run to the compile-time limit with a guard, so that they unroll and the small arrays
    // of the caller stay in registers whatever n_refl is)
#pragma unroll
    for (int i = 0; i < RT_MAX_ORDER; ++i)
        if (i < n_refl && !polygon_contains_point(sc, poly[i], X[i], true, 1e-8)) return false;
    // lines of check_reflections: the emitter, the reflection points, then the receiver or (with an edge) the
    // diffraction point
    d3 prev = tx;
#pragma unroll
    for (int i = 0; i <= RT_MAX_ORDER; ++i) {
        if (i > n_refl) break;
        d3 next = R;
        if (i < n_refl || has_edge) next = X[i < RT_MAX_ORDER ? i : RT_MAX_ORDER - 1];
        if (norm3(prev - next) < min_distance_tol) return false;
        prev = next;
    }
    if (has_edge) {
        // point_on_edge_ (geometry.py:456-479)
        d3 D = X[0];
#pragma unroll
        for (int i = 1; i < RT_MAX_ORDER; ++i)
            if (i == n_refl) D = X[i];
        const d3 P0 = ld3(edge), P1 = ld3(edge + 3);
        d3 fx, fy, fz;
        frame_from_vector(P1 - P0, fx, fy, fz);
        const double tol = 1e-8;
        const double ax = dot(P0, fx), ay = dot(P0, fy), az = dot(P0, fz), bz = dot(P1, fz);
        const double px = dot(D, fx), py = dot(D, fy), pz = dot(D, fz);
        if (!(ax - tol <= px && px <= ax + tol && ay - tol <= py && py <= ay + tol && az - tol <= pz && pz <= bz + tol))
            return false;
    }
    return true;
}

// The found paths sorted by the shape of their system (= their candidate group): lists[g * n ..] receives the indices
// of the paths of group g, counts[g] how many.  A line of sight involved no solver: its record says "no segment to
// test" and its flag "kept".  One reservation per block and group.
__global__ __launch_bounds__(256) void k_rejudge_classify(const int32_t *__restrict__ p_rx, const int32_t *__restrict__ p_group,
                                                          const int32_t *__restrict__ p_cand, int64_t n, int n_groups,
                                                          int32_t *__restrict__ lists, unsigned long long *counts,
                                                          PathRec *__restrict__ recs, uint8_t *__restrict__ flags) {
    __shared__ int block_count[REJ_MAX_GROUPS];
    __shared__ unsigned long long block_base[REJ_MAX_GROUPS];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < REJ_MAX_GROUPS) block_count[threadIdx.x] = 0;
    __syncthreads();
    int g = -1, slot = 0;
    if (i < n) {
        g = rt_group_of(p_group[i]);
        if (g < 0 || g >= n_groups) {
            g = -1;
            PathRec &rec = recs[i];
            rec.rx = p_rx[i];
            rec.group = -1;
            rec.cand = p_cand[i];
            rec.nseg = 0;
            rec.seg0 = 0;
            rec.seg_stride = 0;
            rec.npts = 0;
            flags[i] = 0;
        } else {
            slot = atomicAdd(&block_count[g], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < n_groups && block_count[threadIdx.x] > 0)
        block_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)block_count[threadIdx.x]);
    __syncthreads();
    if (g >= 0) lists[(size_t)g * n + block_base[g] + slot] = (int32_t)i;
}

// The reference's solver call on one path of a group, the checks on the solver's points that need no other polygon, and
// the path's record for the obstruction kernels (flag set = dropped).  Returns false when the solver call ended
// unfinished (nothing is written then).  NR, HE: the shape of the group's systems as template arguments (the method
// then runs in registers), or -1, -1.
// `solve(origin, destination, planes, edge or null, points out, unfinished out)` is the solver call (nested or one-loop
// form, limited or not, fresh or resumed) and returns the solver's verdict.
template <int NR, int HE, class Solve>
__device__ __forceinline__ bool rejudge_one(const SceneView &sc, const ChainSource &chains, const int32_t *__restrict__ edge_ids,
                                            int n_refl_rt, int has_edge_rt, const double *__restrict__ edges, d3 tx,
                                            const double *__restrict__ rx, const int32_t *__restrict__ p_rx,
                                            const int32_t *__restrict__ p_group, const int32_t *__restrict__ p_cand, int64_t i,
                                            int group, Solve solve,
                                            bool write, PathRec *__restrict__ recs, uint8_t *__restrict__ flags) {
    const int n_refl = NR >= 0 ? NR : n_refl_rt;
    const bool has_edge = HE >= 0 ? HE != 0 : has_edge_rt != 0;
    const int64_t c = rt_cand_of(p_group[i], p_cand[i]);
    int32_t poly[RT_MAX_ORDER];
    double planes[4 * RT_MAX_ORDER];
    if (n_refl > 0) chain_load(chains, n_refl, c, poly);
#pragma unroll
    for (int q = 0; q < RT_MAX_ORDER; ++q)
        if (q < n_refl)
            for (int w = 0; w < 4; ++w) planes[4 * q + w] = sc.planes[4 * (size_t)poly[q] + w];
    double edge[6];
    if (has_edge) {
        const double *e = edges + 6 * (size_t)edge_ids[c];
        for (int q = 0; q < 6; ++q) edge[q] = e[q];
    }
    const double origin[3] = {tx.x, tx.y, tx.z};
    const int32_t r = p_rx[i];
    const double dest[3] = {rx[3 * (size_t)r], rx[3 * (size_t)r + 1], rx[3 * (size_t)r + 2]};
    double pts[3 * RT_MAX_ORDER];
    const int n_pts = n_refl + (has_edge ? 1 : 0);
    bool unfinished = false;
    bool ok = solve(origin, dest, planes, has_edge ? edge : nullptr, pts, &unfinished);
    if (unfinished) return false;
    // The reference checks ITS solver's points (ray_tracing.py:310-313): they can sit a few 1e-7 m from the root the
    // closed form gives (the solver stops at xtol), which decides differently when the point is that close to the rim
    // of its polygon or a segment passes that close to a corner.  So the checks are redone on them.
    if (ok) {
        d3 X[RT_MAX_ORDER];
#pragma unroll
        for (int q = 0; q < RT_MAX_ORDER; ++q)
            if (q < n_pts) X[q] = ld3(pts + 3 * q);
        ok = passes_point_checks(sc, tx, ld3(dest), X, poly, n_refl, has_edge, edge);
    }
    if (write) {
        PathRec &rec = recs[i];
        rec.rx = r;
        rec.group = rt_group_word(group, c);
        rec.cand = rt_cand_word(c);
        rec.nseg = n_pts + 1;
        rec.seg0 = 0;
        rec.seg_stride = 0;
        rec.npts = n_pts;
#pragma unroll
        for (int q = 0; q < 3 * RT_MAX_ORDER; ++q)
            if (q < 3 * n_pts) rec.pts[q] = pts[q];  // the reference stores its solver's points
        flags[i] = ok ? 0 : 1;
    }
    return true;
}

// First pass: one thread per path of the group.  A solver run lasts as long as it takes evaluations (10 to 1 000), and
// a wave as long as its longest lane, so a run that has not ended after nfev_limit evaluations is given up here and its
// path handed to the second pass.  The systems of up to four unknowns with a finite-difference Jacobian (REJ_FLAT) run
// the method in its one-loop form (rt_hybr::hybr_flat: the lanes of a wave, which are in different places of the
// method, evaluate together) and leave the state of their run behind, so that the second pass goes on where they
// stopped (the first `state_cap` of a group's late paths have room for it; a later one is run again from its start).
#define REJ_STATE_DOUBLES(NR, HE) RT_FLAT_STATE_DOUBLES(3 * (NR) + (HE))
// waves per SIMD the second pass is compiled for: two (256 registers each) where the method's state fits them without
// spilling -- up to four unknowns --, so that 2 048 late runs go side by side instead of 1 024
#define REJ_LATE_WAVES(NR, HE) ((NR) >= 0 && 3 * (NR) + (HE) <= 4 ? 2 : 1)
// The one-loop form with its hand-over is for the shapes of up to four unknowns (one reflection, with or without a
// diffraction after it).  From six unknowns on its state no longer fits the registers as well as the nested form's
// (330 against 234 for two reflections, 512 and spills for three), a run that has a wave to itself is a quarter
// slower in it, and the first pass gains less than that costs: those shapes keep the nested form, a first pass of
// REJ_NESTED_FIRST_PASS_NFEV evaluations and a second pass that starts the run again (config 2, whose re-judging is one
// double-reflection run of a thousand evaluations: 3.6 ms nested, 4.6 in the one-loop form; config 4: 12.4 and 17.9).
#define REJ_FLAT(NR, HE) ((NR) >= 1 && 3 * (NR) + (HE) <= 4)
#define REJ_NESTED_FIRST_PASS_NFEV 96
template <int NR, int HE>
__global__ __launch_bounds__(64) void k_rejudge_solve(SceneView sc, ChainSource chains, const int32_t *__restrict__ edge_ids,
                                                       int n_refl_rt, int has_edge_rt, const double *__restrict__ edges, d3 tx,
                                                       const double *__restrict__ rx, const int32_t *__restrict__ p_rx,
                                                       const int32_t *__restrict__ p_group,
                                                       const int32_t *__restrict__ p_cand, const int32_t *__restrict__ list,
                                                       const unsigned long long *__restrict__ count, int group, int nfev_limit,
                                                       int32_t *__restrict__ late, unsigned long long *late_count,
                                                       double *__restrict__ states, int state_cap,
                                                       PathRec *__restrict__ recs, uint8_t *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (REJ_FLAT(NR, HE)) {
        if (k >= (int64_t)*count) return;
        const int64_t i = list[k];
        auto claim = [&]() -> double * {  // a place in the late list, and room for the state with it
            const unsigned long long slot = atomicAdd(late_count, 1ull);
            late[slot] = (int32_t)i;
            return slot < (unsigned long long)state_cap ? states + slot * REJ_STATE_DOUBLES(NR, HE) : nullptr;
        };
        auto solve = [&](const double *o, const double *d, const double *planes, const double *edge, double *pts, bool *unfinished) {
            return rt_hybr::solve_shape_flat<NR, HE>(o, d, planes, edge, pts, nullptr, nfev_limit, nullptr, claim, unfinished);
        };
        rejudge_one<NR, HE>(sc, chains, edge_ids, n_refl_rt, has_edge_rt, edges, tx, rx, p_rx, p_group, p_cand, i, group, solve, true, recs,
                            flags);
    } else {
        bool done = true;
        int64_t i = 0;
        if (k < (int64_t)*count) {
            i = list[k];
            auto solve = [&](const double *o, const double *d, const double *planes, const double *edge, double *pts, bool *unfinished) {
                return rt_hybr::solve_shape<NR, HE>(o, d, planes, n_refl_rt, edge, pts, nullptr, nfev_limit, unfinished);
            };
            done = rejudge_one<NR, HE>(sc, chains, edge_ids, n_refl_rt, has_edge_rt, edges, tx, rx, p_rx, p_group, p_cand, i, group, solve,
                                       true, recs, flags);
        }
        // late list: one reservation per wave
        const unsigned long long m = __ballot(!done);
        if (m == 0ull) return;
        unsigned long long base = 0;
        const int lane = threadIdx.x & 63;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(late_count, (unsigned long long)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (!done) late[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// Second pass: the long runs, ONE PATH PER WAVE (every lane makes the same run, lane 0 writes): lanes that run
// different systems take each other's branches as well as their own, and a wave with three long runs lasted five times
// as long as its longest; alone, a run of a thousand evaluations takes what its own chain of operations takes.
template <int NR, int HE>
__global__ __launch_bounds__(64, REJ_LATE_WAVES(NR, HE)) void k_rejudge_solve_late(SceneView sc, ChainSource chains, const int32_t *__restrict__ edge_ids,
                                                            int n_refl_rt, int has_edge_rt, const double *__restrict__ edges,
                                                            d3 tx, const double *__restrict__ rx,
                                                            const int32_t *__restrict__ p_rx, const int32_t *__restrict__ p_group,
                                                            const int32_t *__restrict__ p_cand,
                                                            const int32_t *__restrict__ late,
                                                            const unsigned long long *__restrict__ late_count,
                                                            const double *__restrict__ states, int state_cap, int group,
                                                            PathRec *__restrict__ recs, uint8_t *__restrict__ flags) {
    const unsigned long long n_late = *late_count;
    for (unsigned long long k = blockIdx.x; k < n_late; k += gridDim.x) {
        const int64_t i = late[k];
        if constexpr (REJ_FLAT(NR, HE)) {
            const double *from = k < (unsigned long long)state_cap ? states + k * REJ_STATE_DOUBLES(NR, HE) : nullptr;
            auto solve = [&](const double *o, const double *d, const double *planes, const double *edge, double *pts, bool *unfinished) {
                return rt_hybr::solve_shape_flat<NR, HE>(o, d, planes, edge, pts, nullptr, 0x7fffffff, from, rt_hybr::NoHandOver(),
                                                         unfinished);
            };
            rejudge_one<NR, HE>(sc, chains, edge_ids, n_refl_rt, has_edge_rt, edges, tx, rx, p_rx, p_group, p_cand, i, group, solve,
                                threadIdx.x == 0, recs, flags);
        } else {
            auto solve = [&](const double *o, const double *d, const double *planes, const double *edge, double *pts, bool *unfinished) {
                return rt_hybr::solve_shape<NR, HE>(o, d, planes, n_refl_rt, edge, pts, nullptr, 0x7fffffff, unfinished);
            };
            rejudge_one<NR, HE>(sc, chains, edge_ids, n_refl_rt, has_edge_rt, edges, tx, rx, p_rx, p_group, p_cand, i, group, solve,
                                threadIdx.x == 0, recs, flags);
        }
    }
}

__global__ void k_rejudge_scatter(const uint8_t *__restrict__ flags, const PathRec *__restrict__ recs, int64_t n,
                                  const double *__restrict__ a_pts, int32_t *__restrict__ b_rx, int32_t *__restrict__ b_group,
                                  int32_t *__restrict__ b_cand, double *__restrict__ b_pts, unsigned long long *count) {
    __shared__ int wave_total[4];
    __shared__ unsigned long long block_base;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = i < n && !flags[i];
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wave_total[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = total > 0 ? atomicAdd(count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (!keep) return;
    unsigned long long at = block_base + __popcll(mask & ((1ull << lane) - 1ull));
    for (int q = 0; q < wave; ++q) at += wave_total[q];
    const PathRec &rec = recs[i];
    b_rx[at] = rec.rx;
    b_group[at] = rec.group;
    b_cand[at] = rec.cand;
    // a line of sight has no points of its own: the caller's stay (zeros)
    for (int q = 0; q < 3 * RT_MAX_ORDER; ++q)
        b_pts[at * 3 * RT_MAX_ORDER + q] = q < 3 * rec.npts ? rec.pts[q] : a_pts[(size_t)i * 3 * RT_MAX_ORDER + q];
}

__global__ void k_rejudge_zero(unsigned long long *counts, int n) {
    if ((int)threadIdx.x < n) counts[threadIdx.x] = 0ull;
}

// the kept paths back into the caller's arrays (the count is read on the device: no host wait between the kernels)
__global__ void k_rejudge_copy_back(const unsigned long long *__restrict__ count, const int32_t *__restrict__ b_rx,
                                    const int32_t *__restrict__ b_group, const int32_t *__restrict__ b_cand,
                                    const double *__restrict__ b_pts, int32_t *__restrict__ a_rx, int32_t *__restrict__ a_group,
                                    int32_t *__restrict__ a_cand, double *__restrict__ a_pts) {
    const unsigned long long kept = *count;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < kept;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        a_rx[i] = b_rx[i];
        a_group[i] = b_group[i];
        a_cand[i] = b_cand[i];
        for (int q = 0; q < 3 * RT_MAX_ORDER; ++q) a_pts[i * 3 * RT_MAX_ORDER + q] = b_pts[i * 3 * RT_MAX_ORDER + q];
    }
}

// Re-judges the n paths in the output arrays (device memory) and packs the kept ones to the front; *n_kept = how many.
// Queued on the scene's stream and waited for once.
int rt_rejudge_paths(rt_scene *scene, const rt_candidates *cand, const double *tx, const double *d_rx, int32_t *d_out_rx,
                     int32_t *d_out_group, int32_t *d_out_cand, double *d_out_pts, int64_t n, int64_t *n_kept) {
    *n_kept = n;
    scene->last_rejudge_ms = 0;
    scene->last_rejudge_in = n;
    scene->last_rejudge_kept = n;
    if (n <= 0) return 0;
    const int n_groups = (int)cand->groups.size();
    if (n_groups > REJ_MAX_GROUPS) RT_FAIL("rt_rejudge_paths: too many candidate groups");
    if (n > 0x7fffffffLL) RT_FAIL("rt_rejudge_paths: too many paths");
    hipStream_t st = scene->stream;
    const size_t np = (size_t)n;
    uint8_t *flags = (uint8_t *)rt_scene_temp(scene, np);
    PathRec *recs = (PathRec *)rt_scene_temp(scene, sizeof(PathRec) * np);
    int32_t *lists = (int32_t *)rt_scene_temp(scene, 4 * np * (size_t)std::max(n_groups, 1));
    int32_t *late = (int32_t *)rt_scene_temp(scene, 4 * np * (size_t)std::max(n_groups, 1));
    int32_t *b_rx = (int32_t *)rt_scene_temp(scene, 4 * np), *b_group = (int32_t *)rt_scene_temp(scene, 4 * np),
            *b_cand = (int32_t *)rt_scene_temp(scene, 4 * np);
    double *b_pts = (double *)rt_scene_temp(scene, sizeof(double) * 3 * RT_MAX_ORDER * np);
    // counters of this step: [0 .. n_groups) paths per group, [REJ_MAX_GROUPS] kept paths, then the paths per group
    // handed to the second pass (a block of the scene's temporary memory)
    unsigned long long *counts = (unsigned long long *)rt_scene_temp(scene, sizeof(unsigned long long) * (2 * REJ_MAX_GROUPS + 1));
    if (!flags || !recs || !lists || !late || !b_rx || !b_group || !b_cand || !b_pts || !counts)
        RT_FAIL("rt_rejudge_paths: out of device memory");
    unsigned long long *kept_count = counts + REJ_MAX_GROUPS;
    unsigned long long *late_counts = counts + REJ_MAX_GROUPS + 1;
    // two events of the scene's pool around the kernels (the solve's own timer reads only the events it handed out)
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (scene->timing) {
        for (int k = 0; k < 2; ++k) {
            if (scene->events_used == scene->event_pool.size()) {
                hipEvent_t e = nullptr;
                RT_HIP(hipEventCreate(&e));
                scene->event_pool.push_back(e);
            }
            ev[k] = scene->event_pool[scene->events_used++];
        }
        RT_HIP(hipEventRecord(ev[0], st));
    }
    hipLaunchKernelGGL(k_rejudge_zero, dim3(1), dim3(64), 0, st, counts, 2 * REJ_MAX_GROUPS + 1);
    hipLaunchKernelGGL(k_rejudge_classify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_out_rx, d_out_group, d_out_cand,
                       n, n_groups, lists, counts, recs, flags);
    const d3 TX = ld3(tx);
    int max_nseg = 0;
    // room for the state of the runs that the first pass hands over (one-loop shapes), per group
    double *states[REJ_MAX_GROUPS] = {nullptr};
    int state_cap[REJ_MAX_GROUPS] = {0};
    for (int g = 0; g < n_groups; ++g) {
        const CandGroup &G = cand->groups[g];
        const int unknowns = 3 * G.n_refl + (G.has_edge ? 1 : 0);
        if (G.count == 0 || G.n_refl < 1 || unknowns > 4) continue;  // (REJ_FLAT)
        state_cap[g] = (int)std::min<int64_t>(n, 32768);
        states[g] = (double *)rt_scene_temp(scene, sizeof(double) * RT_FLAT_STATE_DOUBLES(unknowns) * (size_t)state_cap[g]);
        if (!states[g]) state_cap[g] = 0;  // (out of memory: those runs start again in the second pass)
    }
    static const int first_pass_nfev = [] {
        const char *e = getenv("RT_REJ_FIRST_PASS_NFEV");  // (for measurements of the hand-over point)
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : REJ_FIRST_PASS_NFEV;
    }();
    // The groups that end with a diffraction on the scene's stream, the pure reflections beside them on its side stream
    // (same priority): a group's second pass lasts as long as its longest run and fills a fraction of the chip, and
    // the two kinds have about the same to do.  Each stream: first passes, then second passes.
    if (!scene->side_stream) RT_HIP(rt_take_stream(scene->device, &scene->side_stream));
    if (!scene->side_event) RT_HIP(hipEventCreateWithFlags(&scene->side_event, hipEventDisableTiming));
    if (!scene->rejudge_event) RT_HIP(hipEventCreateWithFlags(&scene->rejudge_event, hipEventDisableTiming));
    RT_HIP(hipEventRecord(scene->rejudge_event, st));
    RT_HIP(hipStreamWaitEvent(scene->side_stream, scene->rejudge_event, 0));
    for (int pass = 0; pass < 2; ++pass) {
        for (int g = 0; g < n_groups; ++g) {
            const CandGroup &G = cand->groups[g];
            if (G.count == 0) continue;
            if (pass == 0) max_nseg = std::max(max_nseg, G.n_refl + (G.has_edge ? 1 : 0) + 1);
            hipStream_t where = G.has_edge ? st : scene->side_stream;
            // (a group's share of the paths is known on the device only: the first pass gets a grid for all of them,
            // and the blocks beyond the group's share end at once; the second strides over what was handed to it)
            const dim3 grid1((unsigned)((n + 63) / 64)), grid2((unsigned)std::min<int64_t>(n, 8192)), block(64);
#define REJ_LAUNCH(NR, HE)                                                                                                       \
    if (pass == 0)                                                                                                               \
        hipLaunchKernelGGL((k_rejudge_solve<NR, HE>), grid1, block, 0, where, scene->view, G.source, G.edge_ids, G.n_refl,       \
                           G.has_edge, cand->d_edges, TX, d_rx, d_out_rx, d_out_group, d_out_cand, lists + (size_t)g * np, counts + g, g,     \
                           REJ_FLAT(NR, HE) ? first_pass_nfev : std::min(first_pass_nfev, REJ_NESTED_FIRST_PASS_NFEV),          \
                           late + (size_t)g * np, late_counts + g, states[g], state_cap[g], recs, flags);                        \
    else                                                                                                                         \
        hipLaunchKernelGGL((k_rejudge_solve_late<NR, HE>), grid2, block, 0, where, scene->view, G.source, G.edge_ids, G.n_refl,  \
                           G.has_edge, cand->d_edges, TX, d_rx, d_out_rx, d_out_group, d_out_cand, late + (size_t)g * np, late_counts + g,    \
                           states[g], state_cap[g], g, recs, flags)
            const int shape = G.n_refl * 2 + (G.has_edge ? 1 : 0);
            switch (shape) {
                case 1: REJ_LAUNCH(0, 1); break;
                case 2: REJ_LAUNCH(1, 0); break;
                case 3: REJ_LAUNCH(1, 1); break;
                case 4: REJ_LAUNCH(2, 0); break;
                case 5: REJ_LAUNCH(2, 1); break;
                case 6: REJ_LAUNCH(3, 0); break;
                default: REJ_LAUNCH(-1, -1); break;
            }
#undef REJ_LAUNCH
        }
    }
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(scene->side_event, scene->side_stream));
    RT_HIP(hipStreamWaitEvent(st, scene->side_event, 0));
    // every segment of every path still standing against the whole scene (ray_tracing.py:257-262, 279), last segment
    // first, a launch per segment place; a path found cut is skipped by the launches after
    for (int round = 0; round < max_nseg; ++round)
        if (rt_launch_obstruct(scene, nullptr, n, nullptr, 0, flags, nullptr, 0, 0, recs, round, false, nullptr, tx, d_rx)) return -1;
    hipLaunchKernelGGL(k_rejudge_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flags, recs, n, d_out_pts, b_rx,
                       b_group, b_cand, b_pts, kept_count);
    hipLaunchKernelGGL(k_rejudge_copy_back, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, kept_count,
                       b_rx, b_group, b_cand, b_pts, d_out_rx, d_out_group, d_out_cand, d_out_pts);
    RT_HIP(hipGetLastError());
    if (ev[1]) RT_HIP(hipEventRecord(ev[1], st));
    unsigned long long kept = 0;
    RT_HIP(hipMemcpyAsync(&kept, kept_count, sizeof(kept), hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    if (ev[1]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) scene->last_rejudge_ms = ms;
    }
    *n_kept = (int64_t)kept;
    scene->last_rejudge_kept = (int64_t)kept;
    return 0;
}

extern "C" int rt_solve_last_rejudge(rt_scene *scene, double *ms, int64_t *n_in, int64_t *n_kept) {
    if (!scene) RT_FAIL("rt_solve_last_rejudge: null scene");
    if (ms) *ms = scene->last_rejudge_ms;
    if (n_in) *n_in = scene->last_rejudge_in;
    if (n_kept) *n_kept = scene->last_rejudge_kept;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the stand-alone solver entry points as the reference's own solver answers them (geometry.py:660-687, 718-748, 800-833)
// ---------------------------------------------------------------------------------------------------------------
template <int NR, int HE>
__global__ __launch_bounds__(64) void k_solve_reference(const double *__restrict__ origin, const double *__restrict__ dests,
                                                         int64_t n, const double *__restrict__ planes_in, int k,
                                                         const double *__restrict__ edge_in, double *__restrict__ out,
                                                         uint8_t *__restrict__ ok, int32_t *__restrict__ nfev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int n_refl = NR >= 0 ? NR : k;
    double planes[4 * RT_MAX_ORDER], edge[6];
#pragma unroll
    for (int q = 0; q < 4 * RT_MAX_ORDER; ++q)
        if (q < 4 * n_refl) planes[q] = planes_in[q];
    if (edge_in)
        for (int q = 0; q < 6; ++q) edge[q] = edge_in[q];
    const double a[3] = {origin[0], origin[1], origin[2]};
    const double b[3] = {dests[3 * i], dests[3 * i + 1], dests[3 * i + 2]};
    double pts[3 * RT_MAX_ORDER];
    int evals = 0;
    const bool good = rt_hybr::solve_shape<NR, HE>(a, b, planes, n_refl, edge_in ? edge : nullptr, pts, &evals);
    const int n_pts = n_refl + (edge_in ? 1 : 0);
#pragma unroll
    for (int q = 0; q < 3 * RT_MAX_ORDER; ++q)
        if (q < 3 * n_pts) out[(size_t)i * 3 * n_pts + q] = pts[q];
    ok[i] = good ? 1 : 0;
    if (nfev) nfev[i] = evals;
}

extern "C" int rt_solve_like_reference(int device, const double *origin, const double *dests, int64_t n_dest,
                                       const double *planes, int32_t k, const double *edge, double *out_points,
                                       uint8_t *out_success, int32_t *out_nfev) {
    if (!origin || (n_dest > 0 && (!dests || !out_points || !out_success))) RT_FAIL("rt_solve_like_reference: null argument");
    if (k < 0 || k > RT_MAX_ORDER || (k > 0 && !planes)) RT_FAIL("rt_solve_like_reference: bad plane count");
    const int n_pts = k + (edge ? 1 : 0);
    if (n_pts == 0) RT_FAIL("rt_solve_like_reference: nothing to solve");
    if (n_dest <= 0) return 0;
    RT_HIP(rt_use_device(device));
    double *d_in = nullptr, *d_out = nullptr;
    uint8_t *d_ok = nullptr;
    int32_t *d_nfev = nullptr;
    const size_t n_in = 3 + 3 * (size_t)n_dest + 4 * RT_MAX_ORDER + 6;
    hipError_t e = hipMalloc(&d_in, sizeof(double) * n_in);
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(double) * 3 * n_pts * n_dest);
    if (e == hipSuccess) e = hipMalloc(&d_ok, n_dest);
    if (e == hipSuccess) e = hipMalloc(&d_nfev, sizeof(int32_t) * n_dest);
    double *d_origin = d_in, *d_dests = d_in + 3, *d_planes = d_dests + 3 * n_dest, *d_edge = d_planes + 4 * RT_MAX_ORDER;
    if (e == hipSuccess) e = hipMemcpy(d_origin, origin, sizeof(double) * 3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dests, dests, sizeof(double) * 3 * n_dest, hipMemcpyHostToDevice);
    if (e == hipSuccess && k) e = hipMemcpy(d_planes, planes, sizeof(double) * 4 * k, hipMemcpyHostToDevice);
    if (e == hipSuccess && edge) e = hipMemcpy(d_edge, edge, sizeof(double) * 6, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((n_dest + 63) / 64)), block(64);
#define SOLVE_LAUNCH(NR, HE)                                                                                                  \
    hipLaunchKernelGGL((k_solve_reference<NR, HE>), grid, block, 0, 0, d_origin, d_dests, n_dest, d_planes, k,                \
                       edge ? d_edge : (const double *)nullptr, d_out, d_ok, d_nfev)
        switch (k * 2 + (edge ? 1 : 0)) {
            case 1: SOLVE_LAUNCH(0, 1); break;
            case 2: SOLVE_LAUNCH(1, 0); break;
            case 3: SOLVE_LAUNCH(1, 1); break;
            case 4: SOLVE_LAUNCH(2, 0); break;
            case 5: SOLVE_LAUNCH(2, 1); break;
            case 6: SOLVE_LAUNCH(3, 0); break;
            default: SOLVE_LAUNCH(-1, -1); break;
        }
#undef SOLVE_LAUNCH
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_points, d_out, sizeof(double) * 3 * n_pts * n_dest, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_success, d_ok, n_dest, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_nfev) e = hipMemcpy(out_nfev, d_nfev, sizeof(int32_t) * n_dest, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_ok);
    (void)hipFree(d_nfev);
    if (e != hipSuccess) RT_FAIL(std::string("rt_solve_like_reference: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int rt_scene_set_reference_solver(rt_scene *scene, int on) {
    if (!scene) RT_FAIL("rt_scene_set_reference_solver: null scene");
    scene->reference_solver = on != 0;
    return 0;
}
