// Candidate enumeration and the per-(receiver, candidate) solve of RayTracingProblem.solve
// (reference ray_tracing.py:232-237 line of sight, :243-279 checks, :281-390 enumeration and solve;
//  solvers geometry.py:592-833).
//
// Enumeration: the reference walks chains depth first, receivers innermost.  The set of chains does not depend on
// the receiver, so it is built once per emitter as flat arrays by prefix-sum expansion over the CSR form of the
// visibility matrix: level k = for every chain of level k-1, the polygons visible from its last polygon, ascending —
// the same order the recursion visits them in.
//
// Solve: planar mirrors make the min-path-tracing system linear after unfolding, so the solution of
// geometry.py:635-657 is the image construction: reflect the emitter through the chain's planes once per candidate,
// then per receiver intersect backwards.  The Keller equation of geometry.py:690-715 has the closed form
// z = zA + (zB - zA) rhoA / (rhoA + rhoB) in the edge's frame.  For reflection-then-diffraction candidates the
// reference's system aims the reflections at the edge's first endpoint (geometry.py:821 keeps points[-1] there), which
// makes the reflection points and their checks independent of the receiver: they are done once per candidate.
//
// Work mapping of the candidate kernels: one lane per candidate, all lanes of a wave walk the same receiver (scalar
// loads).  A cheap padded pre-test rejects almost every pair; pairs that pass run the exact arithmetic and checks and
// append a path record plus its segments; the obstruction kernel then tests all segments; a last kernel keeps the
// paths none of whose segments is obstructed.
#include <algorithm>
#include <cstring>

#include "rt_candidates.h"
#include "rt_common.h"

int rt_launch_obstruct(rt_scene *scene, const double *d_segs, int64_t n_seg, const uint8_t *d_mask, int64_t n_masked,
                       uint8_t *d_out, const unsigned long long *d_count, int64_t count_add, int64_t n_hint,
                       const PathRec *d_paths, int round, bool clear_flags, const int32_t *d_index, const double *tx,
                       const double *d_rx);
int rt_launch_list_clear_flags(rt_scene *scene, const uint8_t *d_flags, int64_t n_cap, const unsigned long long *d_count,
                               int64_t count_add, int64_t n_hint, int32_t *d_list, unsigned long long *d_list_count);

#define MIN_DISTANCE_TOL 0.005 /* ray_tracing.py:252 */

struct Queues {
    PathRec *paths;
    double *segs;                  // null: only the records are queued (their segments follow from their points)
    int64_t path_cap, seg_cap;
    unsigned long long *counters;  // [0] paths, [1] segments queued behind the lines of sight
    int64_t base;                  // records and segments [0, base) are the chunk's lines of sight
    uint8_t *hits;                 // obstruction flags, cleared by whoever queues a path
    int hits_per_path;             // one flag per path (else one per segment)
};

// clears the obstruction flags of a path that was just queued
__device__ __forceinline__ void clear_hits(const Queues &q, int64_t path_slot, int64_t seg0, int64_t stride, int nseg) {
    if (q.hits_per_path) {
        q.hits[path_slot] = 0;
    } else {
        for (int i = 0; i < nseg; ++i) q.hits[seg0 + i * stride] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// enumeration kernels
// ---------------------------------------------------------------------------------------------------------------

// out[i] = sum_{j<i} in[j], out[n] = total; one 1024-thread workgroup
__global__ __launch_bounds__(1024) void k_exclusive_scan(const int32_t *__restrict__ in, int64_t n,
                                                          int64_t *__restrict__ out) {
    __shared__ long long wsum[16];
    __shared__ long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const long long v = i < n ? in[i] : 0;
        long long x = v;
        for (int off = 1; off < 64; off <<= 1) {
            long long y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        if (wave == 0) {
            long long s = lane < 16 ? wsum[lane] : 0;
            for (int off = 1; off < 16; off <<= 1) {
                long long y = __shfl_up(s, off);
                if (lane >= off) s += y;
            }
            if (lane < 16) wsum[lane] = s;
        }
        __syncthreads();
        const long long prefix = carry + (wave > 0 ? wsum[wave - 1] : 0) + x - v;
        if (i < n) out[i] = prefix;
        __syncthreads();
        if (tid == 1023) carry = prefix + v;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// deg[i] = number of set bytes in row i of the visibility matrix; one wave per row
__global__ void k_row_degree(const uint8_t *__restrict__ vis, int n, int32_t *__restrict__ deg) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int c = 0;
    for (int j = lane; j < n; j += 64) c += vis[(size_t)row * n + j] ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) deg[row] = c;
}

// col[row_off[i] ...] = visible polygons of row i, ascending; one wave per row
__global__ void k_fill_csr(const uint8_t *__restrict__ vis, int n, const int64_t *__restrict__ row_off,
                           int32_t *__restrict__ col) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int64_t w = row_off[row];
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool f = j < n && vis[(size_t)row * n + j];
        const unsigned long long m = __ballot(f);
        if (f) col[w + __popcll(m & ((1ull << lane) - 1))] = j;
        w += __popcll(m);
    }
}

// flags of the 1-diffraction candidates: an edge whose either polygon the emitter sees (ray_tracing.py:335)
__global__ void k_edge_flags(const uint8_t *__restrict__ ev, const int32_t *__restrict__ pairs, int64_t n_edges,
                             int32_t *__restrict__ flags) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) flags[e] = (ev[pairs[2 * e]] || ev[pairs[2 * e + 1]]) ? 1 : 0;
}

__global__ void k_flags_from_bytes(const uint8_t *__restrict__ b, int64_t n, int32_t *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = b[i] ? 1 : 0;
}

// ordered compaction: out[off[i]] = i where flags[i]
__global__ void k_scatter_indices(const int32_t *__restrict__ flags, const int64_t *__restrict__ off, int64_t n,
                                  int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) out[off[i]] = (int32_t)i;
}

// tcnt[l] = number of sharp edges with a polygon visible from l (container_utils.py:41-46); one wave per polygon
__global__ void k_edge_touch_count(const uint8_t *__restrict__ vis, int n, const int32_t *__restrict__ pairs,
                                   int64_t n_edges, int32_t *__restrict__ tcnt) {
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (l >= n) return;
    const uint8_t *row = vis + (size_t)l * n;
    int c = 0;
    for (int64_t e = lane; e < n_edges; e += 64) c += (row[pairs[2 * e]] || row[pairs[2 * e + 1]]) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) tcnt[l] = c;
}

// per-parent child count of a level: degree (reflection level) or touching edges (diffraction level) of its last polygon
__global__ void k_parent_counts(const int32_t *__restrict__ parents, int64_t n_parents, int klen,
                                const int32_t *__restrict__ per_polygon, int32_t *__restrict__ cnt) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_parents) cnt[p] = per_polygon[parents[p * klen + klen - 1]];
}

__global__ void k_degree_from_offsets(const int64_t *__restrict__ row_off, int n, int32_t *__restrict__ deg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) deg[i] = (int32_t)(row_off[i + 1] - row_off[i]);
}

// children of a reflection level; one thread per child, parent found by binary search in the offsets
__global__ void k_expand_reflect(const int32_t *__restrict__ parents, int64_t n_parents, int klen,
                                 const int64_t *__restrict__ off, const int64_t *__restrict__ row_off,
                                 const int32_t *__restrict__ col, int64_t n_children, int32_t *__restrict__ out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_children) return;
    int64_t lo = 0, hi = n_parents;  // last p with off[p] <= c
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
    }
    const int32_t *par = parents + lo * klen;
    int32_t *o = out + c * (klen + 1);
    for (int k = 0; k < klen; ++k) o[k] = par[k];
    o[klen] = col[row_off[par[klen - 1]] + (c - off[lo])];
}

// children of a diffraction level: (parent chain, edge) for every edge touching what the last polygon sees, in edge
// order; one wave per parent
__global__ void k_expand_edges(const int32_t *__restrict__ parents, int64_t n_parents, int klen,
                               const int64_t *__restrict__ off, const uint8_t *__restrict__ vis, int n,
                               const int32_t *__restrict__ pairs, int64_t n_edges, int32_t *__restrict__ out_chains,
                               int32_t *__restrict__ out_edges) {
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= n_parents) return;
    const int32_t *par = parents + p * klen;
    const uint8_t *row = vis + (size_t)par[klen - 1] * n;
    int64_t w = off[p];
    for (int64_t e0 = 0; e0 < n_edges; e0 += 64) {
        const int64_t e = e0 + lane;
        const bool f = e < n_edges && (row[pairs[2 * e]] || row[pairs[2 * e + 1]]);
        const unsigned long long m = __ballot(f);
        if (f) {
            const int64_t slot = w + __popcll(m & ((1ull << lane) - 1));
            for (int k = 0; k < klen; ++k) out_chains[slot * klen + k] = par[k];
            out_edges[slot] = (int32_t)e;
        }
        w += __popcll(m);
    }
}

// coarse[b] = parent of candidate b * CHAIN_COARSE of a level that is not stored (last p with off[p] <= c)
__global__ void k_coarse_parents(const int64_t *__restrict__ off, int64_t n_parents, int64_t n_children,
                                 int32_t *__restrict__ coarse) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = b * CHAIN_COARSE;
    if (c >= n_children) return;
    int64_t lo = 0, hi = n_parents;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
    }
    coarse[b] = (int32_t)lo;
}

__global__ void k_iota(int32_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

// ---------------------------------------------------------------------------------------------------------------
// enumeration host side
// ---------------------------------------------------------------------------------------------------------------

template <typename T>
static T *cand_alloc(rt_candidates *c, size_t count) {
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(T) * (count ? count : 1)) != hipSuccess) return nullptr;
    c->allocations.push_back(d);
    return (T *)d;
}

static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

#define CAND_CHECK(ptr)                                             \
    if (!(ptr)) {                                                   \
        rt_set_error("rt_enumerate: out of device memory");        \
        rt_candidates_destroy(c);                                   \
        return nullptr;                                             \
    }
#define CAND_HIP(call)                                                                   \
    if ((call) != hipSuccess) {                                                          \
        rt_set_error(std::string("rt_enumerate: ") + #call + " failed: " +               \
                     hipGetErrorString(hipGetLastError()));                              \
        rt_candidates_destroy(c);                                                        \
        return nullptr;                                                                  \
    }

static const int64_t MAX_GROUP_CANDIDATES = (1ll << 31) - 1;

extern "C" rt_candidates *rt_enumerate(rt_scene *scene, const uint8_t *vis, const uint8_t *ev,
                                       const int32_t *edge_pairs, const double *edges, int64_t n_edges,
                                       int32_t max_order, int with_diffraction, int vis_on_device) {
    if (!scene || !vis || !ev) {
        rt_set_error("rt_enumerate: null argument");
        return nullptr;
    }
    if (max_order < 0 || max_order > RT_MAX_ORDER) {
        rt_set_error("rt_enumerate: max_order must be in 0.." + std::to_string(RT_MAX_ORDER));
        return nullptr;
    }
    if (n_edges < 0 || (n_edges > 0 && (!edge_pairs || !edges))) {
        rt_set_error("rt_enumerate: bad edge arrays");
        return nullptr;
    }
    const int n = scene->view.n;
    for (int64_t e = 0; e < n_edges; ++e) {
        if (edge_pairs[2 * e] < 0 || edge_pairs[2 * e] >= n || edge_pairs[2 * e + 1] < 0 || edge_pairs[2 * e + 1] >= n) {
            rt_set_error("rt_enumerate: edge polygon index out of range");
            return nullptr;
        }
    }
    if (rt_use_device(scene->device) != hipSuccess) {
        rt_set_error("rt_enumerate: hipSetDevice failed");
        return nullptr;
    }
    hipStream_t st = scene->stream;
    rt_candidates *c = new rt_candidates();
    c->scene = scene;
    c->max_order = max_order;
    c->n_edges = with_diffraction ? n_edges : 0;
    const int n_groups = max_order == 0 ? 0 : 2 * max_order;
    c->groups.resize(n_groups);

    // a matrix already on this GPU is read where it lies (only inside this call: the candidates keep no pointer to it)
    const uint8_t *d_vis = vis;
    if (!vis_on_device) {
        uint8_t *copy = cand_alloc<uint8_t>(c, (size_t)n * n);
        CAND_CHECK(copy);
        CAND_HIP(hipMemcpyAsync(copy, vis, (size_t)n * n, hipMemcpyHostToDevice, st));
        d_vis = copy;
    }
    c->d_ev = cand_alloc<uint8_t>(c, n);
    CAND_CHECK(c->d_ev);
    CAND_HIP(hipMemcpyAsync(c->d_ev, ev, n, hipMemcpyHostToDevice, st));
    c->d_edges = cand_alloc<double>(c, 6 * (size_t)n_edges);
    c->d_edge_pairs = cand_alloc<int32_t>(c, 2 * (size_t)n_edges);
    CAND_CHECK(c->d_edges);
    CAND_CHECK(c->d_edge_pairs);
    if (n_edges) {
        CAND_HIP(hipMemcpyAsync(c->d_edges, edges, sizeof(double) * 6 * n_edges, hipMemcpyHostToDevice, st));
        CAND_HIP(hipMemcpyAsync(c->d_edge_pairs, edge_pairs, sizeof(int32_t) * 2 * n_edges, hipMemcpyHostToDevice, st));
    }
    if (max_order == 0) {
        CAND_HIP(hipStreamSynchronize(st));
        return c;
    }

    // CSR adjacency of the visibility matrix
    int32_t *deg = cand_alloc<int32_t>(c, n);
    int64_t *row_off = cand_alloc<int64_t>(c, (size_t)n + 1);
    CAND_CHECK(deg);
    CAND_CHECK(row_off);
    hipLaunchKernelGGL(k_row_degree, dim3(blocks_for(n, 4)), dim3(256), 0, st, d_vis, n, deg);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, deg, (int64_t)n, row_off);
    int64_t nnz = 0;
    CAND_HIP(hipMemcpyAsync(&nnz, row_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CAND_HIP(hipStreamSynchronize(st));
    int32_t *col = cand_alloc<int32_t>(c, (size_t)nnz);
    CAND_CHECK(col);
    hipLaunchKernelGGL(k_fill_csr, dim3(blocks_for(n, 4)), dim3(256), 0, st, d_vis, n, row_off, col);

    // scratch for flags / counts / offsets, sized as the levels grow
    auto scan_counts = [&](const int32_t *cnt, int64_t m, int64_t *off, int64_t *total) -> bool {
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, cnt, m, off);
        if (hipMemcpyAsync(total, off + m, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        return hipStreamSynchronize(st) == hipSuccess;
    };

    // level 1: polygons the emitter sees, ascending (ray_tracing.py:284, 326-327)
    {
        int32_t *flags = cand_alloc<int32_t>(c, n);
        int64_t *off = cand_alloc<int64_t>(c, (size_t)n + 1);
        CAND_CHECK(flags);
        CAND_CHECK(off);
        hipLaunchKernelGGL(k_flags_from_bytes, dim3(blocks_for(n, 256)), dim3(256), 0, st, c->d_ev, (int64_t)n, flags);
        int64_t total = 0;
        if (!scan_counts(flags, n, off, &total)) CAND_HIP(hipErrorUnknown);
        CandGroup &g = c->groups[0];
        g.n_refl = 1;
        g.count = total;
        g.chains = cand_alloc<int32_t>(c, (size_t)total);
        CAND_CHECK(g.chains);
        hipLaunchKernelGGL(k_scatter_indices, dim3(blocks_for(n, 256)), dim3(256), 0, st, flags, off, (int64_t)n, g.chains);
        g.source.chains = g.chains;
    }
    // deeper reflection levels (ray_tracing.py:316-322)
    for (int k = 2; k <= max_order; ++k) {
        const CandGroup &par = c->groups[k - 2];
        CandGroup &g = c->groups[k - 1];
        g.n_refl = k;
        if (par.count == 0) continue;
        int32_t *cnt = cand_alloc<int32_t>(c, (size_t)par.count);
        int64_t *off = cand_alloc<int64_t>(c, (size_t)par.count + 1);
        CAND_CHECK(cnt);
        CAND_CHECK(off);
        hipLaunchKernelGGL(k_parent_counts, dim3(blocks_for(par.count, 256)), dim3(256), 0, st, par.chains, par.count,
                           k - 1, deg, cnt);
        int64_t total = 0;
        if (!scan_counts(cnt, par.count, off, &total)) CAND_HIP(hipErrorUnknown);
        // (the deepest level, when it is not stored, may be wider than 32 bits: its paths carry a split index)
        const bool stays_virtual = k == max_order && (int64_t)sizeof(int32_t) * total * k > scene->stored_chain_bytes_limit;
        if (total > (stays_virtual ? RT_MAX_WIDE_CANDIDATES : MAX_GROUP_CANDIDATES)) {
            rt_set_error("rt_enumerate: more than 2^31-1 candidates at reflection order " + std::to_string(k) +
                         (k == max_order ? " (and more than a level that is not stored can hold)" : " (a level that has to be stored)"));
            rt_candidates_destroy(c);
            return nullptr;
        }
        g.count = total;
        // The deepest level is the large one (lln: 1 129, 9.0e5, ~4.5e7 chains at orders 1, 2, 3): above the limit it is
        // not stored.  Every other level is a parent of something (the next reflection level, a diffraction level) and
        // small beside it.
        if (k == max_order && k >= 2 && (int64_t)sizeof(int32_t) * total * k > scene->stored_chain_bytes_limit) {
            const int64_t n_coarse = (total + CHAIN_COARSE - 1) / CHAIN_COARSE;
            int32_t *coarse = cand_alloc<int32_t>(c, (size_t)n_coarse);
            CAND_CHECK(coarse);
            hipLaunchKernelGGL(k_coarse_parents, dim3(blocks_for(n_coarse, 256)), dim3(256), 0, st, off, par.count, total, coarse);
            g.source = ChainSource{nullptr, par.chains, off, row_off, col, par.count, coarse};
            continue;
        }
        g.chains = cand_alloc<int32_t>(c, (size_t)total * k);
        CAND_CHECK(g.chains);
        g.source.chains = g.chains;
        if (total)
            hipLaunchKernelGGL(k_expand_reflect, dim3(blocks_for(total, 256)), dim3(256), 0, st, par.chains, par.count,
                               k - 1, off, row_off, col, total, g.chains);
    }
    // diffraction groups
    for (int m = 0; m < max_order; ++m) {
        CandGroup &g = c->groups[max_order + m];
        g.n_refl = m;
        g.has_edge = 1;
    }
    if (with_diffraction && n_edges > 0) {
        // 1 diffraction (ray_tracing.py:334-350)
        {
            int32_t *flags = cand_alloc<int32_t>(c, (size_t)n_edges);
            int64_t *off = cand_alloc<int64_t>(c, (size_t)n_edges + 1);
            CAND_CHECK(flags);
            CAND_CHECK(off);
            hipLaunchKernelGGL(k_edge_flags, dim3(blocks_for(n_edges, 256)), dim3(256), 0, st, c->d_ev, c->d_edge_pairs,
                               n_edges, flags);
            int64_t total = 0;
            if (!scan_counts(flags, n_edges, off, &total)) CAND_HIP(hipErrorUnknown);
            CandGroup &g = c->groups[max_order];
            g.count = total;
            g.edge_ids = cand_alloc<int32_t>(c, (size_t)total);
            CAND_CHECK(g.edge_ids);
            hipLaunchKernelGGL(k_scatter_indices, dim3(blocks_for(n_edges, 256)), dim3(256), 0, st, flags, off, n_edges,
                               g.edge_ids);
        }
        // m reflections + 1 diffraction, m >= 1 (ray_tracing.py:355-388)
        if (max_order >= 2) {
            int32_t *tcnt = cand_alloc<int32_t>(c, n);
            CAND_CHECK(tcnt);
            hipLaunchKernelGGL(k_edge_touch_count, dim3(blocks_for(n, 4)), dim3(256), 0, st, d_vis, n, c->d_edge_pairs,
                               n_edges, tcnt);
            for (int m = 1; m < max_order; ++m) {
                const CandGroup &par = c->groups[m - 1];
                CandGroup &g = c->groups[max_order + m];
                if (par.count == 0) continue;
                int32_t *cnt = cand_alloc<int32_t>(c, (size_t)par.count);
                int64_t *off = cand_alloc<int64_t>(c, (size_t)par.count + 1);
                CAND_CHECK(cnt);
                CAND_CHECK(off);
                hipLaunchKernelGGL(k_parent_counts, dim3(blocks_for(par.count, 256)), dim3(256), 0, st, par.chains,
                                   par.count, m, tcnt, cnt);
                int64_t total = 0;
                if (!scan_counts(cnt, par.count, off, &total)) CAND_HIP(hipErrorUnknown);
                if (total > MAX_GROUP_CANDIDATES) {
                    rt_set_error("rt_enumerate: more than 2^31-1 candidates with " + std::to_string(m) +
                                 " reflections and a diffraction");
                    rt_candidates_destroy(c);
                    return nullptr;
                }
                g.count = total;
                g.chains = cand_alloc<int32_t>(c, (size_t)total * m);
                g.edge_ids = cand_alloc<int32_t>(c, (size_t)total);
                CAND_CHECK(g.chains);
                CAND_CHECK(g.edge_ids);
                g.source.chains = g.chains;
                if (total)
                    hipLaunchKernelGGL(k_expand_edges, dim3(blocks_for(par.count, 4)), dim3(256), 0, st, par.chains,
                                       par.count, m, off, d_vis, n, c->d_edge_pairs, n_edges, g.chains, g.edge_ids);
            }
        }
    }
    CAND_HIP(hipGetLastError());
    CAND_HIP(hipStreamSynchronize(st));
    return c;
}

extern "C" void rt_candidates_destroy(rt_candidates *c) {
    if (!c) return;
    if (c->scene) (void)rt_use_device(c->scene->device);
    for (void *p : c->allocations) (void)hipFree(p);
    delete c;
}

extern "C" int32_t rt_candidates_group_count(const rt_candidates *c) { return c ? (int32_t)c->groups.size() : -1; }

extern "C" int rt_candidates_counts(const rt_candidates *c, int64_t *counts) {
    if (!c || !counts) RT_FAIL("rt_candidates_counts: null argument");
    for (size_t g = 0; g < c->groups.size(); ++g) counts[g] = c->groups[g].count;
    return 0;
}

extern "C" int rt_candidates_group(const rt_candidates *c, int32_t group, int32_t *chains, int32_t *edge_ids) {
    if (!c) RT_FAIL("rt_candidates_group: null handle");
    if (group < 0 || group >= (int)c->groups.size()) RT_FAIL("rt_candidates_group: group out of range");
    const CandGroup &g = c->groups[group];
    if (g.count > MAX_GROUP_CANDIDATES) RT_FAIL("rt_candidates_group: a level of more than 2^31-1 candidates is not listed");
    RT_HIP(rt_use_device(c->scene->device));
    if (chains && g.n_refl > 0 && g.count > 0) {
        if (g.chains) {
            RT_HIP(hipMemcpy(chains, g.chains, sizeof(int32_t) * g.count * g.n_refl, hipMemcpyDeviceToHost));
        } else {
            // a level that is not stored: written out for this call only
            int32_t *tmp = nullptr;
            RT_HIP(hipMalloc((void **)&tmp, sizeof(int32_t) * (size_t)g.count * g.n_refl));
            const ChainSource &s = g.source;
            hipLaunchKernelGGL(k_expand_reflect, dim3(blocks_for(g.count, 256)), dim3(256), 0, c->scene->stream, s.parents,
                               s.n_parents, g.n_refl - 1, s.off, s.row_off, s.col, g.count, tmp);
            hipError_t e = hipStreamSynchronize(c->scene->stream);
            if (e == hipSuccess) e = hipMemcpy(chains, tmp, sizeof(int32_t) * g.count * g.n_refl, hipMemcpyDeviceToHost);
            (void)hipFree(tmp);
            RT_HIP(e);
        }
    }
    if (edge_ids) {
        if (g.has_edge && g.count > 0)
            RT_HIP(hipMemcpy(edge_ids, g.edge_ids, sizeof(int32_t) * g.count, hipMemcpyDeviceToHost));
        else
            for (int64_t i = 0; i < g.count; ++i) edge_ids[i] = -1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// solve kernels
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ double fast_rcp_s(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);
}

// Room for one path and its nseg segments per calling lane (nseg the same for every calling lane): one pair of
// atomics per wave, both in flight together.
__device__ inline bool reserve(const Queues &q, int nseg, int64_t &path_slot, int64_t &seg_slot) {
    const unsigned long long m = __ballot(1);
    const int lane = (int)(threadIdx.x & 63);
    const unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1));
    const unsigned cnt = (unsigned)__popcll(m);
    unsigned long long bp = 0, bs = 0;
    if (rank == 0) {
        bp = atomicAdd(&q.counters[0], (unsigned long long)cnt);
        bs = atomicAdd(&q.counters[1], (unsigned long long)cnt * (unsigned)nseg);
    }
    const unsigned plo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bp);
    const unsigned phi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bp >> 32));
    const unsigned slo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bs);
    const unsigned shi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bs >> 32));
    path_slot = q.base + (int64_t)((((unsigned long long)phi << 32) | plo) + rank);
    seg_slot = q.base + (int64_t)((((unsigned long long)shi << 32) | slo) + (unsigned long long)rank * (unsigned)nseg);
    return path_slot < q.path_cap && seg_slot + nseg <= q.seg_cap;
}

// Exact image construction for K planes: X[i] on plane i for the path origin -> X[0] .. X[K-1] -> dest.
// img[i] = origin mirrored through planes 0..i; g[i] = d_i + img[i]·n_i.  False when an image segment does not
// cross its plane strictly inside (no specular path over the infinite planes).
template <int K>
__device__ inline bool unfold_exact(const d3 *nrm, const double *g, const d3 *img, d3 dest, d3 *X) {
    d3 T = dest;
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        const d3 dir = T - img[i];
        const double t = -g[i] / dot(dir, nrm[i]);
        if (!(t > 0.0 && t < 1.0)) return false;
        X[i] = img[i] + t * dir;
        T = X[i];
    }
    return true;
}

// Exact solve and checks of one (candidate, receiver) pair of a K-reflection group: true when the path survives them;
// X = its reflection points.
template <int K>
__device__ inline bool reflect_exact(const SceneView &sc, const int32_t *chain, d3 tx, d3 R, d3 *X) {
    int32_t poly[K];
    d3 nrm[K], img[K];
    double g[K];
    d3 prev = tx;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        poly[i] = chain[i];
        const double *pl = sc.planes + 4 * (size_t)poly[i];
        nrm[i] = ld3(pl);
        const double s = pl[3] + dot(prev, nrm[i]);
        img[i] = prev - (2.0 * s) * nrm[i];
        g[i] = -s;
        prev = img[i];
    }
    if (!unfold_exact<K>(nrm, g, img, R, X)) return false;
    // Most pairs that come this far fail because an earlier reflection point misses its polygon by far: the extent of
    // every polygon in its own frame (one record each) settles those before the rings are looked at.  The coordinates
    // are the ones the point-in-polygon test forms, so a point inside a ring is inside its extent.
    bool near = true;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double *ur = sc.uv_recs + 16 * (size_t)poly[i];
        const double u = dot(X[i], ld3(ur)), v = dot(X[i], ld3(ur + 3));
        near = near && u >= ur[6] - 1e-9 && u <= ur[8] + 1e-9 && v >= ur[7] - 1e-9 && v <= ur[9] + 1e-9;
    }
    if (!near) return false;
    // ray_tracing.py:253-255; all K tests are made before looking at any of them, so that their loads overlap
    bool inside = true;
#pragma unroll
    for (int i = 0; i < K; ++i) inside = polygon_contains_point(sc, poly[i], X[i], true, 1e-8) && inside;
    if (!inside) return false;
    // ray_tracing.py:268-272
    d3 a = tx;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (!(norm3(a - X[i]) >= MIN_DISTANCE_TOL)) return false;
        a = X[i];
    }
    return norm3(a - R) >= MIN_DISTANCE_TOL;
}

// The reflection points of a pair that reflect_exact has passed: its image construction alone (same arithmetic, same
// points), without the checks.  For the lanes that write a survivor out, which are not the lanes that solved it.
template <int K>
__device__ inline void reflect_points(const SceneView &sc, const int32_t *chain, d3 tx, d3 R, d3 *X) {
    d3 nrm[K], img[K];
    double g[K];
    d3 prev = tx;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double *pl = sc.planes + 4 * (size_t)chain[i];
        nrm[i] = ld3(pl);
        const double s = pl[3] + dot(prev, nrm[i]);
        img[i] = prev - (2.0 * s) * nrm[i];
        g[i] = -s;
        prev = img[i];
    }
    (void)unfold_exact<K>(nrm, g, img, R, X);
}

// Queue record and segments of a surviving reflection path.
template <int K>
__device__ inline void reflect_emit(const Queues &q, int64_t ps, int64_t ss, int group, int64_t c, int64_t r, d3 tx, d3 R,
                                    const d3 *X) {
    PathRec &rec = q.paths[ps];
    rec.rx = (int32_t)r;
    rec.group = rt_group_word(group, c);
    rec.cand = rt_cand_word(c);
    rec.nseg = K + 1;
    rec.seg0 = ss;
    rec.seg_stride = 1;
    rec.npts = K;
    clear_hits(q, ps, ss, 1, K + 1);
#pragma unroll
    for (int i = 0; i < K; ++i) st3(rec.pts + 3 * i, X[i]);
    if (q.segs) {
        d3 a = tx;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            st3(q.segs + 6 * (ss + i), a);
            st3(q.segs + 6 * (ss + i) + 3, X[i]);
            a = X[i];
        }
        st3(q.segs + 6 * (ss + K), a);
        st3(q.segs + 6 * (ss + K) + 3, R);
    }
}

// Reflection candidates: lane = candidate of a K-reflection group, loop = receivers of the block's tile.
// The loop only runs the padded pre-test (is the last reflection point within the extent of its polygon?); pairs that pass are
// parked in a per-wave LDS queue and solved exactly 64 at a time, so the expensive part runs with all lanes busy
// instead of once per receiver with one or two lanes.  Survivors are parked again (few per wave) and written out at
// the end of the block with ONE reservation in the path and segment queues for the whole block: atomics on one
// address serialise at about 10 ns each, which a reservation per wave and exact batch made the longest part of a
// small launch.  A wave whose survivor list fills up before that writes 64 of them out on its own.
#define REFLECT_MAX_RX_TILE 32
struct ReflectShared {
    uint32_t wave_queue[4][128];  // pairs waiting for the exact solve: lane << 16 | receiver within the tile
    uint32_t survivors[4][128];   // same encoding
    int n_survivors[4];
    long long base[2];
    double rx[3 * REFLECT_MAX_RX_TILE];  // the receivers of the block's tile
    int32_t chain[4][64][RT_MAX_ORDER];  // the polygons of each lane's candidate, for the lanes that solve its pairs
};

template <int K>
__device__ __forceinline__ void reflect_write(const SceneView &sc, const int32_t (*wave_chain)[RT_MAX_ORDER], int group, d3 tx,
                                              const double *__restrict__ rx, int64_t c_base, int64_t r_begin,
                                              const Queues &q, uint32_t e, int64_t ps, int64_t ss) {
    const int64_t rr = r_begin + (e & 0xffffu);
    const d3 R = ld3(rx + 3 * rr);
    d3 X[K];
    reflect_points<K>(sc, wave_chain[e >> 16], tx, R, X);
    reflect_emit<K>(q, ps, ss, group, c_base + (e >> 16), rr, tx, R, X);
}

template <int K>
__device__ __forceinline__ void reflect_block(const SceneView &sc, const ChainSource &source, int64_t count,
                                              int group, d3 tx, const double *__restrict__ rx, int64_t rx0, int64_t rx1,
                                              int rx_tile, const Queues &q, unsigned block_x, ReflectShared &sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r_begin = rx0 + (int64_t)blockIdx.y * rx_tile;
    const int64_t r_end = r_begin + rx_tile < rx1 ? r_begin + rx_tile : rx1;
    // The tile's receivers go to LDS first: the pre-test loop below reads a receiver per pair, the same one in every
    // lane, and fetched from global memory trip by trip that wait was most of the loop (10.6 us of a 28 us launch over
    // 1 024 receivers x 2 937 candidates).  The load is in flight while the lanes form their images of the emitter.
    if ((int)threadIdx.x < 3 * (int)(r_end - r_begin)) sh.rx[threadIdx.x] = rx[3 * r_begin + threadIdx.x];
    const int64_t c_base = (int64_t)block_x * blockDim.x + wave * 64;
    const int64_t c = c_base + lane;
    const bool active = c < count;
    d3 prev = tx, nk = mk3(0, 0, 1);
    double gk = 0;
    int32_t last = 0;
    int32_t mine[K];
#pragma unroll
    for (int i = 0; i < K; ++i) mine[i] = 0;
    if (active) chain_load(source, K, c, mine);
    const int32_t(*wave_chain)[RT_MAX_ORDER] = sh.chain[wave];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        last = mine[i];
        sh.chain[wave][lane][i] = last;
        const double *pl = sc.planes + 4 * (size_t)last;
        nk = ld3(pl);
        const double s = pl[3] + dot(prev, nk);
        prev = prev - (2.0 * s) * nk;
        gk = -s;  // d + img·n = -(d + prev·n)
    }
    const d3 im = prev;
    // the last polygon's own frame and the extent of its ring in it: the crossing point of the mirror plane lies in
    // the plane, so two coordinates settle whether it is near the polygon (for a wall, a rectangle, whether it is in it)
    const double *ur = sc.uv_recs + 16 * (size_t)last;
    const d3 fu = ld3(ur), fv = ld3(ur + 3);
    const double u_lo = ur[6], v_lo = ur[7], u_hi = ur[8], v_hi = ur[9];
    const double imu = dot(im, fu), imv = dot(im, fv);

    __syncthreads();  // sh.rx
    int qn = 0, sn = 0;  // wave-uniform
    uint32_t *wq = sh.wave_queue[wave], *ws = sh.survivors[wave];
    // exact solve of the queue entry e of every lane that has one; survivors go to the wave's list
    auto solve_batch = [&](bool has, uint32_t e) {
        bool ok = false;
        if (has) {
            d3 X[K];
            ok = reflect_exact<K>(sc, wave_chain[e >> 16], tx, ld3(rx + 3 * (r_begin + (e & 0xffffu))), X);
        }
        const unsigned long long m = __ballot(ok);
        if (m == 0) return;
        if (ok) ws[sn + __popcll(m & ((1ull << lane) - 1))] = e;
        sn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (sn >= 64) {
            sn -= 64;
            int64_t ps, ss;
            if (reserve(q, K + 1, ps, ss)) reflect_write<K>(sc, wave_chain, group, tx, rx, c_base, r_begin, q, ws[sn + lane], ps, ss);
            __builtin_amdgcn_wave_barrier();
        }
    };
    // four receivers per trip: their pre-tests are independent, which hides the fp64 latency of one behind the others
    for (int64_t r4 = r_begin; r4 < r_end; r4 += 4) {
        bool pass[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r4 + j < r_end ? r4 + j : r_end - 1;
            const d3 R = ld3(sh.rx + 3 * (r - r_begin));  // the same address in every lane
            const d3 dir = R - im;
            const double t = -gk * fast_rcp_s(dot(dir, nk));
            const double pad = 2e-5 * fmax(fabs(dir.x), fmax(fabs(dir.y), fabs(dir.z))) + 2e-6;
            const double hu = imu + t * dot(dir, fu), hv = imv + t * dot(dir, fv);
            pass[j] = active && r4 + j < r_end && t > -1e-6 && t < 1.0 + 1e-6 && hu >= u_lo - pad && hu <= u_hi + pad &&
                      hv >= v_lo - pad && hv <= v_hi + pad;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = __ballot(pass[j]);
            if (m == 0) continue;
            if (pass[j])
                wq[qn + __popcll(m & ((1ull << lane) - 1))] = ((uint32_t)lane << 16) | (uint32_t)(r4 + j - r_begin);
            qn += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64) {
                qn -= 64;
                solve_batch(true, wq[qn + lane]);
            }
        }
    }
    if (qn > 0) solve_batch(lane < qn, lane < qn ? wq[lane] : 0u);
    // one reservation for what the four waves still hold
    if (lane == 0) sh.n_survivors[wave] = sn;
    __syncthreads();
    const int total = sh.n_survivors[0] + sh.n_survivors[1] + sh.n_survivors[2] + sh.n_survivors[3];
    if (total == 0) return;
    if (threadIdx.x == 0) {
        sh.base[0] = q.base + (long long)atomicAdd(&q.counters[0], (unsigned long long)total);
        sh.base[1] = q.base + (long long)atomicAdd(&q.counters[1], (unsigned long long)total * (unsigned)(K + 1));
    }
    __syncthreads();
    if (sh.base[0] + total > q.path_cap || sh.base[1] + (long long)total * (K + 1) > q.seg_cap) return;  // chunk is redone
    int before = 0;
    for (int w = 0; w < wave; ++w) before += sh.n_survivors[w];
    if (lane < sn)
        reflect_write<K>(sc, wave_chain, group, tx, rx, c_base, r_begin, q, ws[lane], sh.base[0] + before + lane,
                         sh.base[1] + (int64_t)(before + lane) * (K + 1));
}

// ---- large batches: receivers in spatial tiles, whole tiles rejected at once -------------------------------------
// A reflection candidate reaches only the receivers inside the pyramid from its image of the emitter through its last
// polygon, a sliver of a coverage map.  The receivers are therefore sorted into spatial tiles of RX_TILE (cells of a
// uniform grid in Morton order, RX_TILE consecutive receivers per tile, each tile with its box), and a lane first
// tests its candidate against a tile's box with interval arithmetic: the pre-test of the one-by-one loop evaluated for
// every receiver position in the box at once, widened so that it can only err towards keeping the tile.  (candidate,
// tile) pairs that survive are parked and expanded 64 at a time into the per-receiver pre-test; its survivors go to
// the exact solve as before.
#define RX_TILE 16
#ifdef RT_REFLECT_STATS
__device__ unsigned long long g_reflect_stats[4];  // tile tests, tiles kept, receiver pre-tests passed, undecided (D spans 0)
extern "C" int rt_debug_reflect_stats(unsigned long long *out4) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_reflect_stats), sizeof(g_reflect_stats)) != hipSuccess) return -1;
    unsigned long long z[4] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_reflect_stats), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
// ---------------------------------------------------------------------------------------------------------------
// Wall hints.  Ninety-nine in a hundred of the "last point -> receiver" segments that the candidate kernels would queue
// are cut by some wall, and the obstruction walk, which starts at the other end, meets the wall that stands next to the
// RECEIVER last (every segment to a receiver inside a building is such a one).  Near its receiver a segment is as low as the
// receiver, so the first wall in its direction cuts it almost surely.  k_rx_hints notes that wall for 256 directions
// around every receiver (a horizontal ray through the obstruction grid, a thread per direction); the diffraction kernel
// puts a pair's last segment to the wall of its direction with the walk's own exact test and drops the pair when the
// wall cuts it.  A hint proves nothing by itself and a missing or useless one costs one test: the paths found are the same.
// Directions are binned by "diamond angle" (y / (|x| + |y|) by quadrant), which needs no arc tangent.
#define RX_HINT_BINS 256 /* directions per receiver (RT_RX_HINT_BINS overrides, for measurements) */
#define RX_HINT_CELLS 24 /* a ray gives up after this many grid cells (RT_RX_HINT_CELLS) */
#ifndef RX_HINT_NEIGHBOURS
#define RX_HINT_NEIGHBOURS 0 /* the walls of the two bins beside it as well: diffract + 0.6 ms, obstruct - 0.3 */
#endif
__device__ inline int rx_hint_bin(double dx, double dy, int bins) {
    const float fx = (float)dx, fy = (float)dy;
    const float s = fabsf(fx) + fabsf(fy);
    const float a = s > 0.0f ? fy / s : 0.0f;
    const float d = fx >= 0.0f ? (fy >= 0.0f ? a : 4.0f + a) : 2.0f - a;
    const int b = (int)(d * (0.25f * (float)bins));
    return b < 0 ? 0 : (b >= bins ? bins - 1 : b);
}
__global__ __launch_bounds__(256) void k_rx_hints(SceneView sc, const double *__restrict__ rx, int64_t n_rx, int bins,
                                                   int max_cells, int32_t *__restrict__ hints) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rx * bins) return;
    const int64_t r = t / bins;
    const int b = (int)(t - r * bins);
    const d3 A = ld3(rx + 3 * r);
    // the middle of the bin
    const double d = (b + 0.5) * (4.0 / bins);
    double fx, fy;
    if (d < 1.0) { fx = 1.0 - d; fy = d; }
    else if (d < 2.0) { fx = 1.0 - d; fy = 2.0 - d; }
    else if (d < 3.0) { fx = d - 3.0; fy = 2.0 - d; }
    else { fx = d - 3.0; fy = d - 4.0; }
    const double inv_len = 1.0 / sqrt(fx * fx + fy * fy);
    const double range = max_cells * sc.gside;
    const d3 V = mk3(range * fx * inv_len, range * fy * inv_len, 0.0);
    const double pad = 1e-5 * fmax(fabs(V.x), fabs(V.y)) + 1e-6;
    // cell by cell along the ray (Amanatides & Woo): the nearest crossing among the polygons of the first cell that has one
    int ci = (int)floor((A.x - sc.gx0) * sc.ginv), cj = (int)floor((A.y - sc.gy0) * sc.ginv);
    const int si = V.x >= 0.0 ? 1 : -1, sj = V.y >= 0.0 ? 1 : -1;
    const double inv_vx = V.x != 0.0 ? 1.0 / V.x : 0.0, inv_vy = V.y != 0.0 ? 1.0 / V.y : 0.0;
    double t_x = V.x != 0.0 ? (sc.gx0 + (ci + (si > 0 ? 1 : 0)) * sc.gside - A.x) * inv_vx : INFINITY;
    double t_y = V.y != 0.0 ? (sc.gy0 + (cj + (sj > 0 ? 1 : 0)) * sc.gside - A.y) * inv_vy : INFINITY;
    const double dt_x = V.x != 0.0 ? sc.gside * fabs(inv_vx) : INFINITY, dt_y = V.y != 0.0 ? sc.gside * fabs(inv_vy) : INFINITY;
    int best = -1;
    for (int step = 0; step < max_cells + 2 && best < 0; ++step) {
        if (ci >= 0 && ci < sc.gnx && cj >= 0 && cj < sc.gny) {
            const OccCell cell = sc.cells[cj * sc.gnx + ci];
            double best_t = INFINITY;
            if (!(A.z - sc.gz0 - pad > (double)cell.zmax))
                for (int k = cell.begin; k < cell.end; ++k) {
                    const OccRec *rec = sc.cell_recs + k;
                    // (the conservative test alone: for a wall, a rectangle in a vertical plane, it says what the exact
                    // test says, and a hint need not be right)
                    if (!occ_pretest(rec, sc, A, V, pad)) continue;
                    const double den = V.x * rec->nx + V.y * rec->ny + V.z * rec->nz;
                    const double at = den != 0.0 ? -(rec->d + (A.x * rec->nx + A.y * rec->ny + A.z * rec->nz)) / den : 0.0;
                    if (at < best_t) {
                        best_t = at;
                        best = rec->poly;
                    }
                }
        }
        if (t_x < t_y) {
            if (t_x > 1.0) break;
            ci += si;
            t_x += dt_x;
        } else {
            if (t_y > 1.0) break;
            cj += sj;
            t_y += dt_y;
        }
    }
    hints[t] = best;
}

// true when the wall hinted for receiver `r` (place in the array the hints were made from) cuts the segment from P to it
__device__ __forceinline__ bool hinted_wall_cuts(const SceneView &sc, const int32_t *__restrict__ hints, int bins, int64_t r,
                                                 d3 P, d3 R) {
    const d3 ray = R - P;
    const int b = rx_hint_bin(-ray.x, -ray.y, bins);
    const double pad = 1e-5 * fmax(fabs(ray.x), fmax(fabs(ray.y), fabs(ray.z))) + 1e-6;
    const int p = hints[r * bins + b];
    if (p >= 0 && occ_pretest(sc.poly_recs + p, sc, P, ray, pad) && polygon_cuts_segment(sc, p, P, R)) return true;
    if (RX_HINT_NEIGHBOURS) {
        const int p1 = hints[r * bins + (b + 1 < bins ? b + 1 : 0)], p2 = hints[r * bins + (b > 0 ? b - 1 : bins - 1)];
        if (p1 >= 0 && p1 != p && occ_pretest(sc.poly_recs + p1, sc, P, ray, pad) && polygon_cuts_segment(sc, p1, P, R)) return true;
        if (p2 >= 0 && p2 != p && p2 != p1 && occ_pretest(sc.poly_recs + p2, sc, P, ray, pad) && polygon_cuts_segment(sc, p2, P, R))
            return true;
    }
    return false;
}

struct RxTiles {
    const double *rxs;     // receiver positions in tile order
    const int32_t *perm;   // perm[k] = index of sorted receiver k in the caller's array
    const double *boxes;   // [n_tiles * 6] centre and half extent of each tile's box
};

struct ReflectTiledShared {
    uint32_t tile_queue[4][128];   // candidate lane << 16 | tile within the block's range
    uint32_t wave_queue[4][128];   // candidate lane << 26 | receiver within the chunk (sorted order)
    uint32_t survivors[4][128];    // same encoding
    double cand[4][64][19];        // per candidate lane: image, plane normal, g, frame axes and extent of the last polygon
    int32_t chain[4][64][RT_MAX_ORDER];  // the polygons of each lane's candidate
    int n_survivors[4];
    long long base[2];
};

template <int K>
__device__ __forceinline__ void reflect_block_tiled(const SceneView &sc, const ChainSource &source, int64_t count,
                                                    int group, d3 tx, const RxTiles &rt, int64_t r0, int64_t r1,
                                                    int tiles_per_block, const Queues &q, unsigned block_x,
                                                    ReflectTiledShared &sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c_base = (int64_t)block_x * blockDim.x + wave * 64;
    const int64_t c = c_base + lane;
    const bool active = c < count;
    d3 prev = tx, nk = mk3(0, 0, 1);
    double gk = 0;
    int32_t last = 0;
    int32_t mine[K];
#pragma unroll
    for (int i = 0; i < K; ++i) mine[i] = 0;
    if (active) chain_load(source, K, c, mine);
    const int32_t(*wave_chain)[RT_MAX_ORDER] = sh.chain[wave];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        last = mine[i];
        sh.chain[wave][lane][i] = last;
        const double *pl = sc.planes + 4 * (size_t)last;
        nk = ld3(pl);
        const double s = pl[3] + dot(prev, nk);
        prev = prev - (2.0 * s) * nk;
        gk = -s;
    }
    const d3 im = prev;
    const double *ur = sc.uv_recs + 16 * (size_t)last;
    const d3 fu = ld3(ur), fv = ld3(ur + 3);
    const double u_lo = ur[6], v_lo = ur[7], u_hi = ur[8], v_hi = ur[9];
    const double imu = dot(im, fu), imv = dot(im, fv);
    {
        double *cd = sh.cand[wave][lane];
        cd[0] = im.x; cd[1] = im.y; cd[2] = im.z; cd[3] = nk.x; cd[4] = nk.y; cd[5] = nk.z; cd[6] = gk;
        cd[7] = fu.x; cd[8] = fu.y; cd[9] = fu.z; cd[10] = fv.x; cd[11] = fv.y; cd[12] = fv.z;
        cd[13] = u_lo; cd[14] = v_lo; cd[15] = u_hi; cd[16] = v_hi; cd[17] = imu; cd[18] = imv;
    }
    __builtin_amdgcn_wave_barrier();

    const int64_t tile0 = r0 / RX_TILE + (int64_t)blockIdx.y * tiles_per_block;  // r0 is a multiple of RX_TILE
    const int64_t tile_end_all = (r1 + RX_TILE - 1) / RX_TILE;
    const int64_t tile1 = tile0 + tiles_per_block < tile_end_all ? tile0 + tiles_per_block : tile_end_all;
    int tn = 0, qn = 0, sn = 0;  // wave-uniform
    uint32_t *tq = sh.tile_queue[wave], *wq = sh.wave_queue[wave], *ws = sh.survivors[wave];

    auto solve_batch = [&](bool has, uint32_t e) {
        bool ok = false;
        if (has) {
            d3 X[K];
            ok = reflect_exact<K>(sc, wave_chain[e >> 26], tx, ld3(rt.rxs + 3 * (r0 + (e & 0x3ffffffu))), X);
            // (the wall hints of the diffraction kernel, tried here on the last segment: 0.34 M of 5 M walked segments
            // fewer on the lln map, and the kernel 0.4 ms slower for the registers the test takes)
        }
        const unsigned long long m = __ballot(ok);
        if (m == 0) return;
        if (ok) ws[sn + __popcll(m & ((1ull << lane) - 1))] = e;
        sn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (sn >= 64) {
            sn -= 64;
            int64_t ps, ss;
            if (reserve(q, K + 1, ps, ss)) {
                const uint32_t w = ws[sn + lane];
                const int64_t rr = r0 + (w & 0x3ffffffu);
                const d3 R = ld3(rt.rxs + 3 * rr);
                d3 X[K];
                reflect_points<K>(sc, wave_chain[w >> 26], tx, R, X);
                reflect_emit<K>(q, ps, ss, group, c_base + (w >> 26), rt.perm[rr], tx, R, X);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    // the receivers of the tile of every lane that has a (candidate, tile) entry, one by one
    auto expand_tiles = [&](bool has, uint32_t e) {
        const double *cd = sh.cand[wave][has ? (e >> 16) : 0];
        const d3 eim = mk3(cd[0], cd[1], cd[2]), enk = mk3(cd[3], cd[4], cd[5]);
        const double eg = cd[6];
        const d3 efu = mk3(cd[7], cd[8], cd[9]), efv = mk3(cd[10], cd[11], cd[12]);
        const double eu_lo = cd[13], ev_lo = cd[14], eu_hi = cd[15], ev_hi = cd[16], eimu = cd[17], eimv = cd[18];
        const int64_t rb = (tile0 + (e & 0xffffu)) * RX_TILE;
        for (int j = 0; j < RX_TILE; ++j) {
            const int64_t r = rb + j;
            bool pass = false;
            if (has && r < r1) {
                const d3 R = ld3(rt.rxs + 3 * r);
                const d3 dir = R - eim;
                const double t = -eg * fast_rcp_s(dot(dir, enk));
                const double pad = 2e-5 * fmax(fabs(dir.x), fmax(fabs(dir.y), fabs(dir.z))) + 2e-6;
                const double hu = eimu + t * dot(dir, efu), hv = eimv + t * dot(dir, efv);
                pass = t > -1e-6 && t < 1.0 + 1e-6 && hu >= eu_lo - pad && hu <= eu_hi + pad && hv >= ev_lo - pad &&
                       hv <= ev_hi + pad;
            }
            const unsigned long long m = __ballot(pass);
#ifdef RT_REFLECT_STATS
            if (lane == 0 && m != 0) atomicAdd(&g_reflect_stats[2], (unsigned long long)__popcll(m));
#endif
            if (m == 0) continue;
            if (pass) wq[qn + __popcll(m & ((1ull << lane) - 1))] = ((e >> 16) << 26) | (uint32_t)(r - r0);
            qn += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64) {
                qn -= 64;
                solve_batch(true, wq[qn + lane]);
            }
        }
    };

    // does the candidate of this lane reach any receiver position inside the box of tile t?
    auto tile_kept = [&](int64_t t) -> bool {
        const double *bx = rt.boxes + 6 * t;  // wave-uniform
        const double cx = bx[0], cy = bx[1], cz = bx[2], ex = bx[3], ey = bx[4], ez = bx[5];
        // every receiver position R = centre + d, |d| <= half extent:  D = (R - im).n in Dc +- Dr
        const double ux = cx - im.x, uy = cy - im.y, uz = cz - im.z;
        const double Dc = ux * nk.x + uy * nk.y + uz * nk.z;
        const double Dr = (ex * fabs(nk.x) + ey * fabs(nk.y) + ez * fabs(nk.z)) * (1.0 + 1e-9) + 1e-9 * (fabs(Dc) + 1.0);
        bool keep = active;
        // The crossing parameter t = -g / D must lie in the pre-test's range (-1e-6, 1 + 1e-6): D has the sign of -g and
        // |D| >= |g| / (1 + 1e-6), unless g is so small that the other sign could give t > -1e-6 (then the tile is kept
        // without looking further).  Oriented so that D is positive: D in [dlo, dhi].
        const double ag = fabs(gk);
        if (keep && ag > 1e-3) {
            const double sgn = gk < 0.0 ? 1.0 : -1.0;
            const double dhi = sgn * Dc + Dr;
            double dlo = sgn * Dc - Dr;
            dlo = fmax(dlo, ag * (1.0 - 2e-6));
            if (dlo > dhi) {
                keep = false;
            } else {
                double tlo = ag * fast_rcp_s(dhi), thi = ag * fast_rcp_s(dlo);  // widened below by far more than their error
                const double tw = 1e-9 * (tlo + thi) + 1e-12;
                tlo = fmax(tlo - tw, 0.0);  // the true t is positive here
                thi = fmin(thi + tw, 1.0 + 1e-6);
                const double pad = 2e-5 * fmax(fabs(ux) + ex, fmax(fabs(uy) + ey, fabs(uz) + ez)) + 4e-6;
                // in-plane coordinate of the crossing point, a + t w, over t in [tlo, thi] and w in [wc - wr, wc + wr]
                // (w = (R - im).axis over the box): the extremes are at corners of that rectangle
                auto outside = [&](double a, d3 axis, double lo, double hi) {
                    const double wc = ux * axis.x + uy * axis.y + uz * axis.z;
                    const double wr = ex * fabs(axis.x) + ey * fabs(axis.y) + ez * fabs(axis.z);
                    const double w0 = wc - wr, w1 = wc + wr;
                    const double hmin = a + fmin(tlo * w0, thi * w0), hmax = a + fmax(tlo * w1, thi * w1);  // t >= 0
                    const double w = 1e-9 * (fabs(a) + fabs(hmin) + fabs(hmax)) + pad;
                    return hmax + w < lo || hmin - w > hi;
                };
                if (outside(imu, fu, u_lo, u_hi) || outside(imv, fv, v_lo, v_hi)) keep = false;
            }
        }
        return keep;
    };
    // two tiles per trip: their tests are independent chains of fp64 operations
    for (int64_t t2 = tile0; t2 < tile1; t2 += 2) {
        bool kept[2];
        kept[0] = tile_kept(t2);
        kept[1] = t2 + 1 < tile1 && tile_kept(t2 + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool keep = kept[j];
            const int64_t t = t2 + j;
            const unsigned long long m = __ballot(keep);
#ifdef RT_REFLECT_STATS
            if (lane == 0 && t < tile1) {
                atomicAdd(&g_reflect_stats[0], (unsigned long long)__popcll(__ballot(active)));
                atomicAdd(&g_reflect_stats[1], (unsigned long long)__popcll(m));
            }
#endif
            if (m == 0) continue;
            if (keep) tq[tn + __popcll(m & ((1ull << lane) - 1))] = ((uint32_t)lane << 16) | (uint32_t)(t - tile0);
            tn += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (tn >= 64) {
                tn -= 64;
                expand_tiles(true, tq[tn + lane]);
            }
        }
    }
    if (tn > 0) expand_tiles(lane < tn, lane < tn ? tq[lane] : 0u);
    if (qn > 0) solve_batch(lane < qn, lane < qn ? wq[lane] : 0u);
    // one reservation for what the four waves still hold
    if (lane == 0) sh.n_survivors[wave] = sn;
    __syncthreads();
    const int total = sh.n_survivors[0] + sh.n_survivors[1] + sh.n_survivors[2] + sh.n_survivors[3];
    if (total == 0) return;
    if (threadIdx.x == 0) {
        sh.base[0] = q.base + (long long)atomicAdd(&q.counters[0], (unsigned long long)total);
        sh.base[1] = q.base + (long long)atomicAdd(&q.counters[1], (unsigned long long)total * (unsigned)(K + 1));
    }
    __syncthreads();
    if (sh.base[0] + total > q.path_cap || sh.base[1] + (long long)total * (K + 1) > q.seg_cap) return;  // chunk is redone
    int before = 0;
    for (int w = 0; w < wave; ++w) before += sh.n_survivors[w];
    if (lane < sn) {
        const uint32_t w = ws[lane];
        const int64_t rr = r0 + (w & 0x3ffffffu);
        const d3 R = ld3(rt.rxs + 3 * rr);
        d3 X[K];
        reflect_points<K>(sc, wave_chain[w >> 26], tx, R, X);
        reflect_emit<K>(q, sh.base[0] + before + lane, sh.base[1] + (int64_t)(before + lane) * (K + 1), group,
                        c_base + (w >> 26), rt.perm[rr], tx, R, X);
    }
}

// The line of sight of receiver rx0 + i of the chunk as record i of the path queue (group -1, one segment): the
// candidate kernels append behind the chunk's lines of sight (Queues::base).
__device__ __forceinline__ void los_emit_one(d3 tx, const double *__restrict__ rx, int64_t rx0, int64_t i, const Queues &q) {
    PathRec &rec = q.paths[i];
    rec.rx = (int32_t)(rx0 + i);
    rec.group = -1;
    rec.cand = 0;
    rec.nseg = 1;
    rec.seg0 = i;
    rec.seg_stride = 1;
    rec.npts = 0;
    clear_hits(q, i, i, 1, 1);
    for (int k = 0; k < RT_MAX_ORDER * 3; ++k) rec.pts[k] = 0.0;
    if (q.segs) {
        st3(q.segs + 6 * i, tx);
        st3(q.segs + 6 * i + 3, ld3(rx + 3 * (rx0 + i)));
    }
}

// All reflection orders of a solve in ONE launch (a small step is bound by the number of dependent launches): the
// blocks [first_block[k-1], first_block[k]) of grid.x work on the k-reflection group.
struct ReflectGroups {
    ChainSource source[RT_MAX_ORDER];
    int64_t count[RT_MAX_ORDER];
    unsigned first_block[RT_MAX_ORDER + 1];
};

template <int MO>
__global__ __launch_bounds__(256) void k_reflect(SceneView sc, ReflectGroups g, d3 tx, const double *__restrict__ rx,
                                                  int64_t rx0, int64_t rx1, int rx_tile, Queues q) {
    __shared__ ReflectShared wave_queue;
    const unsigned b = blockIdx.x;
    if (b == 0) {  // the lines of sight of this row of blocks' receivers
        const int64_t r = rx0 + (int64_t)blockIdx.y * rx_tile + threadIdx.x;
        if ((int)threadIdx.x < rx_tile && r < rx1) los_emit_one(tx, rx, rx0, r - rx0, q);
    }
    if (b < g.first_block[1]) {
        reflect_block<1>(sc, g.source[0], g.count[0], 0, tx, rx, rx0, rx1, rx_tile, q, b, wave_queue);
    } else if (MO >= 2 && b < g.first_block[2]) {
        reflect_block<(MO >= 2 ? 2 : 1)>(sc, g.source[1], g.count[1], 1, tx, rx, rx0, rx1, rx_tile, q, b - g.first_block[1],
                                         wave_queue);
    } else if (MO >= 3 && b < g.first_block[3]) {
        reflect_block<(MO >= 3 ? 3 : 1)>(sc, g.source[2], g.count[2], 2, tx, rx, rx0, rx1, rx_tile, q, b - g.first_block[2],
                                         wave_queue);
    } else if (MO >= 4) {
        reflect_block<(MO >= 4 ? 4 : 1)>(sc, g.source[3], g.count[3], 3, tx, rx, rx0, rx1, rx_tile, q, b - g.first_block[3],
                                         wave_queue);
    }
}

template <int MO>
__global__ __launch_bounds__(256, MO <= 2 ? 3 : 2) void k_reflect_tiled(SceneView sc, ReflectGroups g, d3 tx, RxTiles rt, int64_t r0,
                                                        int64_t r1, int tiles_per_block, Queues q) {
    __shared__ ReflectTiledShared sh;
    const unsigned b = blockIdx.x;
    if (b < g.first_block[1]) {
        reflect_block_tiled<1>(sc, g.source[0], g.count[0], 0, tx, rt, r0, r1, tiles_per_block, q, b, sh);
    } else if (MO >= 2 && b < g.first_block[2]) {
        reflect_block_tiled<(MO >= 2 ? 2 : 1)>(sc, g.source[1], g.count[1], 1, tx, rt, r0, r1, tiles_per_block, q,
                                               b - g.first_block[1], sh);
    } else if (MO >= 3 && b < g.first_block[3]) {
        reflect_block_tiled<(MO >= 3 ? 3 : 1)>(sc, g.source[2], g.count[2], 2, tx, rt, r0, r1, tiles_per_block, q,
                                               b - g.first_block[2], sh);
    } else if (MO >= 4) {
        reflect_block_tiled<(MO >= 4 ? 4 : 1)>(sc, g.source[3], g.count[3], 3, tx, rt, r0, r1, tiles_per_block, q,
                                               b - g.first_block[3], sh);
    }
}

// ---- receivers into tiles: cell of a uniform grid over the scene (Morton order), counting sort, boxes ------------
__device__ __forceinline__ uint32_t morton_spread(uint32_t v) {
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__device__ __forceinline__ uint32_t rx_cell(const double *__restrict__ rx, int64_t i, double x0, double y0, double inv,
                                            int g_side) {
    const double fx = (rx[3 * i] - x0) * inv, fy = (rx[3 * i + 1] - y0) * inv;
    const int cx = fx > 0.0 ? (fx < (double)(g_side - 1) ? (int)fx : g_side - 1) : 0;  // also NaN -> 0
    const int cy = fy > 0.0 ? (fy < (double)(g_side - 1) ? (int)fy : g_side - 1) : 0;
    return morton_spread((uint32_t)cx) | (morton_spread((uint32_t)cy) << 1);
}
__global__ void k_rx_count(const double *__restrict__ rx, int64_t n, double x0, double y0, double inv, int g_side,
                           int32_t *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[rx_cell(rx, i, x0, y0, inv, g_side)], 1);
}
__global__ void k_rx_scatter(const double *__restrict__ rx, int64_t n, double x0, double y0, double inv, int g_side,
                             const int64_t *__restrict__ cell_off, int32_t *__restrict__ cursor,
                             double *__restrict__ rxs, int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t cell = rx_cell(rx, i, x0, y0, inv, g_side);
    const int64_t pos = cell_off[cell] + atomicAdd(&cursor[cell], 1);
    rxs[3 * pos] = rx[3 * i];
    rxs[3 * pos + 1] = rx[3 * i + 1];
    rxs[3 * pos + 2] = rx[3 * i + 2];
    perm[pos] = (int32_t)i;
}
__global__ void k_rx_boxes(const double *__restrict__ rxs, int64_t n, double *__restrict__ boxes) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t * RX_TILE >= n) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < RX_TILE && t * RX_TILE + j < n; ++j)
        for (int a = 0; a < 3; ++a) {
            const double v = rxs[3 * (t * RX_TILE + j) + a];
            lo[a] = fmin(lo[a], v);
            hi[a] = fmax(hi[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        boxes[6 * t + a] = 0.5 * (lo[a] + hi[a]);
        // half extent rounded up so that the box holds its receivers whatever the rounding of the centre
        boxes[6 * t + 3 + a] = 0.5 * (hi[a] - lo[a]) * (1.0 + 1e-12) + 1e-12 * (fabs(lo[a]) + fabs(hi[a]));
    }
}

// The three steps above in ONE launch of one workgroup, for batches whose grid fits LDS (up to 64 x 64 cells, i.e. 16 384
// receivers): four launches of a few microseconds each cost 75 us with the gaps between them, and every kernel of the
// solve waits for the tiles.
#define RX_TILES_ONE_LAUNCH_CELLS 4096
__global__ __launch_bounds__(1024) void k_rx_tiles(const double *__restrict__ rx, int n, double x0, double y0, double inv,
                                                   int g_side, double *__restrict__ rxs, int32_t *__restrict__ perm,
                                                   double *__restrict__ boxes) {
    __shared__ int32_t cell_at[RX_TILES_ONE_LAUNCH_CELLS];  // count of a cell, then the place of its next receiver
    __shared__ int32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_cells = g_side * g_side;
    for (int c = tid; c < n_cells; c += 1024) cell_at[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&cell_at[rx_cell(rx, i, x0, y0, inv, g_side)], 1);
    __syncthreads();
    // exclusive scan: every thread owns a run of neighbouring cells
    const int per = (n_cells + 1023) / 1024;
    const int c0 = tid * per, c1 = min(n_cells, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += cell_at[c];
    int x = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (wave == 0) {
        int v = lane < 16 ? wsum[lane] : 0;
        for (int off = 1; off < 16; off <<= 1) {
            const int y = __shfl_up(v, off);
            if (lane >= off) v += y;
        }
        if (lane < 16) wsum[lane] = v;
    }
    __syncthreads();
    int at = (wave > 0 ? wsum[wave - 1] : 0) + x - mine;
    for (int c = c0; c < c1; ++c) {
        const int k = cell_at[c];
        cell_at[c] = at;
        at += k;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int pos = atomicAdd(&cell_at[rx_cell(rx, i, x0, y0, inv, g_side)], 1);
        rxs[3 * pos] = rx[3 * i];
        rxs[3 * pos + 1] = rx[3 * i + 1];
        rxs[3 * pos + 2] = rx[3 * i + 2];
        perm[pos] = i;
    }
    __threadfence();
    __syncthreads();
    for (int t = tid; t * RX_TILE < n; t += 1024) {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j = 0; j < RX_TILE && t * RX_TILE + j < n; ++j)
            for (int a = 0; a < 3; ++a) {
                const double v = __builtin_nontemporal_load(rxs + 3 * (t * RX_TILE + j) + a);
                lo[a] = fmin(lo[a], v);
                hi[a] = fmax(hi[a], v);
            }
        for (int a = 0; a < 3; ++a) {
            boxes[6 * t + a] = 0.5 * (lo[a] + hi[a]);
            boxes[6 * t + 3 + a] = 0.5 * (hi[a] - lo[a]) * (1.0 + 1e-12) + 1e-12 * (fabs(lo[a]) + fabs(hi[a]));
        }
    }
}

// Receiver-independent half of an (M reflections + diffraction) candidate: reflection points aimed at the edge's first
// endpoint, their containment and spacing checks.  Candidates that pass go to `alive` with their points and their M
// receiver-independent segments.
struct AliveRec {
    int32_t cand;
    int32_t pad;
    double pts[(RT_MAX_ORDER - 1) * 3];
};

template <int M>
__global__ __launch_bounds__(256) void k_rd_prepare(SceneView sc, const int32_t *__restrict__ chains,
                                                     const int32_t *__restrict__ edge_ids, int64_t count,
                                                     const double *__restrict__ edges, d3 tx, AliveRec *alive,
                                                     double *segs, int64_t cap, unsigned long long *counter) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    d3 nrm[M], img[M];
    double g[M];
    int32_t poly[M];
    d3 prev = tx;
    for (int i = 0; i < M; ++i) {
        poly[i] = chains[c * M + i];
        const double *pl = sc.planes + 4 * (size_t)poly[i];
        nrm[i] = ld3(pl);
        const double s = pl[3] + dot(prev, nrm[i]);
        img[i] = prev - (2.0 * s) * nrm[i];
        g[i] = -s;
        prev = img[i];
    }
    const d3 P0 = ld3(edges + 6 * (size_t)edge_ids[c]);
    d3 X[M];
    if (!unfold_exact<M>(nrm, g, img, P0, X)) return;
    for (int i = 0; i < M; ++i)
        if (!polygon_contains_point(sc, poly[i], X[i], true, 1e-8)) return;
    d3 a = tx;
    for (int i = 0; i < M; ++i) {
        if (!(norm3(a - X[i]) >= MIN_DISTANCE_TOL)) return;
        a = X[i];
    }
    const int64_t slot = (int64_t)atomicAdd(counter, 1ull);
    if (slot >= cap) return;
    alive[slot].cand = (int32_t)c;
    a = tx;
    for (int i = 0; i < M; ++i) {
        st3(alive[slot].pts + 3 * i, X[i]);
        st3(segs + 6 * (slot * M + i), a);
        st3(segs + 6 * (slot * M + i) + 3, X[i]);
        a = X[i];
    }
}

// keeps the alive records none of whose M receiver-independent segments is obstructed
template <int M>
__global__ void k_rd_filter(const AliveRec *__restrict__ in, const uint8_t *__restrict__ hit, int64_t n, AliveRec *out,
                            unsigned long long *counter) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < M; ++k)
        if (hit[i * M + k]) return;
    const int64_t slot = (int64_t)atomicAdd(counter, 1ull);
    out[slot] = in[i];
}

// Receiver-independent part of one diffraction candidate: the edge-aligned frame (geometry.py:736-742), the edge and
// the point before it (emitter or last reflection point) in that frame.
struct DiffSetup {
    double fx[3], fy[3], fz[3];
    double xe, ye, ze, z1;
    double ax, ay, az, rhoA;
    double A[3];
    double n1[3], n2[3];  // normals of the two faces that meet in the edge
    int32_t cand, src;    // candidate index in its group; alive record (M > 0)
    int32_t bld, cell;    // the building both faces belong to when it is one building of at most DIFF_OWN_FACES faces, else -1;
                          // the grid cell of the edge's middle
};

#define DIFF_OWN_FACES 48
template <int M>
__global__ void k_diffract_setup(SceneView sc, const AliveRec *__restrict__ alive, const int32_t *__restrict__ edge_ids,
                                 int64_t count, const double *__restrict__ edges, const int32_t *__restrict__ edge_pairs,
                                 d3 tx, DiffSetup *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    int32_t cand = (int32_t)idx;
    d3 A = tx;
    if (M > 0) {
        cand = alive[idx].cand;
        A = ld3(alive[idx].pts + 3 * (M - 1));
    }
    const double *e = edges + 6 * (size_t)edge_ids[cand];
    const d3 P0 = ld3(e), P1 = ld3(e + 3);
    d3 fx, fy, fz;
    frame_from_vector(P1 - P0, fx, fy, fz);
    DiffSetup S;
    st3(S.fx, fx);
    st3(S.fy, fy);
    st3(S.fz, fz);
    S.xe = dot(P0, fx);
    S.ye = dot(P0, fy);
    S.ze = dot(P0, fz);
    S.z1 = dot(P1, fz);
    S.ax = dot(A, fx);
    S.ay = dot(A, fy);
    S.az = dot(A, fz);
    S.rhoA = sqrt((S.xe - S.ax) * (S.xe - S.ax) + (S.ye - S.ay) * (S.ye - S.ay));
    st3(S.A, A);
    const int p1 = edge_pairs[2 * (size_t)edge_ids[cand]], p2 = edge_pairs[2 * (size_t)edge_ids[cand] + 1];
    st3(S.n1, ld3(sc.planes + 4 * (size_t)p1));
    st3(S.n2, ld3(sc.planes + 4 * (size_t)p2));
    S.bld = -1;
    {
        const d3 mid = 0.5 * (P0 + P1);
        int ci = (int)floor((mid.x - sc.gx0) * sc.ginv), cj = (int)floor((mid.y - sc.gy0) * sc.ginv);
        ci = ci < 0 ? 0 : (ci >= sc.gnx ? sc.gnx - 1 : ci);
        cj = cj < 0 ? 0 : (cj >= sc.gny ? sc.gny - 1 : cj);
        S.cell = cj * sc.gnx + ci;
    }
    if (sc.n_bld > 0 && sc.bld_of[p1] >= 0 && sc.bld_of[p1] == sc.bld_of[p2]) {
        const int b = sc.bld_of[p1];
        if (sc.bld_off[b + 1] - sc.bld_off[b] <= DIFF_OWN_FACES) S.bld = b;
    }
    S.cand = cand;
    S.src = (int32_t)idx;
    out[idx] = S;
}

// The diffraction point of one (candidate, receiver) pair and its receiver-dependent checks.
__device__ __forceinline__ bool keller_point(const DiffSetup &S, d3 R, d3 &D) {
    const d3 fx = ld3(S.fx), fy = ld3(S.fy), fz = ld3(S.fz);
    const double bx = dot(R, fx), by = dot(R, fy), bz = dot(R, fz);
    const double rhoB = sqrt((S.xe - bx) * (S.xe - bx) + (S.ye - by) * (S.ye - by));
    const double sum = S.rhoA + rhoB;
    // Keller cone: (z - zA)/|X - A| = (zB - z)/|B - X|  <=>  z = zA + (zB - zA) rhoA/(rhoA + rhoB)
    const double z = S.az + (bz - S.az) * (S.rhoA / sum);
    if (!(sum > 0.0 && z >= S.ze - 1e-6 && z <= S.z1 + 1e-6)) return false;
    // back to world coordinates (geometry.py:746-748, 831)
    D = (S.xe * fx + S.ye * fy) + z * fz;
    // point_on_edge_ (geometry.py:470-479)
    const double tol = 1e-8;
    const double dx = dot(D, fx), dy = dot(D, fy), dz = dot(D, fz);
    if (!(S.xe - tol <= dx && dx <= S.xe + tol && S.ye - tol <= dy && dy <= S.ye + tol && S.ze - tol <= dz &&
          dz <= S.z1 + tol))
        return false;
    return norm3(ld3(S.A) - D) >= MIN_DISTANCE_TOL;  // ray_tracing.py:268-272 on lines[:-1]
}

// Diffraction candidates: a block takes DIFF_CHUNK candidates and 256 receivers, lane = receiver, every lane of a wave
// on the same candidate.  The pairs that survive are noted in LDS; the block then reserves room for all of them with
// one pair of atomics and writes their records candidate by candidate, receiver by receiver: the segments leave in
// fans (same edge point, neighbouring receivers), which is the order the obstruction kernel likes, first all
// "point before -> edge" segments of a wave, then all "edge -> receiver" ones.
#define DIFF_CHUNK 32
template <int M>
__global__ __launch_bounds__(256) void k_diffract(SceneView sc, const DiffSetup *__restrict__ setup,
                                                   const AliveRec *__restrict__ alive, int64_t count, int group,
                                                   const double *__restrict__ rx, const int32_t *__restrict__ perm,
                                                   int64_t rx0, int64_t rx1, Queues q, int chunk,
                                                   const int32_t *__restrict__ hints, int hint_bins) {
    // chunk <= DIFF_CHUNK candidates per block: fewer when the launch would otherwise have too few blocks for the chip
    __shared__ DiffSetup s_setup[DIFF_CHUNK];
    __shared__ uint16_t s_list[4][DIFF_CHUNK * 64];
    __shared__ int s_count[4];
    __shared__ long long s_base[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c_begin = (int64_t)blockIdx.x * chunk;
    const int n_c = (int)(count - c_begin < chunk ? count - c_begin : chunk);
    {
        const double *src = (const double *)(setup + c_begin);
        double *dst = (double *)s_setup;
        const int n_words = n_c * (int)(sizeof(DiffSetup) / sizeof(double));
        for (int i = threadIdx.x; i < n_words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t r_wave = rx0 + (int64_t)blockIdx.y * 256 + wave * 64;
    const int64_t r = r_wave + lane;
    const bool active = r < rx1;
    const d3 R = ld3(rx + 3 * (active ? r : rx0));
    uint16_t *list = s_list[wave];
    int n = 0;  // wave-uniform
    // Which of the block's candidates can reach any receiver of this wave at all?  Lane c puts candidate c's Keller
    // condition to the box around the wave's receivers (interval arithmetic, widened to err towards yes); when the
    // receivers come in tile order the box is a small patch and most candidates drop out here.
    unsigned long long reach = 0;
    {
        double lo[3], hi[3];
        const double rv[3] = {R.x, R.y, R.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = active ? rv[a] : INFINITY;
            hi[a] = active ? rv[a] : -INFINITY;
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fmin(lo[a], __shfl_xor(lo[a], off));
                hi[a] = fmax(hi[a], __shfl_xor(hi[a], off));
            }
        }
        bool may = false;
        if (lane < n_c && lo[0] <= hi[0]) {
            const DiffSetup &S = s_setup[lane];
            const d3 ctr = mk3(0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]));
            const d3 ext = mk3(0.5 * (hi[0] - lo[0]), 0.5 * (hi[1] - lo[1]), 0.5 * (hi[2] - lo[2]));
            const d3 fx = ld3(S.fx), fy = ld3(S.fy), fz = ld3(S.fz);
            const double cx = dot(ctr, fx), cy = dot(ctr, fy), cz = dot(ctr, fz);
            const double rx_ = ext.x * fabs(fx.x) + ext.y * fabs(fx.y) + ext.z * fabs(fx.z);
            const double ry_ = ext.x * fabs(fy.x) + ext.y * fabs(fy.y) + ext.z * fabs(fy.z);
            const double rz_ = ext.x * fabs(fz.x) + ext.y * fabs(fz.y) + ext.z * fabs(fz.z);
            const double mag = fabs(cx) + fabs(cy) + fabs(cz) + rx_ + ry_ + rz_ + fabs(S.xe) + fabs(S.ye) + fabs(S.az) + 1.0;
            const double slack = 1e-9 * mag + 2e-6;
            // distance of the receivers from the edge's axis: between rho_lo and rho_hi
            const double rho_c = sqrt((S.xe - cx) * (S.xe - cx) + (S.ye - cy) * (S.ye - cy));
            const double rad = sqrt(rx_ * rx_ + ry_ * ry_) + slack;
            const double rho_lo = fmax(rho_c - rad, 0.0), rho_hi = rho_c + rad;
            if (!(S.rhoA + rho_lo > 0.0)) {
                may = true;
            } else {
                // z = zA + (zB - zA) w with w = rhoA / (rhoA + rhoB) in [w_lo, w_hi] (0 <= w <= 1), zB - zA in [d_lo, d_hi]
                const double w_lo = S.rhoA / (S.rhoA + rho_hi) * (1.0 - 1e-12), w_hi = fmin(S.rhoA / (S.rhoA + rho_lo) * (1.0 + 1e-12), 1.0);
                const double d_lo = cz - rz_ - S.az - slack, d_hi = cz + rz_ - S.az + slack;
                const double z_lo = S.az + fmin(d_lo * w_lo, d_lo * w_hi), z_hi = S.az + fmax(d_hi * w_lo, d_hi * w_hi);
                may = z_hi + slack >= S.ze && z_lo - slack <= S.z1;
            }
        }
        reach = __ballot(may);
    }
    for (unsigned long long todo = reach; todo != 0; todo &= todo - 1) {
        const int c = __ffsll((long long)todo) - 1;
        d3 D;
        bool ok = active && keller_point(s_setup[c], R, D);
        // the wall next to the receiver in the direction of the edge point (k_rx_hints).  With it, the two tests below
        // (which are for batches without hints) cost more than they save: lln step 5.69 ms with them, 5.49 without
        if (hints && ok && hinted_wall_cuts(sc, hints, hint_bins, r, D, R)) ok = false;
        // Four pairs in ten leave the edge INTO the building the edge belongs to (the receiver lies behind both of the
        // edge's faces).  Such a last segment is cut by another face of that building, which the obstruction walk
        // would find in its first cells - after the path has been written to the queue and read back.  The building's
        // faces are tested here instead, with the walk's own tests: a pair that one of them cuts is dropped, every
        // other pair goes on as before.  (All lanes of the wave are on the same edge, hence the same faces.)
        const int own = hints ? -1 : s_setup[c].bld;
        if (own >= 0) {
            const d3 ray = R - D;
            const bool inward = ok && dot(ray, ld3(s_setup[c].n1)) < 0.0 && dot(ray, ld3(s_setup[c].n2)) < 0.0;
            if (__any(inward)) {
                const double pad = 1e-5 * fmax(fabs(ray.x), fmax(fabs(ray.y), fabs(ray.z))) + 1e-6;
                const int f_end = sc.bld_off[own + 1];
                for (int fi = sc.bld_off[own]; fi < f_end && __any(inward && ok); ++fi) {
                    const int p = sc.bld_polys[fi];
                    if (inward && ok && occ_pretest(sc.poly_recs + p, sc, D, ray, pad) && polygon_cuts_segment(sc, p, D, R))
                        ok = false;
                }
            }
        }
        // the polygons registered in the grid cell of the edge's middle: what the walk would look at first
        if (!hints && __any(ok)) {
            const OccCell cell = sc.cells[s_setup[c].cell];
            const d3 ray = R - D;
            const double pad = 1e-5 * fmax(fabs(ray.x), fmax(fabs(ray.y), fabs(ray.z))) + 1e-6;
            const int k_end = cell.end - cell.begin > 64 ? cell.begin + 64 : cell.end;
            for (int kk = cell.begin; kk < k_end && __any(ok); ++kk) {
                const OccRec *r = sc.cell_recs + kk;
                if (ok && occ_pretest(r, sc, D, ray, pad) && polygon_cuts_segment(sc, r->poly, D, R)) ok = false;
            }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) list[n + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)((c << 6) | lane);
        n += __popcll(m);
    }
    if (lane == 0) s_count[wave] = n;
    __syncthreads();
    const int total = s_count[0] + s_count[1] + s_count[2] + s_count[3];
    if (total == 0) return;
    if (threadIdx.x == 0) {
        s_base[0] = q.base + (long long)atomicAdd(&q.counters[0], (unsigned long long)total);
        s_base[1] = q.base + (long long)atomicAdd(&q.counters[1], 2ull * (unsigned long long)total);
    }
    __syncthreads();
    if (s_base[0] + total > q.path_cap || s_base[1] + 2 * (long long)total > q.seg_cap) return;  // host redoes the chunk
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_count[w];
    const int64_t p0 = s_base[0] + before, s0 = s_base[1] + 2 * (int64_t)before;
    for (int j = lane; j < n; j += 64) {
        const unsigned e = list[j];
        const DiffSetup &S = s_setup[e >> 6];
        const int64_t rr = r_wave + (e & 63u);
        const d3 RR = ld3(rx + 3 * rr);
        d3 D;
        (void)keller_point(S, RR, D);
        PathRec &rec = q.paths[p0 + j];
        rec.rx = perm ? perm[rr] : (int32_t)rr;  // receivers may come in tile order
        rec.group = group;
        rec.cand = S.cand;
        rec.nseg = 2;
        rec.seg0 = s0 + j;
        rec.seg_stride = n;
        rec.npts = M + 1;
        clear_hits(q, p0 + j, s0 + j, n, 2);
        if (M > 0) {
            const AliveRec &al = alive[c_begin + (e >> 6)];
            for (int i = 0; i < M; ++i) st3(rec.pts + 3 * i, ld3(al.pts + 3 * i));
        }
        st3(rec.pts + 3 * M, D);
        if (q.segs) {
            st3(q.segs + 6 * (s0 + j), ld3(S.A));
            st3(q.segs + 6 * (s0 + j) + 3, D);
            st3(q.segs + 6 * (s0 + n + j), D);
            st3(q.segs + 6 * (s0 + n + j) + 3, RR);
        }
    }
}

// Lines of sight on their own, for a solve without reflection candidates (otherwise k_reflect queues them).
__global__ void k_los_emit(d3 tx, const double *__restrict__ rx, int64_t rx0, int64_t n, Queues q) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) los_emit_one(tx, rx, rx0, i, q);
}

struct OutArrays {
    int32_t *rx, *group, *cand;
    double *pts;
    int64_t cap;
    unsigned long long *count;
};

// Keeps the queued paths none of whose segments is obstructed.  base + counters[0], base + counters[1] = paths and
// segments the producers asked for; when either exceeds its queue the records are incomplete and nothing is done (the host redoes
// the chunk with larger queues).  hit holds one flag per segment, or (per_path) one per path.  A wave looks at 512
// paths and reserves the room of its survivors with one atomic.
// FINALIZE_ITEMS = 8 paths per lane for large batches (few atomics), 1 for small ones (more waves, shorter chains).
// The block that finishes last copies the counts to host_counters (host memory the device can write), so the host
// needs no copy after the launch, and leaves the queue counters and the count of finished blocks (counters[3]) at
// zero for the next launch (the result counter too after a solve's last chunk): a solve starts without a memset.
template <int FINALIZE_ITEMS>
__global__ __launch_bounds__(256) void k_finalize(const PathRec *__restrict__ paths, const uint8_t *__restrict__ hit,
                                                   unsigned long long *counters, int64_t base, int64_t path_cap,
                                                   int64_t seg_cap, int per_path, OutArrays out,
                                                   unsigned long long *host_counters, int last_chunk) {
    const unsigned long long np = counters[0] + (unsigned long long)base, ns = counters[1] + (unsigned long long)base;
    const bool fits = np <= (unsigned long long)path_cap && ns <= (unsigned long long)seg_cap;
    const int64_t n = fits ? (int64_t)np : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One reservation per block and trip (the four waves' survivors added up in LDS): the result counter is one word,
    // and reservations on one word go one after the other at about 7 ns each -- a reservation per wave was most of
    // what this kernel took on the lln map (33 000 of them, 0.22 ms).
    __shared__ int s_total[4];
    __shared__ unsigned long long s_slot0;
    const int64_t per_block = 256 * FINALIZE_ITEMS;
    for (int64_t block_base = (int64_t)blockIdx.x * per_block; block_base < n; block_base += (int64_t)gridDim.x * per_block) {
        const int64_t base = block_base + (int64_t)wave * 64 * FINALIZE_ITEMS;
        bool keep[FINALIZE_ITEMS];
        int before[FINALIZE_ITEMS];
        int total = 0;
        // a lane's items lie one after the other, so that with one flag per path its eight flags are one 8-byte read
        // (read byte by byte, 64 bytes per wave and instruction, the flags of 17 M paths took 0.2 ms)
        const int64_t first = base + (int64_t)lane * FINALIZE_ITEMS;
        unsigned long long flags8 = 0;
        const bool wide = FINALIZE_ITEMS == 8 && per_path && first + 8 <= n;
        if (wide) flags8 = *(const unsigned long long *)(hit + first);  // (hit and base are multiples of 8)
#pragma unroll
        for (int j = 0; j < FINALIZE_ITEMS; ++j) {
            const int64_t i = first + j;
            bool ok = i < n;
            if (ok) {
                if (wide) {
                    ok = ((flags8 >> (8 * j)) & 0xffull) == 0;
                } else if (per_path) {
                    ok = !hit[i];
                } else {
                    const PathRec &p = paths[i];
                    for (int k = 0; k < p.nseg; ++k) ok = ok && !hit[p.seg0 + k * p.seg_stride];
                }
            }
            keep[j] = ok;
            const unsigned long long m = __ballot(ok);
            before[j] = total + __popcll(m & ((1ull << lane) - 1));
            total += __popcll(m);
        }
        if (lane == 0) s_total[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int all = s_total[0] + s_total[1] + s_total[2] + s_total[3];
            s_slot0 = all ? atomicAdd(out.count, (unsigned long long)all) : 0ull;
        }
        __syncthreads();
        unsigned long long slot0 = s_slot0;
        for (int w = 0; w < wave; ++w) slot0 += (unsigned long long)s_total[w];
        __syncthreads();  // (before the next trip writes the two again)
#pragma unroll
        for (int j = 0; j < FINALIZE_ITEMS; ++j) {
            if (!keep[j]) continue;
            const int64_t slot = (int64_t)slot0 + before[j];
            if (slot >= out.cap) continue;
            const PathRec &p = paths[first + j];
            out.rx[slot] = p.rx;
            out.group[slot] = p.group;
            out.cand[slot] = p.cand;
            // (the slots past the path's own points are written as zeros: a record of the queue keeps what an earlier
            // solve left there)
            for (int k = 0; k < RT_MAX_ORDER * 3; ++k)
                out.pts[slot * RT_MAX_ORDER * 3 + k] = k < 3 * p.npts ? p.pts[k] : 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (no fence in front of this count: what the last block reads below are counters that are only ever touched by
        // atomic operations, each of which had returned its value to this block before the block got here, and the
        // path arrays written above are read by nobody before the kernel has ended.  A device-wide fence writes the
        // compute die's L2 back; one per block, 4 096 of them, was 0.2 ms of this kernel's 0.23.)
        const unsigned long long done = atomicAdd(&counters[3], 1ull);
        if (done == (unsigned long long)gridDim.x - 1) {
            __threadfence();
            host_counters[0] = np;
            host_counters[1] = ns;
            host_counters[2] = atomicAdd(&counters[2], 0ull);
            host_counters[3] = counters[8];  // paths listed as clear after the first obstruction round
            counters[8] = 0;
            counters[0] = 0;
            counters[1] = 0;
            counters[3] = 0;
            if (fits && last_chunk) counters[2] = 0;
            __threadfence_system();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// solve host side
// ---------------------------------------------------------------------------------------------------------------

int rt_rejudge_paths(rt_scene *scene, const rt_candidates *cand, const double *tx, const double *d_rx, int32_t *d_out_rx,
                     int32_t *d_out_group, int32_t *d_out_cand, double *d_out_pts, int64_t n, int64_t *n_kept);

struct EventTimer {
    rt_scene *sc;
    std::vector<hipEvent_t> ev[4];
    explicit EventTimer(rt_scene *s) : sc(s) { sc->events_used = 0; }
    hipEvent_t take() {
        if (sc->events_used == sc->event_pool.size()) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            sc->event_pool.push_back(e);
        }
        return sc->event_pool[sc->events_used++];
    }
    void begin(int fam) {
        if (!sc->timing) return;
        hipEvent_t a = take(), b = take();
        (void)hipEventRecord(a, sc->stream);
        ev[fam].push_back(a);
        ev[fam].push_back(b);
    }
    void end(int fam) {
        if (sc->timing) (void)hipEventRecord(ev[fam].back(), sc->stream);
    }
    // after the stream has been synchronised
    void collect(double *ms, int64_t *launches) {
        for (int f = 0; f < 4; ++f) {
            ms[f] = 0;
            launches[f] = (int64_t)ev[f].size() / 2;
            for (size_t i = 0; i + 1 < ev[f].size(); i += 2) {
                float t = 0;
                if (hipEventElapsedTime(&t, ev[f][i], ev[f][i + 1]) == hipSuccess) ms[f] += t;
            }
            ev[f].clear();
        }
        sc->events_used = 0;
    }
};

// false when there is no reflection candidate (nothing launched: the caller queues the lines of sight itself)
static bool launch_reflect(rt_scene *s, const rt_candidates *cand, d3 tx, const double *rx, int64_t rx0, int64_t rx1,
                           int rx_tile, const Queues &q) {
    const int mo = cand->max_order;
    ReflectGroups g{};
    unsigned blocks = 0;
    int top = 0;  // highest order that has candidates
    for (int k = 1; k <= RT_MAX_ORDER; ++k) {
        g.first_block[k - 1] = blocks;
        if (k <= mo && cand->groups[k - 1].count > 0) {
            g.source[k - 1] = cand->groups[k - 1].source;
            g.count[k - 1] = cand->groups[k - 1].count;
            blocks += blocks_for(g.count[k - 1], 256);
            top = k;
        }
    }
    g.first_block[RT_MAX_ORDER] = blocks;
    if (blocks == 0) return false;
    dim3 grid(blocks, blocks_for(rx1 - rx0, rx_tile));
    switch (top) {
        case 1: hipLaunchKernelGGL(k_reflect<1>, grid, dim3(256), 0, s->stream, s->view, g, tx, rx, rx0, rx1, rx_tile, q); break;
        case 2: hipLaunchKernelGGL(k_reflect<2>, grid, dim3(256), 0, s->stream, s->view, g, tx, rx, rx0, rx1, rx_tile, q); break;
        case 3: hipLaunchKernelGGL(k_reflect<3>, grid, dim3(256), 0, s->stream, s->view, g, tx, rx, rx0, rx1, rx_tile, q); break;
        default: hipLaunchKernelGGL(k_reflect<4>, grid, dim3(256), 0, s->stream, s->view, g, tx, rx, rx0, rx1, rx_tile, q); break;
    }
    return true;
}

static bool launch_reflect_tiled(rt_scene *s, const rt_candidates *cand, d3 tx, const RxTiles &rt, int64_t r0, int64_t r1,
                                 const Queues &q) {
    const int mo = cand->max_order;
    ReflectGroups g{};
    unsigned blocks = 0;
    int top = 0;
    for (int k = 1; k <= RT_MAX_ORDER; ++k) {
        g.first_block[k - 1] = blocks;
        if (k <= mo && cand->groups[k - 1].count > 0) {
            g.source[k - 1] = cand->groups[k - 1].source;
            g.count[k - 1] = cand->groups[k - 1].count;
            blocks += blocks_for(g.count[k - 1], 256);
            top = k;
        }
    }
    g.first_block[RT_MAX_ORDER] = blocks;
    if (blocks == 0) return false;
    // tiles per block: enough to spread the set-up of a block's candidates, few enough for some thousand blocks
    const int64_t n_tiles = (r1 - r0 + RX_TILE - 1) / RX_TILE;
    int64_t per_block = n_tiles * (int64_t)blocks / 8192;
    per_block = per_block < 4 ? 4 : (per_block > 64 ? 64 : per_block);
    per_block = per_block > n_tiles ? n_tiles : per_block;
    dim3 grid(blocks, blocks_for(n_tiles, (int)per_block));
    const int tpb = (int)per_block;
    switch (top) {
        case 1: hipLaunchKernelGGL(k_reflect_tiled<1>, grid, dim3(256), 0, s->stream, s->view, g, tx, rt, r0, r1, tpb, q); break;
        case 2: hipLaunchKernelGGL(k_reflect_tiled<2>, grid, dim3(256), 0, s->stream, s->view, g, tx, rt, r0, r1, tpb, q); break;
        case 3: hipLaunchKernelGGL(k_reflect_tiled<3>, grid, dim3(256), 0, s->stream, s->view, g, tx, rt, r0, r1, tpb, q); break;
        default: hipLaunchKernelGGL(k_reflect_tiled<4>, grid, dim3(256), 0, s->stream, s->view, g, tx, rt, r0, r1, tpb, q); break;
    }
    return true;
}

template <int M>
static void launch_diffract(rt_scene *s, const DiffSetup *setup, const AliveRec *alive, int64_t count, int group,
                            const double *rx, const int32_t *perm, int64_t rx0, int64_t rx1, const Queues &q,
                            const int32_t *hints, int hint_bins) {
    int chunk = DIFF_CHUNK;
    while (chunk > 4 && blocks_for(count, chunk) * blocks_for(rx1 - rx0, 256) < 8192) chunk >>= 1;
    dim3 grid(blocks_for(count, chunk), blocks_for(rx1 - rx0, 256));
    hipLaunchKernelGGL(k_diffract<M>, grid, dim3(256), 0, s->stream, s->view, setup, alive, count, group, rx, perm, rx0, rx1, q,
                       chunk, hints, hint_bins);
}

template <int M>
static void launch_diffract_setup(rt_scene *s, const AliveRec *alive, const int32_t *edge_ids, int64_t count,
                                  const double *edges, const int32_t *edge_pairs, d3 tx, DiffSetup *out) {
    hipLaunchKernelGGL(k_diffract_setup<M>, dim3(blocks_for(count, 256)), dim3(256), 0, s->stream, s->view, alive, edge_ids,
                       count, edges, edge_pairs, tx, out);
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int rt_solve_paths(rt_scene *scene, rt_candidates *cand, const double *tx, const double *rx, int64_t n_rx,
                              int32_t *out_rx, int32_t *out_group, int32_t *out_cand, double *out_points,
                              int64_t capacity, int64_t *n_found, int64_t *n_attempted, int on_device) {
    if (!scene || !cand || !tx || !n_found || !n_attempted) RT_FAIL("rt_solve_paths: null argument");
    if (cand->scene != scene) RT_FAIL("rt_solve_paths: candidates belong to another scene");
    if (n_rx < 0 || capacity < 0) RT_FAIL("rt_solve_paths: negative size");
    if (n_rx > 0 && !rx) RT_FAIL("rt_solve_paths: null receivers");
    if (capacity > 0 && (!out_rx || !out_group || !out_cand || !out_points)) RT_FAIL("rt_solve_paths: null output");
    if (n_rx > 0x7fffffffLL) RT_FAIL("rt_solve_paths: too many receivers");
    RT_HIP(rt_use_device(scene->device));
    hipStream_t st = scene->stream;
    const d3 TX = ld3(tx);
    const int n = scene->view.n;
    (void)n;

    int64_t total_cand = 0;
    for (const CandGroup &g : cand->groups) total_cand += g.count;
    *n_attempted = n_rx * (1 + total_cand);
    *n_found = 0;
    for (int f = 0; f < 4; ++f) {
        scene->last_ms[f] = 0;
        scene->last_launches[f] = 0;
    }
    scene->last_segments = 0;
    scene->last_rejudge_ms = 0;
    scene->last_rejudge_in = scene->last_rejudge_kept = 0;
    if (n_rx == 0) return 0;

    // device views of the caller's arrays
    const double *d_rx = rx;
    OutArrays out{};
    // memory of this call only: from the scene's pool (no allocation once the pool holds the sizes of this workload)
    auto temp_alloc = [&](size_t bytes) -> void * { return rt_scene_temp(scene, bytes); };
    auto free_temps = [&]() {
        (void)hipStreamSynchronize(st);  // nothing queued may still be using them
        if (scene->side_stream) (void)hipStreamSynchronize(scene->side_stream);
        rt_scene_temp_release_all(scene);
    };
#define SOLVE_HIP(call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            rt_set_error(std::string("rt_solve_paths: ") + #call + ": " + hipGetErrorString(e_));         \
            free_temps();                                                                                 \
            return -1;                                                                                    \
        }                                                                                                 \
    } while (0)
#define SOLVE_PTR(p)                                                   \
    if (!(p)) {                                                        \
        rt_set_error("rt_solve_paths: out of device memory");          \
        free_temps();                                                  \
        return -1;                                                     \
    }

    unsigned long long *d_counters = scene->d_counters;  // 0 paths, 1 segments, 2 results, 4-5 pre-pass
    unsigned long long *h_counters = scene->h_counters;
    unsigned long long *h_counters_dev = nullptr;  // the same words as the device addresses them
    SOLVE_HIP(hipHostGetDevicePointer((void **)&h_counters_dev, h_counters, 0));
    // the final filter of a chunk leaves the queue counters at zero; only a solve that was cut short does not
    const bool counters_were_reset = scene->counters_dirty;
    if (scene->counters_dirty) SOLVE_HIP(hipMemsetAsync(d_counters, 0, sizeof(unsigned long long) * 16, st));
    if (on_device) {
        out.rx = out_rx;
        out.group = out_group;
        out.cand = out_cand;
        out.pts = out_points;
    } else {
        double *t = (double *)temp_alloc(sizeof(double) * 3 * n_rx);
        SOLVE_PTR(t);
        SOLVE_HIP(hipMemcpyAsync(t, rx, sizeof(double) * 3 * n_rx, hipMemcpyHostToDevice, st));
        d_rx = t;
        out.rx = (int32_t *)temp_alloc(sizeof(int32_t) * capacity);
        out.group = (int32_t *)temp_alloc(sizeof(int32_t) * capacity);
        out.cand = (int32_t *)temp_alloc(sizeof(int32_t) * capacity);
        out.pts = (double *)temp_alloc(sizeof(double) * RT_MAX_ORDER * 3 * capacity);
        SOLVE_PTR(out.rx);
        SOLVE_PTR(out.group);
        SOLVE_PTR(out.cand);
        SOLVE_PTR(out.pts);
    }
    out.cap = capacity;
    out.count = d_counters + 2;

    EventTimer timer(scene);
    // where the main stream stood when this solve began (the side stream's part waits for it)
    if (!scene->rejudge_event) SOLVE_HIP(hipEventCreateWithFlags(&scene->rejudge_event, hipEventDisableTiming));
    hipEvent_t entry_event = scene->rejudge_event;
    SOLVE_HIP(hipEventRecord(entry_event, st));

    const int mo = cand->max_order;
    std::vector<AliveRec *> alive(mo, nullptr);
    std::vector<int64_t> alive_count(mo, 0);
    std::vector<DiffSetup *> diff_setup(mo, nullptr);
    std::vector<int64_t> diff_count(mo, 0);
    bool any_diffraction = false;
    // The receiver-independent half of the diffraction groups (reflection points aimed at the edge, their segments
    // tested, edge frames) is independent of the reflection launch.  On a large solve it runs on the scene's side
    // stream (same priority) beside the first chunk's reflection kernel, which is queued first: its two host waits
    // then cost nothing, where they used to leave the GPU idle for a quarter of a millisecond at the start of every
    // solve.  The main stream waits for it, on the device, before the diffraction kernels.
    auto diffraction_head = [&]() -> int {
        // ---- receiver-independent half of the reflection+diffraction groups -------------------------------------
        for (int m = 1; m < mo; ++m) {
            const CandGroup &g = cand->groups[mo + m];
            if (g.count == 0) continue;
            AliveRec *a0 = (AliveRec *)temp_alloc(sizeof(AliveRec) * g.count);
            AliveRec *a1 = (AliveRec *)temp_alloc(sizeof(AliveRec) * g.count);
            double *segs = (double *)temp_alloc(sizeof(double) * 6 * g.count * m);
            uint8_t *hit = (uint8_t *)temp_alloc((size_t)g.count * m);
            SOLVE_PTR(a0);
            SOLVE_PTR(a1);
            SOLVE_PTR(segs);
            SOLVE_PTR(hit);
            unsigned long long *cnt = d_counters + 4;
            SOLVE_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), scene->stream));
            timer.begin(1);
            dim3 grid(blocks_for(g.count, 256));
            switch (m) {
                case 1: hipLaunchKernelGGL(k_rd_prepare<1>, grid, dim3(256), 0, scene->stream, scene->view, g.chains, g.edge_ids, g.count, cand->d_edges, TX, a0, segs, g.count, cnt); break;
                case 2: hipLaunchKernelGGL(k_rd_prepare<2>, grid, dim3(256), 0, scene->stream, scene->view, g.chains, g.edge_ids, g.count, cand->d_edges, TX, a0, segs, g.count, cnt); break;
                default: hipLaunchKernelGGL(k_rd_prepare<3>, grid, dim3(256), 0, scene->stream, scene->view, g.chains, g.edge_ids, g.count, cand->d_edges, TX, a0, segs, g.count, cnt); break;
            }
            timer.end(1);
            unsigned long long n0 = 0;
            SOLVE_HIP(hipMemcpyAsync(&n0, cnt, sizeof(n0), hipMemcpyDeviceToHost, scene->stream));
            SOLVE_HIP(hipStreamSynchronize(scene->stream));
            if (n0 == 0) continue;
            timer.begin(2);
            if (rt_launch_obstruct(scene, segs, (int64_t)n0 * m, nullptr, 0, hit, nullptr, 0, 0, nullptr, 0, true, nullptr, nullptr, nullptr)) {
                free_temps();
                return -1;
            }
            timer.end(2);
            scene->last_segments += (int64_t)n0 * m;
            timer.begin(3);
            switch (m) {
                case 1: hipLaunchKernelGGL(k_rd_filter<1>, dim3(blocks_for(n0, 256)), dim3(256), 0, scene->stream, a0, hit, (int64_t)n0, a1, cnt + 1); break;
                case 2: hipLaunchKernelGGL(k_rd_filter<2>, dim3(blocks_for(n0, 256)), dim3(256), 0, scene->stream, a0, hit, (int64_t)n0, a1, cnt + 1); break;
                default: hipLaunchKernelGGL(k_rd_filter<3>, dim3(blocks_for(n0, 256)), dim3(256), 0, scene->stream, a0, hit, (int64_t)n0, a1, cnt + 1); break;
            }
            timer.end(3);
            unsigned long long n1 = 0;
            SOLVE_HIP(hipMemcpyAsync(&n1, cnt + 1, sizeof(n1), hipMemcpyDeviceToHost, scene->stream));
            SOLVE_HIP(hipStreamSynchronize(scene->stream));
            alive[m] = a1;
            alive_count[m] = (int64_t)n1;
        }

        // ---- receiver-independent part of every diffraction candidate (edge frames) ------------------------------
        for (int m = 0; m < mo; ++m) {
            const CandGroup &g = cand->groups[mo + m];
            const int64_t cnt = m == 0 ? g.count : alive_count[m];
            diff_count[m] = cnt;
            if (cnt == 0) continue;
            any_diffraction = true;
            diff_setup[m] = (DiffSetup *)temp_alloc(sizeof(DiffSetup) * cnt);
            SOLVE_PTR(diff_setup[m]);
            timer.begin(1);
            switch (m) {
                case 0: launch_diffract_setup<0>(scene, nullptr, g.edge_ids, cnt, cand->d_edges, cand->d_edge_pairs, TX, diff_setup[m]); break;
                case 1: launch_diffract_setup<1>(scene, alive[m], g.edge_ids, cnt, cand->d_edges, cand->d_edge_pairs, TX, diff_setup[m]); break;
                case 2: launch_diffract_setup<2>(scene, alive[m], g.edge_ids, cnt, cand->d_edges, cand->d_edge_pairs, TX, diff_setup[m]); break;
                default: launch_diffract_setup<3>(scene, alive[m], g.edge_ids, cnt, cand->d_edges, cand->d_edge_pairs, TX, diff_setup[m]); break;
            }
            timer.end(1);
        }

        return 0;
    };
    // ---- receivers, in chunks whose survivors fit the workspace ---------------------------------------------
    // One host synchronisation per chunk: the kernels after the candidate kernels take their sizes from the device
    // counters (their grids are sized from what the previous call needed and stride over the rest), and the host
    // reads the counters once at the end of the chunk.
    int64_t chunk = n_rx;
    const int64_t max_pairs_per_launch = 1ll << 36;
    if (total_cand > 0 && chunk * total_cand > max_pairs_per_launch) chunk = std::max<int64_t>(1, max_pairs_per_launch / total_cand);
    int64_t path_cap = std::max<int64_t>(1 << 16, std::min<int64_t>(chunk * std::max<int64_t>(total_cand, 1) / 64 + 4096, 1ll << 26));
    path_cap = std::max<int64_t>(path_cap, scene->hint_paths + scene->hint_paths / 8);
    if (scene->queue_capacity_override > 0) path_cap = scene->queue_capacity_override;
    // receivers per block of a candidate kernel: fewer when the batch is small, so that the launch still has a few
    // hundred blocks (measured on 1 024 receivers x 2 936 candidates: 42 us with 4, 34 us with 32)
    int64_t cand_blocks = 0;
    for (const CandGroup &g : cand->groups) cand_blocks += blocks_for(g.count, 256);
    // (rounded up to the loop's four receivers per trip: rounded down, 1 024 receivers x 13 candidate blocks made 559
    // blocks, three on some compute units and two on the others, and the launch lasted as long as three)
    int rx_tile = (int)std::min<int64_t>(REFLECT_MAX_RX_TILE,
                                         std::max<int64_t>(4, (chunk * std::max<int64_t>(cand_blocks, 1) + 511) / 512));
    rx_tile = std::min(REFLECT_MAX_RX_TILE, (rx_tile + 3) & ~3);
    chunk = std::min<int64_t>(chunk, 65535ll * rx_tile);  // grid.y of the candidate kernels

    // Large batches: the candidate kernels work on receivers sorted into spatial tiles (the reflection kernel rejects
    // whole tiles; both queue the segments of a wave's receivers, now a compact patch, next to each other).
    const bool tiled = total_cand > 0 && (scene->reflect_mode == 2 ||
                                          (scene->reflect_mode == 0 && n_rx >= 128 && n_rx * total_cand >= (1ll << 28)));
    RxTiles tiles{};
    if (tiled) {
        // (a chunk boundary has to fall on a tile boundary; a batch that goes in one chunk keeps its odd receivers: cut
        // down to whole tiles it left a second chunk of a few receivers behind, whose sizes then steered the next solve)
        if (chunk < n_rx || chunk > (1ll << 22)) chunk = std::max<int64_t>(RX_TILE, std::min<int64_t>(chunk, 1ll << 22) / RX_TILE * RX_TILE);
        int g_side = 1;
        while (g_side < 256 && (int64_t)g_side * g_side * 4 < n_rx) g_side *= 2;
        const SceneView &v = scene->view;
        const double extent = std::max(v.gside * v.gnx, v.gside * v.gny);
        const double inv = extent > 0.0 ? g_side / extent : 0.0;
        const int64_t n_cells = (int64_t)g_side * g_side, n_tiles = (n_rx + RX_TILE - 1) / RX_TILE;
        double *rxs = (double *)temp_alloc(sizeof(double) * 3 * n_rx);
        int32_t *perm = (int32_t *)temp_alloc(sizeof(int32_t) * n_rx);
        double *boxes = (double *)temp_alloc(sizeof(double) * 6 * n_tiles);
        SOLVE_PTR(rxs);
        SOLVE_PTR(perm);
        SOLVE_PTR(boxes);
        if (n_cells <= RX_TILES_ONE_LAUNCH_CELLS) {
            timer.begin(0);
            hipLaunchKernelGGL(k_rx_tiles, dim3(1), dim3(1024), 0, st, d_rx, (int)n_rx, v.gx0, v.gy0, inv, g_side, rxs, perm, boxes);
            timer.end(0);
        } else {
            int32_t *counts = (int32_t *)temp_alloc(sizeof(int32_t) * 2 * n_cells);
            int64_t *cell_off = (int64_t *)temp_alloc(sizeof(int64_t) * (n_cells + 1));
            SOLVE_PTR(counts);
            SOLVE_PTR(cell_off);
            SOLVE_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * n_cells, st));
            timer.begin(0);
            hipLaunchKernelGGL(k_rx_count, dim3(blocks_for(n_rx, 256)), dim3(256), 0, st, d_rx, n_rx, v.gx0, v.gy0, inv, g_side, counts);
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, counts, n_cells, cell_off);
            hipLaunchKernelGGL(k_rx_scatter, dim3(blocks_for(n_rx, 256)), dim3(256), 0, st, d_rx, n_rx, v.gx0, v.gy0, inv, g_side,
                               cell_off, counts + n_cells, rxs, perm);
            hipLaunchKernelGGL(k_rx_boxes, dim3(blocks_for(n_tiles, 256)), dim3(256), 0, st, rxs, n_rx, boxes);
            timer.end(0);
        }
        tiles = RxTiles{rxs, perm, boxes};
    }
    // the wall next to every receiver in 256 directions, for the diffraction kernel (large batches over a scene with a grid)
    int32_t *rx_hints = nullptr;
    static const int hint_bins = [] { const char *e = getenv("RT_RX_HINT_BINS"); const int v = e ? atoi(e) : 0; return v >= 4 && v <= 4096 ? v : RX_HINT_BINS; }();
    static const int hint_cells = [] { const char *e = getenv("RT_RX_HINT_CELLS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 1024 ? v : RX_HINT_CELLS; }();
    bool diffraction_groups = false;  // (any_diffraction is only known once the head has run)
    for (int m = 0; m < mo; ++m) diffraction_groups = diffraction_groups || cand->groups[mo + m].count > 0;
    // (a quarter of a GiB of hints at most, i.e. 262 144 receivers at 256 directions: beyond that the diffraction kernel
    // does without them, with the tests it makes itself)
    if (tiled && diffraction_groups && scene->view.cells && scene->view.gnx > 0 && scene->wall_hints && !getenv("RT_NO_RX_HINTS") &&
        n_rx * (int64_t)hint_bins <= (1ll << 26)) {
        rx_hints = (int32_t *)temp_alloc(sizeof(int32_t) * hint_bins * n_rx);
        SOLVE_PTR(rx_hints);
        // (made on the side stream, in front of the head and beside the reflection kernel: the side stream then has to
        // start after the receivers have been put in tile order)
        SOLVE_HIP(hipEventRecord(entry_event, st));
    }
    bool head_done = false;
    if (!tiled) {
        if (diffraction_head()) return -1;
        head_done = true;
    }
    unsigned long long found_before = 0;
    // what the next solve sizes its launches from: the largest chunk of this one, not its last
    int64_t seen_paths = 0, seen_segments = 0, seen_clear = 0;

    for (int64_t r0 = 0; r0 < n_rx;) {
        const int64_t r1 = std::min(n_rx, r0 + chunk);
        const int64_t nr = r1 - r0;
        // workspace layout: paths | segs | hits.  The first nr records of both queues are the chunk's lines of sight.
        if (path_cap < nr + 1024) path_cap = nr + 1024;
        const int64_t seg_hint = scene->hint_segments > 0 ? scene->hint_segments : std::min<int64_t>(path_cap * 3, 1 << 22);
        const int64_t path_hint = scene->hint_paths > 0 ? scene->hint_paths : std::min<int64_t>(path_cap, 1 << 21);
        // Large batches: one obstruction launch per segment place, last segment first, each skipping the paths already
        // blocked (flags per path); only the path records are queued, their segments follow from the records' points.
        // Small batches are launch-bound: one launch over all segments, which are queued too (flags per segment).
        const bool by_rounds = path_hint >= (1 << 15);
        const int64_t seg_cap = by_rounds ? (1ll << 60) : path_cap * 3;
        const int64_t n_flags = by_rounds ? path_cap : seg_cap;
        size_t off_paths = 0;
        size_t off_segs = align_up(off_paths + sizeof(PathRec) * path_cap);
        size_t off_hits = align_up(off_segs + (by_rounds ? 0 : sizeof(double) * 6 * seg_cap));
        size_t off_list = align_up(off_hits + (size_t)n_flags);
        size_t total = align_up(off_list + sizeof(int32_t) * path_cap);
        void *ws = nullptr;
        if (rt_scene_workspace(scene, total, &ws)) {
            free_temps();
            return -1;
        }
        char *w = (char *)ws;
        uint8_t *d_hits = (uint8_t *)(w + off_hits);
        int32_t *d_list = (int32_t *)(w + off_list);  // paths still clear after the first obstruction round

        Queues q{(PathRec *)(w + off_paths), by_rounds ? nullptr : (double *)(w + off_segs), path_cap, seg_cap, d_counters, nr,
                 d_hits, by_rounds ? 1 : 0};

        // Lines of sight: queued like every other path (the first nr records), tested against the polygons the emitter
        // sees only (ray_tracing.py:232-237).  The reflection launch queues them on the way; reflections of every
        // order are one launch.
        scene->counters_dirty = true;
        timer.begin(0);
        // (on tiles the reflection launch takes the receivers [r0, r1) of the sorted order: over the chunks every
        // receiver meets every candidate once all the same; it queues no lines of sight)
        const bool los_queued = tiled ? (launch_reflect_tiled(scene, cand, TX, tiles, r0, r1, q), false)
                                      : launch_reflect(scene, cand, TX, d_rx, r0, r1, rx_tile, q);
        timer.end(0);
        if (!los_queued) {
            timer.begin(3);
            hipLaunchKernelGGL(k_los_emit, dim3(blocks_for(nr, 256)), dim3(256), 0, st, TX, d_rx, r0, nr, q);
            timer.end(3);
        }
        if (!head_done) {
            // beside the reflection launch just queued: the head on the side stream, then the main stream waits for it
            if (!scene->side_stream) SOLVE_HIP(rt_take_stream(scene->device, &scene->side_stream));
            if (!scene->side_event) SOLVE_HIP(hipEventCreateWithFlags(&scene->side_event, hipEventDisableTiming));
            // (the head takes blocks of the scene's temporary memory that a call before this one may have handed back
            // with its kernels still queued on the main stream — the deferred forms of the description and the field
            // step return without waiting: the side stream starts after what the main stream held when this solve began)
            SOLVE_HIP(hipStreamWaitEvent(scene->side_stream, entry_event, 0));
            if (counters_were_reset) {
                // (the reset of all counters queued on the main stream at the start must not land on the head's two)
                SOLVE_HIP(hipStreamSynchronize(st));
            }
            if (rx_hints)
                hipLaunchKernelGGL(k_rx_hints, dim3(blocks_for(n_rx * hint_bins, 256)), dim3(256), 0, scene->side_stream,
                                   scene->view, tiles.rxs, n_rx, hint_bins, hint_cells, rx_hints);
            scene->stream = scene->side_stream;
            const int rc_head = diffraction_head();
            scene->stream = st;
            if (rc_head) return -1;
            SOLVE_HIP(hipEventRecord(scene->side_event, scene->side_stream));
            SOLVE_HIP(hipStreamWaitEvent(st, scene->side_event, 0));
            head_done = true;
        }
        // diffractions
        if (any_diffraction) {
            // on tiles the diffraction kernel takes the receivers in tile order too: the 64 receivers of a wave are then
            // a compact patch, and the segments it queues from one edge point stay together longer in the grid walk
            const double *diff_rx = tiled ? tiles.rxs : d_rx;
            const int32_t *diff_perm = tiled ? tiles.perm : nullptr;
            timer.begin(1);
            for (int m = 0; m < mo; ++m) {
                const int64_t cnt = diff_count[m];
                if (cnt == 0) continue;
                switch (m) {
                    case 0: launch_diffract<0>(scene, diff_setup[m], nullptr, cnt, mo + m, diff_rx, diff_perm, r0, r1, q, rx_hints, hint_bins); break;
                    case 1: launch_diffract<1>(scene, diff_setup[m], alive[m], cnt, mo + m, diff_rx, diff_perm, r0, r1, q, rx_hints, hint_bins); break;
                    case 2: launch_diffract<2>(scene, diff_setup[m], alive[m], cnt, mo + m, diff_rx, diff_perm, r0, r1, q, rx_hints, hint_bins); break;
                    default: launch_diffract<3>(scene, diff_setup[m], alive[m], cnt, mo + m, diff_rx, diff_perm, r0, r1, q, rx_hints, hint_bins); break;
                }
            }
            timer.end(1);
        }
        // obstruction of the queued segments and the final filter, sized on the device
        timer.begin(2);
        int max_nseg = 0;
        if (by_rounds) {
            for (int k = 1; k <= mo; ++k)
                if (cand->groups[k - 1].count > 0) max_nseg = std::max(max_nseg, k + 1);
            if (any_diffraction) max_nseg = std::max(max_nseg, 2);
            // round 0 looks at the last segment of every path; nearly all of a coverage map's paths end there, so the
            // paths still clear are listed and the later rounds look at those only
            int rc = rt_launch_obstruct(scene, q.segs, path_cap, cand->d_ev, 0, d_hits, d_counters, nr, path_hint, q.paths, 0,
                                        false, nullptr, tx, d_rx);
            if (rc == 0 && max_nseg > 1)
                rc = rt_launch_list_clear_flags(scene, d_hits, path_cap, d_counters, nr, path_hint, d_list, d_counters + 8);
            const int64_t list_hint = scene->hint_clear > 0 ? scene->hint_clear : std::min<int64_t>(path_hint, 1 << 20);
            for (int round = 1; rc == 0 && round < max_nseg; ++round)
                rc = rt_launch_obstruct(scene, q.segs, path_cap, cand->d_ev, 0, d_hits, d_counters + 8, 0, list_hint, q.paths,
                                        round, false, d_list, tx, d_rx);
            if (rc) {
                free_temps();
                return -1;
            }
        } else if (rt_launch_obstruct(scene, q.segs, seg_cap, cand->d_ev, nr, d_hits, d_counters + 1, nr, seg_hint, nullptr, 0, false, nullptr,
                                      nullptr, nullptr)) {
            free_temps();
            return -1;
        }
        timer.end(2);
        timer.begin(3);
        if (path_hint >= (1 << 16))
            hipLaunchKernelGGL(k_finalize<8>, dim3(std::min<unsigned>(blocks_for(path_hint, 256 * 8), 4096u)), dim3(256), 0, st,
                               q.paths, d_hits, d_counters, nr, path_cap, seg_cap, by_rounds ? 1 : 0, out, h_counters_dev,
                               r1 == n_rx ? 1 : 0);
        else
            hipLaunchKernelGGL(k_finalize<1>, dim3(blocks_for(path_hint + path_hint / 4 + 256, 256)), dim3(256), 0, st, q.paths,
                               d_hits, d_counters, nr, path_cap, seg_cap, by_rounds ? 1 : 0, out, h_counters_dev,
                               r1 == n_rx ? 1 : 0);
        timer.end(3);
        SOLVE_HIP(hipStreamSynchronize(st));  // the filter's last block has written h_counters
        SOLVE_HIP(hipGetLastError());
        const int64_t n_paths = (int64_t)h_counters[0], n_segs = (int64_t)h_counters[1];
        if (n_paths > path_cap || n_segs > seg_cap) {
            // survivors did not fit: nothing of this chunk reached the outputs; grow the queues and redo the chunk
            const int64_t need = std::max<int64_t>(n_paths, (n_segs + 2) / 3);
            path_cap = need + need / 8 + 1024;
            continue;
        }
        found_before = h_counters[2];
        scene->counters_dirty = false;
        seen_paths = std::max<int64_t>(seen_paths, n_paths);
        seen_segments = std::max<int64_t>(seen_segments, n_segs);
        seen_clear = std::max<int64_t>(seen_clear, (int64_t)h_counters[3]);
        scene->hint_paths = std::max<int64_t>(seen_paths, 1);
        scene->hint_segments = std::max<int64_t>(seen_segments, 1);
        scene->hint_clear = std::max<int64_t>(seen_clear, 1);
        // segments walked: every queued segment on the one-launch route; by rounds, the last segment of every queued
        // path and then, per further round, at most the paths still clear after the first (nearly all of them are
        // walked again; those that have no segment at that place are not: an upper bound a per cent or two high)
        scene->last_segments += by_rounds ? n_paths + (int64_t)h_counters[3] * std::max(0, max_nseg - 1) : n_segs;
        if (getenv("RT_SOLVE_COUNTS"))
            fprintf(stderr, "[solve] receivers %lld: %lld paths queued, %lld clear after the first round, %d rounds\n", (long long)nr,
                    (long long)n_paths, (long long)h_counters[3], max_nseg);
        r0 = r1;
    }
    unsigned long long found = found_before;
    if (scene->reference_solver && found > 0 && (int64_t)found <= capacity) {
        int64_t kept = 0;
        if (rt_rejudge_paths(scene, cand, tx, d_rx, out.rx, out.group, out.cand, out.pts, (int64_t)found, &kept)) {
            free_temps();
            return -1;
        }
        found = (unsigned long long)kept;
    }
    *n_found = (int64_t)found;
    if (!on_device) {
        const int64_t m = std::min<int64_t>((int64_t)found, capacity);
        if (m > 0) {
            SOLVE_HIP(hipMemcpy(out_rx, out.rx, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
            SOLVE_HIP(hipMemcpy(out_group, out.group, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
            SOLVE_HIP(hipMemcpy(out_cand, out.cand, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
            SOLVE_HIP(hipMemcpy(out_points, out.pts, sizeof(double) * RT_MAX_ORDER * 3 * m, hipMemcpyDeviceToHost));
        }
    }
    timer.collect(scene->last_ms, scene->last_launches);
    rt_scene_temp_release_all(scene);  // the stream has been waited for
    return 0;
}

extern "C" int rt_solve_last_timing(rt_scene *scene, double *ms, int64_t *launches, int64_t *segments_tested) {
    if (!scene) RT_FAIL("rt_solve_last_timing: null scene");
    for (int f = 0; f < 4; ++f) {
        if (ms) ms[f] = scene->last_ms[f];
        if (launches) launches[f] = scene->last_launches[f];
    }
    if (segments_tested) *segments_tested = scene->last_segments;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// stand-alone solver entry points (geometry.py:660-687, 718-748, 800-833)
// ---------------------------------------------------------------------------------------------------------------

__global__ void k_solve_standalone(d3 origin, const double *__restrict__ dests, int64_t n, const double *__restrict__ planes,
                                   int k, const double *__restrict__ edge, double *__restrict__ out, uint8_t *__restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d3 dest = ld3(dests + 3 * i);
    d3 nrm[RT_MAX_ORDER], img[RT_MAX_ORDER], X[RT_MAX_ORDER];
    double g[RT_MAX_ORDER];
    d3 prev = origin;
    for (int j = 0; j < k; ++j) {
        nrm[j] = ld3(planes + 4 * j);
        const double s = planes[4 * j + 3] + dot(prev, nrm[j]);
        img[j] = prev - (2.0 * s) * nrm[j];
        g[j] = -s;
        prev = img[j];
    }
    const int npts = k + (edge ? 1 : 0);
    double *o = out + (size_t)i * npts * 3;
    bool good = true;
    d3 T = edge ? ld3(edge) : dest;  // reflections aim at the edge's first endpoint (geometry.py:821)
    for (int j = k - 1; j >= 0; --j) {
        const d3 dir = T - img[j];
        const double t = -g[j] / dot(dir, nrm[j]);
        if (!(t > 0.0 && t < 1.0)) good = false;
        X[j] = img[j] + t * dir;
        T = X[j];
    }
    for (int j = 0; j < k; ++j) st3(o + 3 * j, X[j]);
    if (edge) {
        const d3 A = k > 0 ? X[k - 1] : origin;
        const d3 P0 = ld3(edge), P1 = ld3(edge + 3);
        d3 fx, fy, fz;
        frame_from_vector(P1 - P0, fx, fy, fz);
        const double xe = dot(P0, fx), ye = dot(P0, fy);
        const double ax = dot(A, fx), ay = dot(A, fy), az = dot(A, fz);
        const double bx = dot(dest, fx), by = dot(dest, fy), bz = dot(dest, fz);
        const double rhoA = sqrt((xe - ax) * (xe - ax) + (ye - ay) * (ye - ay));
        const double rhoB = sqrt((xe - bx) * (xe - bx) + (ye - by) * (ye - by));
        const double sum = rhoA + rhoB;
        if (!(sum > 0.0)) good = false;
        const double z = az + (bz - az) * (rhoA / sum);
        st3(o + 3 * k, (xe * fx + ye * fy) + z * fz);
    }
    ok[i] = good ? 1 : 0;
}

static int solve_standalone(int device, const double *origin, const double *dests, int64_t n_dest, const double *planes,
                            int32_t k, const double *edge, double *out_points, uint8_t *out_ok) {
    if (!origin || (n_dest > 0 && (!dests || !out_points || !out_ok))) RT_FAIL("rt_solve_*: null argument");
    if (k < 0 || k > RT_MAX_ORDER || (k > 0 && !planes)) RT_FAIL("rt_solve_*: bad plane count");
    if (n_dest <= 0) return 0;
    RT_HIP(rt_use_device(device));
    const int npts = k + (edge ? 1 : 0);
    if (npts == 0) RT_FAIL("rt_solve_*: nothing to solve");
    double *d_dest = nullptr, *d_planes = nullptr, *d_edge = nullptr, *d_out = nullptr;
    uint8_t *d_ok = nullptr;
    hipError_t e = hipMalloc(&d_dest, sizeof(double) * 3 * n_dest);
    if (e == hipSuccess) e = hipMalloc(&d_planes, sizeof(double) * 4 * (k ? k : 1));
    if (e == hipSuccess) e = hipMalloc(&d_edge, sizeof(double) * 6);
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(double) * 3 * npts * n_dest);
    if (e == hipSuccess) e = hipMalloc(&d_ok, n_dest);
    if (e == hipSuccess) e = hipMemcpy(d_dest, dests, sizeof(double) * 3 * n_dest, hipMemcpyHostToDevice);
    if (e == hipSuccess && k) e = hipMemcpy(d_planes, planes, sizeof(double) * 4 * k, hipMemcpyHostToDevice);
    if (e == hipSuccess && edge) e = hipMemcpy(d_edge, edge, sizeof(double) * 6, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_solve_standalone, dim3(blocks_for(n_dest, 256)), dim3(256), 0, 0, ld3(origin), d_dest, n_dest,
                           d_planes, k, edge ? d_edge : nullptr, d_out, d_ok);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_points, d_out, sizeof(double) * 3 * npts * n_dest, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_ok, d_ok, n_dest, hipMemcpyDeviceToHost);
    (void)hipFree(d_dest);
    (void)hipFree(d_planes);
    (void)hipFree(d_edge);
    (void)hipFree(d_out);
    (void)hipFree(d_ok);
    if (e != hipSuccess) RT_FAIL(std::string("rt_solve_*: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int rt_solve_reflections(int device, const double *origin, const double *dests, int64_t n_dest,
                                    const double *planes, int32_t k, double *out_points, uint8_t *out_ok) {
    return solve_standalone(device, origin, dests, n_dest, planes, k, nullptr, out_points, out_ok);
}

extern "C" int rt_solve_reflections_diffraction(int device, const double *origin, const double *dests, int64_t n_dest,
                                                const double *planes, int32_t k, const double *edge,
                                                double *out_points, uint8_t *out_ok) {
    if (!edge) RT_FAIL("rt_solve_reflections_diffraction: null edge");
    return solve_standalone(device, origin, dests, n_dest, planes, k, edge, out_points, out_ok);
}
