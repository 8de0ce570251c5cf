// Candidate lists of one emitter as the library keeps them (shared by the enumeration / solve and the field step).
#pragma once
#include <vector>

#include "rt_common.h"

// Where the polygon chain of candidate c of a group comes from.  Either the chains are stored (chains[c * k ..]), or —
// for a reflection level too large to be worth storing — they are formed when asked for, from the level before it:
// candidate c is child c - off[p] of the parent chain p with off[p] <= c < off[p + 1], i.e. that chain followed by the
// (c - off[p])-th polygon visible from its last polygon (col[row_off[last] ..], ascending), which is the order the
// reference's recursion visits them in (ray_tracing.py:316-322).
struct ChainSource {
    const int32_t *chains = nullptr;   // [count * k] or null
    const int32_t *parents = nullptr;  // [n_parents * (k - 1)]
    const int64_t *off = nullptr;      // [n_parents + 1] first child of each parent
    const int64_t *row_off = nullptr;  // CSR of the visibility matrix
    const int32_t *col = nullptr;
    int64_t n_parents = 0;
    const int32_t *coarse = nullptr;   // [ceil(count / CHAIN_COARSE)] parent of every CHAIN_COARSE-th candidate: where
                                       // the search for a candidate's parent starts
};
#define CHAIN_COARSE 64

__device__ inline void chain_load(const ChainSource &s, int k, int64_t c, int32_t *poly) {
    if (s.chains) {
        for (int i = 0; i < k; ++i) poly[i] = s.chains[c * k + i];
        return;
    }
    // last p with off[p] <= c: from the parent noted for the candidate's group of CHAIN_COARSE in growing strides
    // (most parents have children, so it is a few steps away), then by halving
    int64_t lo = 0, hi = s.n_parents;
    if (s.coarse) {
        lo = s.coarse[c / CHAIN_COARSE];
        int64_t step = 1;
        while (lo + step < hi && s.off[lo + step] <= c) {
            lo += step;
            step <<= 1;
        }
        if (lo + step < hi) hi = lo + step;
    }
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (s.off[mid] <= c) lo = mid; else hi = mid;
    }
    const int32_t *par = s.parents + lo * (k - 1);
    for (int i = 0; i < k - 1; ++i) poly[i] = par[i];
    poly[k - 1] = s.col[s.row_off[par[k - 2]] + (c - s.off[lo])];
}

struct CandGroup {
    int n_refl = 0;
    int has_edge = 0;
    int64_t count = 0;
    int32_t *chains = nullptr;    // device [count * n_refl]; null for a level whose chains are formed when asked for
    int32_t *edge_ids = nullptr;  // device [count]
    ChainSource source;           // chains, or the level before with the offsets of its children
};

struct rt_candidates {
    rt_scene *scene = nullptr;
    int max_order = 0;
    std::vector<CandGroup> groups;
    int64_t n_edges = 0;
    double *d_edges = nullptr;    // [n_edges * 6]
    int32_t *d_edge_pairs = nullptr;
    uint8_t *d_ev = nullptr;      // [n]
    std::vector<void *> allocations;
    // per-group tables of the path description (reflections, has an edge, chain source, edge ids), copied to the
    // device by the first description and kept: one block, in `allocations`
    void *d_describe_tables = nullptr;
};
