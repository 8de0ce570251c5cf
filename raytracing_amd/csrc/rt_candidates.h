// Candidate lists of one emitter as the library keeps them (shared by the enumeration / solve and the field step).
#pragma once
#include <vector>

#include "rt_common.h"

struct CandGroup {
    int n_refl = 0;
    int has_edge = 0;
    int64_t count = 0;
    int32_t *chains = nullptr;    // device [count * n_refl]
    int32_t *edge_ids = nullptr;  // device [count]
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
};
