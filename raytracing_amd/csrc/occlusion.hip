// Segment-versus-scene obstruction (reference geometry.py:836-898) and point containment (geometry.py:1778-1803).
//
// Work mapping: one lane per segment; every lane of a wave walks the same polygon, so the polygon's plane and box
// are wave-uniform and come through the scalar data path (no per-lane memory traffic at all in the inner loop).
// Per (segment, polygon) the loop does a plane crossing and a padded box test; a pair whose crossing point lies in the
// polygon's padded box is parked in a per-wave LDS queue, and the queue is emptied 64 pairs at a time through the exact
// test of the reference (projection along the segment, extent overlap, boundary-inclusive point-in-polygon,
// 1e-8 < t < 1-1e-8), so the expensive test always runs with full waves.  The padded pre-test can only let too many
// pairs through, never too few, so the result is the exact test's.
#include <algorithm>

#include "rt_common.h"

// What a launch works on: item i is segment i of the queue, or (paths given) the segment `round` places from the end
// of path i, skipped when the path is already known to be blocked; out[i] is set when the item's segment is cut.
// Testing a path's segments last to first, one launch per place, stops at the first cut segment: most paths of a
// coverage map are blocked on the stretch that reaches the receiver, and their other segments are never walked.
// With a polygon mask, the mask applies to the first n_masked segments of the queue, or (paths given) to the paths of
// group -1: the lines of sight, which the reference tests against the polygons the emitter sees only
// (ray_tracing.py:232-237); every other segment is tested against the whole scene.
struct ObstructItems {
    const double *segs;    // the segment queue; null with paths: a path's segments are then formed from its record, the
                           // emitter and the receiver's position (nothing but the records was queued)
    d3 tx;
    const double *rx;
    const PathRec *paths;
    int round;
    int64_t n_masked;
    const int32_t *index;  // when given, work item k is item index[k] (second pass over the stragglers of a first)
    const int32_t *cursor; // when given with index: where item index[k] had got to when it was handed over, 8 ints per
                           // entry (next record, end of the cell's records, column, row, last row of the column,
                           // columns not yet entered, lowest point over the column as a double): the walk goes on from there
    int64_t count_add;     // added to the item count read from the device
};

__device__ __forceinline__ bool item_segment(const ObstructItems &it, int64_t &i, const uint8_t *__restrict__ out,
                                             bool &masked, d3 &A, d3 &B) {
    if (it.index) i = it.index[i];
    masked = i < it.n_masked;
    if (!it.paths) {
        A = ld3(it.segs + 6 * i);
        B = ld3(it.segs + 6 * i + 3);
        return true;
    }
    const PathRec &p = it.paths[i];
    masked = p.group < 0;
    if (it.round >= p.nseg || out[i]) return false;
    if (it.segs) {
        const int64_t seg = p.seg0 + (int64_t)(p.nseg - 1 - it.round) * p.seg_stride;
        A = ld3(it.segs + 6 * seg);
        B = ld3(it.segs + 6 * seg + 3);
    } else {
        const int e = p.npts - it.round;  // the segment ends at point e, npts standing for the receiver
        // (A before B on every route through this function: with the two written in different orders the optimiser
        // merged the stores of the routes through pointers into the two structs and left both in scratch memory,
        // 40 bytes per lane written and read back by every wave)
        const double *b = e >= p.npts ? it.rx + 3 * (int64_t)p.rx : p.pts + 3 * e;
        const d3 bb = ld3(b);
        d3 a = it.tx;
        if (e >= 1) a = ld3(p.pts + 3 * (e - 1));
        A = a;
        B = bb;
    }
    return true;
}

// Runs the exact test for up to 64 parked (item lane, polygon) pairs of a wave, one pair per lane.  BATCHED picks the
// arrangement of the exact test for latency-bound launches.
template <bool BATCHED>
__device__ __forceinline__ void obstruct_drain(const SceneView &sc, const double (*wave_ab)[6],
                                               const int64_t *wave_item, int p_begin,
                                               const uint32_t *entries, int n_entries, int lane,
                                               volatile uint8_t *wave_hit, uint8_t *__restrict__ out) {
    if (lane < n_entries) {
        const uint32_t e = entries[lane];
        const int L = (int)(e & 63u);
        if (!wave_hit[L]) {
            const int p = p_begin + (int)(e >> 6);
            const d3 A = ld3(wave_ab[L]), B = ld3(wave_ab[L] + 3);
            if (BATCHED ? polygon_cuts_segment_batched(sc, p, A, B) : polygon_cuts_segment(sc, p, A, B)) {
                wave_hit[L] = 1;
                out[wave_item[L]] = 1;  // out starts at zero; several polygon chunks may set the same byte
            }
        }
    }
}

// polygon_cuts_segment for a polygon whose plane and ring sit in LDS (ring = nv vertices of 3 doubles).  The box
// shortcut of the global-memory form is left out: it only saves looking at the vertices, which costs little here, and
// never changes the result.
__device__ inline bool polygon_cuts_segment_lds(const double *ring, int nv, d3 n, double d, d3 A, d3 B) {
    const d3 V = B - A;
    d3 fx, fy, fz;
    frame_from_vector(V, fx, fy, fz);
    const double z_min = dot(A, fz), z_max = dot(B, fz);
    const double px = dot(A, fx), py = dot(A, fy);
    double zlo = INFINITY, zhi = -INFINITY;
    bool inside = false, boundary = false;
    const d3 Pl = ld3(ring + 3 * (nv - 1));
    double ax = dot(Pl, fx), ay = dot(Pl, fy);
    for (int k0 = 0; k0 < nv; k0 += 4) {
        d3 Ps[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Ps[j] = ld3(ring + 3 * (k0 + j < nv ? k0 + j : nv - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < nv) {
                const d3 P = Ps[j];
                const double z = dot(P, fz);
                zlo = fmin(zlo, z);
                zhi = fmax(zhi, z);
                const double bx = dot(P, fx), by = dot(P, fy);
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
    const double t = -(d + dot(A, n)) / dot(V, n);
    return 1e-8 < t && t < 1 - 1e-8;
}

#define OBS_STAGE 128
#define OBS_STAGE_VERTS 1024

// obstruct_drain for entries whose polygons are those of the stage in LDS: recs[i], ring of polygon i =
// verts[3 * voff[i] .. 3 * voff[i + 1]); an entry's polygon field counts from the stage's first polygon.
__device__ __forceinline__ void obstruct_drain_lds(const OccRec *recs, const int *voff, const double *verts,
                                                   const double (*wave_ab)[6],
                                                   const int64_t *wave_item, const uint32_t *entries, int n_entries,
                                                   int lane, volatile uint8_t *wave_hit, uint8_t *__restrict__ out) {
    if (lane < n_entries) {
        const uint32_t e = entries[lane];
        const int L = (int)(e & 63u);
        if (!wave_hit[L]) {
            const int i = (int)(e >> 6);
            const OccRec &r = recs[i];
            const int v0 = voff[i], nv = voff[i + 1] - v0;
            if (polygon_cuts_segment_lds(verts + 3 * v0, nv, mk3(r.nx, r.ny, r.nz), r.d, ld3(wave_ab[L]), ld3(wave_ab[L] + 3))) {
                wave_hit[L] = 1;
                out[wave_item[L]] = 1;
            }
        }
    }
}

// Every-polygon walk.  n_dev, when given, holds the segment count on the device (clamped to n_cap; a count above n_cap
// means the producer overflowed its queue and the launch does nothing).  The block stages the pre-test records of its
// polygon range through LDS, OBS_STAGE at a time, so the inner loop reads them at LDS latency, and with them the
// polygons' rings when they fit (OBS_STAGE_VERTS vertices): a small launch is a chain of dependent memory accesses, and
// the exact test of a roof with tens of vertices was the longest link of it (13 of 28 us on 10 000 segments x 237
// polygons).  Pairs parked during a stage are then examined before the next stage replaces the rings.
#define RT_GRID_MIN_POLYGONS 512
#define OBS_PASSES 2
#ifndef OBS_FIRST_PASS_TRIPS
#define OBS_FIRST_PASS_TRIPS 16 /* trips of up to two records each */
#endif
template <bool MASKED>
__global__ __launch_bounds__(256) void k_obstruct(SceneView sc, ObstructItems items, int64_t n_cap,
                                                   const unsigned long long *__restrict__ n_dev,
                                                   const uint8_t *__restrict__ mask, int poly_chunk, uint8_t *out) {
    __shared__ int64_t wave_items[4][64];
    __shared__ double wave_ab[4][64][6];  // the segment of each lane, for the exact tests
    __shared__ OccRec s_rec[OBS_STAGE];
    __shared__ uint8_t s_mask[OBS_STAGE];
    __shared__ double s_verts[3 * OBS_STAGE_VERTS];
    __shared__ int s_voff[OBS_STAGE + 1];
    __shared__ uint32_t wave_queue[4][128];
    __shared__ uint8_t wave_hits[4][64];
    int64_t n_seg = n_cap;
    if (n_dev) {
        const unsigned long long nd = *n_dev + (unsigned long long)items.count_add;
        if (nd > (unsigned long long)n_cap) return;
        n_seg = (int64_t)nd;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *wq = wave_queue[wave];
    volatile uint8_t *wave_hit = wave_hits[wave];
    const int p_begin = blockIdx.y * poly_chunk;
    const int p_end = p_begin + poly_chunk < sc.n ? p_begin + poly_chunk : sc.n;
    for (int64_t tile = (int64_t)blockIdx.x * blockDim.x; tile < n_seg; tile += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s_base = tile + wave * 64;
        const int64_t s = s_base + lane;
        int64_t item = s;
        bool use_mask = false;
        d3 A = mk3(0, 0, 0), B = mk3(1, 0, 0);
        const bool active = s < n_seg && item_segment(items, item, out, use_mask, A, B);
        wave_items[wave][lane] = item;
        st3(wave_ab[wave][lane], A);
        st3(wave_ab[wave][lane] + 3, B);
        const d3 V = B - A;
        const double pad = 1e-5 * fmax(fabs(V.x), fmax(fabs(V.y), fabs(V.z))) + 1e-6;
        bool hit = false;
        bool wave_done = !__any(active);  // wave-uniform
        int qn = 0;                       // wave-uniform
        wave_hit[lane] = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += OBS_STAGE) {
            if (__syncthreads_and(wave_done)) break;  // also: the previous stage has been consumed
            const int cnt = p_end - p0 < OBS_STAGE ? p_end - p0 : OBS_STAGE;
            const int v_begin = sc.vert_off[p0], v_count = sc.vert_off[p0 + cnt] - v_begin;
            const bool rings_staged = v_count <= OBS_STAGE_VERTS;  // block-uniform
            {
                const uint4 *src = (const uint4 *)(sc.poly_recs + p0);
                uint4 *dst = (uint4 *)s_rec;
                for (int i = threadIdx.x; i < cnt * 4; i += blockDim.x) dst[i] = src[i];
                if (MASKED) {
                    for (int i = threadIdx.x; i < cnt; i += blockDim.x) s_mask[i] = mask[p0 + i];
                }
                if (rings_staged) {
                    const double *vs = sc.verts + 3 * (size_t)v_begin;
                    for (int i = threadIdx.x; i < 3 * v_count; i += blockDim.x) s_verts[i] = vs[i];
                    for (int i = threadIdx.x; i <= cnt; i += blockDim.x) s_voff[i] = sc.vert_off[p0 + i] - v_begin;
                }
            }
            __syncthreads();
            if (wave_done) continue;
            const int stage_first = rings_staged ? p0 - p_begin : 0;  // what an entry's polygon field counts from
            if (rings_staged && qn > 0) {  // pairs parked during a stage whose rings did not fit
                obstruct_drain<true>(sc, wave_ab[wave], wave_items[wave], p_begin, wq, qn, lane, wave_hit, out);
                __builtin_amdgcn_wave_barrier();
                qn = 0;
                hit = wave_hit[lane] != 0;
            }
            for (int r = 0; r < cnt; ++r) {
                bool wanted = active && !hit;
                if (MASKED) wanted = wanted && (!use_mask || s_mask[r]);
                const bool cand = wanted && occ_pretest(&s_rec[r], sc, A, V, pad);
                // pairs that pass the padded pre-test are parked and examined exactly 64 at a time, all lanes busy
                const unsigned long long m = __ballot(cand);
                if (m != 0) {
                    if (cand)
                        wq[qn + __popcll(m & ((1ull << lane) - 1))] = ((uint32_t)(p0 + r - p_begin - stage_first) << 6) | (uint32_t)lane;
                    qn += __popcll(m);
                    __builtin_amdgcn_wave_barrier();
                    if (qn >= 64) {
                        qn -= 64;
                        if (rings_staged)
                            obstruct_drain_lds(s_rec, s_voff, s_verts, wave_ab[wave], wave_items[wave], wq + qn, 64, lane,
                                               wave_hit, out);
                        else
                            obstruct_drain<true>(sc, wave_ab[wave], wave_items[wave], p_begin, wq + qn, 64, lane, wave_hit, out);
                        __builtin_amdgcn_wave_barrier();
                        hit = wave_hit[lane] != 0;
                        if (__all(hit || !active)) {
                            qn = 0;
                            wave_done = true;
                            break;
                        }
                    }
                }
            }
            if (rings_staged && qn > 0) {  // before the next stage replaces the rings
                obstruct_drain_lds(s_rec, s_voff, s_verts, wave_ab[wave], wave_items[wave], wq, qn, lane, wave_hit, out);
                __builtin_amdgcn_wave_barrier();
                qn = 0;
                hit = wave_hit[lane] != 0;
                if (__all(hit || !active)) wave_done = true;
            }
        }
        if (qn > 0) obstruct_drain<true>(sc, wave_ab[wave], wave_items[wave], p_begin, wq, qn, lane, wave_hit, out);
        __syncthreads();  // the staging buffer is free again
    }
}

// Grid walk: one lane per segment, each lane visiting only the cells its own segment passes over (within RT_GRID_PAD)
// and testing the polygons registered there.  The walk goes cell column by cell column along the segment's major
// horizontal axis, from the segment's start; within a column the segment spans at most three cells of the minor axis,
// and a cell whose highest polygon lies below the segment's stretch over the column is skipped.  Lanes advance
// independently (a lane's next record comes from its own cell list, through the vector memory path), so the walk is
// efficient when the segments of a wave are neighbours — the solve emits them in fans for that reason.  The pre-test
// and the parked exact tests are the every-polygon kernel's.  The polygons too large for the grid (the ground) are
// tested first by every segment.  Every polygon the exact test could flag has its crossing point inside its own
// padded box, hence inside a cell it is registered in, and that cell is on the segment's walk.
#ifdef RT_OCC_STATS
__device__ unsigned long long g_occ_stats[8];
extern "C" int rt_debug_occlusion_stats(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_occ_stats), sizeof(g_occ_stats)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_occ_stats), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
__device__ inline unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
#endif

#ifndef OBS_GRID_MIN_WAVES
#define OBS_GRID_MIN_WAVES 4
#endif
template <bool MASKED>
__global__ __launch_bounds__(256, OBS_GRID_MIN_WAVES) void k_obstruct_grid(SceneView sc, ObstructItems items, int64_t n_cap,
                                                        const unsigned long long *__restrict__ n_dev,
                                                        const uint8_t *__restrict__ mask, uint8_t *out, int max_trips,
                                                        unsigned long long *late_count, int32_t *late_items,
                                                        int32_t *late_cursor, int split_shift) {
    // split_shift: a launch of few items walks every segment in 2^split_shift pieces, one lane each (a piece = its share
    // of the columns the segment crosses, tested against the whole segment like any other: a polygon reaches the
    // segment inside a cell of some piece).  A launch lasts as long as its longest walk, and with few segments there
    // is nothing else to fill the chip with.  Not combined with the hand-over between passes.
    __shared__ int64_t wave_items[4][64];
    __shared__ double wave_ab[4][64][6];  // the segment of each lane, for the exact tests
    __shared__ uint32_t wave_queue[4][128];
    __shared__ uint8_t wave_hits[4][64];
    int64_t n_seg = n_cap;
    if (n_dev) {
        const unsigned long long nd = *n_dev + (unsigned long long)items.count_add;
        if (nd > (unsigned long long)n_cap) return;
        n_seg = (int64_t)nd;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *wq = wave_queue[wave];
    volatile uint8_t *wave_hit = wave_hits[wave];
    const OccCell global_cell = sc.cells[sc.gnx * sc.gny];
    const int64_t n_lanes = n_seg << split_shift;
    for (int64_t tile = (int64_t)blockIdx.x * blockDim.x; tile < n_lanes; tile += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s_base = tile + wave * 64;
        const int64_t s = s_base + lane;
        int64_t item = s >> split_shift;
        const int piece = (int)(s & ((1 << split_shift) - 1));
        bool use_mask = false;
        d3 A = mk3(0, 0, 0), B = mk3(1, 0, 0);
        const bool active = s < n_lanes && item_segment(items, item, out, use_mask, A, B);
        wave_items[wave][lane] = item;
        st3(wave_ab[wave][lane], A);
        st3(wave_ab[wave][lane] + 3, B);
        const d3 V = B - A;
        const double pad = 1e-5 * fmax(fabs(V.x), fmax(fabs(V.y), fabs(V.z))) + 1e-6;
        // walk set-up: u = major horizontal axis, w = minor, both relative to the grid origin; the columns are visited
        // from the segment's start towards its end
        const bool major_x = fabs(V.x) >= fabs(V.y);
        const double au = major_x ? A.x - sc.gx0 : A.y - sc.gy0, aw = major_x ? A.y - sc.gy0 : A.x - sc.gx0;
        const double du = major_x ? V.x : V.y, dw = major_x ? V.y : V.x;
        const int nu = major_x ? sc.gnx : sc.gny, nw = major_x ? sc.gny : sc.gnx;
        const double inv_du = fabs(du) > 1e-12 ? 1.0 / du : 0.0;
        const double gp = RT_GRID_PAD + 1e-9 * (fabs(au) + fabs(aw) + fabs(du));
        const double fi_lo = floor((fmin(au, au + du) - gp) * sc.ginv), fi_hi = floor((fmax(au, au + du) + gp) * sc.ginv);
        const bool outside = !active || fi_hi < 0.0 || fi_lo > (double)(nu - 1);
        const int i_first = outside ? 0 : (int)fmax(fi_lo, 0.0), i_last = outside ? -1 : (int)fmin(fi_hi, (double)(nu - 1));
        const int step = du >= 0.0 ? 1 : -1;
        int cols_left = i_last - i_first + 1;           // columns not yet entered
        int i = (step > 0 ? i_first : i_last) - step;  // current column
        int j = 1, j_hi = 0;
        int k = active ? global_cell.begin : 0, k_end = active ? global_cell.end : 0;
        const bool resumed = active && items.index && items.cursor;
        if (split_shift > 0 && !resumed) {
            // this piece's columns, counted along the walk; the polygons too large for the grid go with the first piece
            const int lo = (int)(((int64_t)cols_left * piece) >> split_shift);
            const int hi = (int)(((int64_t)cols_left * (piece + 1)) >> split_shift);
            i += step * lo;
            cols_left = hi - lo;
            if (piece != 0) k = k_end = 0;
        }
#ifdef RT_OCC_STATS
        unsigned long long st_cells = 0, st_recs = 0, st_iters = 0, st_exact = 0;
#endif
        double z_low = 0.0;  // lowest point of the segment over the current column, relative to the grid origin
        // moves to the next cell that has records and is not entirely below the segment; false when the walk is over
        auto next_cell = [&]() -> bool {
            while (true) {
                ++j;
                if (j > j_hi) {
                    if (cols_left <= 0) return false;
                    --cols_left;
                    i += step;
                    // minor-axis cells of column i: the segment's extent over the padded slab of the column
                    double ta = 0.0, tb = 1.0;
                    if (inv_du != 0.0) {
                        const double t0 = (i * sc.gside - gp - au) * inv_du, t1 = ((i + 1) * sc.gside + gp - au) * inv_du;
                        ta = fmax(0.0, fmin(t0, t1));
                        tb = fmin(1.0, fmax(t0, t1));
                    }
                    const double wa = aw + ta * dw, wb = aw + tb * dw;
                    const double fj_lo = floor((fmin(wa, wb) - gp) * sc.ginv), fj_hi = floor((fmax(wa, wb) + gp) * sc.ginv);
                    if (fj_hi < 0.0 || fj_lo > (double)(nw - 1)) {
                        j = 1;
                        j_hi = 0;
                        continue;
                    }
                    j = (int)fmax(fj_lo, 0.0);
                    j_hi = (int)fmin(fj_hi, (double)(nw - 1));
                    z_low = fmin(A.z + ta * V.z, A.z + tb * V.z) - sc.gz0 - pad;
                }
                // (fetching the up to three cells of a column together on entering it was slower: 18.8 against 12.5 ms)
                const OccCell cell = sc.cells[major_x ? j * sc.gnx + i : i * sc.gnx + j];
#ifdef RT_OCC_STATS
                ++st_cells;
#endif
                if (cell.end > cell.begin && !(z_low > (double)cell.zmax)) {
                    k = cell.begin;
                    k_end = cell.end;
                    return true;
                }
            }
        };
        bool finished = !active;
        if (resumed) {
            // the walk of an item handed over by the first pass goes on where it stopped; in pieces, the first piece
            // finishes the column the walk was in and every piece takes its share of the columns not yet entered
            const int32_t *c = items.cursor + 8 * (s >> split_shift);
            const int left = c[5];
            const int lo = (int)(((int64_t)left * piece) >> split_shift);
            const int hi = (int)(((int64_t)left * (piece + 1)) >> split_shift);
            i = c[2] + step * lo;
            cols_left = hi - lo;
            if (piece == 0) {
                k = c[0]; k_end = c[1]; j = c[3]; j_hi = c[4];
                z_low = __hiloint2double(c[7], c[6]);
            } else {
                k = k_end = 0;
                j = 1;
                j_hi = 0;
            }
        }
        if (active && k >= k_end) finished = !next_cell();
        int qn = 0;  // wave-uniform
        wave_hit[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        int trips = 0;  // wave-uniform
        // (Fetching the next trip's record before this trip's arithmetic was tried and was slower, 13.2 against 12.5 ms
        // on 25 M segments of lln, as was fetching a second record per trip: the record fetch is not what a trip waits
        // for.  Doubling the pre-test arithmetic costs 23 %.)
        while (__any(!finished) && trips < max_trips) {
            ++trips;
            // two records of the cell per trip where it has two left: what a trip costs beside the pre-tests (moving on
            // to the next cell, the votes, the loop) is then shared by both
            bool cand = false, cand2 = false;
            int poly = 0, poly2 = 0;
            if (!finished) {
                const OccRec *r = sc.cell_recs + k;
                const bool two = k + 1 < k_end;
                poly = r->poly;
                bool wanted = true;
                if (MASKED) wanted = !use_mask || mask[poly] != 0;
                if (wanted) cand = occ_pretest(r, sc, A, V, pad);
                if (two) {
                    poly2 = r[1].poly;
                    bool wanted2 = true;
                    if (MASKED) wanted2 = !use_mask || mask[poly2] != 0;
                    if (wanted2) cand2 = occ_pretest(r + 1, sc, A, V, pad);
                }
                k += two ? 2 : 1;
                if (k >= k_end) finished = !next_cell();
#ifdef RT_OCC_STATS
                st_recs += two ? 2 : 1;
                if (cand) ++st_exact;
                if (cand2) ++st_exact;
#endif
            }
#ifdef RT_OCC_STATS
            if (lane == 0) ++st_iters;
#endif
            auto park = [&](bool c, int p) {
                const unsigned long long m = __ballot(c);
                if (m == 0) return;
                if (c) wq[qn + __popcll(m & ((1ull << lane) - 1))] = ((uint32_t)p << 6) | (uint32_t)lane;
                qn += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64) {
                    qn -= 64;
                    obstruct_drain<false>(sc, wave_ab[wave], wave_items[wave], 0, wq + qn, 64, lane, wave_hit, out);
                    __builtin_amdgcn_wave_barrier();
                    if (wave_hit[lane]) finished = true;
                }
            };
            park(cand, poly);
            park(cand2, poly2);
        }
        if (qn > 0) obstruct_drain<false>(sc, wave_ab[wave], wave_items[wave], 0, wq, qn, lane, wave_hit, out);
        __builtin_amdgcn_wave_barrier();
        if (late_items) {
            // a lane whose walk is not over when the wave's trips are used up hands its item to the second pass, where
            // the stragglers of many waves walk side by side instead of each holding 63 idle lanes
            const bool late = !finished && !wave_hit[lane];
            const unsigned long long lm = __ballot(late);
            if (lm != 0) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(late_count, (unsigned long long)__popcll(lm));
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
                base = ((unsigned long long)hi << 32) | lo;
                if (late) {
                    const unsigned long long at = base + __popcll(lm & ((1ull << lane) - 1));
                    late_items[at] = (int32_t)item;
                    if (late_cursor) {
                        int32_t *c = late_cursor + 8 * at;
                        c[0] = k; c[1] = k_end; c[2] = i; c[3] = j; c[4] = j_hi; c[5] = cols_left;
                        c[6] = __double2loint(z_low); c[7] = __double2hiint(z_low);
                    }
                }
            }
        }
#ifdef RT_OCC_STATS
        {
            const unsigned long long a = wave_sum(active ? 1 : 0), c = wave_sum(st_cells), r = wave_sum(st_recs),
                                     e = wave_sum(st_exact), it = wave_sum(st_iters), h = wave_sum(wave_hit[lane] ? 1 : 0);
            if (lane == 0 && a > 0) {
                atomicAdd(&g_occ_stats[0], a);
                atomicAdd(&g_occ_stats[1], c);
                atomicAdd(&g_occ_stats[2], r);
                atomicAdd(&g_occ_stats[3], e);
                atomicAdd(&g_occ_stats[4], it);
                atomicAdd(&g_occ_stats[5], 1ull);
                atomicAdd(&g_occ_stats[6], h);
            }
        }
#endif
    }
}

__global__ void k_zero_bytes(uint8_t *__restrict__ out, int64_t n_cap, const unsigned long long *__restrict__ n_dev,
                             int64_t count_add) {
    const unsigned long long nd = *n_dev + (unsigned long long)count_add;
    const int64_t n = nd > (unsigned long long)n_cap ? 0 : (int64_t)nd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0;
}

// Obstruction flags of n_item items: segments of the queue, or with d_paths one segment per path (see ObstructItems;
// the flags then accumulate over the rounds).  The flags are cleared first when clear_flags is set; otherwise whoever
// queued the items has cleared them.  d_mask, when given, restricts the polygons
// for the first n_masked segments / the line-of-sight paths.  d_index, when given, lists the items to look at (the
// count then counts list entries).  With d_count the device holds the item
// count less count_add (the sum <= n_item) and the grid is sized for n_hint of them.
int rt_launch_obstruct(rt_scene *scene, const double *d_segs, int64_t n_item, const uint8_t *d_mask, int64_t n_masked,
                       uint8_t *d_out, const unsigned long long *d_count, int64_t count_add, int64_t n_hint,
                       const PathRec *d_paths, int round, bool clear_flags, const int32_t *d_index, const double *tx,
                       const double *d_rx) {
    if (n_item == 0) return 0;
    hipStream_t stream = scene->stream;
    const int block = 256;
    const int64_t n_grid = d_count ? (n_hint < 1 ? 1 : (n_hint > n_item ? n_item : n_hint)) : n_item;
    int64_t grid = (n_grid + block - 1) / block;
    if (grid > (1 << 20)) grid = 1 << 20;  // the kernel strides over the rest
    // few items: split the polygon list over grid.y so the launch still fills the chip (target >= ~4096 waves)
    const int64_t waves = grid * (block / RT_WAVE);
    const int64_t target_waves = 4096;  // measured on 10 402 segments x 237 polygons: 1 024: 34 us, 2 048: 33, 4 096: 30, 8 192: 31
    int64_t ny = waves >= target_waves ? 1 : (target_waves + waves - 1) / waves;
    const int64_t max_ny = (scene->view.n + 7) / 8;
    ny = ny > max_ny ? max_ny : ny;
    ny = ny < 1 ? 1 : ny;
    int poly_chunk = (int)((scene->view.n + ny - 1) / ny);
    ny = (scene->view.n + poly_chunk - 1) / poly_chunk;
    if (clear_flags) {
        if (d_count)
            hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)(grid < 1024 ? grid : 1024)), dim3(block), 0, scene->stream,
                               d_out, n_item, d_count, count_add);
        else
            RT_HIP(hipMemsetAsync(d_out, 0, (size_t)n_item, scene->stream));
    }
    if (!d_segs && !(d_paths && tx && d_rx)) RT_FAIL("rt_launch_obstruct: no segments");
    const ObstructItems items{d_segs, tx ? ld3(tx) : mk3(0, 0, 0), d_rx, d_paths, round, d_mask ? n_masked : 0, d_index,
                              nullptr, d_count ? count_add : 0};
    // small launches on small scenes are latency-bound: there the every-polygon walk, split over grid.y, is quicker
    const bool walk_grid = scene->occlusion_mode == 2 ||
                           (scene->occlusion_mode == 0 && (scene->view.n > RT_GRID_MIN_POLYGONS || n_grid >= (1 << 16)));
    if (walk_grid) {
        // Large launches go in passes: every wave of the first stops after OBS_FIRST_PASS_TRIPS trips and
        // hands the items still walking to a second launch over just those, which walks them to the end (a wave lasts
        // as long as its longest walk, and most walks are short). Measured on 25 M segments of lln: one pass 17.2 ms,
        // two passes 12.6 ms with 24 one-record trips (worse with 8, 12, 16 or 40), three passes (24, 96, rest) 12.9 ms: a handed
        // over item starts its walk again.
        const bool passes = n_grid >= (1 << 20) && n_item <= 0x7fffffffLL;
        if (!passes) {
            // few segments (the later rounds of a solve, a rank's share of a map): walked in pieces
            const int split_shift = n_grid <= (1 << 17) ? 3 : (n_grid <= (1 << 19) ? 2 : 1);
            grid = ((n_grid << split_shift) + block - 1) / block;
            if (d_mask)
                hipLaunchKernelGGL(k_obstruct_grid<true>, dim3((unsigned)grid), dim3(block), 0, stream, scene->view, items,
                                   n_item, d_count, d_mask, d_out, 0x7fffffff, (unsigned long long *)nullptr,
                                   (int32_t *)nullptr, (int32_t *)nullptr, split_shift);
            else
                hipLaunchKernelGGL(k_obstruct_grid<false>, dim3((unsigned)grid), dim3(block), 0, stream, scene->view, items,
                                   n_item, d_count, d_mask, d_out, 0x7fffffff, (unsigned long long *)nullptr,
                                   (int32_t *)nullptr, (int32_t *)nullptr, split_shift);
        } else {
            // two hand-over lists (a pass reads one and fills the other) and their counters
            // (behind the two lists: where each handed-over item had got to, 8 ints per item)
            if (scene->late_cap < 10 * (size_t)n_item) {
                RT_HIP(hipStreamSynchronize(stream));
                if (scene->late_items) (void)hipFree(scene->late_items);
                scene->late_items = nullptr;
                scene->late_cap = 0;
                RT_HIP(hipMalloc((void **)&scene->late_items, sizeof(int32_t) * 10 * (size_t)n_item));
                scene->late_cap = 10 * (size_t)n_item;
            }
            int32_t *cursors = scene->late_items + 2 * (size_t)n_item;
            unsigned long long *counts = scene->d_counters + 6;  // [6], [7]
            RT_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), stream));
            const int trip_limits[OBS_PASSES] = {OBS_FIRST_PASS_TRIPS, 0x7fffffff};
            // about a quarter of the items are handed over; when that is too few to fill the chip they go on in pieces
            const int late_shift = n_grid <= (1 << 22) ? 2 : (n_grid <= (1 << 23) ? 1 : 0);
            for (int pass = 0; pass < OBS_PASSES; ++pass) {
                ObstructItems it = items;
                const unsigned long long *n_in = d_count;
                unsigned g = (unsigned)grid;
                if (pass > 0) {
                    it.index = scene->late_items + (size_t)((pass - 1) & 1) * n_item;
                    it.cursor = pass == 1 ? cursors : nullptr;
                    it.count_add = 0;
                    n_in = counts + ((pass - 1) & 1);
                    g = (unsigned)std::max<int64_t>(1024, (grid >> (2 * pass)) << late_shift);
                }
                const bool last = pass == OBS_PASSES - 1;
                int32_t *late_out = last ? nullptr : scene->late_items + (size_t)(pass & 1) * n_item;
                unsigned long long *count_out = last ? nullptr : counts + (pass & 1);
                if (pass >= 2 && !last) RT_HIP(hipMemsetAsync(count_out, 0, sizeof(unsigned long long), stream));
                if (d_mask)
                    hipLaunchKernelGGL(k_obstruct_grid<true>, dim3(g), dim3(block), 0, stream, scene->view, it, n_item, n_in,
                                       d_mask, d_out, trip_limits[pass], count_out, late_out, pass == 0 ? cursors : (int32_t *)nullptr,
                                       pass == 1 ? late_shift : 0);
                else
                    hipLaunchKernelGGL(k_obstruct_grid<false>, dim3(g), dim3(block), 0, stream, scene->view, it, n_item, n_in,
                                       d_mask, d_out, trip_limits[pass], count_out, late_out, pass == 0 ? cursors : (int32_t *)nullptr,
                                       pass == 1 ? late_shift : 0);
            }
        }
    } else if (d_mask)
        hipLaunchKernelGGL(k_obstruct<true>, dim3((unsigned)grid, (unsigned)ny), dim3(block), 0, scene->stream,
                           scene->view, items, n_item, d_count, d_mask, poly_chunk, d_out);
    else
        hipLaunchKernelGGL(k_obstruct<false>, dim3((unsigned)grid, (unsigned)ny), dim3(block), 0, scene->stream,
                           scene->view, items, n_item, d_count, d_mask, poly_chunk, d_out);
    RT_HIP(hipGetLastError());
    return 0;
}

// list = the items i < *n_dev + count_add whose flag is clear, *list_count = how many.  A thread looks at 16 flags (one
// 16-byte load; flags is 16-byte aligned), a block at 4096 and reserves the room of what it found with one atomic.
__global__ __launch_bounds__(256) void k_list_clear_flags(const uint8_t *__restrict__ flags, int64_t n_cap,
                                                          const unsigned long long *__restrict__ n_dev, int64_t count_add,
                                                          int32_t *__restrict__ list, unsigned long long *list_count) {
    __shared__ int wave_total[4];
    __shared__ unsigned long long block_base;
    const unsigned long long nd = *n_dev + (unsigned long long)count_add;
    if (nd > (unsigned long long)n_cap) return;
    const int64_t n = (int64_t)nd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 4096; base < n; base += (int64_t)gridDim.x * 4096) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 16;
        uint4 v = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        if (i0 + 16 <= n) {
            v = *(const uint4 *)(flags + i0);
        } else if (i0 < n) {
            uint8_t *b = (uint8_t *)&v;
            for (int k = 0; k < 16 && i0 + k < n; ++k) b[k] = flags[i0 + k];
        }
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine += ((w[k] >> (8 * j)) & 0xffu) == 0 ? 1 : 0;
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        const int total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        if (threadIdx.x == 0 && total > 0) block_base = atomicAdd(list_count, (unsigned long long)total);
        __syncthreads();
        if (mine > 0) {
            int before = incl - mine;
            for (int q = 0; q < wave; ++q) before += wave_total[q];
            int64_t pos = (int64_t)block_base + before;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((w[k] >> (8 * j)) & 0xffu) == 0) list[pos++] = (int32_t)(i0 + 4 * k + j);
        }
        __syncthreads();  // wave_total and block_base are reused by the next stretch
    }
}

int rt_launch_list_clear_flags(rt_scene *scene, const uint8_t *d_flags, int64_t n_cap, const unsigned long long *d_count,
                               int64_t count_add, int64_t n_hint, int32_t *d_list, unsigned long long *d_list_count) {
    int64_t grid = (std::max<int64_t>(n_hint, 1) + 4095) / 4096;
    grid = grid > 65536 ? 65536 : grid;
    hipLaunchKernelGGL(k_list_clear_flags, dim3((unsigned)grid), dim3(256), 0, scene->stream, d_flags, n_cap, d_count,
                       count_add, d_list, d_list_count);
    RT_HIP(hipGetLastError());
    return 0;
}

extern "C" int rt_obstruct_segments(rt_scene *scene, const double *segs, int64_t n_seg, const uint8_t *mask,
                                    uint8_t *out, int on_device) {
    if (!scene) RT_FAIL("rt_obstruct_segments: null scene");
    if (n_seg < 0) RT_FAIL("rt_obstruct_segments: negative count");
    if (n_seg == 0) return 0;
    if (!segs || !out) RT_FAIL("rt_obstruct_segments: null array");
    RT_HIP(rt_use_device(scene->device));
    if (on_device) {
        if (rt_launch_obstruct(scene, segs, n_seg, mask, n_seg, out, nullptr, 0, 0, nullptr, 0, true, nullptr, nullptr, nullptr)) return -1;
        RT_HIP(hipStreamSynchronize(scene->stream));
        return 0;
    }
    double *d_segs = nullptr;
    uint8_t *d_out = nullptr, *d_mask = nullptr;
    int rc = -1;
    do {
        if (hipMalloc(&d_segs, sizeof(double) * 6 * n_seg) != hipSuccess) break;
        if (hipMalloc(&d_out, n_seg) != hipSuccess) break;
        if (mask && hipMalloc(&d_mask, scene->view.n) != hipSuccess) break;
        if (hipMemcpyAsync(d_segs, segs, sizeof(double) * 6 * n_seg, hipMemcpyHostToDevice, scene->stream) != hipSuccess)
            break;
        if (mask && hipMemcpyAsync(d_mask, mask, scene->view.n, hipMemcpyHostToDevice, scene->stream) != hipSuccess) break;
        if (rt_launch_obstruct(scene, d_segs, n_seg, d_mask, n_seg, d_out, nullptr, 0, 0, nullptr, 0, true, nullptr, nullptr, nullptr)) break;
        if (hipMemcpyAsync(out, d_out, n_seg, hipMemcpyDeviceToHost, scene->stream) != hipSuccess) break;
        if (hipStreamSynchronize(scene->stream) != hipSuccess) break;
        rc = 0;
    } while (0);
    if (rc) rt_set_error(std::string("rt_obstruct_segments: ") + hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_segs);
    (void)hipFree(d_out);
    (void)hipFree(d_mask);
    return rc;
}

// one thread per polygon, exact test only: the return_polygons=True form
__global__ void k_segment_hits(SceneView sc, d3 A, d3 B, uint8_t *__restrict__ out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < sc.n) out[p] = polygon_cuts_segment(sc, p, A, B) ? 1 : 0;
}

extern "C" int rt_segment_hits(rt_scene *scene, const double *seg, uint8_t *out) {
    if (!scene || !seg || !out) RT_FAIL("rt_segment_hits: null argument");
    RT_HIP(rt_use_device(scene->device));
    uint8_t *d_out = nullptr;
    RT_HIP(hipMalloc(&d_out, scene->view.n));
    hipLaunchKernelGGL(k_segment_hits, dim3((scene->view.n + 255) / 256), dim3(256), 0, scene->stream, scene->view,
                       ld3(seg), ld3(seg + 3), d_out);
    hipError_t e = hipMemcpyAsync(out, d_out, scene->view.n, hipMemcpyDeviceToHost, scene->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(scene->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) RT_FAIL(std::string("rt_segment_hits: ") + hipGetErrorString(e));
    return 0;
}

__global__ void k_contains(SceneView sc, int p, const double *__restrict__ pts, int64_t n, int check_in_plane,
                           double plane_tol, uint8_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = polygon_contains_point(sc, p, ld3(pts + 3 * i), check_in_plane != 0, plane_tol) ? 1 : 0;
}

extern "C" int rt_contains_points(rt_scene *scene, int32_t polygon, const double *points, int64_t n_points,
                                  int check_in_plane, double plane_tol, uint8_t *out, int on_device) {
    if (!scene) RT_FAIL("rt_contains_points: null scene");
    if (polygon < 0 || polygon >= scene->view.n) RT_FAIL("rt_contains_points: polygon index out of range");
    if (n_points < 0) RT_FAIL("rt_contains_points: negative count");
    if (n_points == 0) return 0;
    if (!points || !out) RT_FAIL("rt_contains_points: null array");
    RT_HIP(rt_use_device(scene->device));
    const unsigned grid = (unsigned)((n_points + 255) / 256);
    if (on_device) {
        hipLaunchKernelGGL(k_contains, dim3(grid), dim3(256), 0, scene->stream, scene->view, polygon, points, n_points,
                           check_in_plane, plane_tol, out);
        RT_HIP(hipGetLastError());
        RT_HIP(hipStreamSynchronize(scene->stream));
        return 0;
    }
    double *d_pts = nullptr;
    uint8_t *d_out = nullptr;
    RT_HIP(hipMalloc(&d_pts, sizeof(double) * 3 * n_points));
    hipError_t e = hipMalloc(&d_out, n_points);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts, points, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice, scene->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_contains, dim3(grid), dim3(256), 0, scene->stream, scene->view, polygon, d_pts, n_points,
                           check_in_plane, plane_tol, d_out);
        e = hipMemcpyAsync(out, d_out, n_points, hipMemcpyDeviceToHost, scene->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(scene->stream);
    (void)hipFree(d_pts);
    (void)hipFree(d_out);
    if (e != hipSuccess) RT_FAIL(std::string("rt_contains_points: ") + hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// rt_obstruct_segments for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that
// forbid hipMalloc).  Same flags as rt_obstruct_segments with device pointers; the grid walk always runs in its two
// passes here, and what the passes share — the count and the list of the segments still walking after the first —
// lives in a device array of fixed size that belongs to the library instead of memory allocated per call.
// ---------------------------------------------------------------------------------------------------------------
#define RT_NOALLOC_MAX_SEGMENTS (1 << 22) /* 4 194 304 segments per call: a 16 MiB list */
#define RT_NOALLOC_MAX_DEVICES 64

struct NoAllocScratch {
    unsigned long long late_count;  // segments handed from the first pass to the second
    unsigned long long pad_[31];
    int32_t late_items[RT_NOALLOC_MAX_SEGMENTS];
};

// One copy per GPU, at its own address on each; held until the process ends.  In host code this name is only a handle
// for hipGetSymbolAddress: kernels and copies get the address looked up below.
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];
static NoAllocScratch *g_scratch_of_device[RT_NOALLOC_MAX_DEVICES];

// Looks up this device's copy of the array (called by rt_scene_create with the scene's device current).
int rt_noalloc_setup(int device) {
    if (device < 0 || device >= RT_NOALLOC_MAX_DEVICES) return 0;  // scenes work there, the no-allocation entry does not
    if (g_scratch_of_device[device]) return 0;
    void *ptr = nullptr;
    RT_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) RT_FAIL("rt_noalloc_setup: the scratch array has no device address");
    g_scratch_of_device[device] = (NoAllocScratch *)ptr;
    return 0;
}

// the flags and the hand-over count start at zero (a fill kernel, not a memset node: callers may be capturing)
__global__ __launch_bounds__(256) void k_noalloc_clear(uint8_t *__restrict__ out, int64_t n, unsigned long long *late_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *late_count = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0;
}

extern "C" int rt_obstruct_segments_noalloc(rt_scene *scene, const double *d_segs, int64_t n_seg, const uint8_t *d_mask,
                                            uint8_t *d_out, void *stream) {
    if (!scene) RT_FAIL("rt_obstruct_segments_noalloc: null scene");
    if (n_seg < 0) RT_FAIL("rt_obstruct_segments_noalloc: negative count");
    if (n_seg > RT_NOALLOC_MAX_SEGMENTS)
        RT_FAIL("rt_obstruct_segments_noalloc: " + std::to_string(n_seg) + " segments, at most " +
                std::to_string(RT_NOALLOC_MAX_SEGMENTS) + " per call (the size of the library's fixed scratch array)");
    if (n_seg == 0) return 0;
    if (!d_segs || !d_out) RT_FAIL("rt_obstruct_segments_noalloc: null array");
    NoAllocScratch *scr = scene->device >= 0 && scene->device < RT_NOALLOC_MAX_DEVICES ? g_scratch_of_device[scene->device] : nullptr;
    if (!scr) RT_FAIL("rt_obstruct_segments_noalloc: no scratch array on device " + std::to_string(scene->device));
    RT_HIP(rt_use_device(scene->device));
    hipStream_t st = (hipStream_t)stream;
    const int block = 256;
    int64_t grid = (n_seg + block - 1) / block;
    if (grid > (1 << 20)) grid = 1 << 20;
    hipLaunchKernelGGL(k_noalloc_clear, dim3((unsigned)(grid < 2048 ? grid : 2048)), dim3(block), 0, st, d_out, n_seg,
                       &scr->late_count);
    ObstructItems items{d_segs, mk3(0, 0, 0), nullptr, nullptr, 0, d_mask ? n_seg : 0, nullptr, nullptr, 0};
    // first pass: every wave stops after OBS_FIRST_PASS_TRIPS records per lane and lists the segments still walking
    if (d_mask)
        hipLaunchKernelGGL(k_obstruct_grid<true>, dim3((unsigned)grid), dim3(block), 0, st, scene->view, items, n_seg,
                           (const unsigned long long *)nullptr, d_mask, d_out, OBS_FIRST_PASS_TRIPS, &scr->late_count,
                           scr->late_items, (int32_t *)nullptr, 0);
    else
        hipLaunchKernelGGL(k_obstruct_grid<false>, dim3((unsigned)grid), dim3(block), 0, st, scene->view, items, n_seg,
                           (const unsigned long long *)nullptr, d_mask, d_out, OBS_FIRST_PASS_TRIPS, &scr->late_count,
                           scr->late_items, (int32_t *)nullptr, 0);
    // second pass: the listed segments, walked to the end (the count is read on the device)
    items.index = scr->late_items;
    const unsigned g2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(grid, std::max<int64_t>(1024, grid >> 2)));
    if (d_mask)
        hipLaunchKernelGGL(k_obstruct_grid<true>, dim3(g2), dim3(block), 0, st, scene->view, items, n_seg,
                           (const unsigned long long *)&scr->late_count, d_mask, d_out, 0x7fffffff,
                           (unsigned long long *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 0);
    else
        hipLaunchKernelGGL(k_obstruct_grid<false>, dim3(g2), dim3(block), 0, st, scene->view, items, n_seg,
                           (const unsigned long long *)&scr->late_count, d_mask, d_out, 0x7fffffff,
                           (unsigned long long *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 0);
    RT_HIP(hipGetLastError());
    return 0;
}
