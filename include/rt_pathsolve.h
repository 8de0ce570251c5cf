/*
 * C ABI of the MI355X path-solve library (libraytracing_amd.so).
 *
 * The reference (MaximeLeu/RayTracing) has no FFI layer: its path-solve path is plain Python
 * (raytracing/geometry.py, raytracing/ray_tracing.py, electromagnetism_fun/electromagnetism.py).  Each entry point
 * below names the reference function(s) it stands for; a Python binding only needs ctypes.
 *
 * Conventions
 *   - every function returns 0 on success and a negative code on failure; rt_last_error() gives the message of the
 *     calling thread's last failure; functions returning a handle return NULL on failure;
 *   - all arrays are dense, row-major, caller-owned; doubles are IEEE binary64, indices int32 unless stated;
 *   - `on_device` = 0: the array pointers of that call are host memory (the library copies);
 *     `on_device` = 1: they are device memory of the scene's GPU (nothing is copied, work is queued on the
 *     scene's stream and the call returns after the stream has been synchronised).  The scene's stream is a
 *     non-blocking stream of its own: it is NOT ordered after the caller's streams, the null stream included.
 *     Inputs that kernels or copies of the caller are still producing must be complete first: synchronise the
 *     producing stream, or call rt_scene_wait_for_stream(scene, producing_stream) before the entry point.  Outputs
 *     are complete, for every stream, when the call returns;
 *   - a handle may be used by one host thread at a time; handles of different GPUs are independent.
 */
#ifndef RT_PATHSOLVE_H
#define RT_PATHSOLVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_scene rt_scene;
typedef struct rt_candidates rt_candidates;

#define RT_MAX_ORDER 4 /* interactions per path the solve kernels are instantiated for */

const char *rt_last_error(void);
int rt_device_count(void);

/* ---- scene --------------------------------------------------------------------------------------------------
 * The polygon list RayTracingProblem.__init__ takes from place.get_polygons_list() (ray_tracing.py:40), flattened:
 * vert_off[N+1] CSR offsets, verts[Vtot*3], planes[N*4] = OrientedPolygon.get_parametric() of each polygon
 * (geometry.py:1747-1765, with the sign fix-ups of geometry.py:2360-2363 already applied), material_id[N] index into
 * the ITU material table (materials_properties.py:47-104), building_id[N] (geometry.py:2139).  Host pointers.
 * Frames (geometry.py:1819-1853) and bounding boxes (geometry.py:1437-1465) are derived inside.
 * material_id and building_id may be NULL (concrete; no buildings).  building_id changes no result: polygons with the
 * same id are taken to be the faces of one building, and a diffracted path that leaves an edge into its own building is
 * tested against those faces before it is queued (same tests, same paths, a seventh of the lln map's step saved). */
rt_scene *rt_scene_create(int device, int32_t n_poly, const int32_t *vert_off, const double *verts,
                          const double *planes, const int32_t *material_id, const int32_t *building_id);
void rt_scene_destroy(rt_scene *scene);
int32_t rt_scene_polygon_count(const rt_scene *scene);
/* Orders the scene's stream after everything queued so far on `stream` (a hipStream_t of the scene's GPU; NULL = the
 * null stream): an event recorded there, which the scene's stream waits for on the device; the host does not wait.
 * For on_device = 1 calls whose inputs the caller's own kernels produce.  No reference counterpart. */
int rt_scene_wait_for_stream(rt_scene *scene, void *stream);

/* No counterpart in the reference: a serving loop that calls the path description, the field step and the receiver
 * records one after the other with device pointers (on_device = 1) pays for three waits in which the GPU has nothing
 * queued.  With deferred synchronisation on, those three calls only queue their work on the scene's stream and return;
 * rt_scene_synchronize waits for all of it and reports a failed launch.  Their outputs and the timing of the field
 * step are valid after that call only.  The host-array forms (on_device = 0) wait as before.
 * (ray_tracing.py:201-390 and electromagnetism.py:216-420 run these steps strictly one after the other on the host.) */
int rt_scene_set_deferred_sync(rt_scene *scene, int on);
int rt_scene_synchronize(rt_scene *scene);

/* ---- obstruction: polygons_obstruct_line_path (geometry.py:836-898) ----------------------------------------------
 * segs[n_seg*6] = (A, B) per segment; mask[N] or NULL restricts the test to polygons with mask != 0 (the LOS step
 * passes the emitter visibility, ray_tracing.py:232-236); out[n_seg] = 1 when some polygon obstructs. */
int rt_obstruct_segments(rt_scene *scene, const double *segs, int64_t n_seg, const uint8_t *mask, uint8_t *out,
                         int on_device);
/* rt_obstruct_segments for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference.  Device pointers only (segs[n_seg*6], mask[N] or NULL,
 * out[n_seg]); calls no hipMalloc / hipFree; the kernels are queued on `stream` (a hipStream_t; NULL = the default
 * stream) and the call returns WITHOUT waiting for them.  Same flags as rt_obstruct_segments.  The two passes of the
 * grid walk share a hand-over list that lives in a device array of fixed size owned by the library (16 MiB per GPU,
 * held until the process ends), so n_seg is limited to 4 194 304 per call (a larger count is refused), and two calls
 * whose kernels overlap in time on one GPU race on that array: run them one after the other (same stream, or ordered
 * by events). */
int rt_obstruct_segments_noalloc(rt_scene *scene, const double *segs, int64_t n_seg, const uint8_t *mask, uint8_t *out,
                                 void *stream);
/* return_polygons=True form (geometry.py:868-896): out[N] = 1 for every polygon that cuts the one segment. Host. */
int rt_segment_hits(rt_scene *scene, const double *seg, uint8_t *out);

/* ---- containment: OrientedPolygon.contains_point (geometry.py:1778-1803) -------------------------------------- */
int rt_contains_points(rt_scene *scene, int32_t polygon, const double *points, int64_t n_points, int check_in_plane,
                       double plane_tol, uint8_t *out, int on_device);

/* ---- visibility: polygon_visibility_vector / polygons_visibility_matrix (geometry.py:924-1048) --------------------
 * face_verts[n_face_verts*3], face_normal[3] = the face's get_normal(); self_index = its position in the scene or -1
 * (an emitter screen, ray_tracing.py:216-219); out[N] receives 0/1 (zero-initialised, unlike geometry.py:953-954).
 * rt_visibility_matrix writes the symmetrised N*N matrix (geometry.py:1046). */
int rt_face_visibility(rt_scene *scene, const double *face_verts, int32_t n_face_verts, const double *face_normal,
                       int32_t self_index, int strict, uint8_t *out);
/* several faces in one launch (the six screens of the emitter's cube): face_off[n_faces+1] offsets into face_verts,
 * face_normals[n_faces*3], self_index[n_faces] or NULL (= all -1), out[n_faces*N]. */
int rt_faces_visibility(rt_scene *scene, int32_t n_faces, const int32_t *face_off, const double *face_verts,
                        const double *face_normals, const int32_t *self_index, int strict, uint8_t *out);
int rt_visibility_matrix(rt_scene *scene, int strict, uint8_t *out, int on_device);
/* rows of that matrix before it is symmetrised (geometry.py:1036-1046 for the listed i only): out[n_rows*N], row k =
 * what polygon row_ids[k] sees.  matrix = rows | rows^T once every row is there; a node deals the rows to its GPUs. */
int rt_visibility_rows(rt_scene *scene, int32_t n_rows, const int32_t *row_ids, int strict, uint8_t *out);
/* The matrix (row_ids NULL: rows_out[N*N], symmetrised, as rt_visibility_matrix) or the listed rows (rows_out[n_rows*N],
 * as rt_visibility_rows) together with the visibility vectors of n_faces extra faces (faces_out[n_faces*N], arguments as
 * rt_faces_visibility): the two steps RayTracingProblem.precompute takes one after the other (ray_tracing.py:208, then
 * :210-228 for the emitter's screens).  The screens are painted on a stream of their own while the matrix is painted,
 * instead of holding six compute units busy afterwards.  Host pointers, except the outputs named by outputs_on_device:
 * bit 0 -- rows_out, bit 1 -- faces_out is memory of the scene's GPU, written in place (the matrix then never visits
 * the host and rt_enumerate can read it there, vis_on_device; a rank of a sharded run has its rows and screens painted
 * straight into the buffer its all-gather sends).  Same results as the separate calls. */
int rt_visibility_with_faces(rt_scene *scene, int32_t n_rows, const int32_t *row_ids, int strict, uint8_t *rows_out,
                             int32_t n_faces, const int32_t *face_off, const double *face_verts,
                             const double *face_normals, const int32_t *self_index, int faces_strict, uint8_t *faces_out,
                             int outputs_on_device);
/* Obstruction tests either walk every polygon (mode 1) or a uniform grid over the scene (mode 2); mode 0, the
 * default, picks by scene size.  Both give the same flags; the switch exists so that tests can check one against the
 * other (no reference counterpart). */
int rt_debug_set_occlusion_mode(rt_scene *scene, int mode);
/* The reflection kernel meets the receivers one by one (mode 1) or sorted into spatial tiles, rejecting whole tiles
 * with an interval form of its pre-test (mode 2); mode 0, the default, picks by batch size.  Same paths either way;
 * the switch exists so that tests can check one against the other (no reference counterpart). */
int rt_debug_set_reflect_mode(rt_scene *scene, int mode);
/* In a large batch the diffraction kernel tries each pair's last segment on the wall that stands next to the receiver in
 * the segment's direction before it queues the pair (on = 1, the default), or queues every pair for the obstruction
 * walk (on = 0).  Same paths either way; the switch exists so that tests can check one against the other (no reference
 * counterpart). */
int rt_debug_set_wall_hints(rt_scene *scene, int on);
/* First size of the solve's path queue in records (0 = chosen from the batch).  A queue that turns out too small is
 * grown and the chunk redone; the switch exists so that tests can take that route (no reference counterpart). */
int rt_debug_set_path_queue_capacity(rt_scene *scene, int64_t n_paths);
/* Per-kernel-family timing of rt_solve_paths (rt_solve_last_timing) on (default) or off.  The timing is a pair of
 * events around every launch; a small solve is bound by the latency of its few dependent launches and runs a little
 * faster without them.  rt_solve_last_timing reports zeros while it is off (no reference counterpart). */
int rt_scene_set_timing(rt_scene *scene, int on);
/* rt_enumerate stores the polygon chains of every candidate group except, when it would take more than this many bytes
 * (default 256 MiB), the deepest reflection level: that level's chains are then formed inside the kernels from the level
 * before it (same candidates, same order, same indices).  0 = never store the deepest level.  For tests of that route.
 * No reference counterpart. */
int rt_debug_set_stored_chain_limit(rt_scene *scene, int64_t bytes);
/* The painter cuts a face that has at least min_candidates candidate polygons into tiles (rectangles of its
 * (azimuth, polar) view) of about per_tile candidates each, painted by different workgroups (a polygon is visible iff
 * it shows in some tile); per_tile = 0 paints every face whole; min_candidates = -1 (the default) lets each call
 * choose: tiles when it paints at most 256 faces (the emitter's screens), whole faces when it paints a whole matrix,
 * tiles for the largest faces of a share of one.  Same vectors either way; the switch exists so that tests can check
 * one against the other on small scenes (no reference counterpart). */
int rt_debug_set_visibility_tiling(rt_scene *scene, int32_t min_candidates, int32_t per_tile);
/* Device time, in milliseconds, of the last visibility call on this scene (HIP events on the stream the kernels ran on,
 * while rt_scene_set_timing is on): ms[0] mapping to (azimuth, polar) and ring validity, ms[1] sorting by the painter's
 * key, ms[2] painting, ms[3] all of it, for the rows / the matrix; ms[4..7] the same for the extra faces of
 * rt_visibility_with_faces or for rt_faces_visibility.  No reference counterpart. */
int rt_visibility_last_timing(rt_scene *scene, double *ms8);
/* Work counters of the visibility painter accumulated since the last call, then zeroed (profiling aid, no reference
 * counterpart): candidates, listed canvas polygons, canvas edges entering a candidate, edge cuts, candidates settled by
 * the interior-sample shortcut, polygons shown visible, canvas-edge examinations, edge examinations. out8[8]. */
int rt_debug_visibility_stats(unsigned long long *out8);

/* ---- sharp edges: polygons_sharp_edges_iter (geometry.py:482-565) -----------------------------------------------
 * Writes up to `capacity` edges in ascending (i, j) order: pairs[2*e] = (i, j), edges[6*e] = the shared edge with its
 * lexicographically smaller endpoint first (array_utils.py:9-41).  *count = number found (may exceed capacity: call
 * again with a larger buffer).  Host pointers. */
int rt_sharp_edges(rt_scene *scene, double min_angle_deg, int32_t *pairs, double *edges, int64_t capacity,
                   int64_t *count);

/* ---- stand-alone solvers (geometry.py:660-687, 718-748, 800-833), batched over destinations ------------------------
 * planes[k*4]; out_points[n_dest*k*3] (k reflections) or [n_dest*(k+1)*3] (k reflections then the diffraction point);
 * out_ok[n_dest] = 1 when a specular solution exists (what sol.success means for a solver that cannot mis-converge).
 * Host pointers; runs on `device`. */
int rt_solve_reflections(int device, const double *origin, const double *dests, int64_t n_dest, const double *planes,
                         int32_t k, double *out_points, uint8_t *out_ok);
int rt_solve_reflections_diffraction(int device, const double *origin, const double *dests, int64_t n_dest,
                                     const double *planes, int32_t k, const double *edge, double *out_points,
                                     uint8_t *out_ok);
/* The same three solver functions answered as the reference's own solver answers them: SciPy's root with its default
 * method hybr (MINPACK hybrd; hybrj with the reference's hand-written derivative for the pure diffraction, geometry.py:
 * 710-713, 744), from the reference's first guess (geometry.py:624-630, 742, 821-824), restated on the GPU operation
 * for operation.  k planes (0 for a pure diffraction), edge NULL for pure reflections.  out_points as above: the
 * solver's sol.x reshaped as the reference returns it (the diffraction point mapped back to the world);
 * out_success[n_dest] = sol.success; out_nfev[n_dest] (may be NULL) = the evaluations the method itself made, which is
 * SciPy's sol.nfev less the two calls its wrapper makes to check shapes.  Host pointers; runs on `device`. */
int rt_solve_like_reference(int device, const double *origin, const double *dests, int64_t n_dest, const double *planes,
                            int32_t k, const double *edge, double *out_points, uint8_t *out_success, int32_t *out_nfev);

/* ---- candidate enumeration: the depth-first loops of RayTracingProblem.solve (ray_tracing.py:299-388) -------------
 * vis[N*N] (visibility matrix), ev[N] (emitter visibility), edge_pairs[n_edges*2] / edges[n_edges*6] (sharp edges in
 * dictionary order) — host pointers; with vis_on_device != 0 vis is memory of the scene's GPU instead, read during the
 * call only.  Groups, in the order the reference visits them:
 *   group g = 0 .. max_order-1            : g+1 reflections                 (ray_tracing.py:324-327)
 *   group max_order                       : 1 diffraction                   (ray_tracing.py:334-350)
 *   group max_order + m, m = 1..max_order-1 : m reflections + 1 diffraction (ray_tracing.py:355-388)
 * Within a group candidates are in the reference's visiting order.  with_diffraction = 0 leaves the diffraction
 * groups empty (reflection-only workloads). */
rt_candidates *rt_enumerate(rt_scene *scene, const uint8_t *vis, const uint8_t *ev, const int32_t *edge_pairs,
                            const double *edges, int64_t n_edges, int32_t max_order, int with_diffraction,
                            int vis_on_device);
void rt_candidates_destroy(rt_candidates *cand);
int32_t rt_candidates_group_count(const rt_candidates *cand);
/* counts[group_count] = candidates per group */
int rt_candidates_counts(const rt_candidates *cand, int64_t *counts);
/* chains[count*n_refl] polygon indices and edge_ids[count] (-1 for reflection groups) of one group, host pointers;
 * either may be NULL.  A group of more than 2^31 - 1 candidates is not listed (an error). */
int rt_candidates_group(const rt_candidates *cand, int32_t group, int32_t *chains, int32_t *edge_ids);

/* ---- the solve: LOS + every (receiver, candidate) (ray_tracing.py:232-237, 243-390) ---------------------------------
 * tx[3], rx[n_rx*3].  Valid paths are appended, in no particular order, to
 *   out_rx[cap] receiver index, out_group[cap] group (-1 = line of sight), out_cand[cap] index inside the group,
 *   out_points[cap*RT_MAX_ORDER*3] interaction points (first n of each row used, the rest zero).
 * An index of 2^31 or more (a deepest reflection level that is not stored can be that large: order 4 on a town) is split
 * over the two words: out_cand holds its low 31 bits, bits 8 and up of out_group the rest, the low 8 bits of out_group
 * the group; for every smaller index, i.e. whenever out_group is below 256, the words are group and index as they are.
 * rt_describe_paths takes the words as they come.
 * *n_found = number of valid paths (if > capacity the surplus was dropped: call again with more room);
 * *n_attempted = n_rx * (1 + total candidates), the path-solve count of the metric.
 * on_device selects where rx and the out_* arrays live; the two counters are always host. */
int rt_solve_paths(rt_scene *scene, rt_candidates *cand, const double *tx, const double *rx, int64_t n_rx,
                   int32_t *out_rx, int32_t *out_group, int32_t *out_cand, double *out_points, int64_t capacity,
                   int64_t *n_found, int64_t *n_attempted, int on_device);
/* Reference-compatible mode of rt_solve_paths (off by default).  The reference keeps a candidate only when SciPy's hybr
 * reports success on its system (ray_tracing.py:309, 345, 371 with geometry.py:682, 744, 823), and hybr misses some
 * roots; the kernels find every root in closed form.  With the mode on, every path found is put through a restatement
 * of that solver (MINPACK hybrd / hybrj on the reference's zero functions, first guess, tolerances and arithmetic, on
 * the GPU) and kept only when it reports success and its points pass the reference's checks; the points returned are
 * then the solver's.  n_found counts the kept paths. */
int rt_scene_set_reference_solver(rt_scene *scene, int on);
/* Device time, in milliseconds, and launch count of each kernel family during the last rt_solve_paths call on this
 * scene (measured with HIP events on the scene's stream): [0] reflection candidates, [1] diffraction candidates,
 * [2] segment obstruction, [3] finalisation.  ms[4], launches[4].  *segments_tested = segments the obstruction kernels
 * walked through the scene (a path that is found blocked on its last segment has its other segments never walked,
 * and they are not counted). */
int rt_solve_last_timing(rt_scene *scene, double *ms, int64_t *launches, int64_t *segments_tested);
/* The re-judging step of the last rt_solve_paths call in reference-compatible mode (rt_scene_set_reference_solver):
 * *ms = device time of its kernels (HIP events on the scene's stream; zero with the timing off), *n_in = paths the
 * closed-form solve had found, *n_kept = paths the reference's solver and checks kept (ray_tracing.py:309-314,
 * 345-350, 371-376).  All zero when the mode is off.  Any pointer may be NULL. */
int rt_solve_last_rejudge(rt_scene *scene, double *ms, int64_t *n_in, int64_t *n_kept);

/* ---- field step: my_field_computation (electromagnetism.py:648-802) --------------------------------------------------
 * A path is described by
 *   path_rx[i]            receiver index
 *   path_kind[i]          number of reflections + 256 if it ends with a diffraction (0 = line of sight)
 *   path_chain[i*RT_MAX_ORDER + k]   polygon of reflection k
 *   path_edge_polys[i*2]  the two polygons of the diffracting edge (sharp-edge key order), path_edge_points[i*6] its
 *                         endpoints
 *   path_points[i*RT_MAX_ORDER*3]    interaction points
 * i.e. what the reference's reflections / diffractions containers hold (ray_tracing.py:314, 350, 376).
 * rt_describe_paths fills kind / chain / edge arrays from the (group, candidate) pairs rt_solve_paths returned.
 * rt_fields: per path and frequency field[3] complex (re, im interleaved, world frame) and power (em:779-787); per
 * path tx/rx angles in degrees [tx_el, tx_az, rx_el, rx_az] (em:566-575) and time of flight; per receiver and
 * frequency 10 log10(sum of powers / p_in) (electromagnetism_utils.py:104-115, 180-188), -inf without paths;
 * status[i] != 0 where the reference would raise an AssertionError: 1 incidence angle (em:294), 2 Fresnel
 * coefficient > 1 (em:268), 3 reflected stronger than incident (em:342), 4 wedge angle (em:408), 5 beta_0 (em:413),
 * 6 ray inside the wedge (em:425-426), 7 edge orientation (em:470), 8 diffracted stronger than incident (em:502),
 * 9 concave geometry in the tree loss (geometry.py:916), 10 point at the antenna (electromagnetism_utils.py:151).
 * Antennas are aligned per receiver on its shortest path, a line of sight first (em:610-644, 768-771).
 * Up to 16 frequencies per call; the material table is ITU-R P.2040-2 as in materials_properties.py:47-109. */
typedef struct rt_antenna_params {
    double tx_gain, rx_gain, p_in; /* materials_properties.py:32, 41 */
    int32_t isotropic;             /* materials_properties.py:36 */
    int32_t alpha;                 /* materials_properties.py:37 */
} rt_antenna_params;
/* Optional inputs of rt_fields (NULL pointer or NULL members = the defaults of my_field_computation).  They exist so that
 * the reference's per-step entry points can be served by the same kernel: ElectromagneticField.from_path (em:156-176:
 * antenna aimed by the caller), Reflection.my_reflect (em:314-349: incident field given, first point of the path given)
 * and Diffraction.my_diff (em:453-507).
 *   aim_tx / aim_rx [n_rx*3]  points the two antennas are aimed at, per receiver (both or neither)
 *   path_origin [n_paths*3]   first point of each path instead of tx
 *   path_e_in   [n_paths*6]   complex field (re, im interleaved) arriving at the first interaction instead of the source
 *                             field of the emitter's antenna
 *   n_late_polygons           the scene's last n_late_polygons polygons were added after the paths were found: the
 *                             crowns my_field_computation puts on the tree trunks (em:745, geometry.py:2789-2802).
 *                             Every segment of every path is attenuated by the crowns it crosses (em:179-199 called at
 *                             em:338, 348, 506, 725); a line of sight by whatever it crosses, as before. */
typedef struct rt_field_options {
    const double *aim_tx, *aim_rx, *path_origin, *path_e_in;
    int32_t n_late_polygons;
} rt_field_options;
/* Device time of the last rt_fields call on this scene, in milliseconds (HIP events on the scene's stream, while
 * rt_scene_set_timing is on): ms[0] the field kernel, ms[1] everything the call queued.  No reference counterpart. */
int rt_fields_last_timing(rt_scene *scene, double *ms2);
int rt_describe_paths(rt_candidates *cand, const int32_t *path_group, const int32_t *path_cand, int64_t n_paths,
                      int32_t *out_kind, int32_t *out_chain, int32_t *out_edge_polys, double *out_edge_points,
                      int on_device);
int rt_fields(rt_scene *scene, const double *tx, const double *rx, int64_t n_rx, const int32_t *path_rx,
              const int32_t *path_kind, const int32_t *path_chain, const int32_t *path_edge_polys,
              const double *path_edge_points, const double *path_points, int64_t n_paths, const double *freq_hz,
              int32_t n_freq, const rt_antenna_params *antenna, const rt_field_options *options, double *out_field,
              double *out_power, double *out_angles, double *out_time, int32_t *out_status, double *out_power_db,
              int on_device);


/* ---- one fixed-size record per receiver: what the ranks of a multi-GPU coverage map exchange instead of their paths
 * (the device-side stand-in for merge_solved_problems, multithread_solve.py:45-60).  Device pointers of the scene's GPU
 * only: power_db[n_rx*n_freq] as rt_fields left it, path_rx / path_kind / path_status[n_paths] (status may be NULL).
 * out[n_rx * rt_receiver_record_bytes(n_freq)]: per receiver n_freq doubles (dB per frequency), then int32: paths per
 * path type (line of sight; 1..RT_MAX_ORDER reflections; 0..RT_MAX_ORDER-1 reflections and a diffraction), paths in all,
 * paths with a non-zero field status, largest status code; padded to a multiple of 8 bytes (56 at one frequency).
 * Queued on the scene's stream and waited for. */
int32_t rt_receiver_record_bytes(int32_t n_freq);
int rt_receiver_records(rt_scene *scene, int64_t n_rx, int32_t n_freq, const double *power_db, int64_t n_paths,
                        const int32_t *path_rx, const int32_t *path_kind, const int32_t *path_status, unsigned char *out);

/* ---- text encoding of the result table's array columns (electromagnetism_utils.py:104-117 / :217-269: the CSV holds
 * each small array as numpy.array2string(a, separator=",") without spaces).  values[n_rows*n_cols] real or
 * [n_rows*n_cols*2] complex (re, im interleaved), n_cols <= 16; the rows' strings are written one after the other, each
 * followed by '\n', into out[capacity] (64 bytes per real value always suffice); a row holding a non-finite value gets
 * an empty string.  Returns the bytes written or -1.  Host code, no GPU involved. */
int64_t rt_format_array_rows(const double *values, int64_t n_rows, int32_t n_cols, int32_t is_complex, char *out,
                             int64_t capacity);

/* ---- JSON text of a container of solved paths (RayTracingProblem.to_json, ray_tracing.py:66-132, writes
 * `reflections` and `diffractions` with json.dump: {receiver: {order: [[points, polygon chain(, edge key, edge
 * points)], ...]}}).  The paths come grouped by receiver in the order of keys[n_keys] (every key gets an entry, empty
 * when it has no path) and within a receiver by order; path i has n_points[i] points of three doubles at
 * points[i*point_stride ..] (emitter and receiver included), n_chain[i] polygon indices at chain[i*chain_stride ..] and,
 * when edge_key and edge_points are given, its edge's two polygons edge_key[2i ..] and end points edge_points[6i ..].
 * Writes into out[capacity] the characters json.dumps gives for that dictionary; returns their number, -1 for bad
 * arguments, -2 when the paths are not grouped as said, and -(needed + 16) when capacity is too small.  Host code. */
int64_t rt_format_paths_json(int64_t n_paths, const int64_t *path_rx, const int32_t *path_order, const int32_t *n_points,
                             const double *points, int32_t point_stride, const int32_t *n_chain, const int32_t *chain,
                             int32_t chain_stride, const int32_t *edge_key, const double *edge_points, int64_t n_keys,
                             const int64_t *keys, char *out, int64_t capacity);

/* ---- channel statistics of the result table (electromagnetism_fun/electromagnetism_plots.py:287-460) -----------------
 * Per receiver g, over its paths [group_off[g], group_off[g+1]) (rows of the table in their order, receivers without
 * a path have an empty range): compute_RMS_delay_spreads (:389-400), compute_RMS_angular_spreads (:403-431),
 * compute_percent_coming_from_n_rays (:445-460, n = n_strongest) and compute_angular_windows (:287-387,
 * POWER_PERCENT = power_percent, including the way the reference picks the bundle's centre and reads the two angles).
 * Per path: power [W], time [s], arrival elevation / azimuth [deg] (the table's rx_angles), xyz[3] = the point
 * (sin el cos az, sin el sin az, cos el) as the reference forms it, is_los (path_type == "LOS").
 * out[g*8 + ..]: 0 RMS delay spread [s], 1 RMS azimuth spread, 2 RMS elevation spread [deg], 3 share of the
 * n_strongest strongest rays [%], 4 1.0 if the receiver has a line of sight (the reference reports the share for those
 * only), 5 window azimuth, 6 window elevation [deg] (NaN where the reference fails: a bundle of one ray), 7 number of
 * paths.  Receivers without a path get NaN in 0-6.  All pointers host, or all device with on_device. */
int rt_channel_stats(int device, int64_t n_groups, const int64_t *group_off, const double *power, const double *time,
                     const double *el_deg, const double *az_deg, const double *xyz, const uint8_t *is_los,
                     int32_t n_strongest, double power_percent, double *out, int on_device);

/* ---- radiomap of the result table (electromagnetism_fun/scenario_measures.py:31-88) ----------------------------------
 * What compare_radiomaps computes before it draws: z = griddata(receivers, powers, (x, y), method="linear",
 * fill_value) (:66) and mask = street.contains(Point(x, y)) (:54) for every point of the regular grid (:49-53).
 * values[n_points]: per-receiver power [dB].  The Delaunay triangulation of the receivers is the caller's (Qhull, as
 * scipy holds it): triangles[n_triangles*3] vertex indices and transform[n_triangles*6] = per triangle the row-major
 * inverse of [a - c, b - c] followed by c (scipy.spatial.Delaunay.transform; NaN marks a degenerate triangle).
 * query[n_query*2]: grid points.  out_values[n_query] (or NULL): barycentric interpolation in the triangle that holds
 * the point to 100 machine epsilons, summed in scipy's order without fused operations; fill_value outside every
 * triangle.  ring[n_ring*2] (no closing vertex, no holes) and out_inside[n_query] (or both NULL): 1 where the point is
 * strictly inside the ring, 0 on its boundary and outside (shapely's contains).  Host pointers. */
int rt_radiomap(int device, int64_t n_points, const double *values, int64_t n_triangles, const int32_t *triangles,
                const double *transform, int64_t n_query, const double *query, double fill_value, double *out_values,
                int32_t n_ring, const double *ring, uint8_t *out_inside);

#ifdef __cplusplus
}
#endif
#endif
