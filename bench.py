"""Benchmark of the path-solve hot path on MI355X.

Workload (config 3 of BASELINE.json, the one the metric's target names): lln.geojson, one emitter, a 100 x 100 receiver
lattice, reflections and diffractions up to order 2.  With N GPUs the 10 000 receivers are split into N contiguous
blocks (strong scaling: the whole job is fixed), the scene is replicated, and

    one step = for this rank's block, all on the GPU: lines of sight + every (receiver, candidate) path-solve with its
               checks, the description of the surviving paths, the field step (fields, powers, per-receiver dB), the
               condensation into one fixed-size record per receiver, and ONE all-gather of those records (RCCL), so
               that every rank ends the step holding the whole coverage map.

The precompute of the scene (device scene, sharp edges, visibility matrix and the emitter's screens — their rows dealt
to the ranks and all-gathered — and the candidate enumeration) happens once before the steps; it is timed on its own and
reported with one step as ``coverage_map_wall_s`` (scene file -> per-receiver dB on every rank).

Prints ONE JSON line: whole-job path-solves/s, the roofline of the kernel with the most device time (device times from
HIP events inside the library, on the stream the kernels run on), the other kernels' rooflines, and — at N = 1 — a
secondary block for config 2 (place_levant, 1 024 receivers, reflections to order 2) and a CPU baseline (the oracle's
restatement of the reference's solve, SciPy hybr per candidate, on this host's cores, on a bounded sample).

``--gpus N`` with N > 1 and no launcher around it (no WORLD_SIZE): N worker processes are started here, one per GPU,
before this process has touched a GPU, and rank 0's line is passed on.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (half the 157.3 TFLOP/s fp32 vector rate)
HBM_PEAK_GBS = 8000.0
FREQUENCY_HZ = 12.5e9  # materials_properties.py:23 of the reference
T_PROCESS = 0.0  # when this process entered main()


def committed_counters():
    """Per-kernel counter figures of the committed rocprofv3 --pmc passes of this script (scripts/profile_round3.sh runs
    them, scripts/counters_json.py writes the file): the double-precision operations a launch executes, its HBM bytes
    with FETCH_SIZE doubled as gfx950 needs, launches per step, and where the waves' cycles go."""
    try:
        with open(os.path.join(ROOT, "profiles", "r3_counters.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def launch_workers(args):
    """--gpus N > 1 without a launcher: start the N ranks as fresh processes (this one has made no GPU call) and pass
    rank 0's JSON line on."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not lines:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (exit code {proc.returncode})")
    print(lines[-1])


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (test infrastructure) timed on this host's cores; never part of the measured GPU path
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline_worker(args):
    """Oracle path-solves for one receiver on a sample of the candidates of every group, for about `budget` s."""
    import warnings

    warnings.filterwarnings("ignore")
    polys_data, tx, rx, sample, budget = args
    from oracle import geometry_oracle as go
    from oracle import ray_tracing_oracle as ro

    polys = []
    for pts, par in polys_data:
        p = go.Poly(pts)
        p.parametric = par
        polys.append(p)
    o = ro.RayTracingOracle(tx, polys, rx.reshape(1, 3), precompute=False)
    t0 = time.perf_counter()
    n = 0
    for chain, edge in sample:
        planes = [polys[i].parametric for i in chain]
        if edge is None:  # ray_tracing.py:307-314
            points, sol = go.reflexion_points_from_origin_destination_and_planes(tx, rx, planes)
            if sol.success:
                lines = np.vstack([tx.reshape(1, 3), points, rx.reshape(1, 3)])
                o.check_reflections(lines, list(chain))
        else:  # ray_tracing.py:341-350, 369-376
            points, sol = go.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(tx, rx, planes, edge)
            if sol.success:
                lines = np.vstack([tx.reshape(1, 3), np.asarray(points).reshape(-1, 3), rx.reshape(1, 3)])
                o.check_reflections_and_diffraction(lines, list(chain), edge)
        n += 1
        if time.perf_counter() - t0 > budget:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(place, tx, rx, cand, budget_s=5.0, receivers_per_core=2, per_receiver=4000):
    """The oracle's restatement of the reference's solve (SciPy hybr per candidate, then the checks) on a sample of the
    candidates of EVERY group, each group in proportion to its share of the workload."""
    import multiprocessing as mp

    # the GPU box gives one GPU's job a share of 16 host cores (its os.cpu_count() is the whole machine's)
    host_cpus = os.cpu_count() or 1
    cores = min(16, host_cpus)
    polys_data = [(p.points, p.get_parametric()) for p in place.get_polygons_list()]
    total = int(cand.counts.sum())
    sample, shares = [], []
    for g in range(cand.n_groups):
        count = int(cand.counts[g])
        if count == 0:
            continue
        n_refl, has_edge = cand.group_kind(g)
        take = max(1, round(per_receiver * count / total))
        stride = max(1, count // take)
        chains, edge_ids = cand.group(g)
        picked = 0
        for c in range(0, count, stride):
            chain = tuple(int(v) for v in chains[c]) if n_refl else ()
            sample.append((chain, cand.edges[edge_ids[c]].copy() if has_edge else None))
            picked += 1
        shares.append(f"{'R' * n_refl}{'D' if has_edge else ''} {picked} of {count}")
    # interleaved, so that a receiver that runs out of time has met the groups in proportion all the same
    order = np.random.default_rng(0).permutation(len(sample))
    sample = [sample[i] for i in order]
    rx_ids = np.linspace(0, len(rx) - 1, receivers_per_core * cores).astype(int)
    jobs = [(polys_data, tx, rx[i], sample, budget_s) for i in rx_ids]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    solves = sum(r[0] for r in res)
    busy = sum(r[1] for r in res) / cores  # time per core, all cores busy
    return {"value": solves / busy, "unit": "path-solves/s", "cores": cores, "host_cpu_count": host_cpus, "kind": "port",
            "sample": f"{len(jobs)} receivers x up to {len(sample)} candidates of the same workload, every group in proportion "
                      f"to its share ({'; '.join(shares)}) = {solves} path-solves; oracle = the reference's algorithm with SciPy "
                      f"hybr and its checks, one process per core on the {cores} cores this job may use (os.cpu_count() = "
                      f"{host_cpus}), {busy:.0f} s of work per core ({wall:.0f} s wall with process start-up)"}


# ---------------------------------------------------------------------------------------------------------------
# rooflines
# ---------------------------------------------------------------------------------------------------------------
# kernels whose counters make up a family of the step (names without template arguments, as scripts/counters_json.py keys them)
FAMILY_KERNELS = {
    # (k_rx_hints runs on the side stream beside the reflection kernel: its work falls into that kernel's time)
    "reflect": ("k_reflect_tiled", "k_reflect", "k_rx_count", "k_rx_scatter", "k_rx_boxes", "k_rx_hints"),
    "diffract": ("k_diffract", "k_diffract_setup", "k_rd_prepare", "k_rd_filter"),
    "obstruct": ("k_obstruct_grid", "k_obstruct", "k_list_clear_flags"),
    "rejudge": ("k_rejudge_solve", "k_rejudge_solve_late", "k_rejudge_classify", "k_rejudge_scatter", "k_rejudge_copy_back"),
    "fields": ("k_fields", "k_fields_pass2"),
}


def family_counters(counters, family):
    """Executed fp64 flop and HBM bytes per step of a family, from the committed counter passes: per launch figures times
    the launches per step of the same passes, summed over the family's kernels.  None when the passes are missing."""
    flop = hbm = 0.0
    seen = False
    detail = {}
    for k in FAMILY_KERNELS[family]:
        c = counters.get(k)
        if not c or "fp64_flop_per_launch" not in c:
            continue
        seen = True
        per_step = c.get("launches_per_step", 0.0)
        # (the second pass of the re-judging step runs ONE path per wave, its 64 lanes making the same run: the work
        # counts once, not 64 times)
        once = 1.0 / 64.0 if k == "k_rejudge_solve_late" else 1.0
        flop += c["fp64_flop_per_launch"] * per_step * once
        hbm += c.get("hbm_bytes_per_launch", 0.0) * per_step
        detail[k] = {"launches_per_step": per_step, "fp64_flop_per_launch": c["fp64_flop_per_launch"] * once,
                     "hbm_bytes_per_launch": c.get("hbm_bytes_per_launch"),
                     "lanes_on_share_of_valu_cycles": c.get("lanes_on_share_of_valu_cycles")}
    return (flop, hbm, detail) if seen else (None, None, {})


def kernel_rooflines(n_rx, counts, max_order, n_poly, n_vert, ms, segments_walked, field_bytes, counters, n_valid_paths):
    """One entry per kernel family of a step.  For the arithmetic-bound families `achieved` is the double-precision
    work the lanes EXECUTED (counter passes: fp64 instructions x lanes switched on, per launch x launches per step) over
    the family's mean device time in a step, measured live with HIP events; `frac` = achieved / vector fp64 peak, at most
    1 by construction.  The algorithmic count of the reference's brute force (40 flop per interaction of every
    (receiver, candidate) pair; 13 N flop per segment walked) over the same time is kept beside it as
    `speed_vs_brute_force`: the kernels reject whole receiver tiles and grid cells, so it exceeds 1.  The field kernel is
    priced against HBM by its algorithmic bytes (24 (k + 2) + 4 k + 160 per path of k interactions)."""
    algorithmic = {
        "reflect": 40.0 * n_rx * sum((g + 1) * counts[g] for g in range(max_order)),
        "diffract": 40.0 * n_rx * sum((m + 1) * counts[max_order + m] for m in range(max_order)),
        "obstruct": 13.0 * n_poly * segments_walked,
        "rejudge": None,
    }
    names = {"reflect": "k_reflect_tiled<2> (orders 1 and 2)", "diffract": "k_diffract<0>, k_diffract<1>",
             "obstruct": "k_obstruct_grid", "rejudge": "k_rejudge_solve<shape>, k_rejudge_solve_late<shape>",
             "fields": "k_fields"}
    out = {}
    for fam in ("reflect", "diffract", "obstruct", "rejudge"):
        t_ms = ms[fam]
        flop, hbm, detail = family_counters(counters, fam)
        e = {"kernel": names[fam], "bound": "valu_fp64", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "ms_per_step": t_ms}
        if flop is not None and t_ms > 0:
            e["work_per_step"] = flop
            e["work_kind"] = "fp64 operations executed by switched-on lanes (committed counter passes)"
            e["achieved"] = flop / (t_ms * 1e-3) / 1e12
            e["frac"] = e["achieved"] / FP64_VECTOR_PEAK_TFLOPS
            e["traffic"] = hbm
            e["traffic_note"] = "HBM bytes per step from the committed counter passes: (2 x FETCH_SIZE + WRITE_SIZE) x launches per step"
            e["hbm_achieved_GBs"] = hbm / (t_ms * 1e-3) / 1e9
            e["hbm_frac"] = e["hbm_achieved_GBs"] / HBM_PEAK_GBS
            e["counters_by_kernel"] = detail
            main = counters.get(FAMILY_KERNELS[fam][0 if fam != "rejudge" else 1], {})
            for k in ("wave_cycles_issuing_valu_share", "wave_cycles_issuing_share", "wave_cycles_waiting_share",
                      "wave_cycles_issue_stalled_share", "valu_instructions_per_wave", "fp64_share_of_valu_instructions"):
                if k in main:
                    e[k] = main[k]
        else:
            e.update(work_per_step=0.0, work_kind="no counter passes committed for this family", achieved=0.0, frac=0.0, traffic=None)
        if algorithmic[fam] is not None and t_ms > 0:
            e["algorithmic_flop_per_step"] = algorithmic[fam]
            e["speed_vs_brute_force"] = algorithmic[fam] / (t_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
        out[fam] = e
    out["obstruct"]["segments_walked_per_step"] = segments_walked
    out["obstruct"]["algorithmic_bytes_per_step"] = 49.0 * segments_walked + (64.0 * n_poly + 24.0 * n_vert)
    out["rejudge"]["note"] = ("the reference's solver call (MINPACK hybr restated) on every path found: a chain of dependent "
                              "fp64 operations per path, 10 to 1 000 evaluations long; the second pass runs the long runs one "
                              "per wave, so its time is the longest run's latency, not throughput")
    t_ms = ms["fields"]
    flop, hbm, detail = family_counters(counters, "fields")
    ach = field_bytes / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
    out["fields"] = {"kernel": names["fields"], "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": t_ms,
                     "work_per_step": field_bytes, "work_kind": "algorithmic bytes: 24 (k + 2) + 4 k + 160 per path",
                     "achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": hbm, "paths_per_step": n_valid_paths}
    if flop is not None and t_ms > 0:
        out["fields"]["fp64_executed_TFLOPs"] = flop / (t_ms * 1e-3) / 1e12
        out["fields"]["fp64_frac"] = out["fields"]["fp64_executed_TFLOPs"] / FP64_VECTOR_PEAK_TFLOPS
    return out


def once_per_scene_rooflines(counters, vis_timing):
    """The visibility kernels of the precompute: executed fp64 work of the committed counter passes (one precompute per
    profiled run) over the device time measured live."""
    out = {}
    for name, kernels, key in (("vis_prepare", ("k_vis_prepare", "k_vis_prepare_mid", "k_vis_prepare_big"), "prepare"),
                               ("vis_paint", ("k_vis_paint",), "paint")):
        t_ms = vis_timing["rows"][key] + vis_timing["faces"][key]
        flop = hbm = 0.0
        seen = False
        for k in kernels:
            c = counters.get(k)
            if c and "fp64_flop_per_launch" in c:
                seen = True
                n_launch = c["launches_per_step"] * c.get("steps_profiled", 1.0)
                flop += c["fp64_flop_per_launch"] * n_launch
                hbm += c.get("hbm_bytes_per_launch", 0.0) * n_launch
        ach = flop / (t_ms * 1e-3) / 1e12 if (seen and t_ms > 0) else 0.0
        out[name] = {"kernel": ", ".join(kernels), "bound": "valu_fp64", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "ms": t_ms, "rows_ms": vis_timing["rows"][key],
                     "screens_ms": vis_timing["faces"][key], "work": flop,
                     "work_kind": "fp64 operations executed by switched-on lanes (committed counter passes), whole precompute",
                     "traffic": hbm if seen else None, "once_per_scene": True,
                     "note": "rows and screens run side by side, so the two times overlap and the fraction is a lower bound"}
    return out


# ---------------------------------------------------------------------------------------------------------------
def config2_block(local_rank, dev, steps, reference_mode):
    """Config 2 of BASELINE.json on one GPU (place_levant, 32 x 32 receivers, reflections to order 2): a step is the
    solve alone, 3.0 M path-solves in three dependent launches — the latency floor of the path."""
    import torch

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import geometry as geom

    place, tx, rx = scenarios.levant_grid(32)
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons), device=local_rank)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    ev = np.logical_or.reduce(screens)
    cand = scene.enumerate(vis, ev, np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)), 2, with_diffraction=False)
    scene.set_reference_solver(reference_mode)
    n_rx = len(rx)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    cap = 256 * n_rx
    o_rx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_group = torch.empty_like(o_rx)
    o_cand = torch.empty_like(o_rx)
    o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        return scene.solve_paths_device(cand, tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), o_group.data_ptr(),
                                        o_cand.data_ptr(), o_pts.data_ptr(), cap)

    scene.set_timing(False)
    for _ in range(20):
        found, attempted = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        found, attempted = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    scene.set_timing(True)
    fam = {"reflect": 0.0, "obstruct": 0.0, "finalize": 0.0, "rejudge": 0.0}
    for _ in range(steps):
        step()
        tm = scene.last_timing()
        for k in ("reflect", "obstruct", "finalize"):
            fam[k] += tm["ms"][k] / steps
        fam["rejudge"] += scene.rejudge_timing()["ms"] / steps
    out = {"workload": f"place_levant.geojson ({len(polygons)} polygons), 1 TX, 32x32 RX, reflections up to order 2 "
                       "(config 2 of BASELINE.json); a step is the solve alone",
           "solver_mode": "reference" if reference_mode else "exact", "steps": steps, "ms_per_step": 1e3 * dt, "path_solves_per_step": int(attempted),
           "path_solves_per_s": attempted / dt, "valid_paths": int(found), "kernel_ms_per_step": fam}
    cand.close()
    scene.close()
    return out


def config4_block(local_rank, dev, steps, reference_mode):
    """Config 4 of BASELINE.json on one GPU: ny.geojson tiled 8 x 8 (5 825 polygons), 256 receivers, reflections up to
    order 3 (1.4e8 chains of three, formed in the kernels from the level before instead of being stored).  A step is
    the solve alone."""
    import torch

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import geometry as geom

    place, tx, rx = scenarios.ny_tiled(8)
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons), device=local_rank)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    t0 = time.perf_counter()
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    vis_s = time.perf_counter() - t0
    ev = np.logical_or.reduce(screens)
    t0 = time.perf_counter()
    cand = scene.enumerate(vis, ev, np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)), 3, with_diffraction=False)
    enum_s = time.perf_counter() - t0
    scene.set_reference_solver(reference_mode)
    n_rx = len(rx)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    cap = 64 * n_rx
    o_rx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_group = torch.empty_like(o_rx)
    o_cand = torch.empty_like(o_rx)
    o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        return scene.solve_paths_device(cand, tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), o_group.data_ptr(),
                                        o_cand.data_ptr(), o_pts.data_ptr(), cap)

    scene.set_timing(False)
    for _ in range(2):
        found, attempted = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        found, attempted = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    scene.set_timing(True)
    step()
    tm = scene.last_timing()["ms"]
    tm = {k: float(v) for k, v in tm.items()}
    tm["rejudge"] = scene.rejudge_timing()["ms"]
    out = {"workload": f"ny.geojson tiled 8x8 ({len(polygons)} polygons), 1 TX, {n_rx} RX, reflections up to order 3 (config 4 "
                       "of BASELINE.json); a step is the solve alone",
           "solver_mode": "reference" if reference_mode else "exact", "steps": steps, "ms_per_step": 1e3 * dt,
           "candidates_by_group": [int(c) for c in cand.counts], "path_solves_per_step": int(attempted),
           "path_solves_per_s": attempted / dt, "valid_paths": int(found), "kernel_ms_per_step": tm,
           "visibility_matrix_and_screens_s": round(vis_s, 4), "enumeration_s": round(enum_s, 4)}
    cand.close()
    scene.close()
    return out


def config5_block(local_rank, dev, steps, reference_mode, counters):
    """Config 5 of BASELINE.json on one GPU: place_saint_jean, the five emitters of the reference's measurement campaign,
    5 000 receivers, order 2 with diffraction, fields at seven frequencies from 3.5 to 60 GHz.  A step is, for every
    emitter in turn, solve + path description + field step at all frequencies + per-receiver records."""
    import torch

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import geometry as geom

    place, emitters, rx, freqs = scenarios.saint_jean_sweep()
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons), device=local_rank)
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    scene.set_reference_solver(reference_mode)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    pipes = []
    for tx in emitters:
        cube = geom.Cube.by_point_and_side_length(tx, 0.2)
        ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
        cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
        pipes.append((cand, dp.DevicePipeline(scene, cand, tx, capacity=64 * len(rx))))

    def step():
        n_paths = attempted = 0
        kernel_ms = call_ms = 0.0
        for _, pipe in pipes:
            block = pipe.run(d_rx, freqs, records=True)
            n_paths += block.n_paths
            attempted += block.n_attempted
            kernel_ms += block.fields_timing["kernel"]
            call_ms += block.fields_timing["call"]
        return n_paths, attempted, kernel_ms, call_ms

    scene.set_timing(False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        n_paths, attempted, _, _ = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    scene.set_timing(True)
    n_paths, attempted, kernel_ms, call_ms = step()
    pf = n_paths * len(freqs)
    out = {"workload": f"place_saint_jean ({len(polygons)} polygons), {len(emitters)} TX x {len(rx)} RX x {len(freqs)} frequencies "
                       "(3.5-60 GHz), order 2 with diffraction (config 5 of BASELINE.json); a step = for every emitter solve + "
                       "description + fields at all frequencies + per-receiver records",
           "solver_mode": "reference" if reference_mode else "exact", "steps": steps, "ms_per_step": 1e3 * dt,
           "path_solves_per_step": int(attempted), "path_solves_per_s": attempted / dt, "valid_paths": int(n_paths),
           "path_frequencies_per_step": int(pf), "field_kernels_ms_per_step": kernel_ms, "field_call_ms_per_step": call_ms,
           "path_frequencies_per_s_field_kernels": pf / (kernel_ms * 1e-3) if kernel_ms > 0 else None,
           "path_frequencies_per_s_whole_step": pf / dt}
    c = counters.get("k_fields_sweep") or counters.get("k_fields")
    if c and "fp64_flop_per_path_frequency" in c and kernel_ms > 0:
        flop = c["fp64_flop_per_path_frequency"] * pf
        out["field_kernels_fp64_executed_TFLOPs"] = flop / (kernel_ms * 1e-3) / 1e12
        out["field_kernels_fp64_frac"] = out["field_kernels_fp64_executed_TFLOPs"] / FP64_VECTOR_PEAK_TFLOPS
    for cand, _ in pipes:
        cand.close()
    scene.close()
    return out


def main():
    global T_PROCESS
    T_PROCESS = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-side", type=int, default=100, help="receivers per side of the lattice (100 = the 10k-RX map)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the secondary config-2 block")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the secondary config-4 and config-5 blocks")
    ap.add_argument("--solver-mode", choices=("reference", "exact"), default="reference",
                    help="reference: a found path is kept only when the reference's own solver call (SciPy hybr, restated "
                         "on the GPU) reports success and its points pass the reference's checks, i.e. the reference's path "
                         "set; exact: every path that has a root is kept")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the secondary block timed in the other solver mode")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on cuda:0 (rehearsal on a one-GPU box)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return launch_workers(args)  # no GPU call has been made in this process
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    pg = None
    # (RT_BENCH_GROUP=1 with WORLD_SIZE=1 sets the process group up for one rank as well: a rehearsal of the RCCL
    # branches -- device-tensor gathers, barriers, reductions -- on a box with one GPU)
    use_group = world > 1 or os.environ.get("RT_BENCH_GROUP") == "1"
    if use_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL writes a version banner to stdout when its first communicator comes up; stdout carries the one JSON line
        # and nothing else, so file descriptor 1 points at stderr until the group has done its first collective
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier(device_ids=[local_rank])
            else:
                dist.init_process_group(args.backend)
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        pg = dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import geometry as geom
    from raytracing_amd import multithread_solve as mts

    def barrier():
        if use_group:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- precompute: scene file -> candidates (once per scene and emitter) ----------------------------------------
    # (the HIP runtime's own first allocation and first launch, ~0.15 s per process, are not the pipeline's: done here)
    torch.zeros(1, device=dev).add_(1)
    # (likewise the library's own first use in a process -- its code object is loaded on its first launch, its streams
    # are made with its first scene, ~15 ms -- on a scene of a few polygons that has nothing to do with the map)
    from raytracing_amd import place_utils

    warm = _lib.Scene(geom.scene_arrays_of(place_utils.create_small_place()[0].get_polygons_list()), device=local_rank)
    warm.sharp_edges()
    warm.close()
    if use_group:
        # (and the process group's first all-gather: the collective library sets its channels up on first use; and
        # PyTorch's byte kernels that put the gathered rows in order and symmetrise them: their code objects are
        # loaded on first use too, 50 ms and 16 ms in a trace of the sharded route)
        cdev0 = dev if args.backend == "nccl" else torch.device("cpu")
        first = torch.zeros((3, 4), dtype=torch.uint8, device=cdev0)
        gathered0 = torch.empty((world, 3, 4), dtype=torch.uint8, device=cdev0)
        if args.backend == "nccl":
            dist.all_gather_into_tensor(gathered0.view(world * 3, 4), first)
        else:
            dist.all_gather(list(gathered0.unbind(0)), first)
        rows0 = mts.dealt_rows_in_order(gathered0, 3 * world - 1)
        sym0 = (rows0[:2, :2] | rows0[:2, :2].T).to(dev).contiguous()
        rows0[2:].cpu().numpy().astype(bool), sym0.cpu()
    barrier()
    t_wall = time.perf_counter()
    stamps = {}

    def stamp(name, since):
        stamps[name] = round(time.perf_counter() - since, 4)
        return time.perf_counter()

    # (the scene's arrays come from the scene cache when this file has been loaded before with this loader — build()
    # loads every scene file once —, and through the place's objects otherwise: the same arrays bit for bit)
    loaded, tx = scenarios.lln_scene()
    arrays = loaded.arrays
    n_polygons = arrays.n
    t = stamp("scene_file_s", t_wall)
    # the receivers of the map are placed (host work: a lattice minus the buildings' footprints) on a second thread while
    # the scene goes to the GPU and the GPU paints: the library calls release the interpreter lock.  The thread makes no
    # GPU call and no collective (with several ranks the painting ends in an all-gather, which stays on this thread).
    import threading

    made = {}
    worker = threading.Thread(target=lambda: made.update(rx=scenarios.lattice_receivers(loaded, args.n_side, 1.5)))
    worker.start()
    scene = _lib.Scene(arrays, device=local_rank)
    pairs, edges = scene.sharp_edges()
    t = stamp("device_scene_and_sharp_edges_s", t)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
    vis, screen_vis = mts.sharded_visibility(scene, screens, dist=pg, on_device=True)
    worker.join()
    rx = made["rx"]
    vis_timing = scene.visibility_timing()
    ev = np.logical_or.reduce(screen_vis)
    t = stamp("visibility_matrix_and_emitter_screens_s", t)
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    t = stamp("enumeration_s", t)
    n_total = len(rx)
    begin, end = mts.shard_bounds(n_total, world, rank)
    d_rx = torch.tensor(rx[begin:end], dtype=torch.float64, device=dev).reshape(-1, 3)
    pipe = dp.DevicePipeline(scene, cand, tx, capacity=max(1 << 16, 40 * (end - begin)))
    barrier()
    precompute_s = time.perf_counter() - t_wall

    def step():
        block = pipe.run(d_rx, [FREQUENCY_HZ], records=True)
        records = dp.all_gather_records(dp.receiver_records(block), n_total, pg)
        return block, records

    # ---- the timed steps ---------------------------------------------------------------------------------------------
    def timed_steps(n_warm, n_steps, n_ev):
        """n_warm untimed and n_steps timed steps without the per-launch events, then n_ev more with them (kernel times)."""
        scene.set_timing(False)  # the event pairs around every launch are kept out of the timed region
        for _ in range(n_warm):
            block, records = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            block, records = step()
        barrier()
        elapsed = time.perf_counter() - t0
        scene.set_timing(True)
        fam = {"reflect": 0.0, "diffract": 0.0, "obstruct": 0.0, "finalize": 0.0, "rejudge": 0.0, "fields": 0.0,
               "fields_call": 0.0}
        launches = {"reflect": 0, "diffract": 0, "obstruct": 0, "finalize": 0}
        segments = 0
        rejudged = [0, 0]
        t1 = time.perf_counter()
        for _ in range(n_ev):
            block, records = step()
            for k in launches:
                fam[k] += block.solve_timing["ms"][k] / n_ev
                launches[k] += block.solve_timing["launches"][k]
            fam["rejudge"] += block.solve_timing["rejudge"]["ms"] / n_ev
            rejudged = [block.solve_timing["rejudge"]["n_in"], block.solve_timing["rejudge"]["n_kept"]]
            fam["fields"] += block.fields_timing["kernel"] / n_ev
            fam["fields_call"] += block.fields_timing["call"] / n_ev
            segments += block.solve_timing["segments"]
        torch.cuda.synchronize()
        ms_with_events = 1e3 * (time.perf_counter() - t1) / n_ev
        launches = {k: v / n_ev for k, v in launches.items()}
        return elapsed, fam, launches, segments / n_ev, rejudged, ms_with_events, block, records

    n_ev = max(1, min(args.steps, 10))
    other_mode = None
    if world == 1 and not args.no_other_mode:
        # the other solver mode first, as a secondary block: the headline is measured last, on the state it leaves
        other = "exact" if args.solver_mode == "reference" else "reference"
        scene.set_reference_solver(other == "reference")
        o_elapsed, o_fam, _, _, o_rej, _, o_block, _ = timed_steps(args.warmup, args.steps, n_ev)
        other_mode = {"solver_mode": other, "steps": args.steps, "ms_per_step": 1e3 * o_elapsed / max(1, args.steps),
                      "path_solves_per_s": float(o_block.n_attempted) * max(1, args.steps) / o_elapsed,
                      "valid_paths": int(o_block.n_paths), "kernel_ms_per_step": o_fam}
    scene.set_reference_solver(args.solver_mode == "reference")
    elapsed, fam, launches, segments, rejudged, ms_with_events, block, records = timed_steps(args.warmup, args.steps, n_ev)
    rec = dp.unpack_records(records.cpu().numpy(), 1)
    attempted_local = float(block.n_attempted)
    if use_group:
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed, precompute_s], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, precompute_s = float(t[0].item()), float(t[1].item())
        tot = torch.tensor([attempted_local], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        attempted_all = float(tot.item())
    else:
        attempted_all = attempted_local

    if rank == 0:
        steps = max(1, args.steps)
        counts = [int(c) for c in cand.counts]
        ms_step = 1e3 * elapsed / steps
        # bytes the field kernel must move per path of k interactions: points, chain, edge, outputs (24 (k + 2) + 4 k + 160)
        k_inter = ((block.kind & 255) + (block.kind >> 8)).to(torch.float64)
        field_bytes = float((24.0 * (k_inter + 2) + 4.0 * k_inter + 160.0).sum().item())
        counters = committed_counters()
        by_kernel = kernel_rooflines(end - begin, counts, cand.max_order, n_polygons, int(arrays.verts.shape[0]), fam,
                                     segments, field_bytes, counters, block.n_paths)
        # the visibility kernels of the precompute (this rank's share of the rows and screens)
        by_kernel.update(once_per_scene_rooflines(counters, vis_timing))
        step_kernels = ("reflect", "diffract", "obstruct", "rejudge", "fields")
        dominant = max(step_kernels, key=lambda k: by_kernel[k]["ms_per_step"])
        roof = dict(by_kernel[dominant])
        roof.pop("counters_by_kernel", None)
        roof.setdefault("traffic", None)
        roof["family"] = dominant
        roof["achieved_basis"] = ("work of one step of one rank (see work_kind) / mean device time of the family's launches in a "
                                  "step, measured in this run with HIP events on the library's stream")
        finite = np.isfinite(rec["power_db"][:, 0])
        flagged = int(rec["n_flagged"].sum())
        out = {
            "metric": "path-solves/sec (RX x candidate paths, whole node)",
            "value": attempted_all * steps / elapsed,
            "unit": "path-solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"lln.geojson ({n_polygons} polygons), 1 TX, {args.n_side}x{args.n_side} RX coverage "
                                   "map, reflections + diffractions up to order 2 (config 3 of BASELINE.json); a step = "
                                   "solve + path description + field step + all-gather of one record per receiver",
                       "receivers": n_total, "receivers_per_gpu": end - begin,
                       "candidates_per_receiver": 1 + int(sum(counts)), "candidates_by_group": counts,
                       "path_solves_per_step": int(attempted_all), "valid_paths_rank0": int(block.n_paths),
                       "valid_paths_all_gpus": int(rec["n_paths"].sum()), "frequency_hz": FREQUENCY_HZ,
                       "record_bytes_per_receiver": dp.record_bytes(1), "backend": args.backend if use_group else None,
                       "solver_mode": args.solver_mode,
                       "solver_mode_note": "reference = the reference's path set: every path the closed-form kernels find is "
                                           "put through the reference's own solver call (SciPy hybr, restated on the GPU) and "
                                           "kept only on its success and with its points passing the reference's checks "
                                           "(ray_tracing.py:309-314, 345-350, 371-376); exact = every path that has a root",
                       "paths_before_and_after_rejudge_rank0": rejudged},
            "coverage_map_wall_s": round(precompute_s + ms_step * 1e-3, 4),
            "coverage_map": {"precompute_s": round(precompute_s, 4), "precompute_breakdown_rank0": stamps,
                             "visibility_kernels_ms_rank0": vis_timing, "step_s": round(ms_step * 1e-3, 5),
                             "receivers_with_power": int(finite.sum()),
                             "paths_with_field_status": flagged,
                             "worst_field_status": int(rec["worst_status"].max()) if len(rec["worst_status"]) else 0,
                             "note": "scene file -> per-receiver dB on every rank: scene load, device scene, sharp edges, "
                                     "visibility matrix and emitter screens (dealt to the ranks, all-gathered), "
                                     "enumeration, then one step; max over ranks",
                             "taken_before_the_clock": ["interpreter start and imports", "the HIP runtime's first allocation "
                                                        "and launch", "the library's first use in the process (code object "
                                                        "load, stream creation) on a scene of 31 polygons", "the process "
                                                        "group's first all-gather and PyTorch's byte kernels (several ranks)"],
                             "cold_process_wall_s": round(precompute_s + ms_step * 1e-3 + (t_wall - T_PROCESS), 4),
                             "cold_process_note": "the same map with the clock started when this process entered main(): "
                                                  "imports and first-use costs included"},
            "roofline": roof,
            "roofline_by_kernel": by_kernel,
            "kernel_ms_per_step": fam,
            "kernel_ms_source": f"HIP events around every launch, over {n_ev} more steps after the timed region (which "
                                "runs without them)",
            "ms_per_step_with_events": ms_with_events,
            "step_host_and_collective_ms": ms_step - sum(fam[k] for k in ("reflect", "diffract", "obstruct", "finalize",
                                                                          "rejudge", "fields_call")),
        }
        if other_mode is not None:
            out["other_solver_mode"] = other_mode
        if world == 1 and not args.no_config2:
            out["config2"] = config2_block(local_rank, dev, 200, args.solver_mode == "reference")
            out["config2_other_solver_mode"] = config2_block(local_rank, dev, 200, args.solver_mode != "reference")
        if world == 1 and not args.no_extra_configs:
            out["config4"] = config4_block(local_rank, dev, 5, args.solver_mode == "reference")
            out["config5"] = config5_block(local_rank, dev, 5, args.solver_mode == "reference", counters)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(loaded.place, tx, rx, cand)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)  # (flushed: nothing after this line may cost the line)
    cand.close()
    scene.close()
    if use_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
