"""Benchmark of the path-solve hot path on MI355X.

One step = one pass of the solve (line of sight + every (receiver, candidate) path-solve with its checks) over one
batch of receivers already resident in HBM.  Workload at any GPU count: config 2 of BASELINE.json — place_levant,
one emitter, a 32×32 receiver lattice per GPU, reflections up to order 2 (rank r takes the lattice shifted by a
rank-dependent sub-pitch offset, so every rank has its own 1024 receivers: weak scaling, no data-path collective;
the per-receiver results are combined with one all-gather outside the timed region).

Prints ONE JSON line: whole-job path-solves/s, the roofline of the dominant kernel (device time measured with HIP
events inside the library on the stream the kernels run on) and a CPU baseline (the oracle's restatement of the
reference's solve, SciPy hybr per candidate, on this host's cores, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (half the 157.3 TFLOP/s fp32 vector rate)
HBM_PEAK_GBS = 8000.0


def measured_traffic(family):
    """HBM bytes per launch of the family's kernels from the committed PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate rocprofv3 runs of this script; scripts/pmc_traffic.py turns them into this file)."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(family)
    except (OSError, ValueError):
        return None


def large_workload(local_rank, dev):
    """The same solve on a coverage-map-sized batch: lln.geojson, one emitter, 100x100 receivers, reflections and
    diffractions up to order 2 (config 3 of BASELINE.json on one GPU).  Reported beside the headline, not as it."""
    import torch

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import geometry as geom

    t0 = time.perf_counter()
    stamps = {}

    def stamp(name, since):
        stamps[name] = round(time.perf_counter() - since, 3)
        return time.perf_counter()

    place, tx, rx = scenarios.lln_grid(100)
    polygons = place.get_polygons_list()
    t = stamp("scene_file_s", t0)
    scene = _lib.Scene(geom.scene_arrays_of(polygons), device=local_rank)
    pairs, edges = scene.sharp_edges()
    t = stamp("device_scene_and_sharp_edges_s", t)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.zeros(len(polygons), dtype=bool)
    # the matrix and the emitter's screens in one call, as RayTracingProblem.precompute asks for them
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    for v in screens:
        ev |= v
    t = stamp("visibility_matrix_and_emitter_screens_s", t)
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    stamp("enumeration_s", t)
    precompute_s = time.perf_counter() - t0
    n_rx = len(rx)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    cap = 1 << 21
    o_rx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_group = torch.empty_like(o_rx)
    o_cand = torch.empty_like(o_rx)
    o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device=dev)
    best, timing, found, attempted = None, None, 0, 0
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        found, attempted = scene.solve_paths_device(cand, tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), o_group.data_ptr(),
                                                    o_cand.data_ptr(), o_pts.data_ptr(), cap)
        dt = time.perf_counter() - t
        if best is None or dt < best:
            best, timing = dt, scene.last_timing()
    counts = [int(c) for c in cand.counts]
    mo = cand.max_order
    reflect_flop = 40.0 * n_rx * sum((g + 1) * counts[g] for g in range(mo))
    # the rest of the coverage map, everything staying on the GPU: solve, per-path description, fields; only the
    # per-receiver dB comes to the host
    torch.cuda.synchronize()
    t = time.perf_counter()
    found, _ = scene.solve_paths_device(cand, tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), o_group.data_ptr(),
                                        o_cand.data_ptr(), o_pts.data_ptr(), cap)
    n_dev = int(found)
    d_kind = torch.empty(n_dev, dtype=torch.int32, device=dev)
    d_chain = torch.empty((n_dev, _lib.RT_MAX_ORDER), dtype=torch.int32, device=dev)
    d_epolys = torch.empty((n_dev, 2), dtype=torch.int32, device=dev)
    d_epts = torch.empty((n_dev, 6), dtype=torch.float64, device=dev)
    cand.describe_device(o_group.data_ptr(), o_cand.data_ptr(), n_dev, d_kind.data_ptr(), d_chain.data_ptr(),
                         d_epolys.data_ptr(), d_epts.data_ptr())
    d_field = torch.empty((n_dev, 1, 6), dtype=torch.float64, device=dev)
    d_power = torch.empty((n_dev, 1), dtype=torch.float64, device=dev)
    d_angles = torch.empty((n_dev, 4), dtype=torch.float64, device=dev)
    d_time = torch.empty(n_dev, dtype=torch.float64, device=dev)
    d_status = torch.empty(n_dev, dtype=torch.int32, device=dev)
    d_db = torch.empty((n_rx, 1), dtype=torch.float64, device=dev)
    scene.fields_device(tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), d_kind.data_ptr(), d_chain.data_ptr(),
                        d_epolys.data_ptr(), d_epts.data_ptr(), o_pts.data_ptr(), n_dev, [3.6e9], d_field.data_ptr(),
                        d_power.data_ptr(), d_angles.data_ptr(), d_time.data_ptr(), d_status.data_ptr(), d_db.data_ptr())
    power_db_dev = d_db.cpu().numpy()
    device_route_s = time.perf_counter() - t
    # the same through the host-array entry points (paths to the host, description, fields): must give the same map
    t = time.perf_counter()
    res = scene.solve_paths(cand, tx, rx)
    keep = res["group"] >= 0
    n_paths = len(res["rx"])
    kind = np.zeros(n_paths, dtype=np.int32)
    chain = np.zeros((n_paths, _lib.RT_MAX_ORDER), dtype=np.int32)
    epolys = np.zeros((n_paths, 2), dtype=np.int32)
    epts = np.zeros((n_paths, 6))
    kind[keep], chain[keep], epolys[keep], epts[keep] = cand.describe(res["group"][keep], res["cand"][keep])
    fields = scene.fields(tx, rx, res["rx"], kind, chain, epolys, epts, res["points"], [3.6e9])
    host_route_s = time.perf_counter() - t
    covered = int(np.isfinite(fields["power_db"][:, 0]).sum())
    both = np.isfinite(fields["power_db"][:, 0])
    if not (np.array_equal(both, np.isfinite(power_db_dev[:, 0])) and
            np.allclose(power_db_dev[both, 0], fields["power_db"][both, 0], rtol=0, atol=1e-9)):
        raise RuntimeError("the device route and the host-array route of the coverage map disagree")
    out = {"workload": f"lln.geojson ({len(polygons)} polygons), 1 TX, 100x100 RX, reflections+diffractions up to order 2",
           "precompute_s": round(precompute_s, 3), "precompute_breakdown": stamps, "solve_ms": 1e3 * best, "path_solves": int(attempted),
           "path_solves_per_s": attempted / best, "valid_paths": int(found), "candidates": counts,
           "segments": int(timing["segments"]), "kernel_ms": {k: float(v) for k, v in timing["ms"].items()},
           "reflect_algorithmic_tflops_fp64": reflect_flop / (timing["ms"]["reflect"] * 1e-3) / 1e12,
           "reflect_algorithmic_over_fp64_vector_peak": reflect_flop / (timing["ms"]["reflect"] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
           "reflect_note": "40 flop per interaction of every (receiver, candidate) pair over the kernel time; the kernel "
                           "sorts the receivers into spatial tiles and rejects whole tiles, so this is speed against "
                           "brute force (it can exceed 1), not vector-unit utilisation",
           "obstruct_segments_per_s": timing["segments"] / (timing["ms"]["obstruct"] * 1e-3),
           "coverage_map_wall_s": round(precompute_s + device_route_s, 3),
           "coverage_map_after_precompute_s": {"device_pointers": round(device_route_s, 4),
                                               "host_arrays": round(host_route_s, 4)},
           "coverage_map_note": "scene file -> per-receiver dB on one GPU: scene load, sharp edges, visibility matrix, "
                                "emitter screens, enumeration, then solve, path description and fields at 3.6 GHz "
                                "through the device-pointer entry points (only the dB per receiver comes to the host; "
                                "the host-array route gives the same map and is timed beside it); without the "
                                "reference's CSV / JSON files",
           "receivers_with_power": covered}
    cand.close()
    scene.close()
    return out


def cpu_baseline_worker(args):
    """Oracle path-solves for one receiver on a strided sample of the candidate chains, for about `budget` s."""
    import warnings

    warnings.filterwarnings("ignore")
    polys_data, tx, rx, chains, budget = args
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
    for chain in chains:
        planes = [polys[i].parametric for i in chain]
        points, sol = go.reflexion_points_from_origin_destination_and_planes(tx, rx, planes)
        if sol.success:
            lines = np.vstack([tx.reshape(1, 3), points, rx.reshape(1, 3)])
            o.check_reflections(lines, list(chain))
        n += 1
        if time.perf_counter() - t0 > budget:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(place, tx, rx, chains2, budget_s=5.0, receivers_per_core=4):
    import multiprocessing as mp

    cores = min(16, os.cpu_count() or 1)
    polys_data = [(p.points, p.get_parametric()) for p in place.get_polygons_list()]
    stride = max(1, len(chains2) // 4000)
    sample = [tuple(int(v) for v in c) for c in chains2[::stride]]
    rx_ids = np.linspace(0, len(rx) - 1, receivers_per_core * cores).astype(int)
    jobs = [(polys_data, tx, rx[i], sample, budget_s) for i in rx_ids]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    solves = sum(r[0] for r in res)
    busy = sum(r[1] for r in res) / cores  # time per core, all cores busy
    return {"value": solves / busy, "unit": "path-solves/s", "cores": cores, "kind": "port",
            "sample": f"{len(jobs)} receivers x {len(sample)} order-2 chains (every {stride}th of {len(chains2)}) of the "
                      f"same workload = {solves} path-solves, oracle = reference algorithm with SciPy hybr, one process "
                      f"per core, {busy:.0f} s of work per core ({wall:.0f} s wall with process start-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the coverage-map-sized extra measurement")
    ap.add_argument("--n-side", type=int, default=32)
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="keep the per-launch timing events on during the timed steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on cuda:0 (rehearsal on a one-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the tensors of the collectives live

    from raytracing_amd import _lib, scenarios
    from raytracing_amd import geometry as geom

    t_pre = time.perf_counter()
    place, tx, rx = scenarios.levant_grid(args.n_side)
    if world > 1:  # each rank its own receivers: the lattice moved by a rank-dependent fraction of the pitch
        pitch = rx[1, 0] - rx[0, 0]
        rx = rx + np.array([pitch * rank / (2.0 * world), pitch * rank / (3.0 * world), 0.0])
    polygons = place.get_polygons_list()
    arrays = geom.scene_arrays_of(polygons)
    scene = _lib.Scene(arrays, device=local_rank)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.zeros(len(polygons), dtype=bool)
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    for v in screens:
        ev |= v
    cand = scene.enumerate(vis, ev, np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)), 2, with_diffraction=False)
    precompute_s = time.perf_counter() - t_pre

    n_rx = len(rx)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    cap = 256 * n_rx
    o_rx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_group = torch.empty(cap, dtype=torch.int32, device=dev)
    o_cand = torch.empty(cap, dtype=torch.int32, device=dev)
    o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device=dev)

    def step():
        return scene.solve_paths_device(cand, tx, d_rx.data_ptr(), n_rx, o_rx.data_ptr(), o_group.data_ptr(),
                                        o_cand.data_ptr(), o_pts.data_ptr(), cap)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs without the per-launch events (a pair of event records per launch costs a step of this
    # size a measurable part of its time); the kernel times of the roofline come from the same number of steps run
    # again right after it with the events on.
    scene.set_timing(args.events_in_timed_region)
    for _ in range(args.warmup):
        found, attempted = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        found, attempted = step()
    barrier()
    elapsed = time.perf_counter() - t0
    fam_ms = {"reflect": 0.0, "diffract": 0.0, "obstruct": 0.0, "finalize": 0.0}
    fam_launches = dict.fromkeys(fam_ms, 0)
    segments = 0
    scene.set_timing(True)
    step()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = scene.last_timing()
        for k in fam_ms:
            fam_ms[k] += tm["ms"][k]
            fam_launches[k] += tm["launches"][k]
        segments += tm["segments"]
    torch.cuda.synchronize()
    elapsed_with_events = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(attempted)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        attempted_all = float(tot.item())
        # per-receiver path counts of every rank on every rank: the one collective of the path (outside the timing)
        counts = torch.bincount(o_rx[:found].long(), minlength=n_rx).to(torch.int32).to(cdev)
        gathered = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts)
        paths_all = int(sum(int(g.sum().item()) for g in gathered))
    else:
        attempted_all = float(attempted)
        paths_all = int(found)

    if rank == 0:
        n_poly = len(polygons)
        pairs = n_rx * cand.total
        steps = max(1, args.steps)
        ms = {k: fam_ms[k] / steps for k in fam_ms}
        launches = {k: fam_launches[k] / steps for k in fam_ms}
        seg_per_step = segments / steps
        # algorithmic work per step.  Candidate kernels: 40 flop per interaction of every (receiver, candidate) pair
        # (mirror image, plane crossing, containment and spacing checks).  Obstruction: every segment is read once
        # (48 B) and its flag written once (1 B); the scene's pre-test records (64 B per polygon) and vertices are
        # read once per launch.
        reflect_flop = 40.0 * float(n_rx) * sum((g + 1) * int(cand.counts[g]) for g in range(cand.max_order))
        scene_bytes = 64.0 * n_poly + 24.0 * int(arrays.verts.shape[0])
        obstruct_bytes = 49.0 * seg_per_step + scene_bytes * launches["obstruct"]
        by_kernel = {
            "reflect": {"kernel": "k_reflect<2> (reflection orders 1 and 2 in one launch)", "bound": "valu_fp64",
                        "achieved": reflect_flop / (ms["reflect"] * 1e-3) / 1e12 if ms["reflect"] > 0 else 0.0,
                        "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "ms_per_step": ms["reflect"],
                        "launches_per_step": launches["reflect"], "algorithmic_flop_per_step": reflect_flop},
            "obstruct": {"kernel": "k_obstruct<false>+k_obstruct<true>", "bound": "hbm",
                         "achieved": obstruct_bytes / (ms["obstruct"] * 1e-3) / 1e9 if ms["obstruct"] > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": ms["obstruct"],
                         "launches_per_step": launches["obstruct"], "algorithmic_bytes_per_step": obstruct_bytes,
                         "segment_polygon_tests_per_s": seg_per_step * n_poly / (ms["obstruct"] * 1e-3)
                         if ms["obstruct"] > 0 else 0.0},
        }
        for v in by_kernel.values():
            v["frac"] = v["achieved"] / v["peak"]
        dominant = max(by_kernel, key=lambda k: by_kernel[k]["ms_per_step"])
        roof = dict(by_kernel[dominant])
        detail = measured_traffic(dominant)
        roof["traffic"] = detail["bytes_per_launch"] if detail else None
        roof["traffic_detail"] = detail
        roof["achieved_per_launch_basis"] = "algorithmic bytes (or flop) per step / launches per step / mean launch time"
        roof["note"] = ("launch- and latency-bound at this size (a step is four dependent launches of 2-40 us over 3 M "
                        "pairs); see large_workload for the same kernels on a coverage-map-sized batch")
        out = {
            "metric": "path-solves/sec (RX x candidate paths, whole node)",
            "value": attempted_all * args.steps / elapsed,
            "unit": "path-solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "place_levant.geojson (237 polygons), 1 TX, 32x32 RX lattice per GPU, "
                                   "reflections up to order 2 (config 2 of BASELINE.json)",
                       "receivers_per_gpu": n_rx, "candidates_per_receiver": 1 + cand.total,
                       "candidates_by_order": [int(c) for c in cand.counts[:cand.max_order]],
                       "valid_paths_all_gpus": paths_all, "precompute_s": round(precompute_s, 3)},
            "roofline": roof,
            "roofline_by_kernel": by_kernel,
            "kernel_ms_per_step": ms,
            "kernel_ms_source": "HIP events around every launch, over the same number of steps run again after the "
                                "timed region (which runs without them)",
            "ms_per_step_with_events": 1e3 * elapsed_with_events / steps,
            "segments_per_step": seg_per_step,
            "pairs_per_step": pairs,
        }
        if not args.no_large and world == 1:
            out["large_workload"] = large_workload(local_rank, dev)
        if not args.no_cpu_baseline and world == 1:
            chains2, _ = cand.group(1)
            out["cpu_baseline"] = cpu_baseline(place, tx, rx, chains2)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    cand.close()
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
