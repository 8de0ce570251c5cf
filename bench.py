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


def cpu_baseline(place, tx, rx, chains2, budget_s=12.0):
    import multiprocessing as mp

    cores = min(16, os.cpu_count() or 1)
    polys_data = [(p.points, p.get_parametric()) for p in place.get_polygons_list()]
    stride = max(1, len(chains2) // 4000)
    sample = [tuple(int(v) for v in c) for c in chains2[::stride]]
    rx_ids = np.linspace(0, len(rx) - 1, cores).astype(int)
    jobs = [(polys_data, tx, rx[i], sample, budget_s) for i in rx_ids]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    solves = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": solves / busy, "unit": "path-solves/s", "cores": cores, "kind": "port",
            "sample": f"{cores} receivers x {len(sample)} order-2 chains (every {stride}th of {len(chains2)}) of the same "
                      f"workload, oracle = reference algorithm with SciPy hybr, one process per receiver, "
                      f"{budget_s:.0f} s budget each ({wall:.0f} s wall with process start-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-side", type=int, default=32)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

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
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.zeros(len(polygons), dtype=bool)
    for v in scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False):
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

    for _ in range(args.warmup):
        found, attempted = step()
    fam_ms = {"reflect": 0.0, "diffract": 0.0, "obstruct": 0.0, "finalize": 0.0}
    fam_launches = dict.fromkeys(fam_ms, 0)
    segments = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        found, attempted = step()
        tm = scene.last_timing()
        for k in fam_ms:
            fam_ms[k] += tm["ms"][k]
            fam_launches[k] += tm["launches"][k]
        segments += tm["segments"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(attempted)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        attempted_all = float(tot.item())
        # per-receiver path counts of every rank on every rank: the one collective of the path (outside the timing)
        counts = torch.bincount(o_rx[:found].long(), minlength=n_rx).to(torch.int32)
        gathered = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts)
        paths_all = int(sum(int(g.sum().item()) for g in gathered))
    else:
        attempted_all = float(attempted)
        paths_all = int(found)

    if rank == 0:
        n_poly = len(polygons)
        pairs = n_rx * cand.total
        # algorithmic work per launch: obstruction = 13 flop per (segment, polygon); candidate kernels = 40 flop per
        # interaction of every (receiver, candidate) pair
        flops = {
            "obstruct": 13.0 * n_poly * segments / max(1, args.steps),
            "reflect": 40.0 * float(n_rx) * sum((g + 1) * int(cand.counts[g]) for g in range(cand.max_order)),
        }
        dominant = max(("reflect", "obstruct"), key=lambda k: fam_ms[k])
        per_step_ms = fam_ms[dominant] / args.steps
        achieved = flops[dominant] / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else 0.0
        kernel_names = {"reflect": "k_reflect<1>+k_reflect<2>", "obstruct": "k_obstruct"}
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
            "roofline": {"bound": "valu_fp64", "kernel": kernel_names[dominant], "achieved": achieved,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                         "ms_per_step_by_kernel": {k: fam_ms[k] / args.steps for k in fam_ms},
                         "launches_per_step": {k: fam_launches[k] / args.steps for k in fam_ms},
                         "algorithmic_flop_per_step": flops, "segments_per_step": segments / args.steps,
                         "pairs_per_step": pairs},
        }
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
