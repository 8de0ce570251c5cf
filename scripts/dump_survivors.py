"""Writes the surviving paths of a full-size workload as the GPU finds them (exact-root mode, and reference-compatible
mode beside it) to gpurun_out/survivors_<name>.npz.  tests/golden/make_full_rejudge.py then runs SciPy's hybr on every
one of them on the CPU, exactly as the reference calls it, and writes the fixture the GPU test holds k_rejudge against.

Usage (GPU box):  python scripts/dump_survivors.py lln levant
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_configs as pc  # noqa: E402

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name in sys.argv[1:] or list(pc.FULL_SIZE):
        solved = pc.full_size_solve(name)
        data = {k: solved[k] for k in ("tx", "rx", "order", "with_diffraction", "n_polygons", "ev", "edge_pairs", "edges")}
        for mode in ("exact", "reference"):
            for key, value in solved[mode].items():
                if key not in ("n_attempted", "rejudge"):
                    data[f"{mode}_{key}"] = value
            print(f"[{name}] {mode}: {len(solved[mode]['rx'])} paths of {solved[mode]['n_attempted']} path-solves", flush=True)
        timing = solved["reference"]["rejudge"]
        data["rejudge_timing"] = np.array([timing[k] for k in ("ms", "n_in", "n_kept")], dtype=float)
        path = os.path.join(ROOT, "gpurun_out", f"survivors_{name}.npz")
        np.savez_compressed(path, **data)
        print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB), rejudge {timing}", flush=True)
