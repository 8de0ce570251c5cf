"""Makes tests/golden/lln_matrix_rows.npz — CPU only, test infrastructure.

Every row of the visibility matrix of lln.geojson (7 615 polygons, 58 million entries) as the oracle's painter
(oracle/visibility_oracle.c) decides it, NOT symmetrised: row i = what face i sees.  Stored as packed bits.
About a minute on 8 cores.

Usage:  python tests/golden/make_lln_matrix.py [--procs 8]
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_configs as pc  # noqa: E402

OUT = os.path.join(HERE, "lln_matrix_rows.npz")
_G = {}


def _init():
    from oracle import visibility_oracle as vo

    w = pc.workload("c3_lln")
    _G["polys"] = pc.oracle_polys_with_buildings(w["place"])
    _G["packed"] = vo.PackedScene(_G["polys"])


def _row(i):
    from oracle import visibility_oracle as vo

    return i, vo.polygon_visibility_vector(_G["polys"][i], _G["polys"], self_index=i, packed=_G["packed"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    args = ap.parse_args()
    t0 = time.time()
    n = len(pc.oracle_polys_with_buildings(pc.workload("c3_lln")["place"]))
    rows = np.zeros((n, n), dtype=bool)
    with mp.Pool(args.procs, initializer=_init) as pool:
        for i, row in pool.imap_unordered(_row, range(n), chunksize=16):
            rows[i] = row
    np.savez_compressed(OUT, n=n, bits=np.packbits(rows))
    print(f"{n} rows, {int(rows.sum())} entries set, {int((rows | rows.T).sum())} after symmetrising, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
