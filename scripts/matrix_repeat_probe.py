"""The lln matrix painted over and over: every painting must equal the first (not a test): python scripts/matrix_repeat_probe.py [paintings]"""
import sys

sys.path.insert(0, ".")
import numpy as np

from raytracing_amd import _lib, scenarios

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loaded, tx = scenarios.lln_scene()
scene = _lib.Scene(loaded.arrays)
first = scene.visibility_matrix(False)
bad = 0
for k in range(1, n):
    again = scene.visibility_matrix(False)
    d = int((again != first).sum())
    bad += d
    if d:
        print(f"painting {k}: {d} entries differ", flush=True)
print(f"{n} paintings, {int(first.sum())} visible pairs, {bad} differing entries in all", flush=True)
