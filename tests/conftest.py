import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def oracle_polys(place):
    """The oracle's view of a product place: same vertices, same reported planes."""
    from oracle import geometry_oracle as go

    out = []
    for p in place.get_polygons_list():
        q = go.Poly(p.points, p.part)
        q.parametric = p.get_parametric().copy()
        q.material = p.properties["material"] if p.properties else "concrete"
        out.append(q)
    return out


@pytest.fixture(scope="session")
def places():
    from raytracing_amd import geometry as geom

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = geom.load_place(os.path.join(DATA, name + ".geojson"))
        return cache[name]

    return get


def small_tx_rx(n_rx=16):
    """TX / RX layout of the reference's create_small_place (place_utils.py:66-70)."""
    tx = np.array([3.0, 38.0, 18.0])
    rx = np.array([[5.0, 15.0 - 4.0 * i, 1.5] for i in range(n_rx)])
    return tx, rx
