"""The scene cache (geometry.load_scene): the arrays it hands out for a file it has seen before are, bit for bit, the
arrays the loader makes from the file through the place's objects — on every scene file under data/ — and a change of
the file, the seed or the loader's code makes a new entry instead of reusing the old one."""
import glob
import os
import shutil

import numpy as np
import pytest

from conftest import ROOT

FILES = sorted(glob.glob(os.path.join(ROOT, "data", "*.geojson")) +
               glob.glob(os.path.join(ROOT, "data", "place_saint_jean", "*.geojson")))
FIELDS = ("vert_off", "verts", "planes", "material", "building")


def _same(a, b):
    for k in FIELDS:
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), k
    assert a.n == b.n


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_cached_arrays_are_the_loaders_arrays(path, tmp_path, monkeypatch):
    from raytracing_amd import geometry as geom

    monkeypatch.setattr(geom, "_scene_cache_dir", lambda: str(tmp_path))
    fresh = geom.SceneArrays(geom.load_place(path).get_polygons_list())
    first = geom.load_scene(path)            # not in the cache: made through the place, then stored
    assert len(os.listdir(tmp_path)) == 1
    second = geom.load_scene(path)           # read back
    assert second._place is None             # no place was built for it
    _same(first.arrays, fresh)
    _same(second.arrays, fresh)
    place = geom.load_place(path)
    want_fp = [p.get_top_face().points[:, :2] for p in place.polyhedra]
    assert len(second.footprints) == len(want_fp)
    for got, want in zip(second.footprints, want_fp):
        np.testing.assert_array_equal(got, want)
    lo, hi = place.get_domain()
    np.testing.assert_array_equal(second.get_domain()[0], lo)
    np.testing.assert_array_equal(second.get_domain()[1], hi)
    _same(geom.SceneArrays(second.place.get_polygons_list()), fresh)  # the place is there when asked for


def test_another_seed_file_or_loader_is_another_entry(tmp_path, monkeypatch):
    from raytracing_amd import geometry as geom

    monkeypatch.setattr(geom, "_scene_cache_dir", lambda: str(tmp_path / "cache"))
    os.makedirs(tmp_path / "cache")
    src = os.path.join(ROOT, "data", "small.geojson")
    a = geom.load_scene(src, seed=0)
    geom.load_scene(src, seed=1)
    assert len(os.listdir(tmp_path / "cache")) == 2
    copy = str(tmp_path / "small.geojson")
    shutil.copy(src, copy)
    with open(copy, "a") as f:
        f.write("\n")  # same scene, other bytes
    b = geom.load_scene(copy, seed=0)
    assert len(os.listdir(tmp_path / "cache")) == 3
    _same(a.arrays, b.arrays)
    monkeypatch.setattr(geom, "_LOADER_FINGERPRINT", "another version of the loader")
    geom.load_scene(src, seed=0)
    assert len(os.listdir(tmp_path / "cache")) == 4
    # a damaged entry is made again, not trusted
    monkeypatch.setattr(geom, "_LOADER_FINGERPRINT", None)
    for name in os.listdir(tmp_path / "cache"):
        with open(tmp_path / "cache" / name, "wb") as f:
            f.write(b"not an archive")
    c = geom.load_scene(src, seed=0)
    _same(a.arrays, c.arrays)
    # and without a cache the loader is simply the loader
    d = geom.load_scene(src, use_cache=False)
    _same(a.arrays, d.arrays)
