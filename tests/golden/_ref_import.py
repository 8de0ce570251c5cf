"""Import helper used ONLY by the golden-vector generator scripts in this directory.

The reference package needs numba, shapely and geopandas, none of which is installed here.
numba contributes no arithmetic of its own (its decorators compile the decorated NumPy code
unchanged), so the generator registers pass-through decorator modules under those names and
empty shapely/geopandas modules, then imports the reference's own `raytracing.geometry` from
/root/reference.  Functions that need Shapely/GeoPandas arithmetic are NOT usable this way and
no vector is generated from them.  Nothing in this file is imported by the product, the tests,
smoke() or bench.py; /root/reference does not exist on the GPU box.
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _passthrough_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and isinstance(dargs[0], types.FunctionType) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn

    return wrap


class _AnyType:
    """Stands for a numba type object: callable and subscriptable, never evaluated."""

    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


class _Omitted:
    pass


class _NoneType:
    pass


class _Integer:
    pass


class _UnicodeType:
    pass


def _generated_jit(*dargs, **dkwargs):
    """numba.generated_jit: the decorated function maps argument TYPES to an implementation.
    Emulated by calling it with small type tags derived from the run-time values."""

    def tag(value):
        if value is None:
            return _NoneType()
        if isinstance(value, (int, np.integer)):
            return _Integer()
        if isinstance(value, str):
            return _UnicodeType()
        if isinstance(value, np.ndarray):
            t = types.SimpleNamespace()
            t.is_c_contig = value.flags["C_CONTIGUOUS"]
            t.is_f_contig = value.flags["F_CONTIGUOUS"] and not t.is_c_contig
            return t
        return value

    def wrap(fn):
        import inspect

        sig = inspect.signature(fn)

        def dispatch(*args, **kwargs):
            bound = sig.bind(*args, **kwargs)
            bound.apply_defaults()
            impl = fn(*[tag(v) for v in bound.arguments.values()])
            return impl(*bound.arguments.values())

        dispatch.__name__ = fn.__name__
        return dispatch

    if len(dargs) == 1 and isinstance(dargs[0], types.FunctionType) and not dkwargs:
        return wrap(dargs[0])
    return wrap


def import_reference_geometry():
    if not hasattr(np, "float"):
        np.float = float  # removed from numpy 1.24; the reference still spells it this way
    numba = types.ModuleType("numba")
    numba.njit = _passthrough_decorator
    numba.jit = _passthrough_decorator
    numba.generated_jit = _generated_jit
    for name in ("float64", "int64", "boolean", "string"):
        setattr(numba, name, _AnyType())
    nbtypes = types.ModuleType("numba.types")
    for name in ("UniTuple", "float64", "int64", "string", "boolean", "optional"):
        setattr(nbtypes, name, _AnyType())
    nbtypes.Omitted = _Omitted
    nbtypes.NoneType = _NoneType
    nbtypes.Integer = _Integer
    nbtypes.UnicodeType = _UnicodeType
    numba.types = nbtypes
    sys.modules["numba"] = numba
    sys.modules["numba.types"] = nbtypes

    shapely = types.ModuleType("shapely")
    shgeom = types.ModuleType("shapely.geometry")

    class _Unavailable:
        def __init__(self, *a, **k):
            raise RuntimeError("shapely is not installed: this reference code path cannot be used for vectors")

    shgeom.Polygon = _Unavailable
    shgeom.Point = _Unavailable
    shapely.geometry = shgeom
    sys.modules["shapely"] = shapely
    sys.modules["shapely.geometry"] = shgeom
    sys.modules["geopandas"] = types.ModuleType("geopandas")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import raytracing.geometry as geom

    return geom


def import_reference_electromagnetism():
    """The reference's field module, for the generator only.  Beyond the stand-ins above it needs names from
    shapely.geometry / shapely.ops / geopandas at import time (never called here), `np.complex`, and its own directory
    on sys.path (it imports `electromagnetism_utils` both ways).  Importing it changes the working directory
    (place_utils.py:30); the caller's directory is restored."""
    import os

    geom = import_reference_geometry()
    if not hasattr(np, "complex"):
        np.complex = complex
    shgeom = sys.modules["shapely.geometry"]
    for name in ("LineString", "MultiPolygon"):
        setattr(shgeom, name, shgeom.Polygon)
    ops = types.ModuleType("shapely.ops")
    ops.split = None
    sys.modules["shapely.ops"] = ops
    sys.modules["shapely"].ops = ops
    em_dir = os.path.join(REFERENCE_ROOT, "electromagnetism_fun")
    if em_dir not in sys.path:
        sys.path.insert(0, em_dir)
    cwd = os.getcwd()
    try:
        import electromagnetism_fun.electromagnetism as em
    finally:
        os.chdir(cwd)
    return geom, em
