"""Builds the HIP library in-tree: ``raytracing_amd/libraytracing_amd.so`` for gfx950.

hipcc cross-compiles without a GPU, so this runs anywhere the ROCm toolchain is installed.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libraytracing_amd.so")
SOURCES = ["scene.hip", "occlusion.hip", "solve.hip", "visibility.hip", "sharp_edges.hip", "fields.hip", "channel_stats.hip",
           "radiomap.hip"]
HEADERS = ["rt_common.h", "rt_candidates.h", "rt_fresnel.h", os.path.join("..", "..", "include", "rt_pathsolve.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


FLAGS_STAMP = LIB + ".flags"


def _flags_line():
    return " ".join(FLAGS + os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split())


def _stale():
    if not os.path.exists(LIB):
        return True
    try:
        with open(FLAGS_STAMP) as f:
            if f.read() != _flags_line():
                return True  # built with other flags (e.g. an instrumented build)
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DRT_OCC_STATS for the obstruction work counters
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flags_line())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
